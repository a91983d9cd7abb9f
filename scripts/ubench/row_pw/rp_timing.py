#!/usr/bin/env python3
"""phase stamps of the persistent bf16 row kernel (first wave of workgroup 0): rp_timing.py B T [iters] [lib]
build first: bash scripts/ubench/row_pw/build.sh"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["SAVAD_LIB"] = os.path.abspath(sys.argv[4] if len(sys.argv) > 4 else "scripts/ubench/row_pw/libsavad_rowpw_timing.so")
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features, _lib
B, T = int(sys.argv[1]), int(sys.argv[2])
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"; m.row_mode = 7
x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda()
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 300): m(x)   # long enough for the clocks to ramp
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 16)()
lib.savad_debug_stamps(buf, 16)
w = []
for v in buf: w += [v & 0xffffffff, (v >> 32) & 0xffffffff]
names = ["item parameters", "out-projection (64 MFMAs)", "LayerNorm 1 + b2", "FFN chunk 0 (128 MFMAs)", "FFN chunk 1", "FFN chunk 2", "FFN chunk 3",
         "residual store + LayerNorm 2", "Q block (64 MFMAs + tail)", "K block", "V block + next item's C operands", "prologue / loop edge (up to the item top)"]
tot = sum(w[:12])
nblk = B * ((T + 31) // 32) if T > 32 else (B + 32 // T - 1) // (32 // T)
npairs = (nblk + 7) // 8 * 8 // 2
passes = -(-npairs // 1024)
print(f"B={B} T={T}: first wave of WG 0, {passes} passes, cycles per category of the last launch (total {tot}, {tot / passes:.0f} per item):")
for c, n in enumerate(names): print(f"  {n:40s} {w[c]:9d}  {100.0 * w[c] / max(tot, 1):5.1f} %   {w[c] / passes:8.0f} per item")
