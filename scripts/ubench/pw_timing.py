#!/usr/bin/env python3
"""phase stamps of the persistent bf16 attention kernel (wave 0 of workgroup 0): pw_timing.py B T"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SAVAD_LIB"] = os.path.abspath(sys.argv[4] if len(sys.argv) > 4 else "scripts/ubench/libsavad_timing.so")
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features, _lib
B, T = int(sys.argv[1]), int(sys.argv[2])
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"; m.row_mode = 5
x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda()
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 600): m(x)   # long enough for the clocks to ramp
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 16)()
lib.savad_debug_stamps(buf, 16)
w = []
for v in buf: w += [v & 0xffffffff, (v >> 32) & 0xffffffff]
names = {0: "epilogue end -> item start (incl. kernel prologue)", 24: "  cursor advance", 25: "  next item's parameters", 26: "  wait for the staged Q", 27: "  Q -> AGPRs", 28: "  K(0) reads, init, LDS wait", 29: "  S(0) MFMAs + O zeroing", 1: "  first reference (out-of-line call)", 2: "barrier", 3: "vmcnt(8) wait",
         4: "even step", 5: "odd step", 23: "DMA advance", 6: "last step of an item", 22: "key-split item: prologue", 20: "key-split item: the wave's tiles", 21: "key-split item: combine", 8: "item epilogue", 9: "idle stages", 10: "tail", 11: "stage dispatch"}
tot = sum(w[c] for c in names)
cyc, rt = (w[17] - w[16]) & 0xffffffff, (w[19] - w[18]) & 0xffffffff
print(f"wave 0 of WG 0: {cyc} shader cycles in {rt / 100.0:.1f} us -> {cyc / max(rt, 1) * 100:.0f} MHz")
print("cold calls buf0/buf1/first:", w[12], w[13], w[14], "blocks moved:", w[15])
print(f"B={B} T={T}: wave 0 of WG 0, cycles per category (total {tot}):")
for c, n in names.items(): print(f"  {n:48s} {w[c]:9d}  {100.0 * w[c] / max(tot, 1):5.1f} %")
