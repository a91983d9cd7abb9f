"""Phase stamps of the fp32s fused launch (build: scripts/ubench/build_timing.sh -> libsavad_timing.so); wave 0 of workgroup 0, the last
layer-0 launch of a forward.  python scripts/ubench/phase_timing_f32s.py [B T]"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["SAVAD_LIB"] = os.path.abspath("scripts/ubench/libsavad_timing.so")
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features, _lib
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 800)
m = SelfAttentiveVAD(80, 2, 128, 0.5)   # two layers: the stamps of the first (non-last) fused launch survive the last launch's? no: use L = 2 and read after
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234, num_layers=2).items()})
m = m.cuda().eval(); m.precision = "fp32s"
x = torch.from_numpy(seeded_features(3, (B, T, 80))).cuda()
for _ in range(5): m(x)
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 64)()
lib.savad_debug_stamps(buf, 64)
t = list(buf)
names = {1: "prologue: K0 landed", 2: "scores of tile 0", 3: "key tiles", 4: "context -> triples", 5: "out-projection (+h)", 6: "LN + split", 7: "FFN chunk 0",
         8: "FFN chunk 1", 9: "FFN chunk 2", 10: "FFN chunk 3", 11: "LN + split", 12: "Q slot 0", 13: "Q slot 1", 14: "K slot 0", 15: "K slot 1", 16: "V slot 0", 17: "V slot 1", 18: "end"}
prev = t[0]
print(f"[{B},{T},80] fp32s fused launch, wave 0 of workgroup 0 (s_memtime ticks = shader cycles)")
for i in range(1, 19):
    if t[i]:
        print(f"  {names[i]:24s} {t[i] - prev:8d}")
        prev = t[i]
print(f"  total {t[18] - t[0]}")
print(f"  FFN chunk 0: W1 {t[20] - t[6]}, relu + split {t[21] - t[20]}, W2 {t[7] - t[21]}")
lib.savad_debug_wg_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
wb = (ctypes.c_longlong * 4096)()
lib.savad_debug_wg_stamps(wb, 4096)
import numpy as np
w = np.array(list(wb), dtype=np.int64).reshape(1024, 4)
w = w[w[:, 0] != 0]
dur = w[:, 1] - w[:, 0]
rt0, rt1 = w[:, 2] - w[:, 2].min(), w[:, 3] - w[:, 2].min()   # s_memrealtime: 100 MHz
print(f"workgroups of the last fused launch: {len(w)}; duration cycles min {dur.min()} median {int(np.median(dur))} max {dur.max()}")
print(f"  start spread {rt0.max() / 100:.2f} us; end: first {rt1.min() / 100:.2f} us, median {np.median(rt1) / 100:.2f} us, last {rt1.max() / 100:.2f} us (after the first start)")
order = np.argsort(dur)
print("  slowest workgroups (index, cycles, start us, end us):", [(int(i), int(dur[i]), round(rt0[i] / 100, 1), round(rt1[i] / 100, 1)) for i in order[-6:]])
print("  fastest:", [(int(i), int(dur[i]), round(rt0[i] / 100, 1), round(rt1[i] / 100, 1)) for i in order[:4]])
print("row chain slots 1..7 (wait + barrier, MFMAs + DMA pieces, gap to the next slot):")
for T_ in range(1, 8):
    a, b2, c = t[24 + 3 * T_], t[25 + 3 * T_], t[26 + 3 * T_]
    nxt = t[24 + 3 * (T_ + 1)] if T_ < 7 else 0
    print(f"  slot {T_}: acquire {b2 - a:6d}  gemm {c - b2:6d}  then {nxt - c if nxt else 0:6d}")
