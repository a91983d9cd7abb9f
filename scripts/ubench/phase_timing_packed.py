"""Phase stamps of packed_forward_kernel (build: -DSAVAD_TIMING -> scripts/ubench/libsavad_timing.so); wave 0 of WG 0,
the LAST layer's phases (stamps are overwritten layer by layer)."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["SAVAD_LIB"] = os.path.abspath("scripts/ubench/libsavad_timing.so")
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, _lib
B, T = int(sys.argv[1]), int(sys.argv[2])
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.row_mode = 4
x = torch.randn(B, T, 80, device="cuda")
for _ in range(3): m(x)
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 64)()
lib.savad_debug_stamps(buf, 64)
t = list(buf[48:59])
names = ["input GEMM + PE (<=64 MFMA)", "(layers 0..L-2)", "Q K V^T (192 MFMA)", "score share + exchange (16 MFMA)", "softmax + PV (16 MFMA)",
         "ctx exchange", "out-proj (64 MFMA) + LN2", "FFN (512 MFMA)", "reduce-scatter + residual", "final LN + classifier"]
print(f"[{B},{T},80] packed_forward_kernel, cycles (last layer):")
print(f"  {names[0]:36s} {t[1]-t[0]:8d}")
print(f"  {'layers 0..L-2 + LN1 of the last':36s} {t[2]-t[1]:8d}")
for i in range(2, 9): print(f"  {names[i]:36s} {t[i+1]-t[i]:8d}")
print(f"  {names[9]:36s} {t[10]-t[9]:8d}")
print("  total", t[10] - t[0])
u = list(buf[40:46])
print("  last FFN chunk: wait W1", u[1]-u[0], " W1 MFMAs", u[2]-u[1], " relu+issue", u[3]-u[2], " wait W2", u[4]-u[3], " W2 MFMAs", u[5]-u[4])
