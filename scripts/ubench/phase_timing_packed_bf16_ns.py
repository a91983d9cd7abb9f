"""Phase stamps of packed_forward_kernel_bf16_ns / packed_forward_kernel_f32s_ns (argv[3] = bf16 | fp32s) (build: -DSAVAD_TIMING -> scripts/ubench/libsavad_timing.so); wave 0 of WG 0, the LAST
layer's phases (stamps are overwritten layer by layer)."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["SAVAD_LIB"] = os.path.abspath("scripts/ubench/libsavad_timing.so")
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, _lib
B, T = int(sys.argv[1]), int(sys.argv[2])
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
m = m.cuda().eval(); m.precision = prec; m.row_mode = 8
x = torch.randn(B, T, 80, device="cuda")
for _ in range(3): m(x)
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 64)()
lib.savad_debug_stamps(buf, 64)
t = list(buf[40:53])
names = ["bias staging + input GEMM", "park + request + LN1 exchange (first)", "(layers 0..L-2) + Q K V^T of the last", "barrier (Q/K fragments)", "scores, softmax, PV, ctx exchange (2 barriers)",
         "unpark + out-projection", "LN2 exchange", "FFN1 (4 blocks) + hidden stores", "barrier (hidden)", "FFN2 (32 K-steps)", "after the loop: wait", "final LN + classifier"]
print(f"[{B},{T},80] packed_forward_kernel_{prec}_ns, cycles (last layer's phases):")
for i, n in enumerate(names): print(f"  {n:52s} {t[i+1]-t[i]:8d}")
print("  total", t[12] - t[0])
