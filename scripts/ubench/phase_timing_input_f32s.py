"""Phase stamps of input_qkv_kernel_f32s (timing build: scripts/ubench/build_timing.sh), wave 0 of workgroup 0: python scripts/ubench/phase_timing_input_f32s.py [B T]"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["SAVAD_LIB"] = os.path.abspath("scripts/ubench/libsavad_timing.so")
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, _lib
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 800)
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "fp32s"; m.row_mode = 3
x = torch.randn(B, T, 80, device="cuda")
for _ in range(3): m(x)
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 64)()
lib.savad_debug_stamps(buf, 64)
t = list(buf[57:64])
names = ["ring fill (24 DMA pieces) + bias staging", "features, input weights, bias / PE, input GEMM (120 MFMAs)", "residual store + LayerNorm + split", "Q slots (2 x 96 MFMAs + epilogues)", "K slots", "V^T slots"]
print(f"[{B},{T},80] input_qkv_kernel_f32s, cycles:")
for i, n in enumerate(names): print(f"  {n:60s} {t[i+1]-t[i]:8d}")
print("  total", t[6] - t[0])
