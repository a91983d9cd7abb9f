import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["SAVAD_LIB"] = os.path.abspath("scripts/ubench/libsavad_timing.so")
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, _lib
B, T = int(sys.argv[1]), int(sys.argv[2])
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
x = torch.randn(B, T, 80, device="cuda")
for _ in range(3): m(x)
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 64)()
lib.savad_debug_stamps(buf, 64)
t = list(buf[40:46])
names = ["DMA lanes + first DMA + bias/PE requests", "input projection (160 MFMA)", "h store issue", "LayerNorm", "QKV tail (768 MFMA)"]
print(f"[{B},{T},80] input_qkv_kernel_m, wave 0 of WG 0, cycles:")
for i, nme in enumerate(names):
    print(f"  {nme:42s} {t[i+1]-t[i]:8d}")
print("  total", t[5] - t[0])
