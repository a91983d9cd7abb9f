import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["SAVAD_LIB"] = os.path.abspath("scripts/ubench/libsavad_timing.so")
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"
x = torch.randn(B, 800, 80, device="cuda")
for _ in range(3): m(x)
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 32)()
lib.savad_debug_stamps(buf, 32)
a = list(buf[24:28])
print("B", B, "bf16 attention, wave 0 of WG 0, cycles over 13 stages (25 key tiles):")
for nme, val in zip(["wait+barrier", "dma issue", "2 tiles compute", "loop"], a): print(f"  {nme:18s} {val:8d}  per stage {val/13:8.0f}")
print("  sum", sum(a))
