"""Ring-step accounting of packed_forward_kernel_bf16 (build: -DSAVAD_TIMING -> scripts/ubench/libsavad_timing.so); wave 0 of WG 0."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["SAVAD_LIB"] = os.path.abspath("scripts/ubench/libsavad_timing.so")
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, _lib
B, T, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"; m.row_mode = mode
x = torch.randn(B, T, 80, device="cuda")
for _ in range(3): m(x)
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 64)()
lib.savad_debug_stamps(buf, 64)
t = list(buf[32:36])
print(f"[{B},{T},80] row_mode {mode}: compute {t[0]}  ring wait {t[1]}  DMA issue {t[2]}  total {sum(t)} ticks (36 ring steps)")
