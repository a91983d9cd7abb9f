import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("SAVAD_LIB", os.path.abspath("scripts/ubench/libsavad_timing.so"))
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.row_mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2; m.attention_splits = 1
x = torch.randn(B, 800, 80, device="cuda")
for _ in range(3): m(x)
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 16)()
lib.savad_debug_stamps(buf, 16)
t = list(buf[:9])
names = ["prologue+combine", "out-proj(4 blk)", "LN1", "park stores", "FFN(32 blk)", "resid+store", "LN2", "QKV(12 blk)"]
print("B", B, "row_last kernel stamps (ticks @100MHz?):")
for i, nme in enumerate(names):
    print(f"  {nme:18s} {t[i+1]-t[i]:8d}")
print("  total", t[8] - t[0] if t[8] else t[7]-t[0])
buf2 = (ctypes.c_longlong * 32)()
lib.savad_debug_stamps(buf2, 32)
a = list(buf2[16:22])
print("attention (wave 0 of WG 0), cycles summed over its key tiles:")
for nme, val in zip(["wait+barrier", "dma issue", "S=KQ^T (64 MFMA)", "mask+softmax+rescale", "PV (64 MFMA)", "loop overhead"], a):
    print(f"  {nme:22s} {val:9d}")
print("  sum", sum(a))
