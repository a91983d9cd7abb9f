import ctypes, os, sys, struct
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features, _lib
B, T = int(sys.argv[1]), int(sys.argv[2])
m = SelfAttentiveVAD(80, 1, 128, 0.5)
st = {k: v for k, v in seeded_state_dict(1234).items() if ".layers.1." not in k and ".layers.2." not in k}
m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
m = m.cuda().eval(); m.precision = "bf16"; m.row_mode = 6
x = torch.from_numpy(seeded_features(B + T, (B, T, 80))).cuda()
with torch.no_grad(): y = m(x)
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 64)()
lib.savad_debug_stamps(buf, 64)
f = [struct.unpack("f", struct.pack("I", int(v) & 0xFFFFFFFF))[0] for v in buf]
print("final: la lb ta tb ia ib qbA qbB storeA storeB:", [round(v, 4) for v in f[0:10]])
print("prologue: s0a[0] nega[0] s0b[0] negb[0]:", f[10:14])
for j in range(min((T + 31) // 32, 6)):
    print("step", j, "ra ca0 ea0 nega0 la | rb cb0 lb:", [round(v, 4) for v in f[16 + 8 * j: 24 + 8 * j]])
