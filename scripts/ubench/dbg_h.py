import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["SAVAD_LIB"] = os.path.abspath("scripts/ubench/libsavad_timing.so")
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, _lib
B, T = int(sys.argv[1]), int(sys.argv[2])
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.row_mode = 5; m.attention_splits = 1
x = torch.randn(B, T, 80, device="cuda")
m(x); torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 64)()
lib.savad_debug_stamps(buf, 64)
print(dict(zip(["nq", "a", "HT", "HS", "NT", "NG", "l_run*1000", "T", "B"], list(buf[50:59]))))
