#!/usr/bin/env python3
"""start / end of 128 of the 256 workgroups of the persistent bf16 attention kernel on the 100 MHz clock (a --wgtime build of the stream:
python scripts/gen_attn_pw.py --wgtime --out F; hipcc ... -DSAVAD_TIMING -DSAVAD_PW_INC=F): are there stragglers?   pw_wgtime.py B T [lib]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SAVAD_LIB"] = os.path.abspath(sys.argv[3] if len(sys.argv) > 3 else "scripts/ubench/libsavad_wgtime.so")
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features, _lib
B, T = int(sys.argv[1]), int(sys.argv[2])
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"; m.row_mode = 5
x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
for rep in range(3):
    for _ in range(400): m(x)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.savad_debug_stamps(buf, 64)
    w = np.frombuffer(bytes(buf), dtype=np.uint32)          # index j * 8 + xcd, j < 16
    start, end = (w & 0xffff).astype(np.int64), (w >> 16).astype(np.int64)
    dur = ((end - start) & 0xffff) * 0.01                   # us
    t0 = start.min()
    s_rel, e_rel = ((start - t0) & 0xffff) * 0.01, ((end - t0) & 0xffff) * 0.01
    d = dur.reshape(16, 8)
    print(f"[{B},{T}] launch {rep}: workgroup duration us: min {dur.min():.1f} median {np.median(dur):.1f} max {dur.max():.1f}; "
          f"first start -> last end {e_rel.max():.1f} us; start spread {s_rel.max():.1f} us")
    print("   per XCD median:", np.round(np.median(d, axis=0), 1).tolist())
    print("   per j   median:", np.round(np.median(d, axis=1), 1).tolist())
