import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
from oracle import oracle
st = seeded_state_dict(1234)
m = SelfAttentiveVAD(80,3,128,0.5); m.load_state_dict({k: torch.from_numpy(v) for k,v in st.items()}); m = m.cuda().eval()
m.precision = "bf16"
shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4,7,80)
x = seeded_features(1, shape)
y = m(torch.from_numpy(x).cuda()).cpu().numpy()
ref = oracle.forward(st, x)
print("out nan:", np.isnan(y).sum(), "of", y.size, "maxerr", np.nanmax(np.abs(y-ref)))
ws = m._workspace
B,T = shape[:2]
nblk = B*((T+31)//32) if T>32 else (B + (32//T) - 1)//(32//T)
nblk_pad = (nblk+3)//4*4
hbytes = nblk_pad*32*128*4
h = ws[:hbytes].view(torch.float32).cpu().numpy()
print("h nan:", np.isnan(h).sum(), "absmax", np.nanmax(np.abs(h)))
fb = (nblk_pad+1)*8192
off = hbytes
for name in ("q","k","vt","ctx"):
    t = ws[off:off+fb].view(torch.bfloat16).float().cpu().numpy()
    print(name, "nan:", np.isnan(t[:nblk*4096]).sum(), "absmax", np.nanmax(np.abs(t[:nblk*4096])))
    off += fb
