import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
st = seeded_state_dict(1234, num_layers=1)
m = SelfAttentiveVAD(80,1,128,0.5); m.load_state_dict({k: torch.from_numpy(v) for k,v in st.items()}); m = m.cuda().eval()
m.precision = "bf16"
shape = tuple(int(a) for a in sys.argv[1:4])
x = seeded_features(1, shape)
y = m(torch.from_numpy(x).cuda()).cpu().numpy()
print("out nan:", np.isnan(y).sum(), "of", y.size)
ws = m._workspace
B,T = shape[:2]
nblk = B*((T+31)//32) if T>32 else (B + (32//T) - 1)//(32//T)
nblk_pad = (nblk+3)//4*4
hbytes = nblk_pad*32*128*4
h = ws[:hbytes].view(torch.float32).cpu().numpy().reshape(nblk_pad, 16, 64, 4)
print("h nan per block:", np.isnan(h).reshape(nblk_pad,-1).sum(1))
fb = (nblk_pad+1)*8192
off = hbytes
for name in ("q","k","vt","ctx"):
    t = ws[off:off+fb].view(torch.bfloat16).float().cpu().numpy()[:nblk_pad*4096].reshape(nblk_pad, 8, 64, 8)
    nn = np.isnan(t)
    print(name, "nan per block:", nn.reshape(nblk_pad,-1).sum(1), "nan lanes in block0:", np.flatnonzero(nn[0].any(axis=(0,2)))[:70])
    off += fb
