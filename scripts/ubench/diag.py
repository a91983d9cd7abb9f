import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
from oracle import oracle
st = seeded_state_dict(1234)
m = SelfAttentiveVAD(80,3,128,0.5); m.load_state_dict({k: torch.from_numpy(v) for k,v in st.items()}); m = m.cuda().eval()
for shape in [(4,7,80),(2,40,80),(2,96,80),(8,800,80)]:
    x = seeded_features(1, shape)
    ref = oracle.forward(st, x)
    for rm in (1,2):
        m.row_mode = rm
        y = m(torch.from_numpy(x).cuda()).cpu().numpy()
        print(shape, "row_mode", rm, "maxerr", np.abs(y-ref).max(), "nan", np.isnan(y).sum())
