import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
from oracle import oracle
st = seeded_state_dict(1234)
m = SelfAttentiveVAD(80, 3, 128, 0.5); m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}); m = m.cuda().eval()
verbose = len(sys.argv) > 1
shapes = [(2, 96, 80), (1, 192, 80), (3, 768, 80), (2, 801, 80), (5, 100, 80), (9, 33, 80), (2, 800, 80), (1, 2049, 80)]
if len(sys.argv) > 2: shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[2:]]
for shape in shapes:
    x = seeded_features(sum(shape), shape)
    m.row_mode, m.attention_splits = 5, 1
    with torch.no_grad():
        y = m(features=torch.from_numpy(x).cuda()).cpu().numpy()
    ref = oracle.forward(st, x, threads=16)
    d = np.abs(y - ref).max(axis=2)
    print(shape, "max|dlogp| = %.2e" % np.nanmax(d), "finite", np.isfinite(y).all(), flush=True)
    if verbose:
        for b in range(shape[0]):
            T = shape[1]
            print("   seq", b, " per 32-row block:", ["%.1e" % np.nan_to_num(d[b, i:i + 32], nan=9).max() for i in range(0, T, 32)][:30])
