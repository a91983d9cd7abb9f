import sys, numpy as np, torch
sys.path.insert(0, '.')
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_features, seeded_state_dict
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.to("cuda").eval()
def run(x, prec, rm=0):
    m.precision, m.row_mode = prec, rm
    with torch.no_grad():
        y = m(features=torch.from_numpy(x).to("cuda"))
    torch.cuda.synchronize()
    m.row_mode = 0
    return y.cpu().numpy()
for shape in ((3,1,80),(3,2,80),(1,7,80),(4,7,80),(3,32,80),(1000,7,80)):
    x = seeded_features(400+shape[1], shape)
    a = run(x, "fp32s", 7); b = run(x, "fp32s", 8)
    d = np.abs(a-b).reshape(-1, shape[1], 2).max(axis=2)
    print(shape, "max", d.max(), "rows>1e-5:", (d>1e-5).sum(), "of", d.size, "first bad", np.argwhere(d>1e-5)[:6].tolist())
