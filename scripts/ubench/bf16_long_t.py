import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
from oracle import oracle
st = seeded_state_dict(1234)
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
m = m.cuda().eval()
for T in (800, 1000, 1600):
  for seed in (1, 2, 3):
    B = 28 if T <= 1000 else 8
    x = seeded_features(seed * 1000 + T, (B, T, 80))
    ref = oracle.forward(st, x, threads=32)
    ref64 = oracle.forward(st, x, threads=32, acc64=True)
    out = {}
    for prec, mode in (("fp32", 0), ("bf16", 1), ("bf16", 3), ("bf16", 5)):
        m.precision, m.row_mode = prec, mode
        with torch.no_grad():
            out[(prec, mode)] = m(features=torch.from_numpy(x).cuda()).cpu().numpy()
    e = {k: float(np.abs(v - ref).max()) for k, v in out.items()}
    k = ("bf16", 1)
    idx = np.unravel_index(np.abs(out[k] - ref).argmax(), ref.shape)
    print(T, seed, {f"{a}{b}": f"{v:.2e}" for (a, b), v in e.items()}, "bits 1==3", np.array_equal(out[("bf16", 1)], out[("bf16", 3)]), "1==5", np.array_equal(out[("bf16", 1)], out[("bf16", 5)]),
          "worst at", idx, "ref", ref[idx], "ref64", ref64[idx], "bf16", out[k][idx], "prob err", float(np.abs(np.exp(out[k]) - np.exp(ref)).max()))
