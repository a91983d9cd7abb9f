"""Dev check of precision "fp32s" at T <= 32: the single-launch kernel (row_mode 4), the per-layer launches (row_mode 1) and the automatic
choice against the reference goldens; timings for the 7-frame-window shapes and the reference-mode hour."""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from voice_activity_detection_amd import SelfAttentiveVAD, VADFromScratchPredictor, seeded_features, seeded_state_dict  # noqa: E402

golden = np.load(REPO / "tests" / "golden" / "golden.npz")
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


cases = [("g1_out", 101, (4, 7, 80)), ("g4_B1T7", 77, (1, 7, 80))] + [(f"g4_T{T}", 400 + T, (3, T, 80)) for T in (1, 2, 5, 10, 11, 16, 17, 31, 32)]
for tag, seed, shape in cases:
    x = seeded_features(seed, shape)
    print(f"{tag:10s}", " ".join(f"rm{rm} {np.abs(run(x, 'fp32s', rm) - golden[tag]).max():.2e}" for rm in (0, 1, 7, 8)), flush=True)
x = seeded_features(78, (1000, 7, 80))
for rm in (0, 1, 7, 8):
    y = run(x, "fp32s", rm)
    print("B1000T7 rm", rm, np.abs(y[:8] - golden["g4_B1000T7_head"]).max(), np.abs(y[-8:] - golden["g4_B1000T7_tail"]).max(),
          np.abs(y.astype(np.float64).sum(axis=(1, 2)) - golden["g4_B1000T7_seqsum"]).max())
for tag, n, seed in (("g5", 1022, 500), ("g5b", 2100, 501), ("g5c", 39, 502)):
    feat = seeded_features(seed, (n, 80))
    for rm in (0, 7, 8):
        m.precision, m.row_mode = "fp32s", rm
        probs = VADFromScratchPredictor(m, "cuda").predict_probabilities(feat)
        print(tag, "rm", rm, np.abs(probs - golden[f"{tag}_probs"]).max(), (probs == 0.5).sum() == (golden[f"{tag}_probs"] == 0.5).sum())
    m.row_mode = 0

for B in (100, 1000, 2000, 3000, 4000, 6000, 16384, 65536):
    x = torch.from_numpy(seeded_features(5, (B, 7, 80))).to("cuda")
    line = f"[{B},7,80]"
    for prec, rm in (("fp32", 0), ("fp32s", 0), ("fp32s", 7), ("fp32s", 8), ("fp32s", 1), ("bf16", 0)):
        m.precision, m.row_mode = prec, rm
        with torch.no_grad():
            for _ in range(3):
                m(features=x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                m(features=x)
            e1.record()
            torch.cuda.synchronize()
        line += f"  {prec}/rm{rm} {e0.elapsed_time(e1) / n:.4f} ms"
    m.row_mode = 0
    print(line, flush=True)

N = 360_001
feat = torch.from_numpy(seeded_features(4242, (N, 80))).cuda()
for prec in ("fp32", "fp32s", "bf16"):
    m.precision = prec
    pred = VADFromScratchPredictor(m, "cuda")
    for _ in range(2):
        p1, m1 = pred.predict_probabilities_device(feat)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        p1, m1 = pred.predict_probabilities_device(feat)
    e1.record()
    torch.cuda.synchronize()
    if prec == "fp32":
        ref = p1.clone()
    print(f"reference-mode hour {prec}: {e0.elapsed_time(e1) / 3:.3f} ms   max |dp| vs fp32 {float((p1 - ref).abs().max()):.2e}", flush=True)
