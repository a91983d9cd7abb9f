#!/usr/bin/env python3
"""Golden vectors for model widths other than 128, from the REFERENCE implementation (read-only at /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dmodel.py

vad/models/self_attention.py:7-21 and vad/models/model_factory.py:42-48 accept any d_model (d_ff = 4 d_model, one head);
the reference's own config uses 128.  Only seeds and the reference's numeric outputs are stored (tests/golden/golden_dmodel.npz).
Cases are listed in CASES: (name, d_model, feature_size, num_layers, weight seed, input seed, input shape).
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

sys.path.insert(0, str(Path(__file__).resolve().parent))
from dmodel_cases import CASES  # noqa: E402
from voice_activity_detection_amd.seeded import seeded_features, seeded_state_dict  # noqa: E402


def main():
    from vad.models.self_attention import SelfAttentiveVAD  # reference, unmodified

    torch.manual_seed(0)
    torch.set_num_threads(8)
    g = {}
    for name, d_model, F, L, wseed, xseed, shape in CASES:
        state = seeded_state_dict(wseed, feature_size=F, num_layers=L, d_model=d_model)
        m = SelfAttentiveVAD(F, L, d_model, 0.5)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
        m.eval()
        with torch.no_grad():
            g[name] = m(features=torch.from_numpy(seeded_features(xseed, shape))).numpy()
        if name == "d64":   # rows of the reference's positional-encoding table at another width
            g["d64_pe"] = m.input_layer[1].build_positional_encoding(60).numpy()[0]
    out = Path(__file__).resolve().parent / "golden_dmodel.npz"
    np.savez_compressed(out, **g)
    print(f"wrote {out}: {len(g)} arrays, {out.stat().st_size / 1e3:.1f} kB")


if __name__ == "__main__":
    main()
