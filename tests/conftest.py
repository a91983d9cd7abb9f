import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with np.load(REPO / "tests" / "golden" / "golden.npz") as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def state1234():
    from voice_activity_detection_amd.seeded import seeded_state_dict

    return seeded_state_dict(1234)
