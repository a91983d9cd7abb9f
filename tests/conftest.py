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


REFERENCE_CONFIG = {  # the reference's shipped configuration (tests/configs/vad/train_config.yaml:7-28), as OmegaConf.to_container gives it
    "model": {"name": "self-attention", "self_attention": {"num_layers": 3, "d_model": 128, "dropout": 0.5}},
    "context_resolution": {"context_window_half_frames": 19, "context_window_jump_frames": 9, "context_window_shift_frames": 39},
    "feature_extractor": {"transform": {"name": "log-mel", "n_fft": 512, "hop_ms": 10, "window_ms": 25, "n_mels": 80, "n_mfcc": None},
                          "temporal_differences": False, "stack_differences": False, "cachedir": None},
}


def write_reference_checkpoint(path, state, config=None):
    """A checkpoint with the FULL key set the reference's ModelCheckpointer writes
    (vad/training/checkpointers/model_checkpointer.py:97-110): numpy-scalar metrics (np.mean / roc_auc_score in
    vad/model_runner.py:72-89), epoch, global_step, optimizer / scheduler / grad-scaler state."""
    import copy

    import torch

    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in state.items()}
    ckpt = {
        "state_dict": sd, "epoch": 3, "global_step": 1234, "monitor_metric": "val_auc",
        "metrics": {"val_auc": np.float64(0.91), "val_accuracy": np.float32(0.88), "val_loss": np.mean(np.array([0.3, 0.4])),
                    "val_precision": np.float64(0.9), "val_recall": np.float64(0.8)},
        "config": copy.deepcopy(config if config is not None else REFERENCE_CONFIG),
        "optimizer_state_dict": {"state": {0: {"step": 7, "exp_avg": torch.zeros(3)}}, "param_groups": [{"lr": 1e-4, "params": [0]}]},
        "lr_scheduler_state_dict": {"last_epoch": 3}, "grad_scaler_state_dict": {"scale": 65536.0, "_growth_tracker": 0},
    }
    torch.save(ckpt, path)
    return path


@pytest.fixture(scope="session")
def golden_dmodel():
    """reference outputs for model widths other than 128 (tests/golden/make_golden_dmodel.py)"""
    return np.load(Path(__file__).resolve().parent / "golden" / "golden_dmodel.npz")
