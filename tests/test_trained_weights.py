"""Parity on TRAINED weights (round 5).  The reference ships no checkpoint (tests/checkpoints/vad/sample.checkpoint is in its
.MISSING_LARGE_BLOBS) and its own acceptance test -- AUC > 0.1 on a trained model, tests/test_evaluate.py:11-31 -- cannot run; every
AUC statement of rounds 1-4 was made on seeded random weights, whose AUC sits near chance and whose softmaxes are flat.
tests/golden/make_trained_weights.py fine-tunes the REFERENCE's own model in the build container (stock Adam, the reference's three
labelled recordings) and stores only the state_dict and the reference's outputs for the WhenTheWeatherIsFine clip; here the oracle
(CPU) and the HIP paths (GPU) are held against those: log-probs within north_star's 1e-4 in fp32, per-frame AUC within 1e-3 in
fp32 and bf16 on all three recordings, no saturation of the fp16-parked residual stream, peaked softmaxes included."""
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
DATA = HERE / "golden" / "data"
OFFSETS = np.array([-19, -10, -1, 0, 1, 10, 19])
FILES = [("JamakeSpeechSample/data/sample_93/audio_93.wav", "JamakeSpeechSample/data/sample_93/voice_activity_93.json"),
         ("JamakeSpeechSample/data/sample_95/audio_95.wav", "JamakeSpeechSample/data/sample_95/voice_activity_95.json"),
         ("WhenTheWeatherIsFine/When_the_Weather_Is_Fine_12_4.wav", "WhenTheWeatherIsFine/voice_activity.json")]


@pytest.fixture(scope="module")
def trained():
    with np.load(HERE / "golden" / "trained.npz") as z:
        state = {k[len("state/"):]: z[k] for k in z.files if k.startswith("state/")}
        return state, z["clip_logp"], z["clip_probs"], z["auc_ref"]


def recording(i):
    from oracle import logmel
    from voice_activity_detection_amd.data_models import VoiceActivity
    from voice_activity_detection_amd.features import load_wav_mono16k

    audio = load_wav_mono16k(DATA / FILES[i][0])
    labels = VoiceActivity.load(DATA / FILES[i][1]).to_labels(100)
    feat = logmel.log_mel(audio)
    n = min(len(labels), len(feat))
    return audio, feat[:n].astype(np.float32), labels[:n]


def clip_windows(feat):
    return feat[np.arange(19, len(feat) - 19)[:, None] + OFFSETS[None, :]]


def test_trained_weights_are_a_trained_model(trained):
    state, clip_logp, clip_probs, auc_ref = trained
    assert len(state) == 54 and sum(v.size for v in state.values()) == 605698   # the reference's state_dict
    assert auc_ref[0] > 0.99 and auc_ref[1] > 0.99 and auc_ref[2] > 0.8         # the bar of the round-4 review: AUC_ref > 0.8 on the clip
    p = np.exp(clip_logp[..., 1])
    assert (np.minimum(p, 1 - p) < 0.1).mean() > 0.6                            # confident decisions: not the near-chance scores of random weights


def test_oracle_matches_the_reference_on_trained_weights(trained):
    """the CPU oracle (the checker of every GPU test) against the reference's own outputs on the trained weights"""
    from oracle import oracle
    from voice_activity_detection_amd.metrics import roc_auc

    state, clip_logp, clip_probs, auc_ref = trained
    _, feat, labels = recording(2)
    got = oracle.forward(state, clip_windows(feat))
    assert got.shape == clip_logp.shape and np.abs(got - clip_logp).max() < 2e-5
    probs, mean = oracle.predict_probabilities(state, feat)
    assert np.abs(probs - clip_probs).max() < 2e-5 and np.array_equal(probs == 0.5, clip_probs == 0.5)
    assert abs(roc_auc(labels, mean) - auc_ref[2]) < 1e-6


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (torch.cuda.is_available() is False)")
    return torch


@pytest.fixture(scope="module")
def model(torch_cuda, trained):
    from voice_activity_detection_amd import SelfAttentiveVAD

    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict({k: torch_cuda.from_numpy(v) for k, v in trained[0].items()}, strict=True)
    return m.to("cuda").eval()


@pytest.mark.gpu
@pytest.mark.parametrize("precision,row_mode", [("fp32", 0), ("fp32s", 0), ("fp32s", 3), ("fp32s", 7), ("fp32s", 8), ("fp32s", 1)])
def test_fp32_logprobs_match_the_reference_on_trained_weights(torch_cuda, model, trained, precision, row_mode):
    """north_star: per-frame log-probs within 1e-4 of the reference's, here on a trained model's peaked outputs: the clip's windows
    as a batch (single launch), through the windowed predictor, and as 800-frame sequences against the oracle -- with the exact-fp32
    MFMA and with the split-bf16 operands of precision "fp32s" (automatic schedule; row_mode 3: the split-bf16 kernels at every size --
    the [8,800,80] batch is in the range the automatic schedule hands to the exact-fp32 kernels; both variants of its single launch; its
    per-layer launches)"""
    from oracle import oracle
    from voice_activity_detection_amd import VADFromScratchPredictor

    torch = torch_cuda
    state, clip_logp, clip_probs, _ = trained
    _, feat, _ = recording(2)
    model.precision, model.row_mode = precision, row_mode
    try:
        with torch.no_grad():
            y = model(features=torch.from_numpy(clip_windows(feat)).cuda()).cpu().numpy()
        assert np.abs(y - clip_logp).max() < 1e-4, np.abs(y - clip_logp).max()
        probs = VADFromScratchPredictor(model, "cuda").predict_probabilities(feat)
        assert np.abs(probs - clip_probs).max() < 1e-4 and np.array_equal(probs == 0.5, clip_probs == 0.5)
        _, feat93, _ = recording(0)
        x = np.ascontiguousarray(feat93[:6400].reshape(8, 800, 80))
        with torch.no_grad():
            y800 = model(features=torch.from_numpy(x).cuda()).cpu().numpy()
        assert np.abs(y800 - oracle.forward(state, x, threads=8)).max() < 1e-4
    finally:
        model.precision, model.row_mode = "fp32", 0


@pytest.mark.gpu
def test_auc_parity_on_trained_weights(torch_cuda, model, trained):
    """BASELINE.json: per-frame AUC equal to the reference's within 1e-3 -- on the reference's three labelled recordings with a
    model that separates them (AUC_ref 0.9997 / 0.9998 / 0.86), audio -> device log-mel -> predictor, fp32 and bf16 operands; the
    fp16-parked residual stream of the bf16 path must not saturate on trained activations."""
    from voice_activity_detection_amd import VADFromScratchPredictor
    from voice_activity_detection_amd.features import log_mel
    from voice_activity_detection_amd.metrics import roc_auc

    _, _, _, auc_ref = trained
    pred = VADFromScratchPredictor(model, "cuda")
    for i in range(3):
        audio, _, labels = recording(i)
        feat = log_mel(audio)
        n = len(labels)
        aucs = {}
        for prec in ("fp32", "fp32s", "bf16"):
            model.precision = prec
            try:
                aucs[prec] = roc_auc(labels, pred.predict_probabilities(feat).mean(axis=1)[:n])
            finally:
                model.precision = "fp32"
        assert all(abs(aucs[prec] - auc_ref[i]) < 1e-3 for prec in aucs), (i, auc_ref[i], aucs)
    assert model.residual_saturations() == 0


@pytest.mark.gpu
def test_bf16_on_trained_weights_long_sequences_and_streaming(torch_cuda, model, trained):
    """the bf16 kernels for long sequences (fused / persistent attention, row chain) and the streaming mode on trained weights:
    probabilities within 2x the measured bf16 error of the oracle, decisions equal, streaming AUC within 1e-3 of fp32's"""
    from oracle import oracle
    from voice_activity_detection_amd import StreamingPredictor
    from voice_activity_detection_amd.metrics import roc_auc

    torch = torch_cuda
    state = trained[0]
    _, feat, labels = recording(0)
    x = np.ascontiguousarray(feat[:6400].reshape(8, 800, 80))
    ref = oracle.forward(state, x, threads=8)
    model.precision = "bf16"
    try:
        with torch.no_grad():
            y = model(features=torch.from_numpy(x).cuda()).cpu().numpy()
        # A trained model's logits span -9 .. +9 and bf16 operands move them by up to 0.11 here / 0.37 on the 7-frame windows
        # (scripts/ubench/trained_bf16_probe.py: 2.3 % / 15 % of 1 + |logit|): the log-prob bound of the flat random-weight tests
        # does not apply.  What is held: probabilities within 2x the measured 5.0e-3, every decision equal, AUC below.
        assert np.isfinite(y).all() and np.abs(np.exp(y) - np.exp(ref)).max() < 1e-2, np.abs(np.exp(y) - np.exp(ref)).max()
        assert ((y[..., 1] > y[..., 0]) == (ref[..., 1] > ref[..., 0])).mean() > 0.999
        xw = clip_windows(feat)
        refw = oracle.forward(state, xw, threads=8)
        with torch.no_grad():
            yw = model(features=torch.from_numpy(xw).cuda()).cpu().numpy()
        assert np.abs(np.exp(yw) - np.exp(refw)).max() < 8e-2                    # measured 3.9e-2
        assert ((yw[..., 1] > yw[..., 0]) == (refw[..., 1] > refw[..., 0])).mean() > 0.9995   # measured 0.99998
        sp = StreamingPredictor(model, "cuda", 800, 400, max_batch=256)
        p16 = sp.predict_device(torch.from_numpy(feat).cuda()).cpu().numpy()
        model.precision = "fp32"
        p32 = sp.predict_device(torch.from_numpy(feat).cuda()).cpu().numpy()
    finally:
        model.precision = "fp32"
    assert abs(roc_auc(labels, p16[:len(labels)]) - roc_auc(labels, p32[:len(labels)])) < 1e-3
    assert model.residual_saturations() == 0
