"""The CPU oracle (oracle/savad_oracle.c) against golden vectors captured from the reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import oracle
from voice_activity_detection_amd.seeded import seeded_features, seeded_state_dict

# fp32 oracle vs fp32 torch-CPU reference: both carry ~2e-6 of fp32 noise at T=800
TOL = 2e-5


def test_pe_table(golden):
    pe = oracle.pe(801, 128)
    rows = golden["pe_rows"]
    # 1 ulp of sin/cos plus one 1-ulp frequency (torch's vectorised expf differs from the
    # correctly-rounded exp in 1 of 64 entries) amplified by t <= 800
    assert np.abs(pe[rows] - golden["pe_vals"]).max() < 5e-6


@pytest.mark.parametrize("tag,seed,shape", [("g1_out", 101, (4, 7, 80)), ("g2_out", 102, (2, 800, 80))])
def test_forward_golden(golden, state1234, tag, seed, shape):
    y = oracle.forward(state1234, seeded_features(seed, shape))
    assert y.shape == golden[tag].shape
    assert np.abs(y - golden[tag]).max() < TOL


def test_forward_golden_fp64_truth(golden, state1234):
    y = oracle.forward(state1234, seeded_features(102, (2, 800, 80)), acc64=True)
    assert np.abs(y - golden["g2_out"]).max() < TOL


def test_forward_normal_input(golden, state1234):
    y = oracle.forward(state1234, seeded_features(103, (3, 200, 80), kind="normal"))
    assert np.abs(y - golden["g2n_out"]).max() < TOL


@pytest.mark.parametrize("T", [1, 2, 5, 10, 11, 16, 17, 31, 32, 33, 63, 64, 65, 100, 799, 801])
def test_edge_lengths(golden, state1234, T):
    y = oracle.forward(state1234, seeded_features(400 + T, (3, T, 80)))
    assert np.abs(y - golden[f"g4_T{T}"]).max() < TOL


def test_batch_edges(golden, state1234):
    assert np.abs(oracle.forward(state1234, seeded_features(77, (1, 7, 80))) - golden["g4_B1T7"]).max() < TOL
    y = oracle.forward(state1234, seeded_features(78, (1000, 7, 80)))
    assert np.abs(y[:8] - golden["g4_B1000T7_head"]).max() < TOL
    assert np.abs(y[-8:] - golden["g4_B1000T7_tail"]).max() < TOL
    assert np.abs(y.astype(np.float64).sum(axis=(1, 2)) - golden["g4_B1000T7_seqsum"]).max() < 14 * TOL
    assert oracle.forward(state1234, np.zeros((0, 7, 80), np.float32)).shape == (0, 7, 2)


def test_taps(golden, state1234):
    y, taps = oracle.forward(state1234, seeded_features(600, (2, 40, 80)), taps=True)
    assert np.abs(y - golden["g6_out"]).max() < TOL
    assert np.abs(taps["input_layer"] - golden["g6_input_layer"]).max() < TOL
    assert np.abs(taps["l0_ctx"] - golden["g6_l0_ctx"]).max() < TOL
    assert np.abs(taps["encoder_out"] - golden["g6_encoder_out"]).max() < TOL


def test_peaked_softmax(golden):
    st = seeded_state_dict(4321, gain=4.0)
    assert np.abs(oracle.forward(st, seeded_features(700, (2, 96, 80))) - golden["g7_out"]).max() < 5 * TOL
    assert np.abs(oracle.forward(st, seeded_features(701, (1, 800, 80))) - golden["g7_T800"]).max() < 5 * TOL


def test_other_model_size(golden):
    st = seeded_state_dict(88, feature_size=40, num_layers=2, d_model=128)
    assert np.abs(oracle.forward(st, seeded_features(800, (3, 50, 40))) - golden["g8_F40L2"]).max() < TOL


@pytest.mark.parametrize("F", [257, 13])
def test_odd_feature_sizes(golden, F):
    st = seeded_state_dict(90 + F, feature_size=F, num_layers=1, d_model=128)
    assert np.abs(oracle.forward(st, seeded_features(900 + F, (3, 37, F))) - golden[f"g9_F{F}"]).max() < TOL


def test_window_offsets():
    # vad/predictor.py:57-59,186-212 with the only shipped config (half 19, jump 9)
    assert oracle.window_offsets(19, 9).tolist() == [-19, -10, -1, 0, 1, 10, 19]
    assert len(oracle.window_offsets(19, 9)) == 2 * (19 - 1) // 9 + 3


@pytest.mark.parametrize("tag,n,seed", [("g5", 1022, 500), ("g5b", 2100, 501), ("g5c", 39, 502)])
def test_predictor_level(golden, state1234, tag, n, seed):
    feat = seeded_features(seed, (n, 80))
    probs, mean = oracle.predict_probabilities(state1234, feat)
    assert probs.shape == golden[f"{tag}_probs"].shape
    assert np.abs(probs - golden[f"{tag}_probs"]).max() < TOL
    assert np.abs(mean - golden[f"{tag}_mean"]).max() < TOL
    # unfilled slots are exactly 0.5 (softmax([0,0])) and are averaged in (predictor.py:238-258,:95)
    assert (probs == 0.5).sum() == (golden[f"{tag}_probs"] == 0.5).sum()


def test_g3_config2_checksums(golden, state1234):
    y = oracle.forward(state1234, seeded_features(0, (32, 800, 80)))
    assert np.abs(y[:2] - golden["g3_head"]).max() < TOL
    assert np.abs(y[-2:] - golden["g3_tail"]).max() < TOL
    assert np.abs(y.astype(np.float64).sum(axis=(1, 2)) - golden["g3_seqsum"]).max() < 1600 * TOL


def test_torch_port_matches_golden(golden, state1234):
    import torch

    from oracle import torch_port

    st = {k: torch.from_numpy(v) for k, v in state1234.items()}
    y = torch_port.forward(st, torch.from_numpy(seeded_features(102, (2, 800, 80)))).numpy()
    assert np.abs(y - golden["g2_out"]).max() < TOL
    y = torch_port.forward(st, torch.from_numpy(seeded_features(101, (4, 7, 80)))).numpy()
    assert np.abs(y - golden["g1_out"]).max() < TOL


def test_streaming_mode_definition(state1234):
    # windows [hop*w, hop*w+T), zero-padded tail; every frame covered by 1..T/hop windows
    feat = seeded_features(9, (2100, 80))
    probs, logp = oracle.predict_streaming(state1234, feat, 800, 400)
    assert logp.shape == (5, 800, 2) and probs.shape == (2100,)
    assert oracle.lib().savad_oracle_stream_window_count(3600 * 100 + 1, 800, 400) == 900  # 1 h of audio @100 fps
    # a frame covered by one window only reproduces that window's probability
    p0 = np.exp(logp[0, :400, 1])
    assert np.abs(probs[:400] - p0).max() < 1e-6
    # interior frames: mean of the two covering windows
    p01 = 0.5 * (np.exp(logp[0, 400:800, 1]) + np.exp(logp[1, 0:400, 1]))
    assert np.abs(probs[400:800] - p01).max() < 1e-6


def test_logmel_oracle_properties():
    """oracle/logmel.py restates librosa 0.8.0 defaults (parity UNPINNED: librosa is absent).  What CAN be
    checked without it: frame count, filterbank shape / Slaney normalisation, a pure tone lands in the
    right mel band, the silence floor is log(1e-6)."""
    from oracle import logmel

    M = logmel.mel_filterbank()
    assert M.shape == (80, 257) and M.dtype == np.float32 and (M >= 0).all()
    assert M[:, 0].max() == 0 and M[:, 256].max() == 0           # fmin = 0, fmax = sr/2: edge bins carry no weight
    centres = logmel.mel_to_hz(np.linspace(logmel.hz_to_mel(0.0), logmel.hz_to_mel(8000.0), 82))[1:-1]
    assert abs(centres[0] - 200.0 / 3 * (logmel.hz_to_mel(8000.0) / 81)) < 1e-6  # linear region below 1 kHz
    peak_bins = M.argmax(axis=1)
    assert (np.diff(peak_bins) >= 0).all()
    y = np.zeros(16000, np.float32)
    f0 = logmel.log_mel(y)
    assert f0.shape == (101, 80) and np.allclose(f0, np.log(1e-6), atol=1e-6)
    t = np.arange(16000) / 16000.0
    tone = logmel.log_mel(np.sin(2 * np.pi * 1000.0 * t).astype(np.float32))
    band = int(np.argmin(np.abs(centres - 1000.0)))
    assert abs(int(tone[50].argmax()) - band) <= 1
    assert logmel.frame_count(163414) == 1022  # the reference's test clip


# ---- log-mel row.  librosa itself cannot pin it (absent); these are the independent checks.
def test_logmel_mel_scale_known_answers():
    """The Slaney mel scale of oracle/logmel.py against the known answers printed in librosa 0.8's own docstrings
    (librosa.hz_to_mel(60) = 0.9, hz_to_mel([110, 220, 440]) = [1.65, 3.3, 6.6], mel_to_hz(3) = 200.,
    mel_to_hz([1..5]) = [66.667, 133.333, 200., 266.667, 333.333], mel_frequencies(n_mels=40) = [0., 85.317, 170.635,
    ..., 1024.856, ..., 10096.408, 11025.], filters.mel(22050, 2048)[0, 1] = 0.016) and its closed form."""
    from oracle import logmel

    assert abs(float(logmel.hz_to_mel(60.0)) - 0.9) < 1e-12
    assert np.allclose(logmel.hz_to_mel([110.0, 220.0, 440.0]), [1.65, 3.3, 6.6], atol=1e-12)
    assert abs(float(logmel.mel_to_hz(3.0)) - 200.0) < 1e-9
    assert np.allclose(np.round(logmel.mel_to_hz([1, 2, 3, 4, 5]), 3), [66.667, 133.333, 200.0, 266.667, 333.333])
    mf = logmel.mel_to_hz(np.linspace(logmel.hz_to_mel(0.0), logmel.hz_to_mel(11025.0), 40))
    doc = [0.0, 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49, 1024.856, 1119.114,
           1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393, 2470.47, 2697.686, 2945.799, 3216.731,
           3512.582, 3835.643, 4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009, 7754.107, 8467.272, 9246.028,
           10096.408, 11025.0]
    assert np.abs(np.round(mf, 3) - np.array(doc)).max() < 1.1e-3
    assert round(float(logmel.mel_filterbank(22050, 2048, 128)[0, 1]), 3) == 0.016
    # closed form: 15 mel = 1 kHz, one mel above that multiplies the frequency by 6.4^(1/27)
    assert abs(float(logmel.hz_to_mel(1000.0)) - 15.0) < 1e-12 and abs(float(logmel.mel_to_hz(42.0)) - 6400.0) < 1e-6


def test_logmel_filterbank_closed_form():
    """The 80-band filterbank the kernels use: area-normalised triangles between Slaney-spaced edges -- peak bins,
    support, unit area (up to the 31.25 Hz bin grid) and agreement with a from-the-definition construction."""
    import sys
    from pathlib import Path

    from oracle import logmel

    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    from make_golden_logmel import slaney_hz, slaney_mel, triangle_filterbank

    fb = logmel.mel_filterbank()
    assert fb.shape == (80, 257) and fb.dtype == np.float32 and (fb >= 0).all()
    assert np.abs(fb - triangle_filterbank()).max() < 1e-6
    edges = slaney_hz(np.linspace(0.0, slaney_mel(8000.0), 82))
    bins = np.arange(257) * 31.25
    for i in range(80):
        nz = np.nonzero(fb[i])[0]
        assert bins[nz[0]] > edges[i] - 1e-9 and bins[nz[-1]] < edges[i + 2] + 1e-9          # support = (lower edge, upper edge)
        assert abs(bins[int(np.argmax(fb[i]))] - edges[i + 1]) < 31.25                            # peak at the bin beside the centre frequency
    wide = (edges[2:] - edges[:-2]) > 8 * 31.25                                               # triangles resolved by the grid
    assert wide.sum() > 20 and np.abs((fb.sum(axis=1) * 31.25)[wide] - 1.0).max() < 0.02      # Slaney norm: unit area


def test_logmel_stft_matches_scipy_on_the_reference_clip():
    """oracle/logmel.py on the reference's test clip against the fixture derived from scipy.signal.stft + the
    from-the-definition filterbank (tests/golden/make_golden_logmel.py), and the STFT power of one frame directly."""
    from pathlib import Path

    from oracle import logmel
    from voice_activity_detection_amd.features import load_wav_mono16k

    here = Path(__file__).resolve().parent
    y = load_wav_mono16k(here / "golden" / "data" / "WhenTheWeatherIsFine" / "When_the_Weather_Is_Fine_12_4.wav")
    with np.load(here / "golden" / "golden_logmel.npz") as z:
        frames, want, p100 = z["frames"], z["logmel"], z["power_frame100"]
    got = logmel.log_mel(y)
    assert got.shape == (1022, 80) and want.shape == (128, 80)
    assert np.abs(got[frames] - want).max() < 5e-5  # float32 frames + complex64 rFFT vs float64 scipy: measured 1e-5
    yp = np.pad(y, 256, mode="reflect")
    win = np.zeros(512, np.float32)
    win[56:456] = logmel.hann_periodic(400)
    mine = np.abs(np.fft.rfft(yp[16000:16000 + 512].astype(np.float64) * win)) ** 2
    assert np.abs(mine - p100).max() < 1e-6 * p100.max()


def _dmodel_cases():
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    from dmodel_cases import CASES

    return CASES


@pytest.mark.parametrize("case", _dmodel_cases(), ids=lambda c: c[0])
def test_other_model_widths(golden_dmodel, case):
    """d_model != 128 (vad/models/model_factory.py:42-48 passes any width through): the oracle against the reference's outputs"""
    name, d_model, F, L, wseed, xseed, shape = case
    st = seeded_state_dict(wseed, feature_size=F, num_layers=L, d_model=d_model)
    y = oracle.forward(st, seeded_features(xseed, shape))
    assert y.shape == golden_dmodel[name].shape
    assert np.abs(y - golden_dmodel[name]).max() < TOL


def test_pe_table_other_width(golden_dmodel):
    assert np.abs(oracle.pe(60, 64) - golden_dmodel["d64_pe"]).max() < 5e-6
