"""Post-processing row: the native host functions in libsavad.so against goldens produced
by the reference's own vad/postprocessing/*.py, vad/util/time_utils.py and VADFromScratchPredictor.predict
(tests/golden/make_golden_post.py).  Integer / index work: bit-exact."""
import json
from datetime import timedelta
from pathlib import Path

import numpy as np
import pytest

G = json.loads((Path(__file__).resolve().parent / "golden" / "golden_post.json").read_text())


def test_trim_voice_activity_bit_exact():
    from voice_activity_detection_amd.postprocessing import trim_voice_activity

    for c in G["trim"]:
        out = trim_voice_activity(np.array(c["pred"], dtype=bool), c["min_vally"], c["min_hill"], c["hang_before"], c["hang_over"])
        assert out.dtype == bool and out.astype(int).tolist() == c["out"], c
    # the reference's hang pass only runs when hang_before > 0 (it tests hang_before twice, trim.py:51)
    p = np.array([0, 0, 1, 1, 0, 0, 0], dtype=bool)
    assert trim_voice_activity(p, 0, 0, 0, 2).tolist() == p.tolist()


def test_frames_to_samples_and_segments():
    from voice_activity_detection_amd.postprocessing import convert_frames_to_samples, convert_samples_to_segments

    for c in G["frames_to_samples"]:
        out = convert_frames_to_samples(np.array(c["frames"]), c["sr"], c["hop"], c["win"])
        assert len(out) == c["n_out"] and out.dtype == np.float64
        assert out[:50].tolist() == c["head"] and out[-50:].tolist() == c["tail"] and out[::97].tolist() == c["every97"]
        assert out.sum() == c["sum"]
    for c in G["segments"]:
        s = convert_frames_to_samples(np.array(c["frames"]), c["sr"], c["hop"], c["win"])
        segs = convert_samples_to_segments(s, c["sr"])
        got = [[int(a / timedelta(microseconds=1)), int(b / timedelta(microseconds=1))] for a, b in segs]
        assert got == c["segments_us"]
    assert convert_samples_to_segments(np.zeros(0)) == []


def test_optimal_split_bit_exact():
    from voice_activity_detection_amd.postprocessing import optimal_split_voice_activity

    for c in G["split"]:
        out = optimal_split_voice_activity(np.array(c["pred"], dtype=float), np.array(c["probs"]), c["max_s"], c["sr"])
        assert out.astype(int).tolist() == c["out"]


def test_timecode_format_and_json_roundtrip(tmp_path):
    from voice_activity_detection_amd.data_models import Activity, VoiceActivity, format_timedelta_to_timecode

    for us, text in G["timecode"]:
        assert format_timedelta_to_timecode(timedelta(microseconds=us)) == text
    va = VoiceActivity(timedelta(seconds=10.213), [Activity(timedelta(seconds=0.525), timedelta(seconds=1.225))], 100, [0.16, 0.5])
    va.save(tmp_path / "va.json")
    data = json.loads((tmp_path / "va.json").read_text())
    assert list(data) == ["version", "duration", "activities", "probs_sample_rate", "probs"] and data["version"] == "v0.3"
    assert data["duration"] == "00:00:10.213" and data["activities"] == [{"start": "00:00:00.525", "end": "00:00:01.225"}]
    assert VoiceActivity.load(tmp_path / "va.json") == va
    # the reference's own sample file parses (format example of JSON v0.3: tests/data/WhenTheWeatherIsFine/voice_activity.json)
    ref_file = Path("/root/reference/tests/data/WhenTheWeatherIsFine/voice_activity.json")
    if ref_file.exists():
        assert len(VoiceActivity.load(ref_file).activities) == 5


def _audio(case, seconds):
    n = int(seconds * 16000)
    arng = np.random.default_rng(700 + case)
    t = np.arange(n) / 16000.0
    env = (np.sin(2 * np.pi * 0.7 * t + case) > 0).astype(np.float32)
    return (env * 0.3 * np.sin(2 * np.pi * (200 + 50 * case) * t) + 0.02 * arng.standard_normal(n)).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [0, 1, 2])
def test_predict_matches_reference_predict(case, state1234):
    """VADFromScratchPredictor.predict (vad/predictor.py:77-157): chunking, threshold, trim, frame->sample,
    optimal split, segments, merge, JSON v0.3 -- against the reference's own predict() on the same weights.
    Features are fed from the CPU log-mel oracle (as the golden run did) so that everything AFTER the
    feature matrix is pinned; a second run uses the GPU log-mel front-end end to end."""
    import torch

    from oracle import logmel
    from voice_activity_detection_amd import SelfAttentiveVAD, VADFromScratchPredictor, VADPredictParameters

    c = G["predict"][case]
    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state1234.items()})
    pred = VADFromScratchPredictor(m.cuda().eval(), "cuda")
    audio = _audio(case, c["seconds"])
    params = VADPredictParameters(**c["params"])
    va = pred.predict(audio, params, features_fn=logmel.log_mel)
    got, ref = va.to_json(), c["json"]
    assert got["duration"] == ref["duration"] and got["probs_sample_rate"] == ref["probs_sample_rate"]
    assert got["activities"] == ref["activities"]
    if ref["probs"] is not None:
        assert len(got["probs"]) == len(ref["probs"]) and np.abs(np.array(got["probs"]) - np.array(ref["probs"])).max() < 1e-5
    # end to end on the GPU front-end: same segmentation up to frames whose probability sits on the threshold
    va2 = pred.predict(audio, params).to_json()
    assert va2["duration"] == ref["duration"] and abs(len(va2["activities"]) - len(ref["activities"])) <= 1


@pytest.mark.gpu
def test_predict_cli_writes_json_v03(tmp_path, state1234):
    """`python -m voice_activity_detection_amd predict AUDIO CKPT --output-path ...` (the reference's main.py predict):
    a checkpoint in the reference's format ({"config", "state_dict"}: vad/training/model_checkpointer.py:97-110,
    vad/predictor.py:266-278) and a 16 kHz WAV in, JSON v0.3 out; mirrors tests/test_predict.py:12-30 of the reference."""
    import wave

    import torch

    from voice_activity_detection_amd.__main__ import main
    from voice_activity_detection_amd.data_models import VoiceActivity

    from tests.conftest import write_reference_checkpoint

    write_reference_checkpoint(tmp_path / "m.checkpoint", state1234)
    pcm = (_audio(0, 6.0) * 20000).astype(np.int16)
    with wave.open(str(tmp_path / "a.wav"), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(pcm.tobytes())
    out = tmp_path / "out" / "va.json"
    assert main(["predict", str(tmp_path / "a.wav"), str(tmp_path / "m.checkpoint"), "--output-path", str(out),
                 "--return-probs", "--probs-sample-rate", "100"]) == 0
    data = json.loads(out.read_text())
    assert data["version"] == "v0.3" and data["duration"] == "00:00:06.000" and data["probs_sample_rate"] == 100
    assert len(data["probs"]) == 602 and len(VoiceActivity.load(out).activities) == len(data["activities"])


def test_audio_loading_formats_and_resampling(tmp_path):
    """AudioData.load restated (vad/data_models/audio_data.py:18-34): .pcm, PCM WAV of several widths / channel counts,
    and other sample rates -> float32 mono @16 kHz.  (The resampler's sample values: test_resampler_against_the_resampy_restatement.)"""
    import wave

    import numpy as np

    from voice_activity_detection_amd.features import load_wav_mono16k, resample_to_16k

    t = np.arange(16000) / 16000.0
    tone = 0.5 * np.sin(2 * np.pi * 440.0 * t)
    pcm16 = np.round(tone * 32767).astype("<i2")
    pcm16.tofile(tmp_path / "a.pcm")
    a = load_wav_mono16k(tmp_path / "a.pcm")
    assert a.dtype == np.float32 and a.shape == (16000,) and np.array_equal(a, pcm16.astype(np.float32) / 32768.0)

    def write(name, data, width, ch, rate):
        with wave.open(str(tmp_path / name), "wb") as w:
            w.setnchannels(ch)
            w.setsampwidth(width)
            w.setframerate(rate)
            w.writeframes(data)

    write("mono16.wav", pcm16.tobytes(), 2, 1, 16000)
    assert np.array_equal(load_wav_mono16k(tmp_path / "mono16.wav"), a)
    stereo = np.stack([pcm16, np.zeros_like(pcm16)], axis=1)  # channels are averaged (audio_data.py:26)
    write("stereo16.wav", stereo.astype("<i2").tobytes(), 2, 2, 16000)
    assert np.allclose(load_wav_mono16k(tmp_path / "stereo16.wav"), a / 2, atol=1e-7)
    write("u8.wav", np.round(tone * 127 + 128).astype(np.uint8).tobytes(), 1, 1, 16000)
    assert np.abs(load_wav_mono16k(tmp_path / "u8.wav") - tone).max() < 1.0 / 100
    v24 = np.round(tone * 8388607).astype(np.int32)
    b24 = np.stack([v24 & 255, (v24 >> 8) & 255, (v24 >> 16) & 255], axis=1).astype(np.uint8)
    write("s24.wav", b24.tobytes(), 3, 1, 16000)
    assert np.abs(load_wav_mono16k(tmp_path / "s24.wav") - tone).max() < 1e-6
    write("s32.wav", np.round(tone * 2147483647).astype("<i4").tobytes(), 4, 1, 16000)
    assert np.abs(load_wav_mono16k(tmp_path / "s32.wav") - tone).max() < 1e-6
    # 44.1 kHz and 8 kHz sources: length as ceil(n * 16000 / rate), the 440 Hz tone survives with its amplitude
    for rate in (44100, 48000, 8000, 22050):
        n = rate  # one second
        src = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(n) / rate)
        write(f"r{rate}.wav", np.round(src * 32767).astype("<i2").tobytes(), 2, 1, rate)
        y = load_wav_mono16k(tmp_path / f"r{rate}.wav")
        assert y.dtype == np.float32 and len(y) == 16000
        spec = np.abs(np.fft.rfft(y[2000:14000] * np.hanning(12000)))
        assert abs(np.argmax(spec) * 16000 / 12000 - 440.0) < 2.0
        assert abs(np.abs(y[2000:14000]).max() - 0.5) < 0.01
    assert len(resample_to_16k(np.zeros(44101, np.float32), 44100)) == int(np.ceil(44101 * 16000 / 44100))


def test_resampler_against_the_resampy_restatement():
    """features.resample_to_16k (numpy, vectorised) against oracle/resample.py, the loop-by-loop restatement of what the reference
    calls -- librosa.resample(..., res_type="kaiser_fast") = resampy's windowed-sinc interpolation + fix_length
    (vad/data_models/audio_data.py:27-30) -- on 8 kHz, 44.1 kHz, 48 kHz, 22.05 kHz and an awkward-ratio source: chirp + noise.
    Both are parity-UNPINNED against resampy itself (absent here); as an independent check the band-limited part of the signal
    must also agree with scipy's polyphase resampler (a different filter) away from the edges."""
    import numpy as np
    from scipy.signal import resample_poly

    from oracle import resample as ref
    from voice_activity_detection_amd.features import resample_to_16k

    rng = np.random.default_rng(5)
    for rate, n in ((8000, 1500), (44100, 4000), (48000, 3001), (22050, 2000), (16001, 700), (44100, 1), (8000, 2)):
        t = np.arange(n) / rate
        x = (0.4 * np.sin(2 * np.pi * (300 + 4000 * t) * t) + 0.1 * rng.standard_normal(n)).astype(np.float32)
        want, got = ref.resample(x, rate), resample_to_16k(x, rate)
        assert got.dtype == np.float32 and got.shape == want.shape == (int(np.ceil(n * 16000 / rate)),)
        assert np.abs(got - want).max() < 2e-6, (rate, np.abs(got - want).max())
    # a tone well inside both pass bands: the two resamplers agree to a fraction of a percent in the middle of the signal
    for rate, up, down in ((44100, 160, 441), (8000, 2, 1), (48000, 1, 3)):
        n = rate // 2
        x = (0.5 * np.sin(2 * np.pi * 700.0 * np.arange(n) / rate)).astype(np.float32)
        a, b = resample_to_16k(x, rate), resample_poly(x.astype(np.float64), up, down)
        m = min(len(a), len(b))
        assert np.abs(a[400:m - 400] - b[400:m - 400]).max() < 5e-3


def test_reference_clip_fixture_host_side():
    """The reference's own test clip (tests/golden/data): WAV loader, label expansion and frame geometry."""
    import numpy as np

    from oracle import logmel
    from voice_activity_detection_amd.data_models import VoiceActivity
    from voice_activity_detection_amd.evaluate import load_data_list
    from voice_activity_detection_amd.features import load_wav_mono16k

    root = Path(__file__).resolve().parent / "golden" / "data"
    pairs = load_data_list(root / "eval_list.jsonl")
    assert len(pairs) == 1
    audio = load_wav_mono16k(root / pairs[0]["audio_path"])
    assert audio.dtype == np.float32 and audio.shape == (163414,) and np.abs(audio).max() <= 1.0
    assert logmel.frame_count(len(audio)) == 1022  # -> 984 windows of 7 frames (vad/predictor.py:169)
    va = VoiceActivity.load(root / pairs[0]["voice_activity_path"])
    labels = va.to_labels(100)
    assert len(va.activities) == 5 and len(labels) == 1021 and 0 < labels.mean() < 1
    assert VoiceActivity.from_json(va.to_json()).to_json() == va.to_json()  # JSON v0.3 round trip


def test_voice_activity_formats_match_reference():
    """Every VoiceActivity reader / writer (vad/data_models/voice_activity.py:49-237: JSON v0.1, v0.2 timecode and
    millisecond, v0.3; milliseconds v0.2, v0.3) against documents written and re-read by the reference's own class
    (tests/golden/make_golden_formats.py)."""
    from voice_activity_detection_amd.data_models import VoiceActivity

    cases = json.loads((Path(__file__).resolve().parent / "golden" / "golden_formats.json").read_text())
    assert len(cases) == 8
    for c in cases:
        va = VoiceActivity.from_json(c["docs"]["json_v0.3"])
        assert va.to_json("v0.1") == c["docs"]["json_v0.1"] and va.to_json("v0.2") == c["docs"]["json_v0.2"]
        assert va.to_json() == c["docs"]["json_v0.3"]
        assert va.to_milliseconds("v0.2") == c["docs"]["ms_v0.2"] and va.to_milliseconds("v0.3") == c["docs"]["ms_v0.3"]
        for k, doc in c["docs"].items():
            back = VoiceActivity.from_milliseconds(doc) if k == "ms_v0.3" else VoiceActivity.from_json(doc)
            assert back.to_json() == c["read"][k]["as_v0.3"], k
            lab = back.to_labels(100)
            assert int(lab.sum()) == c["read"][k]["label_sum"] and len(lab) == c["read"][k]["n_labels"]
        assert VoiceActivity.from_milliseconds(c["docs"]["ms_v0.2"]).to_json() == c["read"]["ms_v0.2_via_from_milliseconds"]
    import pytest

    with pytest.raises(NotImplementedError):
        VoiceActivity.from_json({"version": "v9"})


def test_from_checkpoint_reads_reference_checkpoints_and_refuses_other_transforms(tmp_path, state1234):
    """vad/predictor.py:264-280 on a checkpoint with the reference ModelCheckpointer's full key set (numpy-scalar
    metrics, optimizer state ...: torch.load's weights_only default would refuse it), and a clear error -- not a late
    shape error or silently wrong features -- for feature configurations the device front-end does not implement."""
    import copy

    from tests.conftest import REFERENCE_CONFIG, write_reference_checkpoint
    from voice_activity_detection_amd.predictor import VADFromScratchPredictor

    p = VADFromScratchPredictor.from_checkpoint(write_reference_checkpoint(tmp_path / "ok.checkpoint", state1234), "cpu")
    assert (p.hop_ms, p.window_ms, p.context_window_frames) == (10, 25, 7)
    assert sorted(p.model.state_dict()) == sorted(state1234)
    # another model width / depth in the checkpoint's config (vad/models/model_factory.py:42-48 passes them through)
    from voice_activity_detection_amd.seeded import seeded_state_dict

    cfg64 = copy.deepcopy(REFERENCE_CONFIG)
    cfg64["model"]["self_attention"].update(num_layers=2, d_model=64)
    st64 = seeded_state_dict(64, num_layers=2, d_model=64)
    p64 = VADFromScratchPredictor.from_checkpoint(write_reference_checkpoint(tmp_path / "d64.checkpoint", st64, cfg64), "cpu")
    assert (p64.model.d_model, p64.model.num_layers) == (64, 2) and sorted(p64.model.state_dict()) == sorted(st64)
    assert all(np.array_equal(p64.model.state_dict()[k].numpy(), v) for k, v in st64.items())
    for path, value in ((("transform", "hop_ms"), 20), (("transform", "n_fft"), 1024), (("transform", "name"), "mfcc"),
                        (("transform", "n_mels"), 40), (("temporal_differences",), True),
                        (("silence_remover",), {"silence_threshold": 0.1})):
        cfg = copy.deepcopy(REFERENCE_CONFIG)
        node = cfg["feature_extractor"]
        for key in path[:-1]:
            node = node[key]
        node[path[-1]] = value
        with pytest.raises(NotImplementedError, match="unsupported"):
            VADFromScratchPredictor.from_checkpoint(write_reference_checkpoint(tmp_path / "bad.checkpoint", state1234, cfg), "cpu")
