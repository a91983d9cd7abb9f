"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol
include/savad.h declares; the nn.Module mirror has the reference's state_dict surface and refuses
to fall back to the CPU."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    from voice_activity_detection_amd import _lib, build

    build.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    from voice_activity_detection_amd import _lib

    header = (REPO / "include" / "savad.h").read_text()
    declared = set(re.findall(r"\b(savad_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_host_only_entry_points(lib):
    from oracle import oracle

    off = (ctypes.c_int32 * 64)()
    for half, jump in ((19, 9), (5, 1), (10, 3), (1, 1), (0, 4)):
        w = lib.savad_window_offsets(half, jump, off)
        assert list(off[:w]) == oracle.window_offsets(half, jump).tolist()
        assert w == len(np.arange(-half, 0, jump)) + 1 + len(np.arange(1, half + 1, jump))
    assert lib.savad_window_offsets(19, 9, None) == 2 * (19 - 1) // 9 + 3  # vad/predictor.py:57-59
    assert lib.savad_window_offsets(19, 0, None) == -1
    assert b"gfx950" in lib.savad_version()


def test_module_surface_matches_reference_state_dict():
    import torch

    from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, state_dict_spec

    m = SelfAttentiveVAD(feature_size=80, num_layers=3, d_model=128, dropout=0.5)
    keys = [k for k, _, _ in state_dict_spec()]
    assert list(m.state_dict().keys()) == keys and len(keys) == 54
    assert sum(p.numel() for p in m.parameters()) == 605698
    sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(1).items()}
    assert not any(m.load_state_dict(sd, strict=True))
    with pytest.raises(RuntimeError):
        m.load_state_dict({**sd, "bogus": torch.zeros(1)}, strict=True)


def test_no_cpu_fallback():
    import torch

    from voice_activity_detection_amd import SelfAttentiveVAD
    from voice_activity_detection_amd._lib import SavadError

    m = SelfAttentiveVAD(80, 3, 128, 0.5).eval()
    with pytest.raises(SavadError, match="no CPU fallback"):
        m(features=torch.zeros(1, 7, 80))


def test_product_never_imports_the_oracle():
    for p in (REPO / "voice_activity_detection_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".h", ".cpp"):
            assert "oracle" not in p.read_text().replace("# oracle-free", ""), p


def test_seeded_weights_are_stable():
    from voice_activity_detection_amd import seeded_features, seeded_state_dict

    sd = seeded_state_dict(1234)
    assert sd["input_layer.0.weight"].shape == (128, 80) and abs(float(np.abs(sd["input_layer.0.weight"]).max()) - 1 / np.sqrt(80)) < 1e-3
    a = seeded_features(0, (2, 3, 80))
    assert a.dtype == np.float32 and a.min() >= -13.8 and a.max() <= 4.2
    assert np.array_equal(a, seeded_features(0, (2, 3, 80)))


def test_roc_auc_matches_definition():
    from voice_activity_detection_amd.metrics import roc_auc

    rng = np.random.default_rng(0)
    for n in (10, 200):
        y = rng.integers(0, 2, n)
        y[0], y[1] = 0, 1
        s = np.round(rng.random(n), 1)  # many ties
        pos, neg = s[y == 1], s[y == 0]
        brute = ((pos[:, None] > neg[None, :]).sum() + 0.5 * (pos[:, None] == neg[None, :]).sum()) / (len(pos) * len(neg))
        assert abs(roc_auc(y, s) - brute) < 1e-12
    try:
        from sklearn.metrics import roc_auc_score  # what vad/evaluate.py:65 calls
    except Exception:
        return
    y = rng.integers(0, 2, 500)
    s = rng.random(500)
    assert abs(roc_auc(y, s) - roc_auc_score(y, s)) < 1e-12


def test_weight_changes_the_fast_walk_must_see():
    """the per-forward change detection walks cached `_parameters` dicts (model.py: _param_versions): a replaced SUBMODULE and a
    `p.data = tensor` (same Parameter object, same _version, new storage) must both register"""
    import torch

    from voice_activity_detection_amd import SelfAttentiveVAD

    m = SelfAttentiveVAD(80, 1, 128, 0.5)
    v0 = m._param_versions()
    assert m._param_dicts is not None and m._param_versions() == v0
    m.classifier.weight.data = torch.zeros_like(m.classifier.weight)
    v1 = m._param_versions()
    assert v1 != v0 and v1[0] == v0[0] and v1[1] == v0[1]          # identity and version unchanged: only the storage moved
    m.classifier = torch.nn.Linear(128, 2)
    assert m._param_dicts is None and m._param_versions() != v1     # the walk is rebuilt and sees the new module's parameters


def test_module_copies_and_pickles_drop_runtime_state(state1234):
    """copy.deepcopy / pickle of the module (the ctypes handle and the cached workspace are per-process runtime state and
    are recreated lazily), and the weight re-push triggers that do not need a GPU to check."""
    import copy
    import ctypes
    import pickle

    import torch

    from voice_activity_detection_amd import SelfAttentiveVAD

    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state1234.items()})
    m._handle = ctypes.c_void_p(1234)  # what a forward leaves behind (never dereferenced here)
    m._workspace, m._synced_versions = torch.zeros(4), m._param_versions()
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m)), copy.copy(m)):
        assert clone._handle is None and clone._workspace is None and clone._synced_versions is None
        assert all(torch.equal(a, b) for a, b in zip(clone.state_dict().values(), m.state_dict().values()))
        assert clone.precision == "fp32" and clone.feature_size == 80
    synced = m._synced_versions
    m.classifier.bias.data.add_(1.0)  # invisible to (data_ptr, _version) ...
    assert m._param_versions() == synced
    m.eval()                          # ... so mode switches force a re-push
    assert m._synced_versions is None
    m._synced_versions = m._param_versions()
    m.eval()                          # ... but a call that changes nothing (the predictor's, before every batch) must not
    assert m._synced_versions is not None
    with torch.no_grad():
        m.classifier.bias.add_(1.0)   # ordinary in-place edits are seen
    assert m._param_versions() != m._synced_versions
    # children follow the parent's mode even when the parent's own flag did not change
    m.encoder.training = True
    m.eval()
    assert not m.encoder.training and m._synced_versions is not None
    # nn.DataParallel replicas would SHARE the native handle (shallow __dict__ copy, no parameters): refused, clearly
    from voice_activity_detection_amd._lib import SavadError
    with pytest.raises(SavadError, match="DataParallel"):
        m._replicate_for_data_parallel()
    assert m._handle.value == 1234  # untouched
    m._handle = None  # nothing real to destroy


def test_checkpoints_load_without_arbitrary_unpickling(tmp_path, state1234, monkeypatch):
    """A reference-format checkpoint (tensors, containers, numpy-scalar metrics) loads with weights_only=True; a file
    that needs arbitrary unpickling is refused unless the caller opts in (trust_checkpoint / SAVAD_TRUST_CHECKPOINT)."""
    import numpy as np
    import torch

    from tests.conftest import write_reference_checkpoint
    from voice_activity_detection_amd.predictor import VADFromScratchPredictor as P

    monkeypatch.delenv("SAVAD_TRUST_CHECKPOINT", raising=False)
    write_reference_checkpoint(tmp_path / "ok.checkpoint", state1234)
    ck = P._load_checkpoint(tmp_path / "ok.checkpoint", False)
    assert isinstance(ck["metrics"]["val_auc"], np.floating) and set(ck["state_dict"]) == set(state1234)
    torch.save({"config": _Opaque(), "state_dict": {}}, tmp_path / "opaque.checkpoint")
    with pytest.raises(RuntimeError, match="weights_only"):
        P._load_checkpoint(tmp_path / "opaque.checkpoint", False)
    assert isinstance(P._load_checkpoint(tmp_path / "opaque.checkpoint", True)["config"], _Opaque)
    monkeypatch.setenv("SAVAD_TRUST_CHECKPOINT", "1")
    assert isinstance(P._load_checkpoint(tmp_path / "opaque.checkpoint", False)["config"], _Opaque)
    # only an unlisted global falls through to the full unpickle: a missing or truncated file is reported as what it is, trusted or not
    with pytest.raises(FileNotFoundError):
        P._load_checkpoint(tmp_path / "nowhere.checkpoint", True)
    (tmp_path / "cut.checkpoint").write_bytes((tmp_path / "ok.checkpoint").read_bytes()[:100])
    with pytest.raises(Exception) as info:
        P._load_checkpoint(tmp_path / "cut.checkpoint", True)
    assert "weights_only" not in str(info.value)


class _Opaque:
    """stands for a pickled config object (the reference stores an OmegaConf container)"""


@pytest.fixture(scope="module")
def code_object(tmp_path_factory):
    """the device code compiled ONCE for the tests that read it (metadata notes, disassembly)"""
    import shutil
    import sys

    if not (shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists()) or not Path("/opt/rocm/lib/llvm/bin/llvm-readelf").exists():
        pytest.skip("no ROCm toolchain here")
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "scripts"))
    from kernel_resources import compile_code_object

    return compile_code_object(tmp_path_factory.mktemp("co") / "savad.co")


def test_no_kernel_spills(code_object):
    """No kernel of the code object may spill VGPRs or use scratch memory (row_kernel_m did in round 2: 24 spilled
    registers, 100 bytes of scratch).  Parsed from the code object's metadata notes (llvm-readelf --notes)."""
    from kernel_resources import kernel_resources

    res = kernel_resources(co=code_object)
    assert len(res) >= 30
    for name, r in res.items():
        assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, (name, r)
        assert r["vgpr_count"] + 0 <= 512, (name, r)


def test_generated_instruction_stream_is_current():
    """csrc/savad_attn_pw_bf16*.inc are generated: the committed files must be what scripts/gen_attn_pw.py writes"""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, str(REPO / "scripts" / "gen_attn_pw.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_lds_dma_owns_m0(code_object):
    """The LDS-DMA statements of the bf16 kernels set M0 without saving it (savad_kernels_bf16.h, Ring::dma1k; the generated
    stream of savad_attn_pw_bf16.h): legal only while nothing else in those kernels touches M0.  Checked where it can be
    checked: in the disassembly."""
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not Path(objdump).exists():
        pytest.skip("no llvm-objdump here")
    dis = subprocess.run([objdump, "-d", str(code_object)], check=True, capture_output=True, text=True).stdout
    parts = re.split(r"\n[0-9a-f]+ <([^>]+)>:\n", dis)
    seen = 0
    for name, body in zip(parts[1::2], parts[2::2]):
        if "2bf" not in name:  # namespace savad::bf
            continue
        for line in body.splitlines():
            if re.search(r"\bm0\b", line):
                seen += 1
                assert "s_add_u32 m0" in line or "s_mov_b32 m0" in line, (name, line.strip())
    assert seen >= 100


def build_c_client(out_dir):
    """tests/abi_client.c: a plain C99 program written against include/savad.h only, linked against libsavad.so"""
    import shutil
    import subprocess

    from voice_activity_detection_amd import build

    if not shutil.which("gcc"):
        pytest.skip("no gcc here")
    lib = build.build()
    exe = Path(out_dir) / "abi_client"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", f"-I{REPO / 'include'}", str(REPO / "tests" / "abi_client.c"), "-o", str(exe),
                    f"-L{lib.parent}", "-lsavad", "-ldl", "-lm", f"-Wl,-rpath,{lib.parent}"], check=True)
    return exe


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """the drop-in boundary is a C ABI: include/savad.h compiles as pedantic C99 and as C++11, and a C program that uses
    nothing else links against the library (it runs on the GPU box: tests/test_gpu_parity.py::test_c_client_of_the_abi)"""
    import shutil
    import subprocess

    if not shutil.which("gcc"):
        pytest.skip("no gcc here")
    hdr = str(REPO / "include" / "savad.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr], check=True)
    assert build_c_client(tmp_path).exists()


def test_pipelined_wrapper_host_logic(state1234):
    """PipelinedVAD without a GPU: replicas share the parameters and never the runtime state; no CPU fallback either"""
    import torch

    from voice_activity_detection_amd import PipelinedVAD, SelfAttentiveVAD
    from voice_activity_detection_amd._lib import SavadError

    m = SelfAttentiveVAD(80, 3, 128, 0.5).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state1234.items()})
    pipe = PipelinedVAD(m)
    assert pipe.depth == 3 and len(pipe._replicas) == 3 and all(r is not m for r in pipe._replicas)
    assert all(r.classifier.weight is m.classifier.weight and r._handle is None and r._workspace is None for r in pipe._replicas)
    m.precision = "bf16"
    assert PipelinedVAD(m).depth == 2 and PipelinedVAD(m, depth=1)._replicas == [m]
    with pytest.raises(ValueError):
        PipelinedVAD(m, depth=0)
    with pytest.raises(ValueError):
        pipe.set_active(4)
    with pytest.raises(SavadError, match="no CPU fallback"):
        pipe.submit(torch.zeros(1, 7, 80))
    pipe.join()  # nothing in flight: a no-op


def test_reference_mode_host_chunk_plan_covers_every_window():
    """VADFromScratchPredictor.host_chunk_plan (the chunking of predict_audio_host): for every output frame of every chunk, the windows
    the whole recording gives that frame (vad/predictor.py:238-258: window i of the N - 2 half, centred at half + i, reaches frame
    half + i + off) are exactly the windows of the chunk's feature slice that reach it -- nothing missing at a seam, nothing extra at a
    clipped end --, every slice starts on a multiple of 32 // W windows, chunks tile [0, N) and no tail is shorter than half a chunk."""
    from voice_activity_detection_amd.predictor import VADFromScratchPredictor, window_offsets

    for half, jump in ((19, 9), (3, 1), (8, 4)):
        off = [int(o) for o in window_offsets(half, jump)]
        W = len(off)
        G = max(32 // W, 1)
        for N, per in ((1001, 300), (1001, 250), (5000, 4096), (77, 1000), (2 * half, 100), (2 * half + 1, 100), (640, 160), (1, 10)):
            plan = VADFromScratchPredictor.host_chunk_plan(N, half, W, per)
            assert plan[0][0] == 0 and plan[-1][1] == N and all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
            if len(plan) > 1:
                assert plan[-1][1] - plan[-1][0] >= max(per, 4 * half) // 2
            for f0, f1, g0, g1 in plan:
                assert 0 <= g0 <= f0 < f1 <= g1 <= N and g0 % G == 0
                n_local = g1 - g0
                for n in {f0, f0 + 1, (f0 + f1) // 2, f1 - 1} & set(range(f0, f1)):
                    whole = {j for j, o in enumerate(off) if 0 <= n - half - o < N - 2 * half}
                    local = {j for j, o in enumerate(off) if 0 <= (n - g0) - half - o < n_local - 2 * half}
                    assert whole == local, (half, N, per, n, whole, local)


def test_audio_loader_reads_what_the_stdlib_can_decode(tmp_path):
    """features.load_wav_mono16k (AudioData.load, vad/data_models/audio_data.py:18-34, without soundfile): integer PCM WAV of every width,
    IEEE-float and WAVE_FORMAT_EXTENSIBLE WAVs (which the stdlib `wave` refuses), AIFF and Sun AU (big-endian PCM), headerless .pcm,
    stereo averaged to mono; a compressed or foreign container is refused with a clear message."""
    import struct
    import warnings
    import wave

    from voice_activity_detection_amd.features import load_wav_mono16k

    rng = np.random.default_rng(5)
    x = np.clip(rng.standard_normal(4000) * 0.2, -0.99, 0.99).astype(np.float32)
    i16 = np.round(x * 32767).astype(np.int16)
    want16 = i16.astype(np.float32) / 32768.0

    def riff(fmt_body, data):
        body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body + b"LIST" + struct.pack("<I", 4) + b"abcd" + \
               b"data" + struct.pack("<I", len(data)) + data
        return b"RIFF" + struct.pack("<I", len(body)) + body

    with wave.open(str(tmp_path / "a16.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(i16.tobytes())
    assert np.array_equal(load_wav_mono16k(tmp_path / "a16.wav"), want16)
    (tmp_path / "f32.wav").write_bytes(riff(struct.pack("<HHIIHH", 3, 1, 16000, 64000, 4, 32), x.astype("<f4").tobytes()))
    assert np.array_equal(load_wav_mono16k(tmp_path / "f32.wav"), x)
    (tmp_path / "f64.wav").write_bytes(riff(struct.pack("<HHIIHH", 3, 1, 16000, 128000, 8, 64), x.astype("<f8").tobytes()))
    assert np.array_equal(load_wav_mono16k(tmp_path / "f64.wav"), x)
    ext = struct.pack("<HHIIHH", 0xFFFE, 2, 16000, 64000, 4, 16) + struct.pack("<HHI", 22, 16, 3) + struct.pack("<H", 1) + b"\\x00" * 14
    stereo = np.stack([i16, -i16], axis=1)
    (tmp_path / "ext.wav").write_bytes(riff(ext, stereo.astype("<i2").tobytes()))
    assert np.abs(load_wav_mono16k(tmp_path / "ext.wav")).max() < 1e-6     # L + (-L) averages to (almost) nothing
    i24 = (np.round(x * 8388607).astype(np.int32))
    b24 = np.stack([(i24 >> s) & 0xFF for s in (0, 8, 16)], axis=1).astype(np.uint8).tobytes()
    (tmp_path / "a24.wav").write_bytes(riff(struct.pack("<HHIIHH", 1, 1, 16000, 48000, 3, 24), b24))
    assert np.abs(load_wav_mono16k(tmp_path / "a24.wav") - i24.astype(np.float32) / 8388608.0).max() == 0
    i16.tofile(tmp_path / "raw.pcm")
    assert np.array_equal(load_wav_mono16k(tmp_path / "raw.pcm"), want16)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        import aifc
        import sunau
    with aifc.open(str(tmp_path / "a.aiff"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(i16.astype(">i2").tobytes())
    assert np.array_equal(load_wav_mono16k(tmp_path / "a.aiff"), want16)
    with sunau.open(str(tmp_path / "a.au"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.setcomptype("NONE", "not compressed"); w.writeframes(i16.astype(">i2").tobytes())
    assert np.array_equal(load_wav_mono16k(tmp_path / "a.au"), want16)
    (tmp_path / "x.wav").write_bytes(riff(struct.pack("<HHIIHH", 85, 1, 16000, 4000, 1, 0), b"\\x00" * 64))   # MP3-in-WAV
    with pytest.raises(ValueError, match="format code 85"):
        load_wav_mono16k(tmp_path / "x.wav")
    (tmp_path / "y.wav").write_bytes(b"fLaC" + b"\\x00" * 64)
    with pytest.raises(ValueError, match="not a RIFF/WAVE"):
        load_wav_mono16k(tmp_path / "y.wav")
