"""The factored-DFT log-mel kernel (csrc/savad_logmel.h: logmel_fft_kernel) on the CPU: the library's own A-operand tables
(savad_logmel_tables_host) driven through a numpy replay of the kernel's data flow -- the v_mfma_f32_32x32x2_f32 operand and
result layouts, the [k1][frame][re|im][n2] exchange, the fixed register pair -> mel block pattern -- against oracle/logmel.py.
What this pins without a GPU: the 512 = 32 x 16 factorisation, the folded twiddles, the bin ordering and the mirror-bin
bookkeeping of the tables.  The kernel's own indexing is checked on the GPU box (tests/test_gpu_parity.py)."""
import ctypes

import numpy as np
import pytest

HOP = 160
PAIR = [0, 1, 1, 2, 3, 4, 4, 5, 6, 7]   # entry t of a group's mel table -> register pair, mel block (savad.hip: build_fft_tables)
BLOCK = [0, 0, 1, 1, 1, 1, 2, 2, 2, 2]
LANE = np.arange(64)
J, H = LANE & 31, LANE >> 5


@pytest.fixture(scope="module")
def tables():
    from voice_activity_detection_amd import _lib, build

    build.build()
    lib = _lib.load()
    n = [lib.savad_logmel_table_floats(i) for i in range(3)]
    assert n == [4 * 13 * 256, 16 * 4 * 256, 16 * 3 * 256]
    bufs = [np.zeros(k, np.float32) for k in n]
    _lib.check(lib.savad_logmel_tables_host(*[ctypes.c_void_p(b.ctypes.data) for b in bufs]))
    return bufs[0].reshape(4, 13, 64, 4), bufs[1].reshape(16, 4, 64, 4), bufs[2].reshape(16, 3, 64, 4)


def mfma(a, b):
    """D[i][j] = sum_k A[i][k] B[k][j]: lane (i, k) holds A[i][k], lane (j, k) holds B[k][j] (k = lane >> 5)."""
    return np.stack([a[:32], a[32:]], 1).astype(np.float32) @ np.stack([b[:32], b[32:]], 0).astype(np.float32)


def regs(d):
    """register r = 4g + s of lane (j, h) holds D[8g + 4h + s][j]"""
    return np.stack([d[8 * (r >> 2) + 4 * H + (r & 3), J] for r in range(16)])


def replay_tile(ypad, f0, t1, t3, tm):
    f = f0 + J
    yl = np.zeros((16, 32, 2, 16), np.float32)  # LDS: [k1][frame][re|im][n2]
    for w in range(4):  # wave w: n2 = 4w + e
        for e in range(4):
            d = np.zeros((32, 32), np.float32)
            for s in range(13):
                d += mfma(t1[w, s, :, e], ypad[HOP * f + 16 * (3 + 2 * s + H) + 4 * w + e])
            yl[np.arange(16)[:, None], J[None, :], H[None, :], 4 * w + e] = regs(d)
    macc = [np.zeros((32, 32), np.float32) for _ in range(3)]
    for grp in range(16):
        d = np.zeros((32, 32), np.float32)
        for n2 in range(16):
            d += mfma(t3[grp, n2 >> 2, :, n2 & 3], yl[grp, J, H, n2])
        r = regs(d)
        pw = [r[2 * i] ** 2 + r[2 * i + 1] ** 2 for i in range(8)]
        for t in range(10):
            macc[BLOCK[t]] += mfma(tm[grp, t >> 2, :, t & 3], pw[PAIR[t]])
    return np.log(np.concatenate([m.T for m in macc], 1)[:, :80] + np.float32(1e-6))


def test_factored_dft_tables_replay_matches_oracle(tables):
    from oracle import logmel

    t1, t3, tm = tables
    assert not tm[:, 2, :, 2:].any()  # entries 10, 11 of a group are unused
    rng = np.random.default_rng(5)
    n = HOP * 40 + 77
    t = np.arange(n) / 16000.0
    y = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t * (1 + 0.1 * t)) + 0.05 * rng.standard_normal(n)).astype(np.float32)
    y[: n // 3] *= 0.001  # a near-silent stretch: the log(x + 1e-6) floor
    ref = logmel.log_mel(y)
    ypad = np.concatenate([np.pad(y, 256, mode="reflect"), np.zeros(HOP * 32 + 512, np.float32)])
    got = np.concatenate([replay_tile(ypad, f0, t1, t3, tm) for f0 in range(0, len(ref), 32)])[: len(ref)]
    d = np.abs(got - ref)
    assert d.max() < 5e-4 and np.median(d) < 2e-6, (d.max(), np.median(d))  # the GPU test's bounds (tests/test_gpu_parity.py)


def test_every_bin_is_produced_once(tables):
    """Each of the bins 1..255 appears in exactly one (group, slot); the special group holds the 15 multiples of 16."""
    from oracle import logmel

    _, _, tm = tables
    M = logmel.mel_filterbank()
    # a filter's weights over all (group, entry, lane half) slots must add up to its row sum: nothing lost, nothing doubled
    got = np.zeros(96)
    for grp in range(16):
        for t in range(10):
            v = tm[grp, t >> 2, :, t & 3]
            for lane in range(64):
                got[32 * BLOCK[t] + (lane & 31)] += v[lane]
    assert np.allclose(got[:80], M.sum(axis=1), rtol=1e-6) and not got[80:].any()


def test_span_sample_ranges():
    from voice_activity_detection_amd import _lib

    lib = _lib.load()
    first, count = ctypes.c_long(), ctypes.c_long()
    n = 57_600_000
    nf = 1 + n // HOP
    for f0, fc in ((0, 1), (0, 5000), (1, 3), (2, 1), (45000, 45000), (nf - 1, 1), (nf - 4, 4), (nf - 45000, 45000)):
        _lib.check(lib.savad_logmel_span_samples(n, f0, fc, ctypes.byref(first), ctypes.byref(count)))
        a, b = first.value, first.value + count.value
        assert a % 4 == 0 and 0 <= a < b <= n
        lo, hi = HOP * f0 - 208, HOP * (f0 + fc - 1) + 208  # padded index - 256 of the first / one past the last sample read
        need = {min(max(i, -i), 2 * (n - 1) - i) if i >= n else abs(i) for i in (lo, hi - 1, max(lo, 0), min(hi - 1, n - 1))}
        assert all(a <= i < b for i in need), (f0, fc, a, b, need)
        assert a >= max(0, lo) - 3 - (208 if hi > n else 0) and b <= min(n, hi) + (209 if lo < 0 else 0)  # and no more than that
    assert lib.savad_logmel_span_samples(n, nf, 1, ctypes.byref(first), ctypes.byref(count)) != 0
