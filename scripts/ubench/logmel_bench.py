#!/usr/bin/env python3
"""Log-mel front-end timings: the factored DFT (algorithm 0, round 5) beside the DFT as one GEMM (algorithm 1, rounds 1-4),
an hour of audio and a 10 s clip, HIP events around blocks of calls.  usage: python scripts/ubench/logmel_bench.py"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from voice_activity_detection_amd import _lib  # noqa: E402
from voice_activity_detection_amd.features import log_mel  # noqa: E402


def timed(fn, reps, blocks=7):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(blocks):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps)
    return float(np.median(out)), float(min(out))


def main():
    lib = _lib.load()
    if "--one" in sys.argv:   # the profiled command (scripts/profile_logmel.sh): the product algorithm on an hour of audio, 20 calls
        y = torch.from_numpy(np.random.default_rng(0).standard_normal(16000 * 3600, dtype=np.float32) * 0.1).cuda()
        for _ in range(20):
            log_mel(y)
        torch.cuda.synchronize()
        return
    rng = np.random.default_rng(0)
    res = {}
    for name, seconds, reps in (("1h", 3600, 10), ("10s", 10, 200)):
        y = torch.from_numpy(rng.standard_normal(16000 * seconds, dtype=np.float32) * 0.1).cuda()
        outs = {}
        for alg in (0, 1):
            _lib.check(lib.savad_logmel_set_algorithm(alg))
            med, mn = timed(lambda: log_mel(y), reps)
            outs[alg] = log_mel(y)
            n = y.numel()
            res[f"{name}_alg{alg}"] = {"ms": round(med, 4), "ms_min": round(mn, 4),
                                       "algorithmic_GBps": round((n * 4 + (1 + n // 160) * 320) / (med * 1e-3) / 1e9, 1)}
        _lib.check(lib.savad_logmel_set_algorithm(0))
        res[f"{name}_max_abs_diff"] = float((outs[0] - outs[1]).abs().max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
