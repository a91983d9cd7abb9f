#!/usr/bin/env python3
"""One [256,800,80] bf16 batch as ONE forward against the same 256 sequences as two [128,800,80] halves in flight on two streams
(PipelinedVAD depth 2: the memory-bound input / row launches of one half under the attention launches of the other), and as
four quarters.  ms per 256 sequences, HIP events over >= 0.3 s each.   python scripts/ubench/half_batches.py [B T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from voice_activity_detection_amd import PipelinedVAD, SelfAttentiveVAD, seeded_state_dict, seeded_features

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 800)
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"
x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda().to(torch.bfloat16)
out = torch.empty((B, T, 2), device="cuda")

def timed(fn, reps=400):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps // 5): fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / (reps // 5))
    best.sort()
    return best[len(best) // 2], best[0]

with torch.no_grad():
    res = {"one forward of all": timed(lambda: m(features=x, out=out))}
    ref = out.clone()
    for parts in (2, 4):
        pipe = PipelinedVAD(m, depth=parts if parts == 2 else 2)
        pipe.reserve(T, max_batch=B // parts)
        step = B // parts
        def split():
            for i in range(parts):
                pipe.submit(x[i * step:(i + 1) * step], out=out[i * step:(i + 1) * step])
            pipe.join()
        res[f"{parts} parts of {step}, two in flight"] = timed(split)
        torch.cuda.synchronize()
        res[f"{parts} parts: max |dlogp| vs one forward"] = (float((out - ref).abs().max()), 0.0)
    # and two WHOLE batches in flight (throughput regime), per batch
    pipe = PipelinedVAD(m, depth=2); pipe.reserve(T, max_batch=B)
    out2 = torch.empty_like(out)
    def two():
        pipe.submit(x, out=out); pipe.submit(x, out=out2); pipe.join()
    t = timed(two, 200)
    res["two whole batches in flight (per batch)"] = (t[0] / 2, t[1] / 2)
for k, v in res.items():
    print(f"[{B},{T}] bf16  {k:48s} {v[0]:9.4f}  (min {v[1]:.4f})", flush=True)
