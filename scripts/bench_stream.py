#!/usr/bin/env python3
"""BASELINE.json configs[4]: 1 h of synthetic audio (360 001 log-mel frames @100 fps) through the
streaming long-form mode (T=800, hop=400 -> 900 windows), device-resident features -> per-frame
probabilities.  Reports the real-time factor with and without the GPU log-mel front-end.
Usage: bench_stream.py [max_batch=256] [fp32|bf16].  Multi-GPU: launch with torch.distributed.run."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from voice_activity_detection_amd import SelfAttentiveVAD, StreamingPredictor, seeded_state_dict  # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist

    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
m.precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
from voice_activity_detection_amd.features import log_mel  # noqa: E402

N = 3600 * 100 + 1
audio = torch.from_numpy((np.random.default_rng(0).standard_normal(16000 * 3600) * 0.1).astype(np.float32)).cuda()
for _ in range(2):
    feat = log_mel(audio)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    feat = log_mel(audio)
torch.cuda.synchronize()
dt_mel = (time.perf_counter() - t0) / 5
assert feat.shape == (N, 80)
sp = StreamingPredictor(m, "cuda", 800, 400, max_batch=int(sys.argv[1]) if len(sys.argv) > 1 else 256)
for _ in range(2):
    p = sp.predict_device(feat)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    p = sp.predict_device(feat)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
if rank == 0:
    print(json.dumps({"mode": "streaming T=800 hop=400", "precision": m.precision, "audio_seconds": 3600, "frames": N, "windows": 900,
                      "n_gpus": world, "seconds": round(dt, 5), "rtf_without_logmel": dt / 3600.0,
                      "logmel_seconds": round(dt_mel, 5), "rtf_with_logmel": (dt + dt_mel) / 3600.0,
                      "frames_per_s_of_audio": N / dt, "finite": bool(torch.isfinite(p).all().item())}))
if world > 1:
    dist.destroy_process_group()
