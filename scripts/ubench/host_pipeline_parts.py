"""Where StreamingPredictor.predict_audio_host's time goes (1 h of PCM16, bf16 / fp32s): the call from pinned host memory, the same call from
a DEVICE-resident int16 copy (uploads become device-to-device copies: what the chunking itself costs), and predict_audio_device."""
import sys, time
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from voice_activity_detection_amd import SelfAttentiveVAD, StreamingPredictor, seeded_state_dict
import voice_activity_detection_amd.predictor as P

m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
pcm = np.clip(np.round(np.random.default_rng(0).standard_normal(16000 * 3600, dtype=np.float32) * 3276.8), -32768, 32767).astype(np.int16)
pinned = torch.from_numpy(pcm).pin_memory()
dev16 = pinned.cuda()
devf = (dev16.float() / 32768.0).contiguous()

def wall(fn, n=12):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]

orig = StreamingPredictor._host_source
for prec in ("bf16", "fp32s"):
    m.precision = prec
    sp = StreamingPredictor(m, "cuda", 800, 400, max_batch=256)
    a = wall(lambda: sp.predict_audio_device(devf))
    b = wall(lambda: sp.predict_audio_host(pinned))
    StreamingPredictor._host_source = lambda self_or_audio, audio=None: (audio if audio is not None else self_or_audio)   # accept the device tensor: "uploads" are d2d copies
    c = wall(lambda: sp.predict_audio_host(dev16))
    StreamingPredictor._host_source = staticmethod(orig)
    m.batch_invariant = prec == "bf16"
    e = wall(lambda: sp.predict_audio_device(devf))
    d = wall(lambda: sp.predict_audio_host(pinned, ramp=True))
    same = torch.equal(sp.predict_audio_host(pinned, ramp=True), sp.predict_audio_device(devf))
    m.batch_invariant = False
    print(f"{prec}: device-resident one shot {a:.3f} ms | from pinned host {b:.3f} | chunked, source on the device {c:.3f} | "
          f"ramped spans{' (batch_invariant)' if prec == 'bf16' else ''}: device {e:.3f}, from host {d:.3f}, equal bits {same}", flush=True)
