import sys, numpy as np, torch
sys.path.insert(0, '.')
from voice_activity_detection_amd import SelfAttentiveVAD, VADFromScratchPredictor, StreamingPredictor, seeded_state_dict
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
rng = np.random.default_rng(0)
for prec in ("fp32", "fp32s", "bf16"):
    m.precision = prec
    for n in (1, 159, 160, 3200, 6080, 6240, 16000):
        pcm = (rng.standard_normal(n) * 3000).astype(np.int16)
        fd = torch.from_numpy(pcm.astype(np.float32) / 32768.0).cuda()
        pe = VADFromScratchPredictor(m, "cuda")
        pg = VADFromScratchPredictor(m, "cuda", graph=True)
        a, am = pe.predict_audio_device(fd)
        b, bm = pe.predict_audio_host(pcm)
        c, cm = pg.predict_audio_device(fd)
        c2, _ = pg.predict_audio_device(pcm.astype(np.float32) / 32768.0)
        ok = torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, c2) and torch.equal(am, bm) and torch.equal(am, cm)
        sp = StreamingPredictor(m, "cuda", 96, 48, max_batch=4)
        s1 = sp.predict_audio_device(fd); s2 = sp.predict_audio_host(pcm)
        ok2 = torch.equal(s1, s2)
        print(prec, n, tuple(a.shape), "ref-mode equal", ok, "streaming equal", ok2, "finite", bool(torch.isfinite(a).all()), flush=True)
        assert ok and ok2
try:
    VADFromScratchPredictor(m, "cuda").predict_audio_host(np.zeros(0, np.int16))
except Exception as e:
    print("empty:", type(e).__name__, str(e)[:80])
print("edge ok")
