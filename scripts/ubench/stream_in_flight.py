import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, StreamingPredictor, seeded_state_dict
st = seeded_state_dict(1234)
feat = torch.from_numpy(np.random.default_rng(3).uniform(-13.8, 4.2, (360001, 80)).astype(np.float32)).cuda()
for prec in ("fp32", "bf16"):
    m = SelfAttentiveVAD(80, 3, 128, 0.5); m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}); m = m.cuda().eval(); m.precision = prec
    ref = None
    for mb, nf in ((256, 1), (256, 2), (256, 3), (128, 2), (128, 3), (64, 3), (32, 3)):
        sp = StreamingPredictor(m, "cuda", 800, 400, max_batch=mb, in_flight=nf)
        for _ in range(3): p = sp.predict_device(feat)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): p = sp.predict_device(feat)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
        if ref is None: ref = p.clone()
        print(f"{prec} max_batch={mb:4d} in_flight={nf}: {ms:7.3f} ms per hour of audio   same bits as the first: {torch.equal(p, ref)}  max|d|={float((p-ref).abs().max()):.2e}", flush=True)
