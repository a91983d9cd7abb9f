import sys, os
sys.path.insert(0, os.getcwd())
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"
x = torch.from_numpy(seeded_features(1, (256, 800, 80))).cuda().to(torch.bfloat16)
for mode in [int(a) for a in sys.argv[1:]] or (0, 2, 1, 5):
    m.row_mode = mode
    with torch.no_grad():
        for _ in range(20): m(x)
        torch.cuda.synchronize()
        m.set_profiling(10, skip=100)
        for _ in range(110): m(x)
        torch.cuda.synchronize()
    kt = m.kernel_times(); m.set_profiling(0)
    agg = {}
    for n, t in kt: agg.setdefault(n, []).append(t * 1e3)
    print(mode, {n: round(sum(v) / len(v), 1) for n, v in agg.items()}, "forward", round(sum(t for _, t in kt) * 1e3, 1))
