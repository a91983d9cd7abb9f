"""per-stage times (library profiling marks) of the bf16 launch schedules at [256,800,80]: row_mode 0 (and with batch_invariant) / 1 / 2 / 5 / 3"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"
x = torch.from_numpy(seeded_features(1, (256, 800, 80))).cuda().to(torch.bfloat16)
for mode in (0, "0 batch_invariant", 1, 2, 5, 3):
    m.batch_invariant = isinstance(mode, str)
    m.row_mode = 0 if isinstance(mode, str) else mode
    with torch.no_grad():
        for _ in range(50): m(x)
        torch.cuda.synchronize()
        m.set_profiling(64)
        for _ in range(20): m(x)
        torch.cuda.synchronize()
        prof = m.kernel_times()
        m.set_profiling(0)
    print(mode, round(sum(v for _, v in prof) * 1e3, 1), " ".join(f"{n}={v*1e3:.1f}" for n, v in prof), flush=True)
