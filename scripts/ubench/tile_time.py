import os, sys
sys.path.insert(0, os.getcwd())
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"
for B, T in ((256, 800), (267, 768), (245, 800), (384, 800)):
    x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda().to(torch.bfloat16)
    for mode in (1, 5, 0):
        m.row_mode = mode
        with torch.no_grad():
            for _ in range(30): m(x)
            torch.cuda.synchronize()
            m.set_profiling(20, skip=200)
            for _ in range(220): m(x)
            torch.cuda.synchronize()
            kt = m.kernel_times(); m.set_profiling(0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(50): m(x)
            e0.record()
            for _ in range(300): m(x)
            e1.record(); torch.cuda.synchronize()
        by = {}
        for n, t in kt: by.setdefault(n, []).append(t * 1e3)
        print(f"[{B},{T}] row_mode {mode}: forward {e0.elapsed_time(e1) / 300 * 1e3:7.1f} us  " + "  ".join(f"{n} {sum(v)/len(v):.1f}" for n, v in by.items()), flush=True)
