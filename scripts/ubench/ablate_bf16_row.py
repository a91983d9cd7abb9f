"""Kernel times of the bf16 schedule under the ablation builds (results are wrong by design; timing only)."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.getcwd())
    import torch
    from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict
    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
    m = m.cuda().eval(); m.precision = "bf16"; m.row_mode = 1
    x = torch.randn(256, 800, 80, device="cuda")
    with torch.no_grad():
        for _ in range(5): m(x)
        torch.cuda.synchronize()
        m.set_profiling(10)
        for _ in range(10): m(x)
        torch.cuda.synchronize()
        t = m.kernel_times()
    print(f'{os.environ["SAVAD_LIB"].split("ablate")[-1]:12s}', " ".join(f"{n.split('_bf16')[0]}={v*1e3:.0f}" for n, v in t))
else:
    for a in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, SAVAD_LIB=os.path.abspath(f"scripts/ubench/libsavad_ablate{a}.so")))
