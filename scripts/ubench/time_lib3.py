"""fp32 [32,800], bf16 [256,800], fp32 [1000,7] forward times per library (timing only).  usage: time_lib3.py lib.so ..."""
import os, subprocess, sys
if sys.argv[1] == "--one":
    sys.path.insert(0, os.getcwd())
    import torch
    from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict
    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
    m = m.cuda().eval()
    out = []
    for prec, B, T in (("fp32", 32, 800), ("bf16", 256, 800), ("fp32", 1000, 7)):
        m.precision = prec
        x = torch.randn(B, T, 80, device="cuda")
        with torch.no_grad():
            for _ in range(20): m(x)
            torch.cuda.synchronize()
            ts = []
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40): m(x)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 40)
        out.append(f"{prec}[{B},{T}] {sorted(ts)[2]:.4f}")
    print(f"{os.path.basename(os.environ['SAVAD_LIB']):18s}", "  ".join(out))
else:
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, SAVAD_LIB=os.path.abspath(lib)), timeout=120)
