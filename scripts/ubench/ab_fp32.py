"""Same-run A/B of two library builds on the fp32 headline shape.  usage: ab_fp32.py libA.so libB.so [B] [T]"""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.getcwd())
    import torch
    from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict
    B, T = int(sys.argv[2]), int(sys.argv[3])
    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
    m = m.cuda().eval()
    x = torch.randn(B, T, 80, device="cuda")
    with torch.no_grad():
        for _ in range(20): y = m(x)
        torch.cuda.synchronize()
        ts = []
        for rep in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): m(x)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 50)
    print(f"{os.environ['SAVAD_LIB'].split('/')[-1]:20s} median {sorted(ts)[3]:.4f} min {min(ts):.4f} ms  checksum {float(y.double().sum()):.6f}")
else:
    B, T = (sys.argv[3] if len(sys.argv) > 3 else "32"), (sys.argv[4] if len(sys.argv) > 4 else "800")
    for rnd in range(2):
        for lib in sys.argv[1:3]:
            subprocess.run([sys.executable, __file__, "--one", B, T], env=dict(os.environ, SAVAD_LIB=os.path.abspath(lib)))
