#!/usr/bin/env python3
"""bf16 input stage: the persistent weights-resident kernel (the product) against the ring kernel everywhere (a variant build with
-DSAVAD_INPUT_PERSISTENT=0, made here on first use): same-run A/B of the whole forward and of the stage itself (library profiling
marks), and the bits of the output.  input_p_ab.py [B T]"""
import hashlib, os, subprocess, sys
if sys.argv[1:2] == ["--one"]:
    sys.path.insert(0, os.getcwd())
    import torch
    from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
    B, T = int(sys.argv[2]), int(sys.argv[3])
    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
    m = m.cuda().eval(); m.precision = "bf16"
    x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda()
    with torch.no_grad():
        y = m(x); torch.cuda.synchronize()
        digest = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]
        for _ in range(100): m(x)
        torch.cuda.synchronize()
        best = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): m(x)
            e1.record(); torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) / 100)
        m.set_profiling(64)
        for _ in range(20): m(x)
        torch.cuda.synchronize()
        prof = m.kernel_times()
    print(f"{os.path.basename(os.environ.get('SAVAD_LIB', 'libsavad.so')):20s} [{B},{T}] bits {digest}: forward median {sorted(best)[2]*1e3:7.1f} us min {min(best)*1e3:7.1f} us; "
          + " ".join(f"{n}={v*1e3:.1f}" for n, v in prof), flush=True)
else:
    sys.path.insert(0, os.getcwd())
    from voice_activity_detection_amd import build
    ring = build.build_variant("scripts/ubench/abl/libsavad_ring_input.so", ["SAVAD_INPUT_PERSISTENT=0"])
    B, T = (sys.argv[1:3] if len(sys.argv) >= 3 else ("256", "800"))
    for _ in range(2):
        for lib in (str(ring), str(build.build())):
            subprocess.run([sys.executable, __file__, "--one", B, T], env=dict(os.environ, SAVAD_LIB=os.path.abspath(lib)))
