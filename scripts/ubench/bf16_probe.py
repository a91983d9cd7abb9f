"""Timing probe for the bf16 attention kernels (experiments only).  usage: attn2_probe.py <lib.so> <row_mode> [B] [timing]"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["SAVAD_LIB"] = os.path.abspath(sys.argv[1])
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, _lib
mode = int(sys.argv[2]); B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"; m.row_mode = mode
x = torch.from_numpy(np.random.default_rng(0).uniform(-13.8, 4.2, (B, 800, 80)).astype(np.float32)).cuda().to(torch.bfloat16)
with torch.no_grad():
    for _ in range(30): y = m(x)
    torch.cuda.synchronize()
    m.set_profiling(30)
    for _ in range(30): y = m(x)
    torch.cuda.synchronize()
kt = m.kernel_times()
att = [t for n, t in kt if n.startswith("attention")]
print(os.path.basename(sys.argv[1]), "mode", mode, "B", B, "attention us:", " ".join(f"{1e3*t:.1f}" for t in att), "| forward us:", f"{1e3*sum(t for _, t in kt):.1f}", "finite", bool(torch.isfinite(y).all()))
if len(sys.argv) > 4:
    lib = _lib.load()
    lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    buf = (ctypes.c_longlong * 32)()
    lib.savad_debug_stamps(buf, 32)
    names = ["round prologue", "acquire wait", "phase1 (S+exp)", "dma issue", "kload+phase2", "check/tail", "finalize", "other"]
    tot = sum(buf[:8])
    for n, v in zip(names, buf[:8]): print(f"   {n:16s} {v:9d} cycles  {100*v/max(tot,1):5.1f} %")
    print("   total", tot)
