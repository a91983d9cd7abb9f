"""Build libsavad.so (HIP, gfx950) in-tree.  `python -m voice_activity_detection_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
SRC = PKG / "csrc" / "savad.hip"
DEPS = (sorted((PKG / "csrc").glob("*.h")) + sorted((PKG / "csrc").glob("*.hip")) + sorted((PKG / "csrc").glob("*.inc")) +
        [PKG.parent / "include" / "savad.h"])
LIB = PKG / "libsavad.so"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale() -> bool:
    return not LIB.exists() or any(d.stat().st_mtime > LIB.stat().st_mtime for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
           "-Wno-unused-value", str(SRC), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


def build_variant(out: Path, defines=(), force: bool = False, verbose: bool = False) -> Path:
    """The same sources with extra -D switches into `out` (experiments and the fault-injected library of the negative tests:
    tests/fault/; never the product)."""
    out = Path(out)
    if not force and out.exists() and all(d.stat().st_mtime <= out.stat().st_mtime for d in DEPS):
        return out
    out.parent.mkdir(parents=True, exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value",
           *[f"-D{d}" for d in defines], str(SRC), "-o", str(out)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


FAULT_LIB = PKG.parent / "tests" / "fault" / "libsavad_fault17.so"  # SAVAD_FAULT_INJECT=17 (bits 1 and 16): tests/test_gpu_cache_pressure.py
FAULT_DEFINES = ["SAVAD_FAULT_INJECT=17"]


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
