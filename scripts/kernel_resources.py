#!/usr/bin/env python3
"""Per-kernel register / spill / scratch / LDS figures of the device code object (llvm-readelf --notes).
usage: scripts/kernel_resources.py [extra hipcc flags]    -- prints one line per kernel; exit 1 if any kernel spills."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def compile_code_object(co, extra_flags=()):
    """device code of libsavad.so as one gfx950 code object"""
    from voice_activity_detection_amd import build
    subprocess.run([build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "--no-gpu-bundle-output",
                    "-Wno-unused-value", "-w", *extra_flags, str(build.SRC), "-o", str(co)], check=True)
    return co


def kernel_resources(extra_flags=(), co=None):
    with tempfile.TemporaryDirectory() as d:
        if co is None:
            co = compile_code_object(Path(d) / "savad.co", extra_flags)
        notes = subprocess.run([READELF, "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    out = {}
    for entry in re.split(r"\n  - ", notes):
        name = re.search(r"\.name:\s+(\S+)", entry)
        if not name or ".vgpr_count" not in entry:
            continue
        get = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", entry).group(1))
        out[name.group(1)] = {k: get(k) for k in ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count",
                                                  "private_segment_fixed_size", "group_segment_fixed_size")}
    return out


if __name__ == "__main__":
    res = kernel_resources(sys.argv[1:])
    bad = 0
    for name, r in res.items():
        try:
            demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0] or name
        except OSError:
            demangled = name
        print(f"{r['vgpr_count']:4d} vgpr {r['agpr_count']:4d} agpr {r['vgpr_spill_count']:3d} spill {r['private_segment_fixed_size']:4d} scratch "
              f"{r['sgpr_count']:4d} sgpr {r['group_segment_fixed_size']:7d} lds  {demangled}")
        bad += r["vgpr_spill_count"] > 0 or r["private_segment_fixed_size"] > 0
    sys.exit(1 if bad else 0)
