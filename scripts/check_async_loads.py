#!/usr/bin/env python3
"""Static check of the hand-scheduled asynchronous loads in the compiled kernels (gfx950 assembly of csrc/savad.hip).

The weight streams of the N-split kernels (savad_kernels.h: wload_frag / wwait) and a few other places request global loads
in `asm volatile` statements and wait for them many instructions later with a hand-counted `s_waitcnt vmcnt(N)`.  To the
compiler the destination registers of such a statement hold their value as soon as the statement has executed: it is free to
copy them (to AGPRs under register pressure), or -- when the loaded value is never used -- to put something else into them,
while the load is still in flight.  Either is a silent race with the memory system: the copy takes the OLD register contents,
the late load lands on top of the NEW ones.  Both happened (round 4: `row_kernel<true>` and `packed_forward_kernel` issued a
weight block they never waited for and the registers were reused for the reduce-scatter reads; `row_kernel<false>` parked four
registers of a requested block in AGPRs before its wait on the key-split path) and showed up only when other streams' kernels
made the weight loads miss the L2 -- as one wrong 32-row tile in a few percent of the runs of one GPU test.

The check walks every path of every kernel's control-flow graph (path-sensitive, states memoised per basic block) with the
queue of vector-memory loads in flight, retires them as the `s_waitcnt vmcnt(N)` on the path allow (loads return in order;
stores are ignored, which can only keep a load "in flight" longer than it really is), and reports every instruction that reads
or writes a register an asm-issued load has not yet delivered.  Compiler-issued loads are only counted (the compiler waits for
its own).  tests/test_async_load_hazards.py runs it on the kernels in the tree.

    python scripts/check_async_loads.py             # compiles csrc/savad.hip to assembly (cached by source hash) and checks it
    python scripts/check_async_loads.py --asm f.s   # checks an assembly file
"""
from __future__ import annotations

import argparse
import hashlib
import re
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "voice_activity_detection_amd" / "csrc"

_REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
_LOAD = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load)")
_LABEL = re.compile(r"^([.\w$]+):")


def regs(text: str) -> frozenset:
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.update(f"{m.group(1)}{k}" for k in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add(m.group(4) + m.group(5))
    return frozenset(out)


class Insn:
    __slots__ = ("line", "text", "op", "in_asm", "is_load", "dst", "touched", "vmcnt", "target", "kind")

    def __init__(self, line: int, text: str, in_asm: bool):
        self.line, self.text, self.in_asm = line, text, in_asm
        self.op = text.split()[0]
        args = text[len(self.op):]
        self.is_load = bool(_LOAD.match(self.op))
        self.dst = frozenset()
        if self.is_load and "lds" not in self.op and " lds" not in args:
            first = args.split(",")[0]
            self.dst = regs(first)
            args_rest = ",".join(args.split(",")[1:])
            self.touched = regs(args_rest) | self.dst
        else:
            self.touched = regs(args)
        self.vmcnt = None
        if self.op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", text)
            if m:
                self.vmcnt = int(m.group(1))
            elif re.fullmatch(r"s_waitcnt\s+(0|0x0)", text):
                self.vmcnt = 0
        self.target = None
        self.kind = "plain"
        if self.op == "s_branch":
            self.kind, self.target = "jump", args.strip()
        elif self.op.startswith("s_cbranch"):
            self.kind, self.target = "cond", args.strip().split(",")[-1].strip()
        elif self.op in ("s_endpgm", "s_setpc_b64"):
            self.kind = "end"


def kernels(asm_text: str):
    """-> (symbol, [blocks]); a block is (label or None, [Insn])"""
    lines = asm_text.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*; @", lines[i])
        if not m:
            i += 1
            continue
        sym = m.group(1)
        blocks, cur, in_asm = [], (sym, []), False
        i += 1
        while i < len(lines) and not lines[i].strip().startswith(".Lfunc_end"):
            s = lines[i].strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
            elif s.startswith(";;#ASMEND"):
                in_asm = False
            else:
                lm = _LABEL.match(s)
                if lm and not s.startswith(";"):
                    blocks.append(cur)
                    cur = (lm.group(1), [])
                else:
                    code = s.split(";")[0].strip()
                    if code and not code.startswith("."):
                        cur[1].append(Insn(i + 1, code, in_asm))
            i += 1
        blocks.append(cur)
        yield sym, blocks


def check_kernel(blocks, max_states: int = 200000):
    """-> (hazards, truncated); a hazard is (line, text, line of the pending load, its text)"""
    index = {label: k for k, (label, _) in enumerate(blocks)}
    loads = {}
    hazards = {}
    seen = set()
    work = [(0, ())]
    while work:
        k, state = work.pop()
        if (k, state) in seen:
            continue
        if len(seen) >= max_states:
            return sorted(hazards.values()), True
        seen.add((k, state))
        pending = list(state)   # (load line, is_asm) in issue order
        ended = False
        succ = []
        for ins in blocks[k][1]:
            if ins.vmcnt is not None:
                if ins.vmcnt < len(pending):
                    pending = pending[len(pending) - ins.vmcnt:] if ins.vmcnt else []
                continue
            for pl, is_asm in pending:
                if is_asm and ins.touched & loads[pl].dst:
                    hazards.setdefault((ins.line, pl), (ins.line, ins.text, pl, loads[pl].text))
            if ins.is_load:
                loads[ins.line] = ins
                pending.append((ins.line, ins.in_asm))
                if len(pending) > 64:   # the counter saturates at 63: older ones must have been waited for by construction
                    pending = pending[-64:]
            if ins.kind == "jump":
                succ = [index[ins.target]] if ins.target in index else []
                ended = True
                break
            if ins.kind == "cond" and ins.target in index:
                succ.append(index[ins.target])
            if ins.kind == "end":
                ended = True
                succ = []
                break
        if not ended and k + 1 < len(blocks):
            succ.append(k + 1)
        st = tuple(pending)
        for s in succ:
            if (s, st) not in seen:
                work.append((s, st))
    return sorted(hazards.values()), False


def source_hash() -> str:
    h = hashlib.sha256()
    for f in sorted(CSRC.iterdir()):
        if f.suffix in (".h", ".hip", ".inc"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def compile_asm(verbose: bool = False) -> Path:
    """device assembly of csrc/savad.hip with the library's own flags (voice_activity_detection_amd/build.py); cached by source hash"""
    sys.path.insert(0, str(REPO))
    from voice_activity_detection_amd.build import hipcc

    out = Path(tempfile.gettempdir()) / f"savad_{source_hash()}.s"
    if not out.exists():
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-S",
               str(CSRC / "savad.hip"), "-o", str(out) + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, stderr=None if verbose else subprocess.DEVNULL)
        Path(str(out) + ".tmp").rename(out)
    return out


def check_file(path: Path):
    """-> {kernel symbol: (hazards, truncated)} for every kernel with asm-issued loads"""
    report = {}
    for sym, blocks in kernels(path.read_text()):
        if not any(i.is_load and i.in_asm for _, b in blocks for i in b):
            continue
        report[sym] = check_kernel(blocks)
    return report


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--asm", type=Path, help="assembly file to check (default: compile csrc/savad.hip)")
    ap.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args()
    path = args.asm or compile_asm(args.verbose)
    bad = 0
    for sym, (hazards, truncated) in check_file(path).items():
        print(f"{sym}: {len(hazards)} hazard(s)" + (" (path search truncated)" if truncated else ""))
        for line, text, pl, ptext in hazards[: (1000 if args.verbose else 8)]:
            print(f"    line {line}: {text}\n        touches the destination of line {pl}: {ptext}")
        bad += len(hazards) + (1 if truncated else 0)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
