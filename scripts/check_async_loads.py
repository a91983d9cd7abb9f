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

The check runs a data flow over every kernel's control-flow graph: per load that may be in flight, how many vector-memory
operations were issued after it; `s_waitcnt vmcnt(N)` retires the loads at distance >= N (loads return in order); where paths
meet a load keeps its smallest distance (exact per load over all paths, infeasible ones included).  Every instruction that reads
or writes a register of a load still in flight is reported (a younger LOAD into the same register is ordered by the in-order
return and is not).  On compiler assembly only asm-issued loads are checked and stores are ignored (the stricter reading for
hand-placed waits); on the disassembly of a built library (--lib: the artifact that ships) every load is -- the compiler's own
code passes because it waits before it touches a destination -- with stores in the queue, as the compiler's waits assume.
tests/test_async_load_hazards.py runs the --lib form on the library in the tree; both forms report the same 531 / 56 / 464
accesses for `row_kernel<true>` / `row_kernel<false>` / `packed_forward_kernel` as they were before the fix.

Round 5 added two more rules, same data flow:
  * LDS reads issued by hand (`ds_read_b128` in asm statements with counted `s_waitcnt lgkmcnt(N)`: the weight fragments of the bf16
    ring GEMMs, savad_kernels_bf16.h: gemm_ring_t).  The queue holds the DS operations only: scalar-memory loads share the counter
    and may return out of order, but an extra outstanding operation can only make the hardware's wait retire MORE of the (in-order)
    DS operations than the model does -- the model is the conservative side.
  * LDS-DMA destinations (`global_load_lds_*`): a workgroup's waves read an LDS block while another wave may already be
    re-targeting it.  The protocol everywhere in csrc/ is "barrier, then DMA": the barrier a wave passes before it issues a DMA is
    the one every wave reaches only after its last read of the block being replaced.  Rule: on no path may a wave issue a DMA,
    access LDS (ds_read / ds_write) and issue a DMA again without an `s_barrier` in between (a removed or misplaced ring barrier
    shows as exactly that).  Kernels in which one logical request is several statements with LDS reads of OTHER buffers scheduled
    between them are listed in DMA_DISJOINT with the reason.
  * LDS-DMA publication (dma_barriers_in_flight): a wave's DMA pieces must have LANDED -- retired by one of its own counted
    `s_waitcnt vmcnt(N)`, stores counted as the hardware counts them -- before the wave passes the barrier that hands the block to
    the other waves: per DMA instruction the number of barriers it may cross in flight, against a bound per kernel (0 for the
    2-slot rings and the double-buffered tiles: DMA_PUBLISH_BARRIERS lists the others and why).  This is the rule that checks the
    ring waits which leave a wave's own stores in flight (vmcnt(8) instead of vmcnt(0): savad_kernels_bf16.h, Ring::acquire).

    python scripts/check_async_loads.py             # compiles csrc/savad.hip to assembly (cached by source hash) and checks it
    python scripts/check_async_loads.py --asm f.s   # checks an assembly file
    python scripts/check_async_loads.py --lib voice_activity_detection_amd/libsavad.so   # the device code of the built library itself
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
_STORE = re.compile(r"^(global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic|flat_atomic)")
_LABEL = re.compile(r"^([.\w$]+):")
_DSREAD = re.compile(r"^ds_read")
_DSOP = re.compile(r"^ds_")
# kernels whose LDS-DMA target cannot be what the wave reads between its last barrier and the DMA (substring of the symbol -> why)
_F32S_PIECES = ("the twelve 1-KiB pieces of ring slot t + 2 are placed one at a time between the MFMAs of the step that reads slot t, on purpose "
                "(savad_kernels_f32s.h, Ring3): one request, ordered by the barrier at the head of step t -- the slot they fill was last read "
                "in step t - 1, which every wave has left")
DMA_DISJOINT = {
    "21input_qkv_kernel_f32s": _F32S_PIECES,
    "25attention_row_kernel_f32s": _F32S_PIECES,
    "26packed_forward_kernel_f32s": _F32S_PIECES,
    "logmel_fft_kernel": "the DMA fills the sample stage; between barrier 1 and the DMA statements the wave only reads the exchange "
                         "buffer Yl (its reads of the stage lie before barrier 1, the stage's next readers behind barrier 2)",
    "5savad16attention_kernelE": "the next tile's 16 KiB are four dma_piece statements placed between the MFMAs of the tile being "
                                 "computed on purpose (savad_kernels.h): one request, ordered by the barrier at the head of the tile",
    "5savad20attention_row_kernelI": "as attention_kernel (the same tile loop)",
    "attention_pw_kernel_bf16": "a generated instruction stream that interleaves each stage's DMA with the LDS reads of the stage in "
                                "use; its barriers are modelled instruction by instruction in scripts/gfx950_sim.py",
}


# kernels whose HAND-placed waits count the wave's own stores (substring of the symbol -> why): checked with stores in the queue in the
# assembly form too (elsewhere that form ignores them -- the stricter reading; scripts/ubench/vmcnt_order.hip is the hardware check
# that loads and stores retire in one issue order)
STORES_IN_QUEUE = {
    "input_qkv_kernel_bf16_p": "the next block's requests are waited for with vmcnt(24): this block's 24 fragment stores are younger than every one of them (savad_kernels_bf16.h)",
}


def regs(text: str) -> frozenset:
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.update(f"{m.group(1)}{k}" for k in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add(m.group(4) + m.group(5))
    return frozenset(out)


class Insn:
    __slots__ = ("line", "text", "op", "in_asm", "is_load", "is_store", "dst", "touched", "vmcnt", "target", "kind", "is_dsread", "is_ds",
                 "dsdst", "lgkmcnt", "is_dma", "is_barrier")

    def __init__(self, line: int, text: str, in_asm: bool):
        self.line, self.text, self.in_asm = line, text, in_asm
        self.op = text.split()[0]
        args = text[len(self.op):]
        self.is_load = bool(_LOAD.match(self.op))
        self.is_store = bool(_STORE.match(self.op))
        self.dst = frozenset()
        if self.is_load and "lds" not in self.op and " lds" not in args:
            first = args.split(",")[0]
            self.dst = regs(first)
            # (a load into the destination of an older load still in flight is ordered by the in-order return: only its address counts)
            self.touched = regs(",".join(args.split(",")[1:]))
        else:
            self.touched = regs(args)
        self.is_ds = bool(_DSOP.match(self.op))
        self.is_dsread = bool(_DSREAD.match(self.op))
        self.dsdst = regs(args.split(",")[0]) if self.is_dsread else frozenset()
        if self.is_dsread:
            self.touched = regs(",".join(args.split(",")[1:]))
        self.is_dma = self.is_load and ("lds" in self.op or " lds" in args)
        self.is_barrier = self.op == "s_barrier"
        self.lgkmcnt = None
        if self.op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", text)
            if m:
                self.lgkmcnt = int(m.group(1))
            elif re.fullmatch(r"s_waitcnt\s+(0|0x0)", text):
                self.lgkmcnt = 0
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
        elif self.op.startswith("s_cbranch") or self.op == "s_call_b64":   # (a call: into the callee, whose s_setpc ends that path, and past it)
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


def kernels_disassembly(dis_text: str):
    """the same from `llvm-objdump -d --symbolize-operands --no-show-raw-insn` of a code object: no asm markers there, so EVERY
    vector-memory load is checked (the compiler's own must pass as well: it waits before it reuses a destination)"""
    fn = re.compile(r"^[0-9a-f]{16} <([^>]+)>:$")
    blocks, cur, sym = [], None, None
    for n, line in enumerate(dis_text.split("\n"), 1):
        m = fn.match(line)
        if m:
            name = m.group(1)
            if re.fullmatch(r"L\d+", name):
                if cur is not None:
                    blocks.append(cur)
                    cur = (name, [])
                continue
            if sym is not None:
                blocks.append(cur)
                yield sym, blocks
            sym, blocks, cur = name, [], (name, [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        code = line.split("//")[0].strip()
        if code:
            cur[1].append(Insn(n, code, True))
    if sym is not None:
        blocks.append(cur)
        yield sym, blocks


def disassemble_library(lib: Path) -> str:
    """device code of a built libsavad.so: .hip_fatbin -> the gfx950 code object -> llvm-objdump"""
    llvm = Path("/opt/rocm/lib/llvm/bin")
    with tempfile.TemporaryDirectory() as d:
        fat, co = Path(d) / "fat.bin", Path(d) / "dev.co"
        subprocess.run([str(llvm / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
        listing = subprocess.run([str(llvm / "clang-offload-bundler"), "--list", "--type=o", f"--input={fat}"], check=True,
                                 capture_output=True, text=True).stdout.split()
        targets = [t for t in listing if "gfx950" in t]
        if len(targets) != 1:
            raise RuntimeError(f"{lib}: expected one gfx950 code object, found {listing}")
        subprocess.run([str(llvm / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={targets[0]}",
                        f"--output={co}"], check=True)
        return subprocess.run([str(llvm / "llvm-objdump"), "-d", "--symbolize-operands", "--no-show-raw-insn", str(co)], check=True,
                              capture_output=True, text=True).stdout


def check_kernel(blocks, count_stores: bool = False, domain: str = "vm", all_loads: bool = False):
    """-> (hazards, truncated = False); a hazard is (line, text, line of the pending load, its text).

    Data flow over the control-flow graph.  The state is, for every load that may be in flight, the number of vector-memory
    operations issued after it (its distance from the young end of the queue): an issue adds one to every distance, and
    `s_waitcnt vmcnt(N)` retires every load whose distance is >= N (loads return in order).  Where paths meet, a load keeps
    its SMALLEST distance -- the path on which it is youngest is the one on which it stays in flight longest -- which makes the
    result exact per load over all paths of the graph (infeasible ones included: conservative).
    count_stores: stores and atomics take their place in the queue (gfx9 has one counter for both and the compiler's own waits
    rely on their retiring in issue order); without it they are ignored, which is the stricter reading for hand-placed waits."""
    if domain == "dma":
        return check_dma_after_access(blocks), False
    lds = domain == "lgkm"   # the same flow over the DS queue: hand-issued ds_read against s_waitcnt lgkmcnt(N)
    cap = 16 if lds else 64  # the counter's range: anything older has returned
    index = {label: k for k, (label, _) in enumerate(blocks)}
    loads, hazards = {}, {}
    state_in = [None] * len(blocks)
    state_in[0] = {}
    work = [0]

    def merge_into(k, st):
        cur = state_in[k]
        if cur is None:
            state_in[k] = dict(st)
            work.append(k)
            return
        changed = False
        for key, r in st.items():
            if cur.get(key, 1 << 30) > r:
                cur[key] = r
                changed = True
        if changed:
            work.append(k)

    while work:
        k = work.pop()
        st = dict(state_in[k])
        ended = False
        for ins in blocks[k][1]:
            cnt = ins.lgkmcnt if lds else ins.vmcnt
            if cnt is not None:
                st = {key: r for key, r in st.items() if r < cnt}
                if (ins.vmcnt is not None or ins.lgkmcnt is not None):
                    continue
            for pl in st:
                if ins.touched & (loads[pl].dsdst if lds else loads[pl].dst):
                    hazards.setdefault((ins.line, pl), (ins.line, ins.text, pl, loads[pl].text))
                elif lds and not ins.is_dsread and (regs(ins.text[len(ins.op):]) & loads[pl].dsdst):
                    hazards.setdefault((ins.line, pl), (ins.line, ins.text, pl, loads[pl].text))
            queued = ins.is_ds if lds else (ins.is_load or (count_stores and ins.is_store))
            if queued:
                st = {key: r + 1 for key, r in st.items() if r + 1 < cap}   # (older ones have returned: the counter's range)
                mine = (ins.is_dsread and (ins.in_asm or all_loads) and ins.dsdst) if lds else (ins.is_load and ins.in_asm and ins.dst)
                if mine:
                    loads[ins.line] = ins
                    st[ins.line] = 0
            if ins.kind == "jump":
                if ins.target in index:
                    merge_into(index[ins.target], st)
                ended = True
                break
            if ins.kind == "cond" and ins.target in index:   # the taken edge leaves with the queue as it is HERE, not at the block's end
                merge_into(index[ins.target], st)
            if ins.kind == "end":
                ended = True
                break
        if not ended and k + 1 < len(blocks):
            merge_into(k + 1, st)
    return sorted(hazards.values()), False


def check_dma_after_access(blocks):
    """LDS-DMA rule.  Per path a three-state machine: clear -(DMA)-> issued -(ds_read / ds_write)-> issued + accessed; `s_barrier`
    returns to clear; a DMA in the third state is reported: the wave has requested a block, worked on LDS, and requests again without
    having met the other waves in between -- the second request may land on a block they still read.  (A read the compiler hoists
    between a barrier and the first DMA behind it is not reported: the machine is still clear there.)  Forward data flow; where
    paths meet the more advanced state wins."""
    index = {label: k for k, (label, _) in enumerate(blocks)}
    seen = {0: (0, None, None)}
    work = [0]
    hazards = {}

    def merge_into(k, st):
        cur = seen.get(k)
        if cur is None or st[0] > cur[0]:
            seen[k] = st
            work.append(k)

    while work:
        k = work.pop()
        state, dma, acc = seen[k]
        ended = False
        for ins in blocks[k][1]:
            if ins.is_barrier:
                state, dma, acc = 0, None, None
            elif ins.is_dma:
                if state == 2:
                    hazards.setdefault(ins.line, (ins.line, ins.text, acc.line, acc.text + f"   (behind the DMA of line {dma.line}, no s_barrier since)"))
                elif state == 0:
                    state, dma = 1, ins
            elif ins.is_ds and state == 1:
                state, acc = 2, ins
            if ins.kind == "jump":
                if ins.target in index:
                    merge_into(index[ins.target], (state, dma, acc))
                ended = True
                break
            if ins.kind == "cond" and ins.target in index:
                merge_into(index[ins.target], (state, dma, acc))
            if ins.kind == "end":
                ended = True
                break
        if not ended and k + 1 < len(blocks):
            merge_into(k + 1, (state, dma, acc))
    return sorted(hazards.values())


def dma_barriers_in_flight(blocks, count_stores: bool = True, cap: int = 6):
    """LDS-DMA PUBLISH rule.  A wave publishes its share of an LDS block to the other waves by waiting for its own DMA pieces
    (`s_waitcnt vmcnt(N)`, N = the vector-memory operations it issued after them: they retire in order) and then meeting the others
    at an `s_barrier`.  -> {line of a DMA instruction: the largest number of barriers the wave passes, over all paths, while that DMA
    may still be in flight}.  Data flow as in check_kernel, but per DMA a SET of (distance from the young end of the queue, barriers
    passed) pairs -- one per family of paths; a pair that is no younger and has passed no more barriers than another is dropped --
    because the two components must not be mixed across paths (a prologue DMA enters a loop young, and grows old inside it).
    A sufficient vmcnt retires a pair; at a barrier every remaining pair counts one more; `cap` = reported as "never waited for".
    A ring that runs DEPTH blocks ahead legitimately carries a DMA across DEPTH - 1 barriers; one more than that is a block
    published before it has landed (the bound per kernel: DMA_PUBLISH_BARRIERS)."""
    index = {label: k for k, (label, _) in enumerate(blocks)}
    state_in = [None] * len(blocks)
    state_in[0] = {}
    work = [0]
    worst = {}

    def prune(pairs):
        keep = []
        for d, b in sorted(pairs, key=lambda t: (t[0], -t[1])):   # youngest first, then most barriers
            if not any(kd <= d and kb >= b for kd, kb in keep):
                keep.append((d, b))
        return frozenset(keep)

    def merge_into(k, st):
        cur = state_in[k]
        if cur is None:
            state_in[k] = dict(st)
            work.append(k)
            return
        changed = False
        for key, pairs in st.items():
            merged = prune(set(cur.get(key, ())) | set(pairs))
            if merged != cur.get(key):
                cur[key] = merged
                changed = True
        if changed:
            work.append(k)

    while work:
        k = work.pop()
        st = dict(state_in[k])
        ended = False
        for ins in blocks[k][1]:
            if ins.vmcnt is not None:
                st = {key: frozenset(p for p in pairs if p[0] < ins.vmcnt) for key, pairs in st.items()}
                st = {key: pairs for key, pairs in st.items() if pairs}
                continue
            if ins.is_barrier:
                nxt = {}
                for key, pairs in st.items():
                    worst[key] = max(worst.get(key, 0), max(b for _, b in pairs) + 1)
                    moved = frozenset((d, b + 1) for d, b in pairs if b + 1 < cap)
                    if moved:
                        nxt[key] = moved
                st = nxt
            if ins.is_load or (count_stores and ins.is_store):
                st = {key: prune({(d + 1, b) for d, b in pairs if d + 1 < 64}) for key, pairs in st.items()}
                st = {key: pairs for key, pairs in st.items() if pairs}
                if ins.is_dma:
                    st[ins.line] = prune(set(st.get(ins.line, ())) | {(0, 0)})
                    worst.setdefault(ins.line, 0)
            if ins.kind == "jump":
                if ins.target in index:
                    merge_into(index[ins.target], st)
                ended = True
                break
            if ins.kind == "cond" and ins.target in index:
                merge_into(index[ins.target], st)
            if ins.kind == "end":
                ended = True
                break
        if not ended and k + 1 < len(blocks):
            merge_into(k + 1, st)
    return worst


# symbol substring -> barriers a DMA may legitimately cross in flight (default 0: landed before the next barrier), and why
_RUNTIME_WAITS = "the counted ring wait is picked at run time from the number of blocks still to come; the analysis cannot tie that to the loop's issue condition and sees paths that never wait"
_LAST_PASS = "the tile loop's last pass requests nothing (`more` is false) and leaves through the hand-over barrier: the path 'requested, then left' is infeasible"
_RING3 = "the 3-slot ring of the fp32s kernels runs two slots ahead: a slot's pieces cross the barrier of the slot in front of it"
DMA_PUBLISH_BARRIERS = {
    "21input_qkv_kernel_f32s": (1, _RING3),
    "25attention_row_kernel_f32s": (1, _RING3),
    "25attention_row_kernel_f32sILb0ELb0E": (2, _RING3 + "; its prologue requests THREE slots (K(0) alone in front of the stream): the third crosses the first two barriers"),
    "25attention_row_kernel_f32sILb1ELb0E": (2, _RING3 + "; its prologue requests THREE slots (K(0) alone in front of the stream): the third crosses the first two barriers"),
    "26packed_forward_kernel_f32s": (1, _RING3),
    "attention_pw_kernel_bf16": (99, "a generated instruction stream with a three-stage K / V pipeline; its waits and barriers are modelled instruction by instruction in scripts/gfx950_sim.py"),
    "5savad20attention_row_kernelI": (1, _LAST_PASS),
    "25attention_row_kernel_bf16ILb0ELi4E": (1, _LAST_PASS),
    "25attention_row_kernel_bf16ILb1ELi4E": (1, _LAST_PASS),
    "15row_kernel_bf16ILb0ELi8E": (2, "the 4-slot ring of the 8-wave variant runs three blocks ahead: a block's pieces cross the barriers of the two blocks in front of it"),
    "15row_kernel_bf16ILb1ELi8E": (99, _RUNTIME_WAITS),
    "21attention_kernel_bf16ILi8E": (99, _RUNTIME_WAITS),
    "25attention_row_kernel_bf16ILb0ELi8E": (99, _RUNTIME_WAITS),
    "25attention_row_kernel_bf16ILb1ELi8E": (99, _RUNTIME_WAITS),
    "26packed_forward_kernel_bf16ILi4ELi4ELi4E": (99, _RUNTIME_WAITS),
    "26packed_forward_kernel_bf16ILi8ELi4ELi0E": (99, _RUNTIME_WAITS),
}


def check_dma_publish(sym, blocks, count_stores=True):
    allowed = max([v[0] for key, v in DMA_PUBLISH_BARRIERS.items() if key in sym] or [0])
    text = {i.line: i.text for _, b in blocks for i in b}
    return [(line, text[line], line, f"crosses {n} barrier(s) in flight (allowed for this kernel: {allowed}): published before it has landed")
            for line, n in sorted(dma_barriers_in_flight(blocks, count_stores).items()) if n > allowed]


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
    return _check(kernels(path.read_text()), False)


def check_library(lib: Path):
    """the same for the device code inside a built library (every kernel with vector-memory loads, every load)"""
    return _check(kernels_disassembly(disassemble_library(lib)), True)


def _check(kernel_iter, count_stores):
    """{symbol: (hazards, truncated)}: vector-memory loads, hand-issued LDS reads and LDS-DMA destinations together (a hazard's text
    says which: the pending instruction is a global_load, a ds_read, or -- for the DMA rule -- the LDS access not yet fenced)"""
    report = {}
    for sym, blocks in kernel_iter:
        insns = [i for _, b in blocks for i in b]
        hazards = []
        if any(i.is_load and i.in_asm and not i.is_dma for i in insns):
            hazards += check_kernel(blocks, count_stores=count_stores or any(key in sym for key in STORES_IN_QUEUE))[0]
        if any(i.is_dsread and i.in_asm for i in insns):
            hazards += check_kernel(blocks, domain="lgkm", all_loads=count_stores)[0]   # (library form: the compiler's own reads too)
        if any(i.is_dma for i in insns) and not any(key in sym for key in DMA_DISJOINT):
            hazards += check_kernel(blocks, domain="dma")[0]
        if any(i.is_dma for i in insns):
            hazards += check_dma_publish(sym, blocks)
        if hazards or any((i.is_load and i.in_asm) or (i.is_dsread and i.in_asm) for i in insns):
            report[sym] = (sorted(hazards), False)
    return report


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--asm", type=Path, help="assembly file to check (default: compile csrc/savad.hip)")
    ap.add_argument("--lib", type=Path, help="check the device code inside this built library instead (all loads, not only asm-issued ones)")
    ap.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args()
    report = check_library(args.lib) if args.lib else check_file(args.asm or compile_asm(args.verbose))
    bad = 0
    for sym, (hazards, truncated) in report.items():
        print(f"{sym}: {len(hazards)} hazard(s)" + (" (path search truncated)" if truncated else ""))
        for line, text, pl, ptext in hazards[: (1000 if args.verbose else 8)]:
            print(f"    line {line}: {text}\n        touches the destination of line {pl}: {ptext}")
        bad += len(hazards) + (1 if truncated else 0)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
