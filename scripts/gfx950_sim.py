#!/usr/bin/env python3
"""A functional model of the gfx950 instruction SUBSET that scripts/gen_attn_pw.py emits -- test infrastructure, not
product: it runs the generated stream of attention_pw_kernel_bf16 on the CPU (one workgroup = four waves of 64 lanes,
numpy over the lanes) so that the LOGIC of a generator change (item cursors, ring addresses, register assignment,
barrier pairing, counted waits) is checked here, before a GPU call is spent on it.  tests/test_pw_stream_sim.py drives
it against an fp64 attention on small shapes.

What it models
  * registers v0..255 / a0..255 (64 lanes x 32 bit), s0..127, vcc, scc, m0; exec is all ones (the stream never masks);
  * global memory as one flat byte array, LDS as 160 KiB per workgroup;
  * the matrix instruction v_mfma_f32_32x32x16_bf16 with the register layouts of the ISA (A / B: lane (i, h) holds
    k = 8h .. 8h+7 of row / column i; C / D: lane (j, h), register r <-> row 8 (r >> 2) + 4h + (r & 3), column j);
  * ASYNCHRONOUS memory the way the stream's waits assume it: every vector-memory instruction joins the wave's in-order
    queue and takes effect only when an `s_waitcnt vmcnt(N)` retires it (the LATEST moment the hardware allows);
    LDS bytes a DMA piece has been issued for are tracked as in flight, then landed-for-the-issuer, and visible to the
    other waves only after a barrier both have passed -- a ds_read of bytes that are not visible yet, or a DMA piece
    issued over bytes another wave may still read before the next barrier, is reported (`Hazard`);
  * registers that wait for a load (VMEM or LDS) are poisoned until the matching s_waitcnt: reading one is a `Hazard`.
What it does NOT model: timing, and therefore the manually inserted wait states (MFMA -> VALU, VALU -> permlane ...):
those follow the patterns documented at the top of gen_attn_pw.py.  Transcendentals are numpy's (1 ulp from the chip's).
"""
from __future__ import annotations

import re

import numpy as np

LDS_BYTES = 160 * 1024
MASK32 = 0xFFFFFFFF


class Hazard(Exception):
    pass


class SimError(Exception):
    pass


def f2u(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def u2f(x):
    return np.asarray(x, dtype=np.uint32).view(np.float32)


def bf16_round(x):
    """fp32 array -> bf16 bits (uint32 holding 16 bits), round to nearest even, NaN kept quiet"""
    u = f2u(x).astype(np.uint64)
    nan = np.isnan(u2f(u.astype(np.uint32)))
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF
    r = np.where(nan, 0x7FC0, r)
    return r.astype(np.uint32)


def bf16_to_f32(bits16):
    return u2f((np.asarray(bits16, dtype=np.uint32) << 16).astype(np.uint32))


_REG_RE = re.compile(r"^(-?)([vas])(?:(\d+)|\[(\d+):(\d+)\])$")


class Operand:
    __slots__ = ("kind", "base", "n", "neg", "value")

    def __init__(self, kind, base=0, n=1, neg=False, value=0):
        self.kind, self.base, self.n, self.neg, self.value = kind, base, n, neg, value

    def __repr__(self):
        return f"{'-' if self.neg else ''}{self.kind}{self.base}:{self.n}" if self.kind in "vas" else f"{self.kind}({self.value})"


def parse_operand(tok):
    tok = tok.strip()
    m = _REG_RE.match(tok)
    if m:
        neg, kind, single, lo, hi = m.groups()
        if single is not None:
            return Operand(kind, int(single), 1, bool(neg))
        return Operand(kind, int(lo), int(hi) - int(lo) + 1, bool(neg))
    if tok in ("vcc", "m0", "exec", "exec_hi", "exec_lo", "scc"):
        return Operand(tok)
    if tok.startswith(".L"):
        return Operand("label", value=tok)
    try:
        if tok.lower().startswith(("0x", "-0x")):
            return Operand("lit", value=int(tok, 16) & MASK32)
        if any(c in tok for c in ".eE") and not tok.startswith(".L"):
            return Operand("lit", value=int(f2u(np.float32(float(tok)))))
        return Operand("lit", value=int(tok) & MASK32)
    except ValueError:
        raise SimError(f"operand not understood: {tok!r}")


class Instr:
    __slots__ = ("op", "ops", "offset", "text", "line")

    def __init__(self, op, ops, offset, text, line):
        self.op, self.ops, self.offset, self.text, self.line = op, ops, offset, text, line


def parse_program(text, operand_map=None):
    """text of the .inc (or the bare asm) -> (instructions, labels).  operand_map: {"%0": "s[0:1]", ...}"""
    body = text
    if 'R"ASMPW(' in body:
        body = body.split('R"ASMPW(', 1)[1].split(')ASMPW"', 1)[0]
    instrs, labels = [], {}
    for ln, raw in enumerate(body.split("\n"), 1):
        line = raw.split("//", 1)[0].strip()
        if not line:
            continue
        if line.endswith(":"):
            labels[line[:-1]] = len(instrs)
            continue
        if operand_map:
            line = re.sub(r"%(\d+)", lambda m: operand_map["%" + m.group(1)], line)
        parts = line.split(None, 1)
        op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        offset = 0
        mo = re.search(r"\boffset:(\d+)", rest)
        if mo:
            offset = int(mo.group(1))
            rest = rest[:mo.start()] + rest[mo.end():]
        rest = re.sub(r"\b(nt|sc0|sc1|glc|slc)\b", "", rest)   # cache-policy bits: no functional meaning
        if op == "s_waitcnt":
            ops = rest.strip()
        else:
            ops = [parse_operand(t) for t in rest.split(",") if t.strip()] if rest.strip() else []
        instrs.append(Instr(op, ops, offset, line, ln))
    return instrs, labels


class Wave:
    def __init__(self, wid):
        self.wid = wid
        self.v = np.zeros((256, 64), dtype=np.uint32)
        self.a = np.zeros((256, 64), dtype=np.uint32)
        self.s = np.zeros(128, dtype=np.uint64)  # 32-bit values kept in 64-bit cells
        self.vcc = 0
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.vmq = []          # pending vector-memory operations, oldest first
        self.lgkm = []         # pending LDS reads: (kind, base, n)
        self.poison_v = np.zeros(256, dtype=bool)
        self.poison_a = np.zeros(256, dtype=bool)
        self.icount = 0
        self.counts = {}


class Workgroup:
    """four waves, one LDS, shared global memory; run() interleaves the waves from barrier to barrier"""

    def __init__(self, instrs, labels, gmem, nwaves=4, strict=True):
        self.instrs, self.labels, self.gmem = instrs, labels, gmem
        self.lds = np.zeros(LDS_BYTES + 4096, dtype=np.uint8)
        # per LDS dword: 0 = ordinary, else (issuer + 1) with state in lds_state: 1 in flight, 2 landed for the issuer
        self.lds_owner = np.zeros((LDS_BYTES + 4096) // 4, dtype=np.int8)
        self.lds_state = np.zeros((LDS_BYTES + 4096) // 4, dtype=np.int8)
        self.lds_read_epoch = np.full((LDS_BYTES + 4096) // 4, -1, dtype=np.int64)  # barrier epoch of the last ds_read
        self.lds_reader = np.full((LDS_BYTES + 4096) // 4, -1, dtype=np.int8)
        self.epoch = 0
        self.waves = [Wave(w) for w in range(nwaves)]
        self.strict = strict
        self.max_instr = 50_000_000
        self.hazards = []
        self.label_at = {}
        for name, idx in labels.items():
            self.label_at.setdefault(idx, []).append(name)
        self.label_hits = {}

    # ------------------------------------------------------------------ helpers
    def hazard(self, w, ins, msg):
        text = f"wave {w.wid} pc {w.pc} (line {ins.line}: {ins.text}): {msg}"
        if self.strict:
            raise Hazard(text)
        self.hazards.append(text)

    def _check_read(self, w, ins, o):
        if o.kind == "v" and w.poison_v[o.base:o.base + o.n].any():
            self.hazard(w, ins, f"reads v[{o.base}:{o.base + o.n - 1}] while a load into it is outstanding")
        if o.kind == "a" and w.poison_a[o.base:o.base + o.n].any():
            self.hazard(w, ins, f"reads a[{o.base}:{o.base + o.n - 1}] while a load into it is outstanding")

    def rd32(self, w, ins, o):
        """one 32-bit source -> uint32[64] (vector registers) or a python int (scalars, literals)"""
        if o.kind == "v":
            self._check_read(w, ins, o)
            val = w.v[o.base]
        elif o.kind == "a":
            self._check_read(w, ins, o)
            val = w.a[o.base]
        elif o.kind == "s":
            val = int(w.s[o.base]) & MASK32
        elif o.kind == "lit":
            val = o.value
        elif o.kind == "vcc":
            val = w.vcc & MASK32
        elif o.kind == "m0":
            val = w.m0
        else:
            raise SimError(f"cannot read {o} in {ins.text}")
        return val

    def rdf(self, w, ins, o):
        x = self.rd32(w, ins, o)
        f = u2f(np.full(64, x, dtype=np.uint32) if isinstance(x, int) else x)
        return -f if o.neg else f

    def rdu(self, w, ins, o):
        x = self.rd32(w, ins, o)
        return np.full(64, x, dtype=np.uint32) if isinstance(x, int) else x

    def rd64s(self, w, o):
        if o.kind == "s":
            return (int(w.s[o.base]) & MASK32) | ((int(w.s[o.base + 1]) & MASK32) << 32)
        if o.kind == "vcc":
            return w.vcc
        if o.kind == "lit":
            v = o.value
            return v | (0xFFFFFFFF00000000 if v & 0x80000000 else 0)  # sign-extended inline constant
        if o.kind == "exec":
            return (1 << 64) - 1
        raise SimError(f"cannot read 64-bit {o}")

    def wr64s(self, w, o, val):
        val &= (1 << 64) - 1
        if o.kind == "s":
            w.s[o.base] = val & MASK32
            w.s[o.base + 1] = val >> 32
        elif o.kind == "vcc":
            w.vcc = val
        elif o.kind == "exec":
            if val != (1 << 64) - 1:
                raise SimError("exec is modelled as all ones")
        else:
            raise SimError(f"cannot write 64-bit {o}")

    def wrs(self, w, o, val):
        val &= MASK32
        if o.kind == "s":
            w.s[o.base] = val
        elif o.kind == "m0":
            w.m0 = val
        elif o.kind == "vcc":
            w.vcc = val
        elif o.kind in ("exec_hi", "exec_lo"):
            if val != MASK32:
                raise SimError("exec is modelled as all ones")
        else:
            raise SimError(f"cannot write scalar {o}")

    def wrv(self, w, o, arr, idx=0):
        arr = np.asarray(arr)
        if arr.dtype == np.float32:
            arr = arr.view(np.uint32)
        if o.kind == "v":
            w.v[o.base + idx] = arr
            w.poison_v[o.base + idx] = False
        elif o.kind == "a":
            w.a[o.base + idx] = arr
            w.poison_a[o.base + idx] = False
        else:
            raise SimError(f"cannot write vector {o}")

    # ------------------------------------------------------------------ memory
    def gaddr(self, w, ins, voff, sbase):
        base = self.rd64s(w, sbase)
        off = self.rdu(w, ins, voff).astype(np.int64)
        return base + off + ins.offset

    def gread16(self, addrs):
        if (addrs < 4096).any() or (addrs + 16 > self.gmem.size).any():
            raise SimError(f"global read out of range: {addrs.min():#x}..{addrs.max():#x}")
        idx = addrs[:, None] + np.arange(16)[None, :]
        return self.gmem[idx].copy().view(np.uint32).reshape(64, 4)

    def retire(self, w, keep):
        while len(w.vmq) > keep:
            op = w.vmq.pop(0)
            kind = op[0]
            if kind == "load":
                _, o, data = op
                for k in range(4):
                    if o.kind == "v":
                        w.v[o.base + k] = data[:, k]
                        w.poison_v[o.base + k] = False
                    else:
                        w.a[o.base + k] = data[:, k]
                        w.poison_a[o.base + k] = False
            elif kind == "dma":
                _, lds_addr, data = op
                self.lds[lds_addr:lds_addr + 1024] = data.reshape(-1).view(np.uint8)
                d0 = lds_addr // 4
                self.lds_state[d0:d0 + 256] = 2
            elif kind == "store":
                pass

    def lds_touch_read(self, w, ins, addrs, nbytes):
        """addrs: int64[64] byte addresses; checks visibility, records the read"""
        if (addrs < 0).any() or (addrs + nbytes > LDS_BYTES).any():
            raise SimError(f"LDS read out of range {addrs.min()}..{addrs.max()} in {ins.text}")
        d = (addrs[:, None] // 4 + np.arange(nbytes // 4)[None, :]).reshape(-1)
        own, st = self.lds_owner[d], self.lds_state[d]
        bad = (st == 1) | ((st == 2) & (own != w.wid + 1))
        if bad.any():
            k = int(np.argmax(bad))
            self.hazard(w, ins, f"ds_read of LDS byte {int(d[k]) * 4}: DMA piece of wave {int(own[k]) - 1} is "
                                f"{'in flight' if st[k] == 1 else 'landed but not published by a barrier'}")
        self.lds_read_epoch[d] = self.epoch
        self.lds_reader[d] = w.wid

    def lds_touch_write(self, w, ins, d0, ndw, what):
        """a write (DMA issue or ds_write) over dwords a DIFFERENT wave has read since the last barrier may overtake the read"""
        sl = slice(d0, d0 + ndw)
        clash = (self.lds_read_epoch[sl] == self.epoch) & (self.lds_reader[sl] != w.wid)
        if clash.any():
            k = int(np.argmax(clash))
            self.hazard(w, ins, f"{what} over LDS byte {(d0 + k) * 4}, which wave {int(self.lds_reader[d0 + k])} read in this barrier epoch")

    # ------------------------------------------------------------------ execution
    def run(self):
        total = 0
        while True:
            progressed = False
            for w in self.waves:
                if w.done or w.at_barrier:
                    continue
                self.run_wave(w)
                progressed = True
            live = [w for w in self.waves if not w.done]
            if not live:
                break
            if all(w.at_barrier for w in live):
                if len(live) != len(self.waves):
                    raise Hazard(f"barrier reached by waves {[w.wid for w in live]} after waves "
                                 f"{[w.wid for w in self.waves if w.done]} ended")
                # publish: pieces that landed for their issuer become visible to everybody
                self.lds_owner[self.lds_state == 2] = 0
                self.lds_state[self.lds_state == 2] = 0
                self.epoch += 1
                for w in live:
                    w.at_barrier = False
                continue
            if not progressed:
                raise SimError("no progress")
            total += 1
        for w in self.waves:
            if w.vmq and any(op[0] != "store" for op in w.vmq):
                pending = [op[0] for op in w.vmq if op[0] != "store"]
                if any(p == "load" for p in pending):
                    raise Hazard(f"wave {w.wid} ended with loads outstanding")

    def run_wave(self, w):
        instrs = self.instrs
        n = len(instrs)
        while True:
            if w.pc >= n:
                w.done = True
                return
            ins = instrs[w.pc]
            if w.pc in self.label_at:
                for name in self.label_at[w.pc]:
                    self.label_hits[name] = self.label_hits.get(name, 0) + 1
            w.icount += 1
            if w.icount > self.max_instr:
                raise SimError(f"wave {w.wid}: instruction budget exceeded (endless loop?) at {ins.text}")
            w.counts[ins.op] = w.counts.get(ins.op, 0) + 1
            nxt = self.step(w, ins)
            if nxt == "barrier":
                w.pc += 1
                w.at_barrier = True
                return
            w.pc = w.pc + 1 if nxt is None else nxt

    def step(self, w, ins):  # noqa: C901 -- a flat dispatch is the clearest form here
        op, o = ins.op, ins.ops
        R, F, U = self.rd32, self.rdf, self.rdu
        # ---------------------------------------------------------------- scalar ALU
        if op == "s_nop":
            return None
        if op == "s_mov_b32":
            self.wrs(w, o[0], R(w, ins, o[1]))
            return None
        if op == "s_mov_b64":
            self.wr64s(w, o[0], self.rd64s(w, o[1]))
            return None
        if op in ("s_add_u32", "s_addc_u32", "s_sub_u32", "s_subb_u32", "s_add_i32", "s_sub_i32"):
            a, b = R(w, ins, o[1]), R(w, ins, o[2])
            if op == "s_add_u32":
                r = a + b
                w.scc = int(r > MASK32)
            elif op == "s_addc_u32":
                r = a + b + w.scc
                w.scc = int(r > MASK32)
            elif op == "s_sub_u32":
                r = a - b
                w.scc = int(b > a)
            elif op == "s_subb_u32":
                r = a - b - w.scc
                w.scc = int(b + w.scc > a)
            else:
                sa, sb = _s32(a), _s32(b)
                r = sa + sb if op == "s_add_i32" else sa - sb
                w.scc = int(not (-2**31 <= r < 2**31))
            self.wrs(w, o[0], r)
            return None
        if op == "s_mul_i32":
            self.wrs(w, o[0], (_s32(R(w, ins, o[1])) * _s32(R(w, ins, o[2]))))
            return None
        if op == "s_mul_hi_u32":
            self.wrs(w, o[0], (R(w, ins, o[1]) * R(w, ins, o[2])) >> 32)
            return None
        if op in ("s_lshl_b32", "s_lshr_b32", "s_and_b32", "s_or_b32", "s_xor_b32"):
            a, b = R(w, ins, o[1]), R(w, ins, o[2])
            r = {"s_lshl_b32": (a << (b & 31)), "s_lshr_b32": a >> (b & 31), "s_and_b32": a & b, "s_or_b32": a | b,
                 "s_xor_b32": a ^ b}[op] & MASK32
            w.scc = int(r != 0)
            self.wrs(w, o[0], r)
            return None
        if op in ("s_or_b64", "s_and_b64"):
            a, b = self.rd64s(w, o[1]), self.rd64s(w, o[2])
            r = (a | b) if op == "s_or_b64" else (a & b)
            w.scc = int(r != 0)
            self.wr64s(w, o[0], r)
            return None
        if op in ("s_min_u32", "s_max_u32", "s_min_i32", "s_max_i32"):
            a, b = R(w, ins, o[1]), R(w, ins, o[2])
            if op.endswith("i32"):
                ka, kb = _s32(a), _s32(b)
            else:
                ka, kb = a, b
            first = ka <= kb if "min" in op else ka >= kb
            w.scc = int(first)
            self.wrs(w, o[0], a if first else b)
            return None
        if op == "s_ff1_i32_b32":
            x = R(w, ins, o[1])
            self.wrs(w, o[0], (x & -x).bit_length() - 1 if x else MASK32)
            return None
        if op == "s_cselect_b32":
            self.wrs(w, o[0], R(w, ins, o[1]) if w.scc else R(w, ins, o[2]))
            return None
        if op.startswith("s_cmp_") and op.endswith("_u64"):
            a, b = self.rd64s(w, o[0]), self.rd64s(w, o[1])
            w.scc = int((a == b) if "eq" in op else (a != b))
            return None
        if op.startswith("s_cmp_"):
            a, b = R(w, ins, o[0]), R(w, ins, o[1])
            if op.endswith("i32"):
                a, b = _s32(a), _s32(b)
            cond = op[6:8]
            w.scc = int({"eq": a == b, "lg": a != b, "lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b}[cond])
            return None
        if op == "s_bitcmp1_b32":
            w.scc = int((R(w, ins, o[0]) >> (R(w, ins, o[1]) & 31)) & 1)
            return None
        if op == "s_bitcmp0_b32":
            w.scc = int(not ((R(w, ins, o[0]) >> (R(w, ins, o[1]) & 31)) & 1))
            return None
        if op == "s_branch":
            return self.labels[o[0].value]
        if op == "s_cbranch_scc1":
            return self.labels[o[0].value] if w.scc else None
        if op == "s_cbranch_scc0":
            return None if w.scc else self.labels[o[0].value]
        if op == "s_call_b64":
            self.wr64s(w, o[0], w.pc + 1)
            return self.labels[o[1].value]
        if op == "s_setpc_b64":
            return self.rd64s(w, o[0])
        if op == "s_barrier":
            return "barrier"
        if op == "s_waitcnt":
            txt = ins.ops
            m = re.search(r"vmcnt\((\d+)\)", txt)
            if m:
                self.retire(w, int(m.group(1)))
            m = re.search(r"lgkmcnt\((\d+)\)", txt)
            if m:
                keep = int(m.group(1))
                while len(w.lgkm) > keep:
                    kind, base, cnt = w.lgkm.pop(0)
                    (w.poison_v if kind == "v" else w.poison_a)[base:base + cnt] = False
            return None
        if op in ("s_memtime", "s_memrealtime", "s_setprio", "s_sleep"):
            return None
        # ---------------------------------------------------------------- vector ALU
        if op == "v_mov_b32":
            self.wrv(w, o[0], U(w, ins, o[1]).copy())
            return None
        if op in ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_max_f32", "v_min_f32"):
            a, b = F(w, ins, o[1]), F(w, ins, o[2])
            with np.errstate(all="ignore"):
                r = {"v_add_f32": np.add, "v_sub_f32": np.subtract, "v_mul_f32": np.multiply, "v_max_f32": _fmax,
                     "v_min_f32": _fmin}[op](a, b)
            self.wrv(w, o[0], r.astype(np.float32))
            return None
        if op in ("v_pk_add_f32", "v_pk_mul_f32"):   # two fp32 per lane in an aligned register pair (default op_sel: low with low, high with high)
            for o_ in o[:3]:
                if o_.kind != "v" or o_.n != 2 or o_.base % 2:
                    raise SimError(f"{ins.text}: packed fp32 operands are aligned VGPR pairs here")
            self._check_read(w, ins, o[1])
            self._check_read(w, ins, o[2])
            fn = np.add if op == "v_pk_add_f32" else np.multiply
            with np.errstate(all="ignore"):
                res = [fn(u2f(w.v[o[1].base + h]), u2f(w.v[o[2].base + h])).astype(np.float32) for h in range(2)]
            for h in range(2):
                self.wrv(w, o[0], res[h], idx=h)
            return None
        if op == "v_max3_f32":
            self.wrv(w, o[0], _fmax(_fmax(F(w, ins, o[1]), F(w, ins, o[2])), F(w, ins, o[3])).astype(np.float32))
            return None
        if op in ("v_fma_f32", "v_fmac_f32"):
            if op == "v_fma_f32":
                a, b, c = F(w, ins, o[1]), F(w, ins, o[2]), F(w, ins, o[3])
            else:
                a, b, c = F(w, ins, o[1]), F(w, ins, o[2]), F(w, ins, o[0])
            with np.errstate(all="ignore"):
                r = (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
            self.wrv(w, o[0], r)
            return None
        if op == "v_exp_f32":
            with np.errstate(all="ignore"):
                self.wrv(w, o[0], np.exp2(F(w, ins, o[1]).astype(np.float64)).astype(np.float32))
            return None
        if op == "v_rcp_f32":
            with np.errstate(all="ignore"):
                self.wrv(w, o[0], (np.float32(1.0) / F(w, ins, o[1])).astype(np.float32))
            return None
        if op == "v_cvt_pk_bf16_f32":
            lo, hi = bf16_round(F(w, ins, o[1])), bf16_round(F(w, ins, o[2]))
            self.wrv(w, o[0], (lo | (hi << 16)).astype(np.uint32))
            return None
        if op == "v_cndmask_b32":
            a, b = U(w, ins, o[1]), U(w, ins, o[2])
            sel = o[3]
            mask = self.rd64s(w, sel)
            bits = np.array([(mask >> i) & 1 for i in range(64)], dtype=bool)
            self.wrv(w, o[0], np.where(bits, b, a).astype(np.uint32))
            return None
        if op.startswith("v_cmp_"):
            kind, ty = op[6:-4], op[-3:]
            dst = o[0]
            if ty == "f32":
                a, b = F(w, ins, o[1]), F(w, ins, o[2])
            elif ty == "i32":
                a, b = U(w, ins, o[1]).view(np.int32), U(w, ins, o[2]).view(np.int32)
            else:
                a, b = U(w, ins, o[1]), U(w, ins, o[2])
            with np.errstate(all="ignore"):
                res = {"lt": lambda: a < b, "le": lambda: a <= b, "gt": lambda: a > b, "ge": lambda: a >= b, "eq": lambda: a == b,
                       "lg": lambda: a != b, "ne": lambda: a != b, "nge": lambda: ~(a >= b), "ngt": lambda: ~(a > b),
                       "nlt": lambda: ~(a < b), "nle": lambda: ~(a <= b)}[kind]()
            val = 0
            for i in np.nonzero(res)[0]:
                val |= 1 << int(i)
            self.wr64s(w, dst, val)
            return None
        if op in ("v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32"):
            a, b = U(w, ins, o[1]).astype(np.uint64), U(w, ins, o[2]).astype(np.uint64)
            r = {"v_add_u32": lambda: a + b, "v_sub_u32": lambda: a - b, "v_and_b32": lambda: a & b, "v_or_b32": lambda: a | b,
                 "v_lshlrev_b32": lambda: b << (a & 31), "v_lshrrev_b32": lambda: b >> (a & 31)}[op]()
            self.wrv(w, o[0], (r & MASK32).astype(np.uint32))
            return None
        if op == "v_accvgpr_read_b32":
            self.wrv(w, o[0], U(w, ins, o[1]).copy())
            return None
        if op == "v_accvgpr_write_b32":
            self.wrv(w, o[0], U(w, ins, o[1]).copy())
            return None
        if op == "v_permlane32_swap_b32":
            a, b = U(w, ins, o[0]).copy(), U(w, ins, o[1]).copy()
            a_hi = a[32:].copy()
            a[32:] = b[:32]
            b[:32] = a_hi
            self.wrv(w, o[0], a)
            self.wrv(w, o[1], b)
            return None
        if op == "v_readlane_b32":
            self.wrs(w, o[0], int(U(w, ins, o[1])[R(w, ins, o[2]) & 63]))
            return None
        if op == "v_writelane_b32":
            arr = U(w, ins, o[0]).copy() if not (o[0].kind == "v" and w.poison_v[o[0].base]) else np.zeros(64, np.uint32)
            arr[R(w, ins, o[2]) & 63] = R(w, ins, o[1])
            self.wrv(w, o[0], arr)
            return None
        if op == "v_div_scale_f32":   # operands in the normal range: no scaling, vcc / sdst clear
            self.wrv(w, o[0], F(w, ins, o[2]).copy())
            self.wr64s(w, o[1], 0)
            return None
        if op == "v_div_fmas_f32":
            a, b, c = F(w, ins, o[1]), F(w, ins, o[2]), F(w, ins, o[3])
            self.wrv(w, o[0], (a.astype(np.float64) * b + c).astype(np.float32))
            return None
        if op == "v_div_fixup_f32":   # (quotient, denominator, numerator)
            with np.errstate(all="ignore"):
                self.wrv(w, o[0], (F(w, ins, o[3]) / F(w, ins, o[2])).astype(np.float32))
            return None
        if op == "v_mfma_f32_32x32x16_bf16":
            self.mfma(w, ins)
            return None
        # ---------------------------------------------------------------- LDS
        if op == "ds_read_b128":
            dst, addr = o[0], U(w, ins, o[1]).astype(np.int64) + ins.offset
            self.lds_touch_read(w, ins, addr, 16)
            idx = addr[:, None] + np.arange(16)[None, :]
            data = self.lds[idx].copy().view(np.uint32).reshape(64, 4)
            for k in range(4):
                self.wrv(w, dst, data[:, k], k)
            (w.poison_v if dst.kind == "v" else w.poison_a)[dst.base:dst.base + 4] = True
            w.lgkm.append((dst.kind, dst.base, 4))
            return None
        if op in ("ds_write_b128", "ds_write_b64", "ds_write_b32"):
            nreg = {"ds_write_b128": 4, "ds_write_b64": 2, "ds_write_b32": 1}[op]
            addr = U(w, ins, o[0]).astype(np.int64) + ins.offset
            if (addr < 0).any() or (addr + 4 * nreg > LDS_BYTES).any():
                raise SimError(f"LDS write out of range in {ins.text}")
            src = o[1]
            self._check_read(w, ins, src)
            bank = w.v if src.kind == "v" else w.a
            data = np.stack([bank[src.base + k] for k in range(nreg)], axis=1)  # [64][nreg]
            for lane in range(64):
                d0 = int(addr[lane]) // 4
                if self.lds_state[d0:d0 + nreg].any():
                    self.hazard(w, ins, f"ds_write over LDS byte {d0 * 4} that a DMA piece owns")
                self.lds_touch_write(w, ins, d0, nreg, "ds_write")
            idx = addr[:, None] + np.arange(4 * nreg)[None, :]
            self.lds[idx] = data.copy().view(np.uint8).reshape(64, 4 * nreg)
            # a ds_write is visible to the other waves after the writer's lgkmcnt(0) + a barrier: modelled like a DMA piece
            for lane in range(64):
                d0 = int(addr[lane]) // 4
                self.lds_owner[d0:d0 + nreg] = w.wid + 1
                self.lds_state[d0:d0 + nreg] = 2
            w.lgkm.append(("v", 0, 0))
            return None
        if op in ("ds_read_b32", "ds_read_b64"):
            nreg = 1 if op == "ds_read_b32" else 2
            dst, addr = o[0], U(w, ins, o[1]).astype(np.int64) + ins.offset
            self.lds_touch_read(w, ins, addr, 4 * nreg)
            idx = addr[:, None] + np.arange(4 * nreg)[None, :]
            data = self.lds[idx].copy().view(np.uint32).reshape(64, nreg)
            for k in range(nreg):
                self.wrv(w, dst, data[:, k], k)
            (w.poison_v if dst.kind == "v" else w.poison_a)[dst.base:dst.base + nreg] = True
            w.lgkm.append((dst.kind, dst.base, nreg))
            return None
        # ---------------------------------------------------------------- vector memory
        if op == "global_load_dwordx4":
            dst = o[0]
            data = self.gread16(self.gaddr(w, ins, o[1], o[2]))
            (w.poison_v if dst.kind == "v" else w.poison_a)[dst.base:dst.base + 4] = True
            w.vmq.append(("load", dst, data))
            return None
        if op == "global_store_dwordx4":
            addrs = self.gaddr(w, ins, o[0], o[2])
            src = o[1]
            self._check_read(w, ins, src)
            bank = w.v if src.kind == "v" else w.a
            data = np.stack([bank[src.base + k] for k in range(4)], axis=1)
            if (addrs < 4096).any() or (addrs + 16 > self.gmem.size).any():
                raise SimError(f"global store out of range in {ins.text}")
            idx = addrs[:, None] + np.arange(16)[None, :]
            self.gmem[idx] = data.copy().view(np.uint8).reshape(64, 16)
            w.vmq.append(("store",))
            return None
        if op == "global_store_dword":
            addrs = self.gaddr(w, ins, o[0], o[2])
            data = U(w, ins, o[1])
            idx = addrs[:, None] + np.arange(4)[None, :]
            self.gmem[idx] = data.copy().view(np.uint8).reshape(64, 4)
            w.vmq.append(("store",))
            return None
        if op == "global_load_lds_dwordx4":
            addrs = self.gaddr(w, ins, o[0], o[1])
            data = self.gread16(addrs)
            lds_addr = w.m0 + ins.offset       # + lane * 16: the piece is lane-linear
            if lds_addr < 0 or lds_addr + 1024 > LDS_BYTES or lds_addr % 16:
                raise SimError(f"LDS-DMA destination {lds_addr} out of range in {ins.text}")
            d0 = lds_addr // 4
            if self.lds_state[d0:d0 + 256].any():
                self.hazard(w, ins, f"DMA piece over LDS bytes {lds_addr}.. that an earlier piece still owns")
            self.lds_touch_write(w, ins, d0, 256, "DMA piece issued")
            self.lds_owner[d0:d0 + 256] = w.wid + 1
            self.lds_state[d0:d0 + 256] = 1
            w.vmq.append(("dma", lds_addr, data))
            return None
        raise SimError(f"instruction not modelled: {ins.text}")

    def mfma(self, w, ins):
        D, A, B, C = ins.ops
        for x in (A, B):
            self._check_read(w, ins, x)

        def frag(x):
            bank = w.v if x.kind == "v" else w.a
            regs = np.stack([bank[x.base + k] for k in range(4)], axis=1)       # [64 lanes][4]
            lo, hi = regs & 0xFFFF, regs >> 16
            vals = np.empty((64, 8), dtype=np.float32)
            vals[:, 0::2] = bf16_to_f32(lo)
            vals[:, 1::2] = bf16_to_f32(hi)
            return np.concatenate([vals[:32], vals[32:]], axis=1)              # [32 rows][16 k]

        am, bm = frag(A), frag(B)
        with np.errstate(all="ignore"):
            prod = am.astype(np.float64) @ bm.astype(np.float64).T            # [i][j]
        if C.kind == "lit":
            cm = np.zeros((32, 32), dtype=np.float64)
            if C.value != 0:
                raise SimError("non-zero literal C operand")
        else:
            self._check_read(w, ins, C)
            bank = w.v if C.kind == "v" else w.a
            cm = np.empty((32, 32), dtype=np.float64)
            for r in range(16):
                reg = u2f(bank[C.base + r])
                for h in range(2):
                    cm[8 * (r >> 2) + 4 * h + (r & 3), :] = reg[32 * h:32 * h + 32]
        with np.errstate(all="ignore"):
            dm = (prod + cm).astype(np.float32)
        for r in range(16):
            out = np.empty(64, dtype=np.float32)
            for h in range(2):
                out[32 * h:32 * h + 32] = dm[8 * (r >> 2) + 4 * h + (r & 3), :]
            self.wrv(w, D, out, r)


def _s32(x):
    x &= MASK32
    return x - (1 << 32) if x & 0x80000000 else x


def _fmax(a, b):
    """v_max_f32: the number when one operand is NaN"""
    with np.errstate(all="ignore"):
        return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.maximum(a, b)))


def _fmin(a, b):
    with np.errstate(all="ignore"):
        return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.minimum(a, b)))
