#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 instruction stream of `row_pw_kernel_bf16`
(an EXPERIMENT: scripts/ubench/row_pw/README.md has the measurements and why it is not in the product): the bf16 ROW stage of one encoder layer
(vad/modeling/transformer.py:160-215 of the reference: attention out-projection + residual, LayerNorm, feed-forward +
residual, then the NEXT layer's LayerNorm + QKV projection) as a PERSISTENT workgroup of 4 waves, one per SIMD, each
wave owning a PAIR of 32-row blocks (64 rows), with the whole register file owned by the stream below.

    bash scripts/ubench/row_pw/build.sh     # generates gen/*.inc, patches a copy of savad.hip (row_mode 7), builds libsavad_rowpw*.so

Why: row_kernel_bf16 (32 rows per wave, two workgroups per CU, 2-slot weight ring one block ahead) needs 1 KiB of LDS
reads per 32-cycle MFMA on every SIMD -- the CU's whole 128 B/clk -- streams the layer's 384 KiB of weights once per FOUR
blocks, and waits for the ring at every one of its 12 blocks.  Here every weight fragment read from LDS feeds two MFMAs
(the pair), the weights stream once per EIGHT blocks through a 4-slot ring two blocks ahead that never drains (the
stream is periodic: every item uses the same 12 blocks), and the loads of the next item (context, residual) are issued
a whole item ahead.

Arithmetic: identical, operation for operation, to row_stage_bf16 as hipcc 7.2 compiles it (the LayerNorm reductions
follow the compiled order: sequential row sum, pairwise fma / mul + add sum of squares, IEEE sqrt and division
sequences), so the two kernels agree bit for bit -- which is how this one is tested.

Work items: pair p = blocks 2p, 2p+1; pass k of workgroup g gives wave w the pair k * 4 * grid + 4 g + w (the number of
pairs is a multiple of 4, so the four waves of a workgroup always run the same number of passes).

Registers (asm-owned: v0..v239, a0..a255, s24..s31, s34..s99)
  v[0:127]    O: residual stream / FFN output accumulators o[r][nb][16]; the QKV accumulators at the end of an item
  v[128:191]  FFN1 accumulators of HALF a chunk ac[r][q][16] (64 hidden units); LayerNorm deviations; bias staging
  v[192:223]  BT0 / BT1: C operands of the first MFMA of a chain (bias, or residual + bias); LayerNorm scalars
  v[224:231]  staging (packing -> accvgpr_write / global stores);  v[232:239] addresses
  a[0:63]     XP: LayerNorm output as B-operand fragments xp[r][ks][4]
  a[64:127]   AP: relu(FFN1) fragments ap[r][ks][4]; at the item seam the NEXT item's context fragments
  a[128:191]  WF: 16 weight-fragment slots (ds_read_b128 targets)
  a[192:255]  H: the NEXT item's residual rows as loaded (fp16, 8 per register)
Hazards are stated by hand as in gen_attn_pw.py; s_waitcnt lgkmcnt / vmcnt are COMPUTED by resolve() from the issue
order (LDS reads and vector-memory operations return in order), over the prologue plus two unrolled items so that the
counts hold across the loop edge.
"""
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE / "gen" / "savad_row_pw_bf16.inc"          # written by build.sh, not tracked
OUT_TIMING = HERE / "gen" / "savad_row_pw_bf16_timing.inc"

FRAG, BLK, RINGBLK = 1024, 8192, 32768
D, DFF = 128, 512
NRING, AHEAD, NBLK = 4, 2, 12

A_XP, A_AP, A_WF, A_H = 0, 64, 128, 192
V_O, V_AC = 0, 128
V_BT = [192, 208]
V_T = 224
V_OFF = [232, 233, 234, 235]     # lane * 16 + {0, 4096, 8192, 12288}
V_RING = [236, 237]              # LDS base + lane * 16 (+ 65536): ring slots 0-1 / 2-3
V_BIAS = 238                     # LDS address of the bias area + 16 h
V_BIASN = 239                    # LDS address of the V bias + 4 (lane & 31)
# LayerNorm / store scalars (alias BT0 / BT1: no MFMA chain is being started while they live)
V_C65 = 207                      # 65504.0 (store_hblock's clamp); the per-block LayerNorm / store scalars: ln_scalars()
V_R4, V_R5 = 220, 221            # saturation path scratch

S_WS, S_FR = 26, 28   # prologue only: workspace / fragment bases.  s[24:29] and s[82:83]: VALU compare results of the LayerNorms
S_BCTX, S_BH, S_BQ, S_BK, S_BV = 36, 38, 40, 42, 44
S_WO, S_W1, S_W2, S_WN = 46, 48, 50, 52
S_SAT = 54
S_QSC = 56            # s[56:57] = qscale twice (v_pk_mul_f32 operand)
S_LDSW = 58           # LDS base + w * 8192
S_NBLK, S_NITEMS, S_STRIDE, S_PAIR, S_NPAIR = 59, 60, 61, 62, 63
S_HC, S_Q, S_K, S_V, S_CTXN, S_HN = 64, 66, 68, 70, 72, 74
S_SRC = 76
S_T0, S_T1, S_T2, S_T3 = 78, 79, 80, 81
S_CMP2 = 30          # s[30:31]: exec save of the saturation path
S_PREV, S_TMPD = 84, 85   # timing builds
S_TM = 86
S_ACC = 88            # timing builds: s88..s99 = cycles per category
TIMING = False

LBO, LB1, LB2, LBN = 0, D * 4, (D + DFF) * 4, (2 * D + DFF) * 4      # byte offsets inside the bias area
BIAS_AREA = NRING * RINGBLK


def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def ar(b, n=1):
    return f"a{b}" if n == 1 else f"a[{b}:{b + n - 1}]"


def sr(b, n=1):
    return f"s{b}" if n == 1 else f"s[{b}:{b + n - 1}]"


class Ins:
    __slots__ = ("text", "kind", "tag", "need_lds", "need_vm")

    def __init__(self, text, kind="x", tag=None, need_lds=(), need_vm=()):
        self.text, self.kind, self.tag, self.need_lds, self.need_vm = text, kind, tag, tuple(need_lds), tuple(need_vm)


def X(text, **kw):       # one-instruction op (a filler op is a LIST of Ins that stays together)
    return [Ins(text, "x", **kw)]


def L(text, tag=None, **kw):
    return [Ins(text, "lds", tag, **kw)]


def M(text, tag=None, **kw):
    return [Ins(text, "vmem", tag, **kw)]


def spread(gaps, ops, lo, hi):
    """ops (kept in order) evenly over gaps[lo:hi]"""
    lo = max(lo, 0)
    hi = min(hi, len(gaps))
    n = hi - lo
    assert n > 0, (lo, hi, len(gaps))
    for k, op in enumerate(ops):
        gaps[lo + (k * n) // len(ops)].append(op)


def emit_seg(body, mfmas, gaps):
    assert len(gaps) == len(mfmas) + 1
    for k, mf in enumerate(mfmas):
        for op in gaps[k]:
            body.extend(op)
        body.append(mf)
    for op in gaps[-1]:
        body.extend(op)


# ------------------------------------------------------------------------------------------------ register helpers
def O(r, nb):
    return V_O + r * 64 + nb * 16


def AC(r, q):
    return V_AC + r * 32 + q * 16


def XP(r, ks):
    return A_XP + r * 32 + ks * 4


def AP(r, ks):
    return A_AP + r * 32 + ks * 4


def goff(r, k):
    """(offset VGPR, immediate) of 1 KiB fragment k (0..7) of block r of a pair"""
    return V_OFF[2 * r + (k >> 2)], (k & 3) * FRAG


def ring_addr(slot, frag):
    return V_RING[slot >> 1], (slot & 1) * RINGBLK + frag * FRAG


def mfma(d, a, b, c, need=(), need_vm=()):
    return Ins(f"v_mfma_f32_32x32x16_bf16 {d}, {a}, {b}, {c}", "mfma", need_lds=need, need_vm=need_vm)


# ------------------------------------------------------------------------------------------------ DMA / acquire
def dma_src(t):
    """(base SGPR pair, byte offset) of this wave's 8 KiB segment of ring block t"""
    if t == 0:
        return S_WO, 0
    if t < 9:
        c = (t - 1) >> 1
        return (S_W2, c * BLK) if (t - 1) & 1 else (S_W1, c * RINGBLK)
    return S_WN, (t - 9) * RINGBLK


def dma_ops(t):
    """the 8 LDS-DMA pieces (1 KiB each) of this wave's segment of ring block t (0..11, periodic) as filler ops"""
    t %= NBLK
    base, off = dma_src(t)
    slot = t % NRING
    ops = []
    for half in range(2):
        head = [Ins(f"s_add_u32 {sr(S_SRC)}, {sr(base)}, {off}"), Ins(f"s_addc_u32 {sr(S_SRC + 1)}, {sr(base + 1)}, 0"),
                Ins(f"s_add_u32 m0, {sr(S_LDSW)}, {slot * RINGBLK + half * 4096}"), Ins("s_nop 0"),
                Ins(f"global_load_lds_dwordx4 {vr(V_OFF[half])}, {sr(S_SRC, 2)}", "vmem")]
        ops.append(head)
        for k in range(1, 4):
            last = half == 1 and k == 3
            ops.append(M(f"global_load_lds_dwordx4 {vr(V_OFF[half])}, {sr(S_SRC, 2)} offset:{k * FRAG}", tag=f"dma{t}" if last else None))
    return ops


def acquire(t):
    """ring block t has landed for every wave (the wait count is computed by resolve())"""
    return [Ins("s_barrier", "x", need_vm=(f"dma{t % NBLK}",))]


def stamp(body, cat):
    """timing builds: shader cycles since the previous stamp are added to category `cat` (an SGPR each)"""
    if not TIMING:
        return
    for t in (f"s_memtime {sr(S_TM, 2)}", "s_waitcnt lgkmcnt(0)", f"s_sub_u32 {sr(S_TMPD)}, {sr(S_TM)}, {sr(S_PREV)}",
              f"s_add_u32 {sr(S_ACC + cat)}, {sr(S_ACC + cat)}, {sr(S_TMPD)}", f"s_mov_b32 {sr(S_PREV)}, {sr(S_TM)}"):
        body.append(Ins(t, "drain" if "lgkmcnt" in t else "x"))


# ------------------------------------------------------------------------------------------------ building blocks
def bias_reads(dst, byte_off, tag):
    """bias_block(lds + byte_off / 4, h) -> v[dst:dst+15]: features 8 g + 4 h + s of a 32-feature block"""
    return [L(f"ds_read_b128 {vr(dst + 4 * g, 4)}, {vr(V_BIAS)} offset:{byte_off + 32 * g}", tag=f"{tag}_{g}") for g in range(4)]


def bias_tags(tag):
    return tuple(f"{tag}_{g}" for g in range(4))


V_BSTAGE = 112   # = O(1, 3): free while the out-projection's last chain has not started, and at the item seam


def h_prep_ops(nb, blocks=(0, 1)):
    """BT[r] = (f32(h rows of block r, feature block nb) + 0) + bo block, r = 0, 1  (load_hblock on zeros, then the bias: the
    compiled order); the fp16 rows sit in v[V_AC + 32 r + 8 nb ..+7] as loaded, register k = values 2k, 2k+1; the bias block is
    requested first (LDS latency behind the conversions) into v[V_BSTAGE..+15]"""
    ops = bias_reads(V_BSTAGE, LBO + 128 * nb, f"bo{nb}")
    for r in blocks:
        bt, src = V_BT[r], V_AC + 32 * r + 8 * nb
        for k in range(8):
            ops.append(X(f"v_cvt_f32_f16_e32 {vr(bt + 2 * k)}, {vr(src + k)}", need_vm=(f"h{r}",) if k == 0 else ()))
            ops.append(X(f"v_cvt_f32_f16_sdwa {vr(bt + 2 * k + 1)}, {vr(src + k)} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"))
        for j in range(8):
            ops.append(X(f"v_pk_add_f32 {vr(bt + 2 * j, 2)}, {vr(bt + 2 * j, 2)}, 0 op_sel_hi:[1,0]"))
    for r in blocks:
        for j in range(8):
            ops.append(X(f"v_pk_add_f32 {vr(V_BT[r] + 2 * j, 2)}, {vr(V_BT[r] + 2 * j, 2)}, {vr(V_BSTAGE + 2 * j, 2)}",
                         need_lds=(f"bo{nb}_{j // 2}",) if (r == blocks[0] and j % 2 == 0) else ()))
    return ops  # 4 + 2 x 24 + 16 = 68


def frag_read(slot, frag, wf, tag):
    addr, off = ring_addr(slot, frag)
    return L(f"ds_read_b128 {ar(A_WF + 4 * wf, 4)}, {vr(addr)} offset:{off}", tag=tag)


LA = 6


class Seg:
    """MFMAs mf[0..n) with fillers gaps[k] in front of mf[k] (gaps[n] behind the last); reads[f] feeds mf[2f], mf[2f+1] and is
    issued LA fragments ahead -- the first LA (`head`) by whatever runs before this segment"""

    def __init__(self, t, mf, reads):
        self.t, self.mf, self.reads = t, mf, reads
        self.gaps = [[] for _ in range(len(mf) + 1)]
        for f, op in enumerate(reads[LA:]):
            self.gaps[2 * f].append(op)
        self.head = reads[:LA]

    def put(self, ops, lo, hi):
        spread(self.gaps, ops, lo, hi)

    def emit(self, body):
        emit_seg(body, self.mf, self.gaps)


def interleave(a, b):
    out = []
    for k in range(max(len(a), len(b))):
        if k < len(a):
            out.append(a[k])
        if k < len(b):
            out.append(b[k])
    return out


def rstd_seq(sc):
    """x = ss / 128 + eps (contracted) -> IEEE sqrt -> IEEE 1 / x, the compiled sequences; scalars sc = dict of registers;
    result in sc['RS'] (the low half of an aligned pair).  A VALU-written SGPR mask is read two wait states later at the
    earliest; the division's v_div_scale / v_div_fmas pair communicates through VCC and stays one group."""
    R0, R1, R2, R3, R4, R5, SS, RS, CA, CB = (sc[k] for k in ("R0", "R1", "R2", "R3", "R4", "R5", "SS", "RS", "CMPA", "CMPB"))
    groups = [
        [f"v_mov_b32 {vr(R0)}, 0x3727c5ac"],
        [f"v_fmac_f32 {vr(R0)}, 0x3c000000, {vr(SS)}"],
        [f"v_mul_f32 {vr(R1)}, 0x4f800000, {vr(R0)}"],
        [f"v_mov_b32 {vr(R2)}, 0xf800000"],
        [f"v_cmp_gt_f32 {sr(CB, 2)}, {vr(R2)}, {vr(R0)}", "s_nop 1", f"v_cndmask_b32 {vr(R0)}, {vr(R0)}, {vr(R1)}, {sr(CB, 2)}"],
        [f"v_sqrt_f32 {vr(R1)}, {vr(R0)}", "s_nop 0"],
        [f"v_add_u32 {vr(R2)}, -1, {vr(R1)}"],
        [f"v_fma_f32 {vr(R3)}, -{vr(R2)}, {vr(R1)}, {vr(R0)}"],
        [f"v_cmp_ge_f32 {sr(CA, 2)}, 0, {vr(R3)}", f"v_add_u32 {vr(R3)}, 1, {vr(R1)}", "s_nop 0",
         f"v_cndmask_b32 {vr(R2)}, {vr(R1)}, {vr(R2)}, {sr(CA, 2)}"],
        [f"v_fma_f32 {vr(R1)}, -{vr(R3)}, {vr(R1)}, {vr(R0)}"],
        [f"v_cmp_lt_f32 {sr(CA, 2)}, 0, {vr(R1)}", "s_nop 1", f"v_cndmask_b32 {vr(R1)}, {vr(R2)}, {vr(R3)}, {sr(CA, 2)}"],
        [f"v_mul_f32 {vr(R2)}, 0x37800000, {vr(R1)}"],
        [f"v_cndmask_b32 {vr(R1)}, {vr(R1)}, {vr(R2)}, {sr(CB, 2)}"],
        [f"v_mov_b32 {vr(R2)}, 0x260"],
        [f"v_cmp_class_f32 {sr(CA, 2)}, {vr(R0)}, {vr(R2)}", "s_nop 1", f"v_cndmask_b32 {vr(R0)}, {vr(R1)}, {vr(R0)}, {sr(CA, 2)}"],
        [f"v_div_scale_f32 {vr(R1)}, {sr(CA, 2)}, {vr(R0)}, {vr(R0)}, 1.0",
         f"v_rcp_f32 {vr(R2)}, {vr(R1)}",
         "s_nop 0",
         f"v_fma_f32 {vr(R3)}, -{vr(R1)}, {vr(R2)}, 1.0",
         f"v_fmac_f32 {vr(R2)}, {vr(R3)}, {vr(R2)}",
         f"v_div_scale_f32 {vr(R3)}, vcc, 1.0, {vr(R0)}, 1.0",
         f"v_mul_f32 {vr(R4)}, {vr(R3)}, {vr(R2)}",
         f"v_fma_f32 {vr(R5)}, -{vr(R1)}, {vr(R4)}, {vr(R3)}",
         f"v_fmac_f32 {vr(R4)}, {vr(R5)}, {vr(R2)}",
         f"v_fma_f32 {vr(R1)}, -{vr(R1)}, {vr(R4)}, {vr(R3)}",
         "s_nop 1",
         f"v_div_fmas_f32 {vr(R1)}, {vr(R1)}, {vr(R2)}, {vr(R4)}",
         f"v_div_fixup_f32 {vr(RS)}, {vr(R1)}, {vr(R0)}, 1.0"],
    ]
    return [[Ins(t) for t in g] for g in groups]


def ln_scalars(r):
    b = V_BT[r]
    return {"S": b, "TMP": b + 1, "MEAN": b + 2, "SS": b + 4, "M2": b + 5, "R0": b + 6, "R1": b + 7, "R2": b + 8, "R3": b + 9, "R4": b + 10,
            "R5": b + 11, "DT": b + 12, "AMAX": b + 14, "CMPA": (24, 28)[r], "CMPB": (26, 82)[r]}


def layernorm_ops(r, keep):
    """LayerNorm (no affine part) of block r: x = O[r] -> XP[r] (bf16 B-operand fragments: xp[ks] = values 8 ks .. 8 ks + 7),
    operation order of layernorm_regs() as compiled.  keep: the deviations are kept in v[V_AC..+63]; else they are
    recomputed for the scaling pass (same operation, same bits) so that both blocks' chains can be interleaved."""
    sc = ln_scalars(r)
    S, TMP, MEAN, SS, M2, DT = (sc[k] for k in ("S", "TMP", "MEAN", "SS", "M2", "DT"))
    x0 = O(r, 0)
    ops = [X(f"v_add_f32 {vr(S)}, 0, {vr(x0)}")]
    for e in range(1, 64):
        ops.append(X(f"v_add_f32 {vr(S)}, {vr(x0 + e)}, {vr(S)}"))
    ops.append([Ins(f"v_mov_b32 {vr(TMP)}, {vr(S)}"), Ins("s_nop 1"), Ins(f"v_permlane32_swap_b32 {vr(S)}, {vr(TMP)}")])
    ops.append(X(f"v_add_f32 {vr(S)}, {vr(S)}, {vr(TMP)}"))
    ops.append(X(f"v_mul_f32 {vr(MEAN)}, 0x3c000000, {vr(S)}"))

    def dev(p):
        d = V_AC + 2 * p if keep else DT
        return d, X(f"v_pk_add_f32 {vr(d, 2)}, {vr(x0 + 2 * p, 2)}, {vr(MEAN, 2)} op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")

    # ss = fma(d0, d0, d1 * d1); then per pair: ss = fma(d_even, d_even, ss); ss = d_odd * d_odd + ss
    for p in range(32):
        d, op = dev(p)
        ops.append(op)
        ops.append(X(f"v_mul_f32 {vr(M2)}, {vr(d + 1)}, {vr(d + 1)}"))
        if p == 0:
            ops.append(X(f"v_fma_f32 {vr(SS)}, {vr(d)}, {vr(d)}, {vr(M2)}"))
        else:
            ops.append(X(f"v_fmac_f32 {vr(SS)}, {vr(d)}, {vr(d)}"))
            ops.append(X(f"v_add_f32 {vr(SS)}, {vr(M2)}, {vr(SS)}"))
    ops.append([Ins(f"v_mov_b32 {vr(TMP)}, {vr(SS)}"), Ins("s_nop 1"), Ins(f"v_permlane32_swap_b32 {vr(SS)}, {vr(TMP)}")])
    ops.append(X(f"v_add_f32 {vr(SS)}, {vr(SS)}, {vr(TMP)}"))
    RS = MEAN if keep else sc["R2"]     # rstd: low half of an aligned pair (the mean is needed again when the deviations are recomputed)
    ops += rstd_seq(dict(sc, RS=RS))
    for p in range(32):
        t = V_T + 4 * r + (p & 3)
        if keep:
            d = V_AC + 2 * p
            grp = []
        else:
            d, op = dev(p)
            grp = list(op)
        grp += [Ins(f"v_pk_mul_f32 {vr(d, 2)}, {vr(d, 2)}, {vr(RS, 2)} op_sel_hi:[1,0]"),
                Ins(f"v_cvt_pk_bf16_f32 {vr(t)}, {vr(d)}, {vr(d + 1)}"), Ins(f"v_accvgpr_write_b32 {ar(A_XP + 32 * r + p)}, {vr(t)}")]
        ops.append(grp)
    return ops


def layernorm_pair():
    return interleave(layernorm_ops(0, True), layernorm_ops(1, False))


def store_h_ops(r):
    """store_hblock(O[r]): |x| maximum (v_max3 chain), clamp to +-65504 (v_med3), fp16 pairs, 8 stores"""
    sc = ln_scalars(r)
    AMAX, R4, R5 = sc["AMAX"], sc["R4"], sc["R5"]
    x0 = O(r, 0)
    ops = [X(f"v_max3_f32 {vr(AMAX)}, |{vr(x0)}|, 0, |{vr(x0 + 1)}|")]
    for p in range(1, 32):
        ops.append(X(f"v_max3_f32 {vr(AMAX)}, {vr(AMAX)}, |{vr(x0 + 2 * p)}|, |{vr(x0 + 2 * p + 1)}|"))
    for k in range(8):      # fragment k = values 8k .. 8k+7 of the block
        t = V_T + 4 * r
        grp = []
        for q in range(4):
            a, b = x0 + 8 * k + 2 * q, x0 + 8 * k + 2 * q + 1
            grp += [Ins(f"v_med3_f32 {vr(R4)}, {vr(a)}, {sr(S_T3)}, {vr(V_C65)}"), Ins(f"v_med3_f32 {vr(R5)}, {vr(b)}, {sr(S_T3)}, {vr(V_C65)}"),
                    Ins(f"v_cvt_pk_f16_f32 {vr(t + q)}, {vr(R4)}, {vr(R5)}")]
        off, imm = goff(r, k)
        grp.append(Ins(f"global_store_dwordx4 {vr(off)}, {vr(t, 4)}, {sr(S_HC, 2)} offset:{imm}", "vmem"))
        ops.append(grp)
    return ops


def relu_pack_ops(r, q, nbl):
    """ap[2 nbl + j] = bf16(max(ac[r][q], 0)) -> AP[r][2 nbl + j]  (relu in place: the accumulators are dead afterwards)"""
    ops = []
    a0 = AC(r, q)
    for p in range(8):
        t = V_T + (p & 7)
        ops.append([Ins(f"v_max_f32 {vr(a0 + 2 * p)}, 0, {vr(a0 + 2 * p)}"), Ins(f"v_max_f32 {vr(a0 + 2 * p + 1)}, 0, {vr(a0 + 2 * p + 1)}"),
                    Ins(f"v_cvt_pk_bf16_f32 {vr(t)}, {vr(a0 + 2 * p)}, {vr(a0 + 2 * p + 1)}"),
                    Ins(f"v_accvgpr_write_b32 {ar(A_AP + 32 * r + 8 * nbl + p)}, {vr(t)}")])
    return ops


def qkv_epilogue_ops(rb, r, nbl):
    """scale (Q only), pack and store the two fragments of accumulator block (r, nbl) of QKV ring block rb"""
    a0 = O(r, nbl)
    base = (S_Q, S_K, S_V)[rb]
    ops = []
    if rb == 0:
        for p in range(8):
            ops.append(X(f"v_pk_mul_f32 {vr(a0 + 2 * p, 2)}, {vr(a0 + 2 * p, 2)}, {sr(S_QSC, 2)}"))
    for j in range(2):
        t = V_T + 4 * j
        grp = [Ins(f"v_cvt_pk_bf16_f32 {vr(t + q)}, {vr(a0 + 8 * j + 2 * q)}, {vr(a0 + 8 * j + 2 * q + 1)}") for q in range(4)]
        off, imm = goff(r, 2 * nbl + j)
        grp.append(Ins(f"global_store_dwordx4 {vr(off)}, {vr(t, 4)}, {sr(base, 2)} offset:{imm}", "vmem"))
        ops.append(grp)
    return ops


def next_loads(kind, ks_range):
    """the NEXT item's context fragments (-> AP, the B operands of its out-projection) / residual rows (-> v[V_AC..], fp16)"""
    ops = []
    for r in range(2):
        for k in ks_range:
            off, imm = goff(r, k)
            if kind == "ctx":
                ops.append(M(f"global_load_dwordx4 {ar(AP(r, k), 4)}, {vr(off)}, {sr(S_CTXN, 2)} offset:{imm}", tag=f"ctx{r}_{k}"))
            else:
                ops.append(M(f"global_load_dwordx4 {vr(V_AC + 32 * r + 4 * k, 4)}, {vr(off)}, {sr(S_HN, 2)} offset:{imm}", tag=f"h{r}" if k == ks_range[-1] else None))
    return ops


CTX_TAGS = {r: tuple(f"ctx{r}_{k}" for k in range(8)) for r in range(2)}


# ------------------------------------------------------------------------------------------------ segments
def seg_out():
    """ring block 0: o[r] = (h[r] + bo) + ctx[r] Wo^T; the C operands of nb = 0 were prepared at the end of the previous
    item (or by the prologue)"""
    mf, reads = [], []
    for nb in range(4):
        for ks in range(8):
            f = nb * 8 + ks
            tag = f"wo{f}"
            reads.append(frag_read(0, f, f % 16, tag))
            for r in range(2):
                c = vr(V_BT[r], 16) if ks == 0 else vr(O(r, nb), 16)
                mf.append(mfma(vr(O(r, nb), 16), ar(A_WF + 4 * (f % 16), 4), ar(AP(r, ks), 4), c, need=(tag,) if r == 0 else (),
                               need_vm=(f"ctx{r}_{ks}",) if nb == 0 else ()))
    s = Seg(0, mf, reads)
    for nb in range(1, 4):   # C operands of the next feature block while this one's chain runs
        s.put(h_prep_ops(nb), 16 * (nb - 1) + 3, 16 * nb - 1)
    return s


def ln1_phase(body):
    body.append(Ins("s_nop 7"))
    body.append(Ins("s_nop 7"))
    for op in layernorm_pair():
        body.extend(op)
    # o = h1 + b2: the residual stream enters the FFN2 accumulators (the 128 bias values of a lane through v[V_AC..+63], free again)
    for nb in range(4):
        for op in bias_reads(V_AC + 16 * nb, LB2 + 128 * nb, f"b2_{nb}"):
            body.extend(op)
    for nb in range(4):
        for r in range(2):
            for j in range(8):
                body.extend(X(f"v_pk_add_f32 {vr(O(r, nb) + 2 * j, 2)}, {vr(O(r, nb) + 2 * j, 2)}, {vr(V_AC + 16 * nb + 2 * j, 2)}",
                              need_lds=(f"b2_{nb}_{j // 2}",) if (r == 0 and j % 2 == 0) else ()))
    for op in bias_reads(V_BT[0], LB1 + 0, "b1_0_0") + bias_reads(V_BT[1], LB1 + 128, "b1_0_1"):
        body.extend(op)
    stamp(body, 2)


def seg_ffn_a(c):
    """chunk c, segment A (ring block 1 + 2c = W1 chunk): FFN1 of half 0 (hidden units 0..63 of the chunk)"""
    t1 = 1 + 2 * c
    s1 = t1 % NRING
    mf, reads = [], []
    for q in range(2):
        for ks in range(8):
            f = q * 8 + ks
            tag = f"w1_{c}_{f}"
            reads.append(frag_read(s1, f, f % 16, tag))
            for r in range(2):
                cc = vr(V_BT[q], 16) if ks == 0 else vr(AC(r, q), 16)
                need = ((tag,) if r == 0 else ()) + (bias_tags(f"b1_{c}_{q}") if ks == 0 and r == 0 else ())
                mf.append(mfma(vr(AC(r, q), 16), ar(A_WF + 4 * (f % 16), 4), ar(XP(r, ks), 4), cc, need=need))
    s = Seg(t1, mf, reads)
    s.put(relu_pack_ops(0, 0, 0) + relu_pack_ops(1, 0, 0), 19, 32)
    return s


def seg_ffn_b(c):
    """chunk c, segment B (ring block 2 + 2c = W2 chunk; the W1 chunk stays resident): FFN2 half 0, FFN1 half 1, FFN2 half 1"""
    t1, t2 = 1 + 2 * c, 2 + 2 * c
    s1, s2 = t1 % NRING, t2 % NRING
    mf, reads = [], []
    fcount = 0

    def ffn2(half):
        nonlocal fcount
        for ks in range(4 * half, 4 * half + 4):
            for nb in range(4):
                tag = f"w2_{c}_{nb}_{ks}"
                wf = fcount % 16
                fcount += 1
                reads.append(frag_read(s2, nb * 8 + ks, wf, tag))
                for r in range(2):
                    mf.append(mfma(vr(O(r, nb), 16), ar(A_WF + 4 * wf, 4), ar(AP(r, ks), 4), vr(O(r, nb), 16), need=(tag,) if r == 0 else ()))

    ffn2(0)
    for q in range(2):
        for ks in range(8):
            f = (2 + q) * 8 + ks
            tag = f"w1_{c}_{f}"
            wf = fcount % 16
            fcount += 1
            reads.append(frag_read(s1, f, wf, tag))
            for r in range(2):
                cc = vr(V_BT[q], 16) if ks == 0 else vr(AC(r, q), 16)
                need = ((tag,) if r == 0 else ()) + (bias_tags(f"b1_{c}_{2 + q}") if ks == 0 and r == 0 else ())
                mf.append(mfma(vr(AC(r, q), 16), ar(A_WF + 4 * wf, 4), ar(XP(r, ks), 4), cc, need=need))
    ffn2(1)
    s = Seg(t2, mf, reads)
    s.put(relu_pack_ops(0, 1, 1) + relu_pack_ops(1, 1, 1), 3, 15)
    # bias of half 1's chains (BT0 / BT1 were consumed by the first MFMAs of segment A's chains)
    s.put(bias_reads(V_BT[0], LB1 + 512 * c + 256, f"b1_{c}_2") + bias_reads(V_BT[1], LB1 + 512 * c + 384, f"b1_{c}_3"), 4, 20)
    s.put(relu_pack_ops(0, 0, 2) + relu_pack_ops(1, 0, 2), 51, 63)
    s.put(relu_pack_ops(0, 1, 3) + relu_pack_ops(1, 1, 3), 67, 79)
    if c < 3:   # bias of the next chunk's first half
        s.put(bias_reads(V_BT[0], LB1 + 512 * (c + 1), f"b1_{c + 1}_0") + bias_reads(V_BT[1], LB1 + 512 * (c + 1) + 128, f"b1_{c + 1}_1"), 66, 90)
    else:       # the next item's context, K-steps 0..3 (AP[r][0..3] were last read by MFMA 31)
        s.put(next_loads("ctx", range(0, 4)), 36, 92)
    return s


def end_phase(body):
    """the finished residual rows go back to HBM (fp16), LayerNorm of the next layer"""
    for _ in range(3):
        body.append(Ins("s_nop 7"))
    for op in next_loads("ctx", range(4, 8)):
        body.extend(op)
    body.append(Ins(f"s_mov_b32 {sr(S_T3)}, 0xc77fe000"))
    body.append(Ins(f"v_mov_b32 {vr(V_C65)}, 0x477fe000"))
    for op in interleave(store_h_ops(0), store_h_ops(1)):
        body.extend(op)
    for r in range(2):
        sat, back = f".Lrp_sat{r}", f".Lrp_satback{r}"
        body.append(Ins(f"v_cmp_nle_f32 vcc, {vr(ln_scalars(r)['AMAX'])}, {vr(V_C65)}"))
        body.append(Ins("s_nop 0"))
        body.append(Ins("s_cmp_lg_u64 vcc, 0"))
        body.append(Ins(f"s_cbranch_scc1 {sat}"))
        body.append(Ins(f"{back}:", "label"))
    for op in layernorm_pair():
        body.extend(op)
    for op in bias_reads(V_BT[0], LBN + 0, "bq0_0") + bias_reads(V_BT[1], LBN + 128, "bq0_1"):
        body.extend(op)
    stamp(body, 7)


def qkv_c_ops(rb, nbl):
    bt = V_BT[nbl & 1]
    if rb < 2:
        return bias_reads(bt, LBN + 512 * rb + 128 * nbl, f"bq{rb}_{nbl}")
    ops = [L(f"ds_read_b32 {vr(bt)}, {vr(V_BIASN)} offset:{128 * nbl}", tag=f"bv_{nbl}")]
    ops.append([Ins(f"v_mov_b32 {vr(bt + 1)}, {vr(bt)}", need_lds=(f"bv_{nbl}",))])
    for e in range(2, 16):
        ops.append(X(f"v_mov_b32 {vr(bt + e)}, {vr(bt)}"))
    return ops


def seg_qkv(rb):
    """ring block 9 + rb: Q (pre-scaled), K in the transposed form, V^T in the swapped form; accumulators = O's registers"""
    t = 9 + rb
    slot = t % NRING
    mf, reads = [], []
    for nbl in range(4):
        for ks in range(8):
            f = nbl * 8 + ks
            tag = f"wn{rb}_{f}"
            reads.append(frag_read(slot, f, f % 16, tag))
            for r in range(2):
                bt = V_BT[nbl & 1]
                c = vr(bt, 16) if ks == 0 else vr(O(r, nbl), 16)
                need = (tag,) if r == 0 else ()
                if ks == 0 and r == 0:
                    need += bias_tags(f"bq{rb}_{nbl}") if rb < 2 else (f"bv_{nbl}",)
                wfr, xpr = ar(A_WF + 4 * (f % 16), 4), ar(XP(r, ks), 4)
                a_op, b_op = (wfr, xpr) if rb < 2 else (xpr, wfr)
                mf.append(mfma(vr(O(r, nbl), 16), a_op, b_op, c, need=need))
    s = Seg(t, mf, reads)
    # C operands: n-blocks 0 / 1 were prepared by the previous segment; 2 / 3 here once BT0 / BT1 have been consumed
    def put_c(rb_, nbl, lo, hi):
        ops = qkv_c_ops(rb_, nbl)
        if rb_ < 2:
            s.put(ops, lo, hi)
        else:   # the broadcast waits for its one LDS read: request it a few MFMAs earlier
            s.put(ops[:1], lo, lo + 1)
            s.put(ops[1:], lo + 5, hi)

    put_c(rb, 2, 3, 20)
    put_c(rb, 3, 19, 36)
    if rb < 2:
        put_c(rb + 1, 0, 36, 50)
        put_c(rb + 1, 1, 50, 63)
    for nbl in range(3):   # epilogue of chain nbl under the MFMAs of chain nbl + 1
        s.put(qkv_epilogue_ops(rb, 0, nbl) + qkv_epilogue_ops(rb, 1, nbl), 16 * (nbl + 1) + 3, 16 * (nbl + 2) - 1)
    if rb > 0:             # the previous block's last chain: its registers are rewritten by this block's MFMA 48
        s.put(qkv_epilogue_ops(rb - 1, 0, 3) + qkv_epilogue_ops(rb - 1, 1, 3), 4, 16)
    if rb == 0:            # the next item's residual rows (the FFN1 accumulators are dead until the next item's chunk 0)
        s.put(next_loads("h", range(0, 8)), 6, 60)
    return s


def item_params():
    """per item: store / prefetch bases from S_PAIR and the next pair (this one again when there is none: the loads stay,
    their results are never used)"""
    t = []
    t.append(f"s_add_u32 {sr(S_NPAIR)}, {sr(S_PAIR)}, {sr(S_STRIDE)}")
    t.append(f"s_cmp_lt_u32 {sr(S_NPAIR)}, {sr(S_NITEMS)}")
    t.append(f"s_cselect_b32 {sr(S_T0)}, {sr(S_NPAIR)}, {sr(S_PAIR)}")
    for dst, base, src in ((S_HC, S_BH, S_PAIR), (S_Q, S_BQ, S_PAIR), (S_K, S_BK, S_PAIR), (S_V, S_BV, S_PAIR), (S_CTXN, S_BCTX, S_T0), (S_HN, S_BH, S_T0)):
        t.append(f"s_lshr_b32 {sr(S_T2)}, {sr(src)}, 18")
        t.append(f"s_lshl_b32 {sr(S_T1)}, {sr(src)}, 14")
        t.append(f"s_add_u32 {sr(dst)}, {sr(base)}, {sr(S_T1)}")
        t.append(f"s_addc_u32 {sr(dst + 1)}, {sr(base + 1)}, {sr(S_T2)}")
    return [Ins(x) for x in t]


def zero_pad_ctx():
    """blocks past nblk (the padding of the block space): their context was never written -- zero fragments, as
    row_kernel_bf16 does"""
    out = []
    for r in range(2):
        skip = f".Lrp_nz{r}"
        out += [Ins(f"s_lshl_b32 {sr(S_T1)}, {sr(S_PAIR)}, 1"), Ins(f"s_add_u32 {sr(S_T1)}, {sr(S_T1)}, {r}"),
                Ins(f"s_cmp_lt_u32 {sr(S_T1)}, {sr(S_NBLK)}"), Ins(f"s_cbranch_scc1 {skip}")]
        out += [Ins(f"v_accvgpr_write_b32 {ar(AP(r, 0) + k)}, 0", need_vm=CTX_TAGS[r] if k == 0 else ()) for k in range(32)]
        out.append(Ins(f"{skip}:", "label"))
    return out


def wire(segs, k, e_gap, dma_lo, dma_hi, head_lo):
    """inside segment k: acquire the NEXT segment's ring block early, issue the block three ahead behind that barrier, and
    request the next segment's first weight fragments"""
    cur, nxt = segs[k], segs[(k + 1) % len(segs)]
    cur.put([acquire(nxt.t)], e_gap, e_gap + 1)
    cur.put(dma_ops(nxt.t + AHEAD), dma_lo, dma_hi)
    for j, op in enumerate(nxt.head):
        cur.gaps[min(head_lo + j, len(cur.gaps) - 1)].append(op)


def build_item():
    out = seg_out()
    fa = [seg_ffn_a(c) for c in range(4)]
    fb = [seg_ffn_b(c) for c in range(4)]
    qkv = [seg_qkv(rb) for rb in range(3)]
    segs = [out] + [x for c in range(4) for x in (fa[c], fb[c])] + qkv
    for k, sg in enumerate(segs):
        n = len(sg.mf)
        if n == 64:
            wire(segs, k, 38, 39, 56, 56)
        elif n == 32:   # W1 segments
            wire(segs, k, 18, 19, 30, 26)
        else:           # W2 segments: the W1 chunk is read until MFMA 63; the slot the new block lands in is the W1 chunk's
            wire(segs, k, 66, 67, 88, 88)
    body = []
    stamp(body, 11)
    body += item_params()
    body += zero_pad_ctx()
    stamp(body, 0)
    out.emit(body)
    stamp(body, 1)
    ln1_phase(body)
    for c in range(4):
        fa[c].emit(body)
        fb[c].emit(body)
        stamp(body, 3 + c)
    end_phase(body)
    for rb in range(3):
        qkv[rb].emit(body)
        if rb == 2:
            for _ in range(3):
                body.append(Ins("s_nop 7"))
            for op in qkv_epilogue_ops(2, 0, 3) + qkv_epilogue_ops(2, 1, 3) + h_prep_ops(0):
                body.extend(op)
        stamp(body, 8 + rb)
    return body, out


def build_prologue(first_head):
    """operands -> fixed homes, constants, the first two ring blocks, the first item's context and residual rows"""
    p = []

    def i(t, kind="x", **kw):
        p.append(Ins(t, kind, **kw))

    # operands: %0 workspace, %1..%5 KiB offsets of ctx / h / q / k / vt, %6 fragment base, %7..%10 KiB offsets of Wo / W1 / W2 /
    # Wqkv', %11 qscale, %12 saturation counter, %13 LDS base, %14 w, %15 nblk, %16 pairs, %17 first pair, %18 stride, %19 lane*16, %20 dbg
    i(f"s_mov_b64 {sr(S_WS, 2)}, %0")
    i(f"s_mov_b64 {sr(S_FR, 2)}, %6")
    for dst, off in ((S_BCTX, "%1"), (S_BH, "%2"), (S_BQ, "%3"), (S_BK, "%4"), (S_BV, "%5")):
        i(f"s_lshr_b32 {sr(S_T1)}, {off}, 22")
        i(f"s_lshl_b32 {sr(S_T0)}, {off}, 10")
        i(f"s_add_u32 {sr(dst)}, {sr(S_WS)}, {sr(S_T0)}")
        i(f"s_addc_u32 {sr(dst + 1)}, {sr(S_WS + 1)}, {sr(S_T1)}")
    for dst, off, wmul in ((S_WO, "%7", BLK), (S_W1, "%8", BLK), (S_W2, "%9", RINGBLK), (S_WN, "%10", BLK)):
        i(f"s_lshr_b32 {sr(S_T1)}, {off}, 22")
        i(f"s_lshl_b32 {sr(S_T0)}, {off}, 10")
        i(f"s_add_u32 {sr(dst)}, {sr(S_FR)}, {sr(S_T0)}")
        i(f"s_addc_u32 {sr(dst + 1)}, {sr(S_FR + 1)}, {sr(S_T1)}")
        i(f"s_mul_i32 {sr(S_T0)}, %14, {wmul}")          # this wave's segment
        i(f"s_add_u32 {sr(dst)}, {sr(dst)}, {sr(S_T0)}")
        i(f"s_addc_u32 {sr(dst + 1)}, {sr(dst + 1)}, 0")
    i(f"s_mov_b32 {sr(S_QSC)}, %11")
    i(f"s_mov_b32 {sr(S_QSC + 1)}, %11")
    i(f"s_mov_b64 {sr(S_SAT, 2)}, %12")
    i(f"s_lshl_b32 {sr(S_T0)}, %14, 13")
    i(f"s_add_u32 {sr(S_LDSW)}, %13, {sr(S_T0)}")
    i(f"s_mov_b32 {sr(S_NBLK)}, %15")
    i(f"s_mov_b32 {sr(S_NITEMS)}, %16")
    i(f"s_mov_b32 {sr(S_PAIR)}, %17")
    i(f"s_mov_b32 {sr(S_STRIDE)}, %18")
    i(f"v_mov_b32 {vr(V_OFF[0])}, %19")
    i(f"s_mov_b32 {sr(S_T0)}, %13")
    i("s_mov_b64 exec, -1")
    for k in range(1, 4):
        i(f"v_add_u32 {vr(V_OFF[k])}, {4096 * k}, {vr(V_OFF[0])}")
    i(f"v_add_u32 {vr(V_RING[0])}, {sr(S_T0)}, {vr(V_OFF[0])}")
    i(f"v_add_u32 {vr(V_RING[1])}, 65536, {vr(V_RING[0])}")
    i(f"v_lshrrev_b32 {vr(V_T)}, 9, {vr(V_OFF[0])}")           # h = lane >> 5
    i(f"v_lshlrev_b32 {vr(V_T)}, 4, {vr(V_T)}")
    i(f"v_add_u32 {vr(V_BIAS)}, {sr(S_T0)}, {vr(V_T)}")
    i(f"v_add_u32 {vr(V_BIAS)}, {BIAS_AREA}, {vr(V_BIAS)}")
    i(f"v_lshrrev_b32 {vr(V_T)}, 2, {vr(V_OFF[0])}")           # 4 * lane
    i(f"v_and_b32 {vr(V_T)}, 124, {vr(V_T)}")                  # 4 * (lane & 31)
    i(f"v_add_u32 {vr(V_BIASN)}, {sr(S_T0)}, {vr(V_T)}")
    i(f"v_add_u32 {vr(V_BIASN)}, {BIAS_AREA + LBN + 2 * D * 4}, {vr(V_BIASN)}")
    if TIMING:
        for c in range(12):
            i(f"s_mov_b32 {sr(S_ACC + c)}, 0")
        i(f"s_memtime {sr(S_TM, 2)}")
        i("s_waitcnt lgkmcnt(0)")
        i(f"s_mov_b32 {sr(S_PREV)}, {sr(S_TM)}")
    i(f"s_cmp_lt_u32 {sr(S_PAIR)}, {sr(S_NITEMS)}")
    i("s_cbranch_scc0 .Lrp_exit")
    for t in (0, 1, 2):
        for op in dma_ops(t):
            p.extend(op)
    # first item: S_CTXN / S_HN = this pair
    for dst, base in ((S_CTXN, S_BCTX), (S_HN, S_BH)):
        i(f"s_lshr_b32 {sr(S_T2)}, {sr(S_PAIR)}, 18")
        i(f"s_lshl_b32 {sr(S_T1)}, {sr(S_PAIR)}, 14")
        i(f"s_add_u32 {sr(dst)}, {sr(base)}, {sr(S_T1)}")
        i(f"s_addc_u32 {sr(dst + 1)}, {sr(base + 1)}, {sr(S_T2)}")
    for op in next_loads("ctx", range(0, 8)) + next_loads("h", range(0, 8)):
        p.extend(op)
    i("s_waitcnt vmcnt(0)", "drainvm")
    for op in h_prep_ops(0):
        p.extend(op)
    i("s_waitcnt lgkmcnt(0)", "drain")
    i("s_barrier")
    for op in first_head:
        p.extend(op)
    return p


def saturation_sub(r):
    """out of line: exact count of the elements of O[r] outside +-65504 (NaN included), one atomic per lane that has any;
    the vector-memory queue is drained afterwards so that the counted vmcnt waits of the main stream stay valid"""
    x0 = O(r, 0)
    t = [f".Lrp_sat{r}:", f"v_mov_b32 {vr(V_R4)}, 0"]
    for e in range(64):
        t.append(f"v_cmp_nle_f32 vcc, |{vr(x0 + e)}|, {vr(V_C65)}")
        t.append(f"v_addc_co_u32 {vr(V_R4)}, vcc, 0, {vr(V_R4)}, vcc")
    t += [f"v_cmp_ne_u32 vcc, 0, {vr(V_R4)}", f"s_and_saveexec_b64 {sr(S_CMP2, 2)}, vcc", f"v_mov_b32 {vr(V_R5)}, 0",
          f"global_atomic_add {vr(V_R5)}, {vr(V_R4)}, {sr(S_SAT, 2)}", f"s_mov_b64 exec, {sr(S_CMP2, 2)}", "s_waitcnt vmcnt(0)",
          f"s_branch .Lrp_satback{r}"]
    return t


def resolve(prologue, item):
    """insert the counted s_waitcnt instructions; returns (prologue lines, item lines of the steady state)"""
    state = {"lds_seq": 0, "lds_tag": {}, "lds_done": -1, "vm_seq": 0, "vm_tag": {}, "vm_done": -1}

    def run(ins_list):
        lines = []
        for ins in ins_list:
            if ins.need_lds:
                need = max(state["lds_tag"][t] for t in ins.need_lds)
                if need > state["lds_done"]:
                    n = min(state["lds_seq"] - 1 - need, 15)
                    lines.append(f"\ts_waitcnt lgkmcnt({n})")
                    state["lds_done"] = state["lds_seq"] - 1 - n
            if ins.need_vm:
                need = max(state["vm_tag"][t] for t in ins.need_vm)
                if need > state["vm_done"]:
                    n = min(state["vm_seq"] - 1 - need, 63)
                    lines.append(f"\ts_waitcnt vmcnt({n})")
                    state["vm_done"] = state["vm_seq"] - 1 - n
            lines.append(ins.text if ins.kind == "label" else "\t" + ins.text)
            if ins.kind == "lds":
                if ins.tag:
                    state["lds_tag"][ins.tag] = state["lds_seq"]
                state["lds_seq"] += 1
            elif ins.kind == "vmem":
                if ins.tag:
                    state["vm_tag"][ins.tag] = state["vm_seq"]
                state["vm_seq"] += 1
            elif ins.kind == "drain":
                state["lds_done"] = state["lds_seq"] - 1
            elif ins.kind == "drainvm":
                state["vm_done"] = state["vm_seq"] - 1
        return lines

    pl = run(prologue)
    run(item)
    state["lds_done"] = min(state["lds_done"], state["lds_seq"] - 1)
    il = run(item)
    il2 = run(item)
    assert il == il2, "the wait counts do not reach a steady state"
    return pl, il


def emit_all():
    item, out = build_item()
    pl, il = resolve(build_prologue(out.head), item)
    lines = ["\t// generated by scripts/ubench/row_pw/gen_row_pw.py -- do not edit"] + pl
    lines.append(".Lrp_item:")
    lines += il
    lines.append(f"\ts_mov_b32 {sr(S_PAIR)}, {sr(S_NPAIR)}")
    lines.append(f"\ts_cmp_lt_u32 {sr(S_PAIR)}, {sr(S_NITEMS)}")
    lines.append("\ts_cbranch_scc1 .Lrp_item")
    lines.append(".Lrp_exit:")
    lines.append("\ts_waitcnt vmcnt(0)")
    if TIMING:   # the first wave of the first workgroup reports its per-category cycles (32-bit words 0..11 of g_savad_dbg)
        lines += [f"\ts_cmp_eq_u32 %17, 0", "\ts_cbranch_scc0 .Lrp_nodbg", f"\tv_mov_b32 {vr(V_T)}, 0"]
        for c in range(12):
            lines += [f"\tv_mov_b32 {vr(V_T + 1)}, {sr(S_ACC + c)}", f"\tglobal_store_dword {vr(V_T)}, {vr(V_T + 1)}, %20 offset:{4 * c}"]
        lines += ["\ts_waitcnt vmcnt(0)", ".Lrp_nodbg:"]
    lines.append("\ts_branch .Lrp_end")
    for r in range(2):
        for t in saturation_sub(r):
            lines.append(t if t.endswith(":") else "\t" + t)
    lines.append(".Lrp_end:")
    return lines


def render(lines):
    return ("// generated by scripts/ubench/row_pw/gen_row_pw.py -- do not edit (python scripts/ubench/row_pw/gen_row_pw.py rewrites it)\n"
            "R\"ASMRP(\n" + "\n".join(lines) + "\n)ASMRP\"\n")


def main():
    global TIMING
    if "--out" in sys.argv:
        TIMING = "--timing" in sys.argv
        Path(sys.argv[sys.argv.index("--out") + 1]).write_text(render(emit_all()))
        return
    TIMING = True
    timing_text = render(emit_all())
    TIMING = False
    lines = emit_all()
    text = render(lines)
    if "--check" in sys.argv:
        stale = [f for f, t in ((OUT, text), (OUT_TIMING, timing_text)) if not f.exists() or f.read_text() != t]
        if stale:
            print(f"stale: {[str(f) for f in stale]}: run python scripts/ubench/row_pw/gen_row_pw.py", file=sys.stderr)
            sys.exit(1)
        return
    OUT.parent.mkdir(exist_ok=True)
    OUT.write_text(text)
    OUT_TIMING.write_text(timing_text)
    print(f"{OUT}: {len(lines)} lines", file=sys.stderr)


if __name__ == "__main__":
    main()
