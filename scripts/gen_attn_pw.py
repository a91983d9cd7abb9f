#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 instruction stream of `attention_pw_kernel_bf16`
(voice_activity_detection_amd/csrc/savad_attn_pw_bf16.inc): the bf16 flash-attention stage
(vad/modeling/transformer.py:305-346,351-363 of the reference) as a PERSISTENT workgroup of 4 waves, one per SIMD,
each wave owning 64 query rows (two 32-row query blocks A / B), with the whole 512-entry register file owned by the
instruction stream below (no compiler-allocated register inside it).

    python scripts/gen_attn_pw.py            # rewrites the .inc
    python scripts/gen_attn_pw.py --check    # exit 1 when the committed .inc is stale (tests/test_abi_and_host.py)

Data layout: savad_kernels_bf16.h (fragment-major q / k / v^T / ctx, 1 KiB per K-step fragment of 32 rows).
Arithmetic: identical, operation for operation, to attention_kernel_bf16 (online softmax in the base-2 domain relative
to a per-row reference that rides in as the C operand of the first S^T MFMA; the reference moves when a row maximum
drifts 2^16 above it) -- the two kernels produce the same bits, which is how this one is tested.

Structure
  work items : (sequence b, group g of 8 query blocks); a workgroup walks its items (all full groups first, then the
               ragged tail groups); sequences with b % 8 == xcd stay on one XCD, so the groups of a sequence share
               its K / V^T in that XCD's L2.
  K / V^T    : 64 keys (K 2 blocks | V^T 2 blocks = 32 KiB) per LDS stage, ring of 4 stages, filled by LDS-DMA
               (global_load_lds_dwordx4, 1 KiB per instruction, 8 per wave and stage) three stages ahead of the
               compute; the stream runs continuously ACROSS items (a second item cursor feeds the DMA).
  per stage  : ONE barrier; two steps (key blocks) of 32 MFMAs each:
               phase A  16 x  S^T(i+1) = K(i+1) Q^T    beside  exp / row sum / bf16 pack of tile i, V^T(i) reads
               phase B  16 x  O^T += V^T(i) P^T(i)     beside  K(i+2) reads, row maxima of tile i+1, DMA pieces
  registers  : a[0:127] O^T (A, B), a[128:191] Q (A, B), a[192:223] K fragments, a[224:255] V^T fragments;
               v[0:63] two score tiles per query block (ping-pong), v[64:95] -reference vectors, then P, exps, staging.
Hazards stated by hand (CDNA3/4 ISA, manually inserted wait states): an MFMA result is read by a VALU instruction no
sooner than 8 MFMAs (or 3 x s_nop 7) later; a transcendental result is never consumed by the next instruction; 2 wait
states before v_permlane32_swap reads a VALU result; s_nop 0 between an M0 write and the LDS-DMA that reads it.
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "voice_activity_detection_amd" / "csrc" / "savad_attn_pw_bf16.inc"
OUT_CLOB = ROOT / "voice_activity_detection_amd" / "csrc" / "savad_attn_pw_bf16_clobbers.inc"

# ---------------------------------------------------------------------------------------------- register map
A_O = {"A": 0, "B": 64}        # O^T accumulators: 4 feature blocks x 16
A_Q = {"A": 128, "B": 160}     # Q fragments: 8 x 4
A_K = 192                      # K fragments of the tile whose scores are computed next: 8 x 4
A_V = 224                      # V^T fragments of the tile whose probabilities are consumed next: 8 x 4
V_S = {(0, "A"): 0, (0, "B"): 16, (1, "A"): 32, (1, "B"): 48}   # score tiles [buffer][query block]
V_NEGM = {"A": 64, "B": 80}    # -reference of the lane's query row, replicated in 16 registers (MFMA C operand)
V_P = {"A": 96, "B": 104}      # probabilities as bf16 B-operand fragments (2 x 4 registers)
V_E = {"A": 112, "B": 128}     # exponentials of the tile being normalised (fp32)
V_L = {"A": 144, "B": 145}     # this lane's half of the running row sums
V_RS = {"A": 146, "B": 147}    # row sum of the current tile
V_MX = {"A": 148, "B": 149}    # row maxima
V_T0, V_T1, V_T2, V_T3, V_T4, V_T5 = 150, 151, 152, 153, 154, 155
V_OFF = [158, 159, 160, 161]   # lane * 16 + k * 4096 (DMA / Q / ctx offsets); V_OFF[0] = lane * 16
V_LANE16 = V_OFF[0]
V_ADDR_V = 162                 # LDS address of the compute stage: slot base + lane * 16
V_ADDR_K = 163                 # LDS address of the NEXT stage
V_H4 = 164                     # 4 * (lane >> 5)
V_M = 165                      # lane & 31
V_LIM = 166                    # key limit of a ragged tile for this lane
V_INV = {"A": 167, "B": 168}
V_D = 169                      # cold path: reference shift
V_AL = 170                     # cold path: rescale factor
V_NEGB = 171                   # -1e30 (masked keys)
V_QS = 176                     # Q staging for the next item: 16 fragments x 4 = v[176:239]
V_CTX = {"A": 32, "B": 96}     # packed context of the finished item (32 registers each): registers dead at a seam
S_RET = 24                     # s[24:25] return address of the cold-path subroutines
S_QF, S_KF, S_VTF, S_CTXF = 36, 38, 40, 42
S_B, S_T, S_QB, S_NST, S_NGF, S_TAILQ = 44, 45, 46, 47, 48, 49
S_XCD, S_J, S_BX, S_STRIDE, S_DQ, S_DR = 50, 51, 52, 53, 54, 55
S_W, S_LDS = 56, 57
S_C16, S_CM16, S_NEGBIG = 58, 59, 60
S_CC = 61                      # compute cursor [phase, bi, g, valid]: s[61:64]
S_DC = 65                      # dma cursor: s[65:68]
S_DS = 69                      # dma: stage within its item
S_DKS, S_DVS = 70, 72          # dma: K / V^T source of the stage being filled (64 bit each, this wave's KiB)
S_DLDS = 74                    # dma: LDS byte address of the stage being filled (+ w KiB)
S_DSTREAM = 75                 # dma: stream stage counter
S_CSTREAM = 76                 # compute: stream stage counter
S_STEP = 77                    # compute: key block index of the next step
S_QSRC = 78                    # s[78:79]: Q fragments of block A of the NEXT item; s[80:81]: of block B
S_QSRCB = 80
S_CDST = 82                    # s[82:83]: ctx destination of block A of the finished item; s[84:85] block B
S_CDSTB = 84
S_FLAGS = 86                   # current item: bit 0 wave active, bit 1 block B live
S_PEND = 87                    # finished item waiting for its stores: bit 0 block A, bit 1 block B
S_QPEND = 88                   # 1: Q of the next item still has to be requested
S_T0, S_T1, S_T2, S_T3, S_T4, S_T5 = 89, 90, 91, 94, 92, 93   # s[S_T4:S_T5] is used as a 64-bit pair
S_FORCE = 96                   # s[96:97]: all ones when tile i+1 must take the cold path (ragged last key block)
S_QA = 98                      # first live query block of the wave (index within the sequence)
S_RAG = 99                     # T & 31
S_VALID = {"A": 28, "B": 29}   # rows of block A / B that exist (T - 32 qb), for the store mask
S_NFLAGS = 30                  # flags of the NEXT item (compute cursor is advanced early)
S_NQA = 31
S_SEQBLK = 26                  # first block of the current item's sequence (block space)
S_NSEQBLK = 27

NEG_BIG_BITS = 0xF149F2CA      # -1.0e30f
BLK, FRAG, STAGE, NRING = 8192, 1024, 32768, 4


class Asm:
    def __init__(self):
        self.lines = []
        self.n = 0

    def i(self, text):
        self.lines.append("\t" + text)

    def label(self, name):
        self.lines.append(name + ":")

    def c(self, text):
        self.lines.append("\t// " + text)

    def uniq(self, stem):
        self.n += 1
        return f".Lpw_{stem}_{self.n}"


def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def ar(b, n=1):
    return f"a{b}" if n == 1 else f"a[{b}:{b + n - 1}]"


def sr(b, n=1):
    return f"s{b}" if n == 1 else f"s[{b}:{b + n - 1}]"


# ------------------------------------------------------------------------------------------------ building blocks
def softmax_ops(buf, blk):
    """exp / row sum / pack of one query block's score tile as single instructions in dependency order:
    e_r = 2^s_r ; rs = ((e_0 + e_1) + e_2) + ... (sequential, as attention_kernel_bf16) ; l += rs ; P = bf16(e)."""
    S, E, P, RS = V_S[(buf, blk)], V_E[blk], V_P[blk], V_RS[blk]
    ops = [f"v_exp_f32 {vr(E + r)}, {vr(S + r)}" for r in range(16)]
    for r in range(1, 16):
        ops.append(f"v_add_f32 {vr(RS)}, {vr(E) if r == 1 else vr(RS)}, {vr(E + r)}")
        if r % 2 == 1:
            ops.append(f"v_cvt_pk_bf16_f32 {vr(P + r // 2)}, {vr(E + r - 1)}, {vr(E + r)}")
    ops.append(f"v_add_f32 {vr(V_L[blk])}, {vr(V_L[blk])}, {vr(RS)}")
    return ops  # 16 + 15 + 8 + 1 = 40


def max_ops(buf, blk, dst):
    S = V_S[(buf, blk)]
    ops = [f"v_max3_f32 {vr(dst)}, {vr(S)}, {vr(S + 1)}, {vr(S + 2)}"]
    for r in range(3, 15, 2):
        ops.append(f"v_max3_f32 {vr(dst)}, {vr(dst)}, {vr(S + r)}, {vr(S + r + 1)}")
    ops.append(f"v_max_f32 {vr(dst)}, {vr(dst)}, {vr(S + 15)}")
    return ops  # 8


def half_exchange(dst, tmp, op):
    """dst = op(lower half's value, upper half's value) on both lanes of a row"""
    return [f"v_mov_b32 {vr(tmp)}, {vr(dst)}", "s_nop 1", f"v_permlane32_swap_b32 {vr(dst)}, {vr(tmp)}",
            f"{op} {vr(dst)}, {vr(dst)}, {vr(tmp)}"]


def dma_piece_ops(k, half):
    """one 1 KiB LDS-DMA piece of the stage being filled: half 0 = K, 1 = V^T; k = 0..3 (KiB w + 4k of the half)"""
    src = S_DKS if half == 0 else S_DVS
    return [f"s_add_u32 m0, {sr(S_DLDS)}, {half * 16384 + k * 4096}", "s_nop 0",
            f"global_load_lds_dwordx4 {vr(V_OFF[k])}, {sr(src, 2)}"]


def mfma_s(buf, blk, ks, c0=False):
    D = V_S[(buf, blk)]
    C = "0" if c0 else (vr(V_NEGM[blk], 16) if ks == 0 else vr(D, 16))
    return f"v_mfma_f32_32x32x16_bf16 {vr(D, 16)}, {ar(A_K + 4 * ks, 4)}, {ar(A_Q[blk] + 4 * ks, 4)}, {C}"


def mfma_pv(blk, nbd, j):
    O = A_O[blk] + 16 * nbd
    return f"v_mfma_f32_32x32x16_bf16 {ar(O, 16)}, {ar(A_V + 4 * (2 * nbd + j), 4)}, {vr(V_P[blk] + 4 * j, 4)}, {ar(O, 16)}"


def kread(f, addr, off):
    return f"ds_read_b128 {ar(A_K + 4 * f, 4)}, {vr(addr)} offset:{off + f * FRAG}"


def vread(f, off):
    return f"ds_read_b128 {ar(A_V + 4 * f, 4)}, {vr(V_ADDR_V)} offset:{16384 + off + f * FRAG}"


# ------------------------------------------------------------------------------------------------ item cursors
def emit_cursor_next(a, c, tag):
    """advance cursor c = [phase, bi, g, valid]: full groups (phase 0) with stride S_STRIDE in (bi, g) order, then the
    tail groups (phase 1), one per sequence"""
    ph, bi, g, valid = c, c + 1, c + 2, c + 3
    l_tail, l_done, l_end, l_chk = (a.uniq(tag + x) for x in ("tail", "done", "end", "chk"))
    a.i(f"s_cmp_eq_u32 {sr(ph)}, 0")
    a.i(f"s_cbranch_scc0 {l_tail}")
    a.i(f"s_add_u32 {sr(g)}, {sr(g)}, {sr(S_DR)}")
    a.i(f"s_add_u32 {sr(bi)}, {sr(bi)}, {sr(S_DQ)}")
    a.i(f"s_cmp_ge_u32 {sr(g)}, {sr(S_NGF)}")
    a.i(f"s_cbranch_scc0 {l_chk}")
    a.i(f"s_sub_u32 {sr(g)}, {sr(g)}, {sr(S_NGF)}")
    a.i(f"s_add_u32 {sr(bi)}, {sr(bi)}, 1")
    a.label(l_chk)
    a.i(f"s_cmp_lt_u32 {sr(bi)}, {sr(S_BX)}")
    a.i(f"s_cbranch_scc1 {l_end}")
    a.i(f"s_mov_b32 {sr(ph)}, 1")       # full groups exhausted: first tail group
    a.i(f"s_mov_b32 {sr(bi)}, {sr(S_J)}")
    a.i(f"s_mov_b32 {sr(g)}, {sr(S_NGF)}")
    a.i(f"s_cmp_eq_u32 {sr(S_TAILQ)}, 0")
    a.i(f"s_cbranch_scc1 {l_done}")
    a.i(f"s_cmp_lt_u32 {sr(bi)}, {sr(S_BX)}")
    a.i(f"s_cbranch_scc1 {l_end}")
    a.i(f"s_branch {l_done}")
    a.label(l_tail)
    a.i(f"s_add_u32 {sr(bi)}, {sr(bi)}, {sr(S_STRIDE)}")
    a.i(f"s_cmp_lt_u32 {sr(bi)}, {sr(S_BX)}")
    a.i(f"s_cbranch_scc1 {l_end}")
    a.label(l_done)
    a.i(f"s_mov_b32 {sr(valid)}, 0")
    a.label(l_end)


def emit_cursor_init(a, c, tag):
    """first item of this workgroup; bi, g preloaded with divmod(J, NGF)"""
    ph, bi, g, valid = c, c + 1, c + 2, c + 3
    l_tail, l_done, l_end = (a.uniq(tag + x) for x in ("itail", "idone", "iend"))
    a.i(f"s_mov_b32 {sr(valid)}, 1")
    a.i(f"s_mov_b32 {sr(ph)}, 0")
    a.i(f"s_cmp_eq_u32 {sr(S_NGF)}, 0")
    a.i(f"s_cbranch_scc1 {l_tail}")
    a.i(f"s_cmp_lt_u32 {sr(bi)}, {sr(S_BX)}")
    a.i(f"s_cbranch_scc1 {l_end}")
    a.label(l_tail)
    a.i(f"s_mov_b32 {sr(ph)}, 1")
    a.i(f"s_mov_b32 {sr(bi)}, {sr(S_J)}")
    a.i(f"s_mov_b32 {sr(g)}, {sr(S_NGF)}")
    a.i(f"s_cmp_eq_u32 {sr(S_TAILQ)}, 0")
    a.i(f"s_cbranch_scc1 {l_done}")
    a.i(f"s_cmp_lt_u32 {sr(bi)}, {sr(S_BX)}")
    a.i(f"s_cbranch_scc1 {l_end}")
    a.label(l_done)
    a.i(f"s_mov_b32 {sr(valid)}, 0")
    a.label(l_end)


def emit_seq_block(a, c, dst):
    """dst = (8 * bi + xcd) * QB: first block of the cursor's sequence"""
    a.i(f"s_lshl_b32 {sr(dst)}, {sr(c + 1)}, 3")
    a.i(f"s_add_u32 {sr(dst)}, {sr(dst)}, {sr(S_XCD)}")
    a.i(f"s_mul_i32 {sr(dst)}, {sr(dst)}, {sr(S_QB)}")


def emit_block_addr(a, dst, base, blk_s, tmp_hi, tmp_lo):
    """s[dst:dst+1] = s[base:base+1] + blk_s * 8192   (64 bit)"""
    a.i(f"s_mul_hi_u32 {sr(tmp_hi)}, {sr(blk_s)}, {BLK}")
    a.i(f"s_mul_i32 {sr(tmp_lo)}, {sr(blk_s)}, {BLK}")
    a.i(f"s_add_u32 {sr(dst)}, {sr(base)}, {sr(tmp_lo)}")
    a.i(f"s_addc_u32 {sr(dst + 1)}, {sr(base + 1)}, {sr(tmp_hi)}")


def emit_dma_item_setup(a):
    """K / V^T source of stage 0 of the DMA cursor's item (+ this wave's KiB)"""
    emit_seq_block(a, S_DC, S_T0)
    emit_block_addr(a, S_DKS, S_KF, S_T0, S_T2, S_T1)
    emit_block_addr(a, S_DVS, S_VTF, S_T0, S_T2, S_T1)
    a.i(f"s_lshl_b32 {sr(S_T1)}, {sr(S_W)}, 10")
    a.i(f"s_add_u32 {sr(S_DKS)}, {sr(S_DKS)}, {sr(S_T1)}")
    a.i(f"s_addc_u32 {sr(S_DKS + 1)}, {sr(S_DKS + 1)}, 0")
    a.i(f"s_add_u32 {sr(S_DVS)}, {sr(S_DVS)}, {sr(S_T1)}")
    a.i(f"s_addc_u32 {sr(S_DVS + 1)}, {sr(S_DVS + 1)}, 0")
    a.i(f"s_mov_b32 {sr(S_DS)}, 0")


def emit_dma_lds(a):
    """S_DLDS = LDS base + (stream stage & 3) * 32 KiB + w KiB"""
    a.i(f"s_and_b32 {sr(S_T0)}, {sr(S_DSTREAM)}, {NRING - 1}")
    a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_T0)}, 15")
    a.i(f"s_lshl_b32 {sr(S_T1)}, {sr(S_W)}, 10")
    a.i(f"s_add_u32 {sr(S_T0)}, {sr(S_T0)}, {sr(S_T1)}")
    a.i(f"s_add_u32 {sr(S_DLDS)}, {sr(S_LDS)}, {sr(S_T0)}")


def emit_dma_advance(a):
    """after the 8 pieces of a stage have been issued: next stage of the item, or stage 0 of the next item"""
    l_next, l_end = a.uniq("dnext"), a.uniq("dend")
    a.i(f"s_add_u32 {sr(S_DSTREAM)}, {sr(S_DSTREAM)}, 1")
    emit_dma_lds(a)
    a.i(f"s_add_u32 {sr(S_DS)}, {sr(S_DS)}, 1")
    a.i(f"s_cmp_lt_u32 {sr(S_DS)}, {sr(S_NST)}")
    a.i(f"s_cbranch_scc0 {l_next}")
    a.i(f"s_add_u32 {sr(S_DKS)}, {sr(S_DKS)}, {2 * BLK}")
    a.i(f"s_addc_u32 {sr(S_DKS + 1)}, {sr(S_DKS + 1)}, 0")
    a.i(f"s_add_u32 {sr(S_DVS)}, {sr(S_DVS)}, {2 * BLK}")
    a.i(f"s_addc_u32 {sr(S_DVS + 1)}, {sr(S_DVS + 1)}, 0")
    a.i(f"s_branch {l_end}")
    a.label(l_next)
    emit_cursor_next(a, S_DC, "d")
    a.i(f"s_cmp_eq_u32 {sr(S_DC + 3)}, 0")
    a.i(f"s_cbranch_scc1 {l_end}")
    emit_dma_item_setup(a)
    a.label(l_end)


def emit_item_params(a, c, flags, qa, seqblk):
    """item of cursor c -> flags (bit 0: wave has a query block, bit 1: it has two), qa = first query block of the
    wave, seqblk = first block of the sequence"""
    emit_seq_block(a, c, seqblk)
    a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(c + 2)}, 3")          # first query block of the group
    a.i(f"s_sub_u32 {sr(S_T1)}, {sr(S_QB)}, {sr(S_T0)}")   # query blocks left in the sequence
    a.i(f"s_min_u32 {sr(S_T1)}, {sr(S_T1)}, 8")
    a.i(f"s_lshl_b32 {sr(S_T2)}, {sr(S_W)}, 1")
    a.i(f"s_add_u32 {sr(qa)}, {sr(S_T0)}, {sr(S_T2)}")
    a.i(f"s_sub_i32 {sr(S_T1)}, {sr(S_T1)}, {sr(S_T2)}")   # live blocks of this wave (may be <= 0)
    a.i(f"s_max_i32 {sr(S_T1)}, {sr(S_T1)}, 0")
    a.i(f"s_min_i32 {sr(S_T1)}, {sr(S_T1)}, 2")
    a.i(f"s_mov_b32 {sr(flags)}, 0")
    a.i(f"s_cmp_ge_u32 {sr(S_T1)}, 1")
    a.i(f"s_cselect_b32 {sr(flags)}, 1, 0")
    a.i(f"s_cmp_ge_u32 {sr(S_T1)}, 2")
    a.i(f"s_cselect_b32 {sr(S_T2)}, 2, 0")
    a.i(f"s_or_b32 {sr(flags)}, {sr(flags)}, {sr(S_T2)}")


def emit_q_request(a):
    """request the NEXT item's Q fragments into the staging registers (S_NFLAGS / S_NQA / S_NSEQBLK describe it)"""
    l_skip, l_b = a.uniq("qskip"), a.uniq("qb")
    a.i(f"s_bitcmp1_b32 {sr(S_NFLAGS)}, 0")
    a.i(f"s_cbranch_scc0 {l_skip}")
    a.i(f"s_add_u32 {sr(S_T3)}, {sr(S_NSEQBLK)}, {sr(S_NQA)}")
    emit_block_addr(a, S_QSRC, S_QF, S_T3, S_T2, S_T1)
    a.i(f"s_bitcmp1_b32 {sr(S_NFLAGS)}, 1")
    a.i(f"s_cselect_b32 {sr(S_T0)}, 1, 0")                 # block B = A + 1 when it exists, else A again
    a.i(f"s_add_u32 {sr(S_T3)}, {sr(S_T3)}, {sr(S_T0)}")
    emit_block_addr(a, S_QSRCB, S_QF, S_T3, S_T2, S_T1)
    a.i("s_nop 4")
    for blk, src in (("A", S_QSRC), ("B", S_QSRCB)):
        for f in range(8):
            dst = V_QS + (0 if blk == "A" else 32) + 4 * f
            a.i(f"global_load_dwordx4 {vr(dst, 4)}, {vr(V_OFF[f // 4])}, {sr(src, 2)} offset:{(f % 4) * FRAG}")
    a.label(l_skip)


def emit_ctx_stores(a):
    """stores of the finished item's packed context (S_PEND bits), 8 x 1 KiB per block"""
    for blk, bit, dst in (("A", 0, S_CDST), ("B", 1, S_CDSTB)):
        l_skip = a.uniq("stskip")
        a.i(f"s_bitcmp1_b32 {sr(S_PEND)}, {bit}")
        a.i(f"s_cbranch_scc0 {l_skip}")
        for f in range(8):
            a.i(f"global_store_dwordx4 {vr(V_OFF[f // 4])}, {vr(V_CTX[blk] + 4 * f, 4)}, {sr(dst, 2)} offset:{(f % 4) * FRAG}")
        a.label(l_skip)
    a.i(f"s_mov_b32 {sr(S_PEND)}, 0")


def emit_stage_addrs(a):
    """LDS addresses of the compute stage (V_ADDR_V) and of the one behind it (V_ADDR_K)"""
    a.i(f"s_and_b32 {sr(S_T0)}, {sr(S_CSTREAM)}, {NRING - 1}")
    a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_T0)}, 15")
    a.i(f"s_add_u32 {sr(S_T0)}, {sr(S_T0)}, {sr(S_LDS)}")
    a.i(f"s_add_u32 {sr(S_T1)}, {sr(S_CSTREAM)}, 1")
    a.i(f"s_and_b32 {sr(S_T1)}, {sr(S_T1)}, {NRING - 1}")
    a.i(f"s_lshl_b32 {sr(S_T1)}, {sr(S_T1)}, 15")
    a.i(f"s_add_u32 {sr(S_T1)}, {sr(S_T1)}, {sr(S_LDS)}")
    a.i(f"v_add_u32 {vr(V_ADDR_V)}, {sr(S_T0)}, {vr(V_LANE16)}")
    a.i(f"v_add_u32 {vr(V_ADDR_K)}, {sr(S_T1)}, {vr(V_LANE16)}")


def emit_barrier(a):
    """own share of stream stage (compute stage + 1) landed, then everybody's; afterwards the slot of the stage before
    the compute stage may be refilled.  While the DMA cursor is live the 8 newest requests are the stage after that."""
    l_z, l_b = a.uniq("wz"), a.uniq("wb")
    a.i(f"s_cmp_eq_u32 {sr(S_DC + 3)}, 0")
    a.i(f"s_cbranch_scc1 {l_z}")
    a.i("s_waitcnt vmcnt(8)")
    a.i(f"s_branch {l_b}")
    a.label(l_z)
    a.i("s_waitcnt vmcnt(0)")
    a.label(l_b)
    a.i("s_barrier")


def emit_post_barrier(a):
    """once per item, right after the barrier of its first stage: stores of the previous item, Q request for the next"""
    l_skip = a.uniq("pbskip")
    a.i(f"s_cmp_eq_u32 {sr(S_QPEND)}, 0")
    a.i(f"s_cbranch_scc1 {l_skip}")
    emit_ctx_stores(a)
    emit_q_request(a)
    a.i(f"s_mov_b32 {sr(S_QPEND)}, 0")
    a.label(l_skip)


# ------------------------------------------------------------------------------------------------ one step
def emit_step(a, par, has_next, vblk, kblk_next, dma, advance, check):
    """Key block i (parity par: its scores sit in buffer par; the next tile's go to 1 - par).
    has_next : S^T(i+1) is computed (K(i+1) fragments are in a[192:223])
    vblk     : LDS offset of V^T(i) inside the compute stage (0 / 8192)
    kblk_next: LDS offset of K(i+2) inside the next stage, or None
    dma      : list of (k, half) pieces of the stage being filled issued in this step
    advance  : the DMA cursor moves on after the last piece
    check    : the row maxima of tile i+1 are evaluated and the cold path called when a reference has to move"""
    cur, nxt = par, 1 - par
    smA, smB = softmax_ops(cur, "A"), softmax_ops(cur, "B")
    gaps = [[] for _ in range(32)]
    for n, f in enumerate([0, 2, 4, 6, 1, 3, 5, 7]):  # V^T fragments in the order phase B consumes them
        gaps[n].append(vread(f, vblk))
    pos = 0
    for n in range(14):                               # softmax A: 40 ops over gaps 0..13
        take = 3 if n < 12 else 2
        gaps[n].extend(smA[pos:pos + take])
        pos += take
    assert pos == 40
    for n in range(8):                                # softmax B: exponentials in gaps 8..15 ...
        gaps[8 + n].extend(smB[2 * n:2 * n + 2])
    chain = smB[16:]                                  # ... chain + packing in gaps 16..23 (PV of block A)
    for n in range(8):
        gaps[16 + n].extend(chain[3 * n:3 * n + 3])
    if kblk_next is not None:
        for f in range(8):
            gaps[16 + f].append(kread(f, V_ADDR_K, kblk_next))
    if has_next and check:
        mA, mB = max_ops(nxt, "A", V_MX["A"]), max_ops(nxt, "B", V_MX["B"])
        seq = []
        for x, y in zip(mA, mB):
            seq += [x, y]
        seq.append(f"v_max_f32 {vr(V_T1)}, {vr(V_MX['A'])}, {vr(V_MX['B'])}")
        seq += half_exchange(V_T1, V_T0, "v_max_f32")
        seq.append(f"v_cmp_lt_f32 vcc, {sr(S_C16)}, {vr(V_T1)}")
        per = (len(seq) + 7) // 8
        for n in range(8):
            gaps[24 + n].extend(seq[per * n:per * (n + 1)])
    mf = []
    if has_next:
        for ks in range(8):
            mf.append(mfma_s(nxt, "A", ks))
            mf.append(mfma_s(nxt, "B", ks))
    else:
        mf += [None] * 16
    for blk in ("A", "B"):
        for j in range(2):
            for nbd in range(4):
                mf.append(mfma_pv(blk, nbd, j))
    dma_gaps = {4: [25, 27, 29, 31], 8: [17, 19, 21, 23, 25, 27, 29, 31], 0: []}[len(dma)]
    a.i("s_waitcnt lgkmcnt(0)")       # K(i+1) fragments, requested one phase ago
    for n in range(32):
        if n == 16:
            a.i("s_waitcnt lgkmcnt(0)")  # V^T(i) fragments
        if mf[n] is not None:
            a.i(mf[n])
        for op in gaps[n]:
            a.i(op)
        if n in dma_gaps:
            for op in dma_piece_ops(*dma[dma_gaps.index(n)]):
                a.i(op)
    if advance:
        emit_dma_advance(a)
    if has_next and check:
        l_cont = a.uniq("cont")
        a.i(f"s_or_b64 {sr(S_T4, 2)}, vcc, {sr(S_FORCE, 2)}")
        a.i(f"s_cbranch_scc0 {l_cont}")
        a.i(f"s_call_b64 {sr(S_RET, 2)}, .Lpw_cold_{nxt}")
        a.label(l_cont)


K_PIECES = [(k, 0) for k in range(4)]
V_PIECES = [(k, 1) for k in range(4)]


def emit_stage(a, kind, with_dma):
    """kind: 'mid' (two steps, both followed by another tile), 'last2' (QB even: the second step is the item's
    last), 'last1' (QB odd: a single step, the item's last)"""
    emit_barrier(a)
    emit_post_barrier(a)
    emit_stage_addrs(a)
    # the ragged last key block (T % 32 != 0) goes through the cold path, which masks it
    if kind == "mid":
        # tile 2s+1 is never the last here; tile 2s+2 is the last one iff S_STEP + 3 == QB
        emit_step(a, 0, True, 0, 0, K_PIECES if with_dma else [], False, True)
        emit_force_for(a, 3)   # FORCE for the check of the second step: tile S_STEP + 2
        emit_step(a, 1, True, BLK, BLK, V_PIECES if with_dma else [], with_dma, True)
    elif kind == "last2":
        emit_force_for(a, 2)   # the second tile of the stage is the last one
        emit_step(a, 0, True, 0, None, K_PIECES if with_dma else [], False, True)
        emit_step(a, 1, False, BLK, None, V_PIECES if with_dma else [], with_dma, False)
    else:
        emit_step(a, 0, False, 0, None, (K_PIECES + V_PIECES) if with_dma else [], with_dma, False)


def emit_force_for(a, delta):
    """S_FORCE = all ones when (S_STEP + delta == QB) and the sequence has a ragged last key block"""
    a.i(f"s_add_u32 {sr(S_T0)}, {sr(S_STEP)}, {delta}")
    a.i(f"s_cmp_eq_u32 {sr(S_T0)}, {sr(S_QB)}")
    a.i(f"s_cselect_b32 {sr(S_T0)}, {sr(S_RAG)}, 0")
    a.i(f"s_cmp_lg_u32 {sr(S_T0)}, 0")
    a.i(f"s_cselect_b64 {sr(S_FORCE, 2)}, -1, 0")


# ------------------------------------------------------------------------------------------------ cold path
def emit_cold(a, buf, first):
    """Reference move of the freshly computed score tile in buffer `buf` (scores relative to the current reference):
    ragged last key block masked first; first = the item's first tile (reference := row maximum when it is more than
    2^16 away from 0, nothing to rescale); otherwise O and l of rows whose maximum exceeds the reference by 2^16 are
    rescaled.  Operation for operation online_softmax_shifted() of savad_kernels_bf16.h.  Entered with every MFMA
    that wrote the tile (and, when not first, every PV MFMA) at least 8 MFMAs or 24 wait states behind."""
    l_nomask = a.uniq("nomask")
    a.i("s_nop 7")
    a.i("s_nop 7")
    a.i("s_nop 7")
    a.i(f"s_cmp_eq_u64 {sr(S_FORCE, 2)}, 0")
    a.i(f"s_cbranch_scc1 {l_nomask}")
    # lim = T - 32 * tile - 4h = RAG - 4h for the last tile: key 8(r>>2) + (r&3) exists iff it is < lim
    a.i(f"v_sub_u32 {vr(V_LIM)}, {sr(S_RAG)}, {vr(V_H4)}")
    for blk in ("A", "B"):
        S = V_S[(buf, blk)]
        for r in range(16):
            kidx = 8 * (r >> 2) + (r & 3)
            a.i(f"v_cmp_lt_i32 vcc, {kidx}, {vr(V_LIM)}")
            a.i(f"v_cndmask_b32 {vr(S + r)}, {vr(V_NEGB)}, {vr(S + r)}, vcc")
    a.label(l_nomask)
    for blk in ("A", "B"):
        S = V_S[(buf, blk)]
        l_skip = a.uniq("coldskip")
        for op in max_ops(buf, blk, V_MX[blk]):
            a.i(op)
        for op in half_exchange(V_MX[blk], V_T0, "v_max_f32"):
            a.i(op)
        # move = mx > 16 (|| first && mx < -16); d = move ? mx : 0
        a.i(f"v_cmp_lt_f32 vcc, {sr(S_C16)}, {vr(V_MX[blk])}")
        if first:
            a.i(f"v_cmp_gt_f32 {sr(S_T4, 2)}, {sr(S_CM16)}, {vr(V_MX[blk])}")
            a.i(f"s_or_b64 vcc, vcc, {sr(S_T4, 2)}")
        a.i(f"s_cmp_eq_u64 vcc, 0")
        a.i(f"s_cbranch_scc1 {l_skip}")
        a.i(f"v_cndmask_b32 {vr(V_D)}, 0, {vr(V_MX[blk])}, vcc")
        if not first:
            a.i(f"v_exp_f32 {vr(V_AL)}, -{vr(V_D)}")
            a.i("s_nop 0")
            a.i(f"v_mul_f32 {vr(V_L[blk])}, {vr(V_L[blk])}, {vr(V_AL)}")
            for r in range(64):
                a.i(f"v_accvgpr_read_b32 {vr(V_T2)}, {ar(A_O[blk] + r)}")
                a.i("s_nop 0")
                a.i(f"v_mul_f32 {vr(V_T2)}, {vr(V_T2)}, {vr(V_AL)}")
                a.i(f"v_accvgpr_write_b32 {ar(A_O[blk] + r)}, {vr(V_T2)}")
        for r in range(16):
            a.i(f"v_sub_f32 {vr(S + r)}, {vr(S + r)}, {vr(V_D)}")
        for r in range(16):
            a.i(f"v_sub_f32 {vr(V_NEGM[blk] + r)}, {vr(V_NEGM[blk] + r)}, {vr(V_D)}")
        a.label(l_skip)
    a.i("s_nop 1")
    a.i(f"s_setpc_b64 {sr(S_RET, 2)}")


# ------------------------------------------------------------------------------------------------ item prologue / epilogue
def emit_item_prologue(a):
    """Q of this item from staging into a[128:191]; S^T(0) with C = 0 beside the zeroing of O; reference of tile 0"""
    l_q8, l_qd = a.uniq("q8"), a.uniq("qd")
    a.i(f"s_cmp_eq_u32 {sr(S_DC + 3)}, 0")
    a.i(f"s_cbranch_scc1 {l_q8}")
    a.i("s_waitcnt vmcnt(8)")   # the staged Q is older than the newest stage of DMA pieces
    a.i(f"s_branch {l_qd}")
    a.label(l_q8)
    a.i("s_waitcnt vmcnt(0)")
    a.label(l_qd)
    for r in range(64):
        a.i(f"v_accvgpr_write_b32 {ar(A_Q['A'] + r)}, {vr(V_QS + r)}")
    emit_stage_addrs(a)
    for f in range(8):
        a.i(kread(f, V_ADDR_V, 0))
    for blk in ("A", "B"):
        a.i(f"v_mov_b32 {vr(V_L[blk])}, 0")
        for r in range(16):
            a.i(f"v_mov_b32 {vr(V_NEGM[blk] + r)}, 0")
    a.i("s_waitcnt lgkmcnt(0)")
    n = 0
    for ks in range(8):
        for blk in ("A", "B"):
            a.i(mfma_s(0, blk, ks, c0=(ks == 0)))
            for _ in range(8):
                a.i(f"v_accvgpr_write_b32 {ar(n)}, 0")
                n += 1
    assert n == 128
    for f in range(8):
        a.i(kread(f, V_ADDR_V, BLK))   # K(1): the stage's second block
    # tile 0 is the last tile only when QB == 1, which this kernel never sees (T > 32)
    a.i(f"s_mov_b64 {sr(S_FORCE, 2)}, 0")
    a.i(f"s_call_b64 {sr(S_RET, 2)}, .Lpw_coldfirst_0")
    a.i(f"s_mov_b32 {sr(S_STEP)}, 0")


def emit_item_epilogue(a):
    """normalise O by the row sums, pack to bf16 fragments into the staging registers of the stores, describe them"""
    a.i("s_nop 7")
    a.i("s_nop 7")
    a.i("s_nop 7")
    for blk in ("A", "B"):
        # total row sum: both halves
        a.i(f"v_mov_b32 {vr(V_T0)}, {vr(V_L[blk])}")
        a.i("s_nop 1")
        a.i(f"v_permlane32_swap_b32 {vr(V_L[blk])}, {vr(V_T0)}")
        a.i(f"v_add_f32 {vr(V_T0)}, {vr(V_L[blk])}, {vr(V_T0)}")
        x, out = V_T0, V_INV[blk]
        t0, t1, t2, t3 = V_T1, V_T2, V_T3, V_T4
        a.i(f"v_div_scale_f32 {vr(t0)}, {sr(S_T4, 2)}, {vr(x)}, {vr(x)}, 1.0")
        a.i(f"v_rcp_f32 {vr(t1)}, {vr(t0)}")
        a.i("s_nop 0")
        a.i(f"v_fma_f32 {vr(t2)}, -{vr(t0)}, {vr(t1)}, 1.0")
        a.i(f"v_fmac_f32 {vr(t1)}, {vr(t2)}, {vr(t1)}")
        a.i(f"v_div_scale_f32 {vr(t2)}, vcc, 1.0, {vr(x)}, 1.0")
        a.i(f"v_mul_f32 {vr(t3)}, {vr(t2)}, {vr(t1)}")
        a.i(f"v_fma_f32 {vr(out)}, -{vr(t0)}, {vr(t3)}, {vr(t2)}")
        a.i(f"v_fmac_f32 {vr(t3)}, {vr(out)}, {vr(t1)}")
        a.i(f"v_fma_f32 {vr(t2)}, -{vr(t0)}, {vr(t3)}, {vr(t2)}")
        a.i("s_nop 1")
        a.i(f"v_div_fmas_f32 {vr(t2)}, {vr(t2)}, {vr(t1)}, {vr(t3)}")
        a.i(f"v_div_fixup_f32 {vr(out)}, {vr(t2)}, {vr(x)}, 1.0")
        # rows of the block that do not exist (ragged last query block) store exact zeros
        a.i(f"v_cmp_gt_i32 vcc, {sr(S_VALID[blk])}, {vr(V_M)}")
        a.i("s_nop 1")
        for nbd in range(4):
            for e in range(8):
                r0 = 16 * nbd + 2 * e
                a.i(f"v_accvgpr_read_b32 {vr(V_T1)}, {ar(A_O[blk] + r0)}")
                a.i(f"v_accvgpr_read_b32 {vr(V_T2)}, {ar(A_O[blk] + r0 + 1)}")
                a.i(f"v_mul_f32 {vr(V_T1)}, {vr(V_T1)}, {vr(V_INV[blk])}")
                a.i(f"v_mul_f32 {vr(V_T2)}, {vr(V_T2)}, {vr(V_INV[blk])}")
                a.i(f"v_cvt_pk_bf16_f32 {vr(V_T1)}, {vr(V_T1)}, {vr(V_T2)}")
                a.i(f"v_cndmask_b32 {vr(V_CTX[blk] + 8 * nbd + e)}, 0, {vr(V_T1)}, vcc")


def emit_all():
    a = Asm()
    a.c("generated by scripts/gen_attn_pw.py -- do not edit")
    # ---- inputs -> fixed homes (operands: see savad_attn_pw_bf16.h)
    a.i(f"s_mov_b64 {sr(S_QF, 2)}, %0")
    a.i(f"s_mov_b64 {sr(S_KF, 2)}, %1")
    a.i(f"s_mov_b64 {sr(S_VTF, 2)}, %2")
    a.i(f"s_mov_b64 {sr(S_CTXF, 2)}, %3")
    a.i(f"s_mov_b32 {sr(S_B)}, %4")
    a.i(f"s_mov_b32 {sr(S_T)}, %5")
    a.i(f"s_mov_b32 {sr(S_XCD)}, %6")
    a.i(f"s_mov_b32 {sr(S_J)}, %7")
    a.i(f"s_mov_b32 {sr(S_STRIDE)}, %8")
    a.i(f"s_mov_b32 {sr(S_CC + 1)}, %9")     # bi0
    a.i(f"s_mov_b32 {sr(S_CC + 2)}, %10")    # g0
    a.i(f"s_mov_b32 {sr(S_DQ)}, %11")
    a.i(f"s_mov_b32 {sr(S_DR)}, %12")
    a.i(f"s_mov_b32 {sr(S_W)}, %13")
    a.i(f"s_mov_b32 {sr(S_LDS)}, %14")
    a.i(f"v_mov_b32 {vr(V_LANE16)}, %15")
    a.i(f"s_mov_b64 exec, -1")
    # ---- derived constants
    a.i(f"s_add_u32 {sr(S_QB)}, {sr(S_T)}, 31")
    a.i(f"s_lshr_b32 {sr(S_QB)}, {sr(S_QB)}, 5")
    a.i(f"s_add_u32 {sr(S_NST)}, {sr(S_QB)}, 1")
    a.i(f"s_lshr_b32 {sr(S_NST)}, {sr(S_NST)}, 1")
    a.i(f"s_lshr_b32 {sr(S_NGF)}, {sr(S_QB)}, 3")
    a.i(f"s_and_b32 {sr(S_TAILQ)}, {sr(S_QB)}, 7")
    a.i(f"s_and_b32 {sr(S_RAG)}, {sr(S_T)}, 31")
    # sequences of this XCD: b = 8 bi + xcd < B
    a.i(f"s_add_u32 {sr(S_BX)}, {sr(S_B)}, 7")
    a.i(f"s_sub_u32 {sr(S_BX)}, {sr(S_BX)}, {sr(S_XCD)}")
    a.i(f"s_lshr_b32 {sr(S_BX)}, {sr(S_BX)}, 3")
    a.i(f"s_mov_b32 {sr(S_C16)}, 0x41800000")
    a.i(f"s_mov_b32 {sr(S_CM16)}, 0xc1800000")
    a.i(f"s_mov_b32 {sr(S_NEGBIG)}, {hex(NEG_BIG_BITS)}")
    for k in range(1, 4):
        a.i(f"v_add_u32 {vr(V_OFF[k])}, {k * 4096}, {vr(V_LANE16)}")
    a.i(f"v_lshrrev_b32 {vr(V_M)}, 4, {vr(V_LANE16)}")        # lane
    a.i(f"v_lshrrev_b32 {vr(V_H4)}, 5, {vr(V_M)}")            # h
    a.i(f"v_lshlrev_b32 {vr(V_H4)}, 2, {vr(V_H4)}")           # 4 h
    a.i(f"v_and_b32 {vr(V_M)}, 31, {vr(V_M)}")
    a.i(f"v_mov_b32 {vr(V_NEGB)}, {sr(S_NEGBIG)}")
    a.i(f"s_mov_b32 {sr(S_PEND)}, 0")
    a.i(f"s_mov_b32 {sr(S_QPEND)}, 0")
    a.i(f"s_mov_b32 {sr(S_DSTREAM)}, 0")
    a.i(f"s_mov_b32 {sr(S_CSTREAM)}, 0")
    # ---- cursors
    a.i(f"s_mov_b32 {sr(S_DC + 1)}, {sr(S_CC + 1)}")
    a.i(f"s_mov_b32 {sr(S_DC + 2)}, {sr(S_CC + 2)}")
    emit_cursor_init(a, S_CC, "c")
    emit_cursor_init(a, S_DC, "d")
    a.i(f"s_cmp_eq_u32 {sr(S_CC + 3)}, 0")
    a.i("s_cbranch_scc1 .Lpw_end")
    # first item: its Q, then three stages of the K / V^T stream
    emit_item_params(a, S_CC, S_NFLAGS, S_NQA, S_NSEQBLK)
    emit_q_request(a)
    emit_dma_item_setup(a)
    emit_dma_lds(a)
    for _ in range(3):
        l_skip = a.uniq("pdskip")
        a.i(f"s_cmp_eq_u32 {sr(S_DC + 3)}, 0")
        a.i(f"s_cbranch_scc1 {l_skip}")
        for half in (0, 1):
            for k in range(4):
                for op in dma_piece_ops(k, half):
                    a.i(op)
        emit_dma_advance(a)
        a.label(l_skip)
    # stage 0 has landed for everybody before the first item's prologue reads K(0), K(1) from it
    l_z, l_b = a.uniq("iwz"), a.uniq("iwb")
    a.i(f"s_cmp_eq_u32 {sr(S_DC + 3)}, 0")
    a.i(f"s_cbranch_scc1 {l_z}")
    a.i("s_waitcnt vmcnt(16)")
    a.i(f"s_branch {l_b}")
    a.label(l_z)
    a.i("s_waitcnt vmcnt(0)")
    a.label(l_b)
    a.i("s_barrier")

    # ================================================================================ item loop
    a.label(".Lpw_item")
    # the item the compute cursor points at was described as "next" by the previous round
    a.i(f"s_mov_b32 {sr(S_FLAGS)}, {sr(S_NFLAGS)}")
    a.i(f"s_mov_b32 {sr(S_QA)}, {sr(S_NQA)}")
    a.i(f"s_mov_b32 {sr(S_SEQBLK)}, {sr(S_NSEQBLK)}")
    # rows that exist in block A / B: T - 32 qb
    a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_QA)}, 5")
    a.i(f"s_sub_i32 {sr(S_VALID['A'])}, {sr(S_T)}, {sr(S_T0)}")
    a.i(f"s_sub_i32 {sr(S_VALID['B'])}, {sr(S_VALID['A'])}, 32")
    # ctx destinations of THIS item are needed when it is finished; the pending stores of the previous item still use
    # S_CDST: they are issued in this item's first stage, BEFORE the epilogue overwrites them.
    # advance the compute cursor and describe the next item (its Q is requested behind this item's first barrier)
    emit_cursor_next(a, S_CC, "c")
    a.i(f"s_mov_b32 {sr(S_NFLAGS)}, 0")
    l_nonext = a.uniq("nonext")
    a.i(f"s_cmp_eq_u32 {sr(S_CC + 3)}, 0")
    a.i(f"s_cbranch_scc1 {l_nonext}")
    emit_item_params(a, S_CC, S_NFLAGS, S_NQA, S_NSEQBLK)
    a.label(l_nonext)
    a.i(f"s_mov_b32 {sr(S_QPEND)}, 1")
    a.i(f"s_bitcmp1_b32 {sr(S_FLAGS)}, 0")
    a.i("s_cbranch_scc0 .Lpw_idle_item")

    emit_item_prologue(a)
    a.label(".Lpw_stage")
    a.i(f"s_sub_u32 {sr(S_T0)}, {sr(S_QB)}, {sr(S_STEP)}")
    a.i(f"s_cmp_gt_u32 {sr(S_T0)}, 2")
    a.i("s_cbranch_scc1 .Lpw_mid")
    a.i(f"s_cmp_eq_u32 {sr(S_T0)}, 2")
    a.i("s_cbranch_scc1 .Lpw_last2")
    a.i("s_branch .Lpw_last1")
    for kind in ("mid", "last2", "last1"):
        a.label(f".Lpw_{kind}")
        a.i(f"s_cmp_eq_u32 {sr(S_DC + 3)}, 0")
        a.i(f"s_cbranch_scc1 .Lpw_{kind}_nodma")
        emit_stage(a, kind, True)
        a.i(f"s_branch .Lpw_{kind}_done")
        a.label(f".Lpw_{kind}_nodma")
        emit_stage(a, kind, False)
        a.label(f".Lpw_{kind}_done")
        a.i(f"s_add_u32 {sr(S_CSTREAM)}, {sr(S_CSTREAM)}, 1")
        if kind == "mid":
            a.i(f"s_add_u32 {sr(S_STEP)}, {sr(S_STEP)}, 2")
            a.i("s_branch .Lpw_stage")
        else:
            a.i("s_branch .Lpw_item_done")
    a.label(".Lpw_item_done")
    emit_item_epilogue(a)
    # describe the stores (issued behind the next barrier, or at the end of the kernel)
    a.i(f"s_add_u32 {sr(S_T3)}, {sr(S_SEQBLK)}, {sr(S_QA)}")
    emit_block_addr(a, S_CDST, S_CTXF, S_T3, S_T2, S_T1)
    a.i(f"s_add_u32 {sr(S_T3)}, {sr(S_T3)}, 1")
    emit_block_addr(a, S_CDSTB, S_CTXF, S_T3, S_T2, S_T1)
    a.i(f"s_and_b32 {sr(S_PEND)}, {sr(S_FLAGS)}, 3")
    a.i("s_branch .Lpw_next_item")

    # ---- a wave without a query block in this item: keeps the stream and the barriers going
    a.label(".Lpw_idle_item")
    a.i(f"s_mov_b32 {sr(S_STEP)}, 0")
    a.label(".Lpw_idle_stage")
    emit_barrier(a)
    emit_post_barrier(a)
    l_nod = a.uniq("idlenodma")
    a.i(f"s_cmp_eq_u32 {sr(S_DC + 3)}, 0")
    a.i(f"s_cbranch_scc1 {l_nod}")
    for half in (0, 1):
        for k in range(4):
            for op in dma_piece_ops(k, half):
                a.i(op)
    emit_dma_advance(a)
    a.label(l_nod)
    a.i(f"s_add_u32 {sr(S_CSTREAM)}, {sr(S_CSTREAM)}, 1")
    a.i(f"s_add_u32 {sr(S_STEP)}, {sr(S_STEP)}, 2")
    a.i(f"s_cmp_lt_u32 {sr(S_STEP)}, {sr(S_QB)}")
    a.i("s_cbranch_scc1 .Lpw_idle_stage")

    a.label(".Lpw_next_item")
    a.i(f"s_cmp_eq_u32 {sr(S_CC + 3)}, 0")
    a.i("s_cbranch_scc0 .Lpw_item")
    # ---- the last item's stores
    emit_ctx_stores(a)
    a.label(".Lpw_end")
    a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a.i("s_branch .Lpw_exit")
    # ---- cold paths (subroutines)
    for buf in (0, 1):
        a.label(f".Lpw_cold_{buf}")
        emit_cold(a, buf, False)
    a.label(".Lpw_coldfirst_0")
    emit_cold(a, 0, True)
    a.label(".Lpw_exit")
    return a


def render(a):
    body = "\n".join(x for x in a.lines if x is not None)
    return ("// generated by scripts/gen_attn_pw.py -- do not edit (python scripts/gen_attn_pw.py rewrites it)\n"
            "R\"ASMPW(\n" + body + "\n)ASMPW\"\n")


def render_clobbers():
    """every register the stream names: v0..v239 (v240..v255 stay with the compiler for the input operands), a0..a255,
    s24..s31, s34..s99 (s32 / s33 are reserved by the ABI and not used), vcc, scc"""
    regs = [f"v{i}" for i in range(240)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(24, 32)] + [f"s{i}" for i in range(34, 100)] + ["vcc", "scc", "memory"]
    lines, cur = [], ""
    for r in regs:
        item = f'"{r}", '
        if len(cur) + len(item) > 118:
            lines.append(cur.rstrip())
            cur = ""
        cur += item
    lines.append(cur.rstrip().rstrip(","))
    return "// generated by scripts/gen_attn_pw.py -- do not edit\n" + "\n".join(lines) + "\n"


def main():
    a = emit_all()
    text, clob = render(a), render_clobbers()
    if "--check" in sys.argv:
        if not OUT.exists() or OUT.read_text() != text or not OUT_CLOB.exists() or OUT_CLOB.read_text() != clob:
            print(f"{OUT} is stale: run python scripts/gen_attn_pw.py", file=sys.stderr)
            sys.exit(1)
        return
    OUT.write_text(text)
    OUT_CLOB.write_text(clob)
    print(f"{OUT}: {len(a.lines)} lines", file=sys.stderr)


if __name__ == "__main__":
    main()
