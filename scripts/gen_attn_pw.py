#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 instruction stream of `attention_pw_kernel_bf16`
(voice_activity_detection_amd/csrc/savad_attn_pw_bf16.inc): the bf16 flash-attention stage
(vad/modeling/transformer.py:305-346,351-363 of the reference) as a PERSISTENT workgroup of 4 waves, one per SIMD,
each wave owning 64 query rows (two 32-row query blocks A / B), with the whole 512-entry register file owned by the
instruction stream below (no compiler-allocated register inside it).

    python scripts/gen_attn_pw.py            # rewrites the .inc files
    python scripts/gen_attn_pw.py --check    # exit 1 when a committed .inc is stale (tests/test_abi_and_host.py)
    python scripts/gen_attn_pw.py --out F [--ablate MASK] [--timing | --count] [--split-max N] [--pad N]   # experiments
                                             # (scripts/ubench/build_timing.sh: the phase-stamp build, -DSAVAD_PW_INC)

    python scripts/pw_sim.py B T [...]       # runs the stream on the functional model of scripts/gfx950_sim.py (CPU)

Data layout: savad_kernels_bf16.h (fragment-major q / k / v^T / ctx, 1 KiB per K-step fragment of 32 rows).
Arithmetic: ordinary items are identical, operation for operation, to attention_kernel_bf16 (online softmax in the
base-2 domain relative to a per-row reference that rides in as the C operand of the first S^T MFMA; the reference moves
when a row maximum drifts 2^16 above it) -- same bits; key-split tail items (below) sum a row's keys in four partial
softmaxes that meet in LDS: their rows agree with attention_kernel_bf16 to the bf16 rounding of the context.

Structure
  work items : (sequence b, group g of 8 query blocks); a workgroup walks the full groups of its XCD with a stride;
               sequences with b % 8 == xcd stay on one XCD, so the groups of a sequence share its K / V^T in that
               XCD's L2.  A sequence's ragged TAIL group runs right in front of its first full group, on the same
               workgroup (round 3 ran all tails last: every tail then streamed its sequence's 400 KiB of K / V^T from
               the fabric a second time, all workgroups at once -- 87.5 us against 78.0 at [256,800], same box).
               A tail group of one or two query blocks is a KEY-SPLIT item: all four waves take the same block(s)
               and a quarter of the key blocks each, with all 128 features (round 3's feature split had every wave
               repeat the scores and the exponentials); the partial (O, l, reference) meet through the LDS slot of
               the item's last stage.  Larger tail groups run as ordinary items with idle waves.
  K / V^T    : 64 keys (K 2 blocks | V^T 2 blocks = 32 KiB) per LDS stage, ring of 5 stages (the whole LDS), filled by
               LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction, one M0 write per half stage: the immediate
               offset moves source AND destination) FOUR stages ahead of the compute; the stream runs continuously
               ACROSS items (a second item cursor feeds it), and once it is exhausted the DMA keeps its cadence of 8
               pieces per stage with the last valid sources (they land in slots no later stage reads), so that every
               counted vmcnt keeps its meaning without a second code path.
  per stage  : ONE barrier; two steps (key blocks).  Step i is a two-stage pipeline:
               first half   O^T += V^T(i-1) P^T(i-1)   beside  K(i+1) reads, exponentials of tile i, row sum of block A
               second half  S^T(i+1) = K(i+1) Q^T      beside  V^T(i) reads, row sum of block B, bf16 packing, DMA
                                                               pieces, stream advance, ring-address rotation
               then ONE reference check: a row sum above 0.94 * 2^16 is the only way a score can have outrun the
               reference by 2^16; the out-of-line path applies online_softmax_shifted()'s own test and, where a
               reference moves, rescales O / l, RECOMPUTES the next tile's scores against it (bit-exactness: the
               reference kernel would have) and redoes the tile's exponentials.
  registers  : a[0:127] O^T (A, B), a[128:191] Q (A, B), a[192:223] K fragments, a[224:255] V^T fragments;
               v[0:63] two score tiles per query block (ping-pong), v[64:95] -reference vectors, then P, exps, staging.
Hazards stated by hand (CDNA3/4 ISA, manually inserted wait states): an MFMA result is read by a VALU instruction no
sooner than 2 MFMAs + 16 wait states (or 3 x s_nop 7) later; a transcendental result is never consumed by the next
instruction; 2 wait states before v_permlane32_swap reads a VALU result; s_nop 0 between an M0 write and the LDS-DMA
that reads it; s_nop 1 between a VALU write and the MFMA that reads it as C.
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "voice_activity_detection_amd" / "csrc" / "savad_attn_pw_bf16.inc"
OUT_CLOB = ROOT / "voice_activity_detection_amd" / "csrc" / "savad_attn_pw_bf16_clobbers.inc"
# the same stream with EVERY tail group as an ordinary item (SPLIT_MAX = 0): operation for operation the arithmetic of attention_kernel_bf16 for
# every frame, i.e. results that do not depend on which attention kernel a batch size selects (savad_set_batch_invariant; 9 % slower per launch)
OUT_NOSPLIT = ROOT / "voice_activity_detection_amd" / "csrc" / "savad_attn_pw_bf16_nosplit.inc"

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
V_RSP = {"A": 172, "B": 174}   # --rowsum pk: the two halves of the packed row sum (aligned pairs; not in timing builds: V_ACC)
V_T0, V_T1, V_T2, V_T3, V_T4, V_T5 = 150, 151, 152, 153, 154, 155
V_LDSL = 156                   # LDS base + lane * 16
V_KSADDR = 157                 # key-split items: LDS address of this wave's key block (K image; V^T image 16 KiB behind it)
V_OFF = [158, 159]             # lane * 16, lane * 16 + 4096 (global offsets of Q loads / ctx stores / DMA pieces)
V_LANE16 = V_OFF[0]
V_SCR16, V_SCR8 = 160, 161     # key-split items: combine scratch (the slot of the item's last stage) + lane * 16 / + lane * 8
V_ADDR_V = 162                 # LDS address of the compute stage: slot base + lane * 16
V_ADDR_K = 163                 # LDS address of the stage behind it
V_H4 = 164                     # 4 * (lane >> 5)
V_M = 165                      # lane & 31
V_LIM = 166                    # key limit of a ragged tile for this lane
V_INV = {"A": 167, "B": 168}
V_D = 169                      # cold path: reference shift
V_AL = 170                     # cold path: rescale factor
V_NEGB = 171                   # -1e30 (masked keys)
V_ACC = 172                    # timing builds: lane c = cycles of category c, lane 31 = previous stamp
V_QS = 176                     # Q staging for the next item: 16 fragments x 4 = v[176:239]
V_CTX = {"A": 32, "B": 96}     # packed context of the finished item (32 registers each): registers dead at a seam
S_RET = 24                     # s[24:25]: return address of the out-of-line subroutines (they never nest)
S_SEQBLK, S_NSEQBLK = 26, 27   # first block (block space) of the current / next item's sequence
S_VALID = {"A": 28, "B": 29}   # rows of block A / B that exist (T - 32 qb)
S_NFLAGS, S_NQA = 30, 31       # the NEXT item (the compute cursor is advanced at the start of an item)
S_QF, S_KF, S_VTF, S_CTXF = 36, 38, 40, 42
S_RAG, S_T, S_QB, S_NST, S_NGF, S_TAILQ = 44, 45, 46, 47, 48, 49
S_XCD, S_J, S_BX, S_STRIDE, S_DQ, S_DR = 50, 51, 52, 53, 54, 55
S_W, S_LDS = 56, 57
S_QA = 60                      # first query block of the wave in this item
S_CC = 61                      # compute cursor [phase, bi, g, valid]: s[61:64]
S_DC = 65                      # dma cursor: s[65:68]
S_DS = 69                      # dma: stage (within its item) that is issued next
S_DKS, S_DVS = 70, 72          # dma: K / V^T source of that stage (64 bit each, + this wave's 4 KiB)
S_DLDS = 74                    # dma: LDS byte address it goes to (+ this wave's 4 KiB)
S_DSTREAM = 75                 # dma: stream stage counter
S_KS = 76                      # LDS byte offset of the stage behind the compute stage (K reads)
S_CNT = 77                     # 'mid' stages left in the item
S_DBASE = 78                   # dma: LDS base the ring slot offset is added to (this wave's share; moved once the stream is exhausted)
S_TD = 79                      # dma: scratch of the in-gap stream advance
S_CDST, S_CDSTB = 82, 84       # 64 bit each: ctx destination of blocks A / B of the finished item
S_FLAGS = 86                   # current item: bit 0 wave has a block, bit 1 it has two, bit 2 feature-split tail item
S_PEND = 87                    # finished item waiting for its stores: bits as S_FLAGS
S_QPEND = 88                   # 1: the once-per-item block (stores, Q request) still has to run behind a barrier
S_T0, S_T1, S_T2, S_T3, S_T4, S_T5 = 89, 90, 91, 94, 92, 93   # s[S_T4:S_T5] is used as a 64-bit pair
S_LDSW = 95                    # LDS base + w * 4096
S_KB = 96                      # key-split items: the key block this wave takes from the current pair of stages
S_KFIRST = 97                  # key-split items: 1 until this wave has seen its first key block
S_SCRB = 98                    # key-split items: LDS byte address of the combine scratch
S_KPEND = 99                   # key-split items: 1 while a tile's probabilities wait for their PV MFMAs
S_ATTSH, S_ATTMASK = 34, 35    # a sequence's tail item is attached to its full group g = (bi >> S_ATTSH) & S_ATTMASK
S_PRE = 80                     # 1: the previous item's seam already moved this item's Q into place and computed S^T(0)
FLT_MAX_BITS = "0x7f7fffff"
S_TM = 58                      # timing builds: s[58:59]
TIMING = False
WGTIME = False       # timing builds (--wgtime): no phase stamps; wave 0 of every workgroup with j < 16 stores its (start, end) on the 100 MHz clock
PAD = 0              # experiments: s_nop 0 instructions in front (shifts the stream by 4 bytes each)
COUNT_ONLY = False   # timing builds: only the cold-path call counters, no stamps
SPLIT_MAX = 2   # tail groups of up to this many query blocks are key-split items (experiments: 0, 1)
ATTACH = True   # a sequence's tail item runs right in front of its first full item (False: all tail items last, round 3's order)
NT = 3          # bit 0: Q loads, bit 1: ctx stores carry the non-temporal bit -- they have no reuse, K / V^T have (same time at
                # [256,800], -5 % at [384,800], 8 MB less fabric traffic per launch; --nt 0 for the plain loads / stores)
POLICY = {1: " nt", 2: " sc0 sc1", 3: " sc0 sc1 nt", 4: " sc1", 5: " sc0"}   # experiments: --qpol / --cpol pick another cache policy


def q_policy():
    return QPOL if QPOL is not None else (" nt" if NT & 1 else "")


def ctx_policy():
    return CPOL if CPOL is not None else (" nt" if NT & 2 else "")


QPOL = CPOL = None
BALANCE = 0     # experiments (--balance N): 1 the first half of a step (K reads, exponentials, block A's row sum beside the PV MFMAs) is laid
                # out by estimated ISSUE cost per MFMA gap instead of by instruction count; 2 the second half too
ROWSUM = "seq"  # experiments (--rowsum pk): a tile's row sum as 7 packed adds (two exponentials per instruction, v_pk_add_f32) + 1 add
                # instead of 15 sequential adds -- 14 VALU instructions less per wave and step; changes the summation order (the
                # first-generation kernel's is the sequential one), so every row's low bits
SEAM = True     # an ordinary item's last PV MFMAs / epilogue share their MFMA gaps with the NEXT item's Q move and S^T(0) (False: round 3)
ABLATE = 0   # experiments (results WRONG): 1 no DMA pieces, 2 no barrier, 4 no row maxima / reference check, 8 no exp / sum / pack,
             # 16 no LDS operand reads

CHECK_BITS = "0x47700000"      # 61440.0 = 0.9375 * 2^16: a row whose maximum exceeds the reference by MORE than 2^16 has an
                               # exponential >= 2^16 (1 - ulp) in its sum, whatever v_exp_f32 rounds to: the check is a strict
                               # superset of online_softmax_shifted()'s own test, which the out-of-line path then applies exactly
NEG_BIG_BITS = 0xF149F2CA      # -1.0e30f
BLK, FRAG, STAGE, NRING = 8192, 1024, 32768, 5     # 5 stages of 32 KiB: the whole LDS of a CU
AHEAD = NRING - 1              # the DMA runs this many stages ahead of the compute; once the stream is exhausted it keeps its
                               # cadence (8 pieces per stage, so every counted vmcnt keeps its meaning) with the last valid
                               # sources: those pieces land in ring slots no later stage reads


class Asm:
    def __init__(self):
        self.lines = []
        self.tail = []   # out-of-line blocks, appended behind the main stream
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

    def ool_call(self, target):
        """a not-taken-in-the-common-case branch on SCC == 1 to an out-of-line stub that calls subroutine `target`
        (return address in S_RET) and comes back behind the branch"""
        stub, back = self.uniq("stub"), self.uniq("back")
        self.i(f"s_cbranch_scc1 {stub}")
        self.label(back)
        self.tail += [stub + ":", f"\ts_call_b64 {sr(S_RET, 2)}, {target}", f"\ts_branch {back}"]


def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def ar(b, n=1):
    return f"a{b}" if n == 1 else f"a[{b}:{b + n - 1}]"


def sr(b, n=1):
    return f"s{b}" if n == 1 else f"s[{b}:{b + n - 1}]"


def stamp(a, cat):
    """timing builds: cycles since the previous stamp are added to category `cat` (SCC is clobbered; LDS reads drained)"""
    if not TIMING or COUNT_ONLY:
        return
    a.i(f"s_memtime {sr(S_TM, 2)}")
    a.i("s_waitcnt lgkmcnt(0)")
    a.i(f"v_readlane_b32 {sr(S_TM + 1)}, {vr(V_ACC)}, 31")
    a.i(f"s_sub_u32 {sr(S_TM + 1)}, {sr(S_TM)}, {sr(S_TM + 1)}")
    a.i(f"v_writelane_b32 {vr(V_ACC)}, {sr(S_TM)}, 31")
    a.i(f"v_readlane_b32 {sr(S_TM)}, {vr(V_ACC)}, {cat}")
    a.i(f"s_add_u32 {sr(S_TM)}, {sr(S_TM)}, {sr(S_TM + 1)}")
    a.i(f"v_writelane_b32 {vr(V_ACC)}, {sr(S_TM)}, {cat}")


# ------------------------------------------------------------------------------------------------ building blocks
def softmax_ops(buf, blk):
    """exp / row sum / pack of one query block's score tile as single instructions in dependency order:
    e_r = 2^s_r ; rs = ((e_0 + e_1) + e_2) + ... (sequential, as attention_kernel_bf16) ; l += rs ; P = bf16(e)."""
    S, E, P, RS = V_S[(buf, blk)], V_E[blk], V_P[blk], V_RS[blk]
    ops = [f"v_exp_f32 {vr(E + r)}, {vr(S + r)}" for r in range(16)]
    if ROWSUM == "pk":   # (e_0 + e_2 + ... + e_14) + (e_1 + e_3 + ... + e_15), the two chains side by side in one register pair
        assert not TIMING, "--rowsum pk uses the timing builds' accumulator register"
        R2 = V_RSP[blk]
        for k in range(1, 8):
            ops.append(f"v_pk_add_f32 {vr(R2, 2)}, {vr(E, 2) if k == 1 else vr(R2, 2)}, {vr(E + 2 * k, 2)}")
        ops.append(f"v_add_f32 {vr(RS)}, {vr(R2)}, {vr(R2 + 1)}")
        for k in range(8):
            ops.append(f"v_cvt_pk_bf16_f32 {vr(P + k)}, {vr(E + 2 * k)}, {vr(E + 2 * k + 1)}")
        ops.append(f"v_add_f32 {vr(V_L[blk])}, {vr(V_L[blk])}, {vr(RS)}")
        return ops  # 16 + 8 + 8 + 1 = 33
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


def mfma_pv(blk, nbd, j, c0=False):
    O = A_O[blk] + 16 * nbd
    return f"v_mfma_f32_32x32x16_bf16 {ar(O, 16)}, {ar(A_V + 4 * (2 * nbd + j), 4)}, {vr(V_P[blk] + 4 * j, 4)}, {'0' if c0 else ar(O, 16)}"


def kread(f, addr, off):
    return f"ds_read_b128 {ar(A_K + 4 * f, 4)}, {vr(addr)} offset:{off + f * FRAG}"


def vread(f, off):
    return f"ds_read_b128 {ar(A_V + 4 * f, 4)}, {vr(V_ADDR_V)} offset:{16384 + off + f * FRAG}"


# ------------------------------------------------------------------------------------------------ item cursors
def emit_cursor_next(a, c, tag):
    """advance cursor c = [ph, bi, g, valid].  ph 0: the item is the full group (bi, g), reached with stride S_STRIDE in
    (bi, g) order; ph 2: the item is the TAIL group of sequence bi, running right in front of that sequence's first
    full group (bi, 0), which follows it on this workgroup -- the tail streams the sequence's K / V^T through the L2 just
    before (and while) the three full groups read it; ph 1: sequences without a full group (QB < 8): tail groups only,
    one per sequence.  (ATTACH = False: round 3's order, all full groups, then all tail groups in phase 1.)"""
    ph, bi, g, valid = c, c + 1, c + 2, c + 3
    l_tail, l_done, l_end, l_chk, l_full = (a.uniq(tag + x) for x in ("tail", "done", "end", "chk", "full"))
    if ATTACH:
        a.i(f"s_cmp_eq_u32 {sr(ph)}, 2")
        a.i(f"s_cbranch_scc0 {l_full}")
        a.i(f"s_mov_b32 {sr(ph)}, 0")       # the tail's own full group (bi, g) comes next
        a.i(f"s_branch {l_end}")
        a.label(l_full)
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
    if ATTACH:
        a.i(f"s_cbranch_scc0 {l_done}")
        emit_attach(a, c, l_end)
        a.i(f"s_branch {l_end}")
    else:
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


def emit_attach(a, c, l_end):
    """the cursor has just reached the full group (bi, g): when it is the one the sequence's tail group is attached to,
    the tail group runs first (ph 2).  Which one: g = (bi >> S_ATTSH) & S_ATTMASK -- with d = gcd(NGF, workgroups per XCD) > 1
    a workgroup sees the same g (mod d) in every round, and 'always group 0' would hand all tails to 1 / d of the workgroups
    ([256,600]: 4 items on the even workgroups, 2 on the odd ones); the attached group's residue mod d follows the rounds
    instead, so that every workgroup takes its turn.  d = 1 (T = 800: NGF = 3): group 0, a tail on every workgroup."""
    ph, bi, g = c, c + 1, c + 2
    a.i(f"s_cmp_eq_u32 {sr(S_TAILQ)}, 0")
    a.i(f"s_cbranch_scc1 {l_end}")
    a.i(f"s_lshr_b32 {sr(S_T0)}, {sr(bi)}, {sr(S_ATTSH)}")
    a.i(f"s_and_b32 {sr(S_T0)}, {sr(S_T0)}, {sr(S_ATTMASK)}")
    a.i(f"s_cmp_lg_u32 {sr(g)}, {sr(S_T0)}")
    a.i(f"s_cbranch_scc1 {l_end}")
    a.i(f"s_mov_b32 {sr(ph)}, 2")


def emit_cursor_init(a, c, tag):
    """first item of this workgroup; bi, g preloaded with divmod(J, NGF)"""
    ph, bi, g, valid = c, c + 1, c + 2, c + 3
    l_tail, l_done, l_end = (a.uniq(tag + x) for x in ("itail", "idone", "iend"))
    a.i(f"s_mov_b32 {sr(valid)}, 1")
    a.i(f"s_mov_b32 {sr(ph)}, 0")
    a.i(f"s_cmp_eq_u32 {sr(S_NGF)}, 0")
    a.i(f"s_cbranch_scc1 {l_tail}")
    a.i(f"s_cmp_lt_u32 {sr(bi)}, {sr(S_BX)}")
    if ATTACH:
        a.i(f"s_cbranch_scc0 {l_done}")     # no full group for this workgroup: no tail group either (they come attached)
        emit_attach(a, c, l_end)
        a.i(f"s_branch {l_end}")
    else:
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


# ------------------------------------------------------------------------------------------------ DMA stream
def dma_half_ops(half):
    """this wave's 4 KiB of one half stage (0: K, 1: V^T): one M0 write, four pieces told apart by the immediate offset
    (it moves the global source AND the LDS destination: scripts/ubench/dma_issue.hip) -> [m0 write, piece x 4]"""
    src = S_DKS if half == 0 else S_DVS
    m0 = f"s_mov_b32 m0, {sr(S_DLDS)}" if half == 0 else f"s_add_u32 m0, {sr(S_DLDS)}, 16384"
    return [m0, "s_nop 0"] + [f"global_load_lds_dwordx4 {vr(V_OFF[0])}, {sr(src, 2)} offset:{k * FRAG}" for k in range(4)]


def emit_dma_half(a, half):
    for op in dma_half_ops(half):
        a.i(op)


def emit_dma_item_setup(a):
    """K / V^T source of stage 0 of the DMA cursor's item (+ this wave's 4 KiB)"""
    emit_seq_block(a, S_DC, S_T0)
    emit_block_addr(a, S_DKS, S_KF, S_T0, S_T2, S_T1)
    emit_block_addr(a, S_DVS, S_VTF, S_T0, S_T2, S_T1)
    a.i(f"s_lshl_b32 {sr(S_T1)}, {sr(S_W)}, 12")
    a.i(f"s_add_u32 {sr(S_DKS)}, {sr(S_DKS)}, {sr(S_T1)}")
    a.i(f"s_addc_u32 {sr(S_DKS + 1)}, {sr(S_DKS + 1)}, 0")
    a.i(f"s_add_u32 {sr(S_DVS)}, {sr(S_DVS)}, {sr(S_T1)}")
    a.i(f"s_addc_u32 {sr(S_DVS + 1)}, {sr(S_DVS + 1)}, 0")
    a.i(f"s_mov_b32 {sr(S_DS)}, 0")


def dma_slot_ops():
    """the next stage of the stream goes to the next ring slot (S_DSTREAM = slot index 0 .. NRING - 1)"""
    return [f"s_add_u32 {sr(S_DSTREAM)}, {sr(S_DSTREAM)}, 1\n\ts_cmp_eq_u32 {sr(S_DSTREAM)}, {NRING}\n\ts_cselect_b32 {sr(S_DSTREAM)}, 0, {sr(S_DSTREAM)}",
            f"s_lshl_b32 {sr(S_TD)}, {sr(S_DSTREAM)}, 15", f"s_add_u32 {sr(S_DLDS)}, {sr(S_DBASE)}, {sr(S_TD)}"]


def dma_advance_ops(a):
    """after both halves of a stage have been issued.  A list of items that may be placed in MFMA gaps (each item
    keeps its SCC producer and consumer together).  The in-line part ALWAYS runs: sources += one stage, next ring slot;
    when the item's last stage has been issued, an out-of-line call first points the sources one stage in front of the
    next item's first stage -- or, once the stream is exhausted, keeps them where they are and moves the LDS base so
    that the ring arithmetic lands in the dump slot."""
    stub, back = a.uniq("dstub"), a.uniq("dback")
    a.tail += [stub + ":", f"\ts_call_b64 {sr(S_RET, 2)}, .Lpw_dma_next_item", f"\ts_branch {back}"]
    return [f"s_add_u32 {sr(S_DS)}, {sr(S_DS)}, 1\n\ts_cmp_ge_u32 {sr(S_DS)}, {sr(S_NST)}\n\ts_cbranch_scc1 {stub}\n{back}:",
            f"s_add_u32 {sr(S_DKS)}, {sr(S_DKS)}, {2 * BLK}\n\ts_addc_u32 {sr(S_DKS + 1)}, {sr(S_DKS + 1)}, 0",
            f"s_add_u32 {sr(S_DVS)}, {sr(S_DVS)}, {2 * BLK}\n\ts_addc_u32 {sr(S_DVS + 1)}, {sr(S_DVS + 1)}, 0"] + dma_slot_ops()


def emit_dma_advance(a):
    for op in dma_advance_ops(a):
        a.i(op)


def emit_back_one_stage(a):
    for r in (S_DKS, S_DVS):
        a.i(f"s_sub_u32 {sr(r)}, {sr(r)}, {2 * BLK}")
        a.i(f"s_subb_u32 {sr(r + 1)}, {sr(r + 1)}, 0")


def emit_dma_next_item_sub(a):
    a.label(".Lpw_dma_next_item")
    l_dump = a.uniq("dump")
    a.i(f"s_cmp_eq_u32 {sr(S_DC + 3)}, 0")
    a.i(f"s_cbranch_scc1 {l_dump}")
    emit_cursor_next(a, S_DC, "d")
    a.i(f"s_cmp_eq_u32 {sr(S_DC + 3)}, 0")
    a.i(f"s_cbranch_scc1 {l_dump}")
    emit_dma_item_setup(a)
    emit_back_one_stage(a)
    a.i(f"s_setpc_b64 {sr(S_RET, 2)}")
    a.label(l_dump)   # stream exhausted: same cadence, the last valid sources again, ring slots nobody reads any more
    emit_back_one_stage(a)
    a.i(f"s_setpc_b64 {sr(S_RET, 2)}")


# ------------------------------------------------------------------------------------------------ items
def emit_item_params(a, c, flags, qa, seqblk):
    """item of cursor c -> flags (bit 0: wave has a query block, bit 1: it has two, bit 2: feature-split tail item),
    qa = first query block of the wave, seqblk = first block of the sequence.
    A tail group of one or two query blocks is a FEATURE-SPLIT item: every wave takes the same block(s), computes the
    scores and the softmax redundantly, and owns one of the four 32-feature blocks of the context."""
    l_split, l_end = a.uniq("ipsplit"), a.uniq("ipend")
    emit_seq_block(a, c, seqblk)
    a.i(f"s_cmp_eq_u32 {sr(c)}, 2")                         # an attached tail item: group NGF, whatever g says
    a.i(f"s_cselect_b32 {sr(S_T0)}, {sr(S_NGF)}, {sr(c + 2)}")
    a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_T0)}, 3")            # first query block of the group
    a.i(f"s_sub_u32 {sr(S_T1)}, {sr(S_QB)}, {sr(S_T0)}")   # query blocks left in the sequence
    a.i(f"s_min_u32 {sr(S_T1)}, {sr(S_T1)}, 8")
    a.i(f"s_cmp_le_u32 {sr(S_T1)}, {SPLIT_MAX}")
    a.i(f"s_cbranch_scc1 {l_split}")
    a.i(f"s_lshl_b32 {sr(S_T2)}, {sr(S_W)}, 1")
    a.i(f"s_add_u32 {sr(qa)}, {sr(S_T0)}, {sr(S_T2)}")
    a.i(f"s_sub_i32 {sr(S_T1)}, {sr(S_T1)}, {sr(S_T2)}")   # live blocks of this wave (may be <= 0)
    a.i(f"s_max_i32 {sr(S_T1)}, {sr(S_T1)}, 0")
    a.i(f"s_min_i32 {sr(S_T1)}, {sr(S_T1)}, 2")
    a.i(f"s_cmp_ge_u32 {sr(S_T1)}, 1")
    a.i(f"s_cselect_b32 {sr(flags)}, 1, 0")
    a.i(f"s_cmp_ge_u32 {sr(S_T1)}, 2")
    a.i(f"s_cselect_b32 {sr(S_T2)}, 2, 0")
    a.i(f"s_or_b32 {sr(flags)}, {sr(flags)}, {sr(S_T2)}")
    a.i(f"s_branch {l_end}")
    a.label(l_split)
    a.i(f"s_mov_b32 {sr(qa)}, {sr(S_T0)}")
    a.i(f"s_cmp_ge_u32 {sr(S_T1)}, 2")
    a.i(f"s_cselect_b32 {sr(flags)}, 7, 5")
    a.label(l_end)


def emit_q_request(a):
    """request the NEXT item's Q fragments into the staging registers (S_NFLAGS / S_NQA / S_NSEQBLK describe it)"""
    l_skip = a.uniq("qskip")
    a.i(f"s_bitcmp1_b32 {sr(S_NFLAGS)}, 0")
    a.i(f"s_cbranch_scc0 {l_skip}")
    a.i(f"s_add_u32 {sr(S_T3)}, {sr(S_NSEQBLK)}, {sr(S_NQA)}")
    for blk in ("A", "B"):
        if blk == "B":
            a.i(f"s_bitcmp1_b32 {sr(S_NFLAGS)}, 1")
            a.i(f"s_cselect_b32 {sr(S_T0)}, 1, 0")             # block B = A + 1 when it exists, else A again
            a.i(f"s_add_u32 {sr(S_T3)}, {sr(S_T3)}, {sr(S_T0)}")
        emit_block_addr(a, S_T4, S_QF, S_T3, S_T2, S_T1)
        for f in range(8):
            dst = V_QS + (0 if blk == "A" else 32) + 4 * f
            a.i(f"global_load_dwordx4 {vr(dst, 4)}, {vr(V_OFF[f // 4])}, {sr(S_T4, 2)} offset:{(f % 4) * FRAG}" + q_policy())
    a.label(l_skip)


def emit_ctx_stores(a):
    """stores of the finished item's packed context (S_PEND: S_FLAGS of that item).  Ordinary item: 8 fragments per
    block; feature-split item: this wave's two fragments (feature block w) per block, packed at the front."""
    l_split, l_end = a.uniq("stsplit"), a.uniq("stend")
    a.i(f"s_bitcmp1_b32 {sr(S_PEND)}, 2")
    a.i(f"s_cbranch_scc1 {l_split}")
    for blk, bit, dst in (("A", 0, S_CDST), ("B", 1, S_CDSTB)):
        l_skip = a.uniq("stskip")
        a.i(f"s_bitcmp1_b32 {sr(S_PEND)}, {bit}")
        a.i(f"s_cbranch_scc0 {l_skip}")
        for f in range(8):
            a.i(f"global_store_dwordx4 {vr(V_OFF[f // 4])}, {vr(V_CTX[blk] + 4 * f, 4)}, {sr(dst, 2)} offset:{(f % 4) * FRAG}" + ctx_policy())
        a.label(l_skip)
    a.i(f"s_branch {l_end}")
    a.label(l_split)   # S_CDST / S_CDSTB already point at fragment 2w of the block
    for blk, bit, dst in (("A", 0, S_CDST), ("B", 1, S_CDSTB)):
        l_skip = a.uniq("stskip")
        a.i(f"s_bitcmp1_b32 {sr(S_PEND)}, {bit}")
        a.i(f"s_cbranch_scc0 {l_skip}")
        for f in range(2):
            a.i(f"global_store_dwordx4 {vr(V_OFF[0])}, {vr(V_CTX[blk] + 4 * f, 4)}, {sr(dst, 2)} offset:{f * FRAG}" + ctx_policy())
        a.label(l_skip)
    a.label(l_end)
    a.i(f"s_mov_b32 {sr(S_PEND)}, 0")


def emit_post_barrier_sub(a):
    """once per item, behind the barrier of its first stage: stores of the previous item, Q request for the next"""
    a.label(".Lpw_post_barrier")
    emit_ctx_stores(a)
    emit_q_request(a)
    a.i(f"s_mov_b32 {sr(S_QPEND)}, 0")
    a.i(f"s_setpc_b64 {sr(S_RET, 2)}")


def rotate_ops():
    """the stage behind the compute stage becomes the compute stage"""
    return [f"v_mov_b32 {vr(V_ADDR_V)}, {vr(V_ADDR_K)}",
            f"s_add_u32 {sr(S_KS)}, {sr(S_KS)}, {STAGE}\n\ts_cmp_eq_u32 {sr(S_KS)}, {NRING * STAGE}\n\ts_cselect_b32 {sr(S_KS)}, 0, {sr(S_KS)}",
            f"v_add_u32 {vr(V_ADDR_K)}, {sr(S_KS)}, {vr(V_LDSL)}"]


def emit_stage_top(a):
    """own share of the stage behind the compute stage landed (the newest 8 (AHEAD - 2) requests are the stages after it),
    then everybody's; afterwards the slot of the stage before the compute stage may be refilled"""
    a.i(f"s_waitcnt vmcnt({8 * (AHEAD - 2)})")
    stamp(a, 3)
    if not ABLATE & 2:
        a.i("s_barrier")
    a.i(f"s_cmp_lg_u32 {sr(S_QPEND)}, 0")
    a.ool_call(".Lpw_post_barrier")


# ------------------------------------------------------------------------------------------------ one step
def spread(gaps, ops, lo, hi):
    """ops (kept in order) evenly over gaps[lo:hi]"""
    n = hi - lo
    if n <= 0 or not ops:
        gaps[max(lo, 0) if lo < len(gaps) else len(gaps) - 1].extend(ops)
        return
    for k, op in enumerate(ops):
        gaps[lo + (k * n) // len(ops)].append(op)


ISSUE_COST = {"ds_read_b128": 15, "v_exp_f32": 8, "global_load_lds_dwordx4": 8}   # cycles a wave cannot issue its next MFMA (estimates; others 4)


def op_cost(op):
    return sum(ISSUE_COST.get(ln.split()[0], 4) for ln in op.split("\n\t"))


def spread_balanced(gaps, ops, lo, hi):
    """ops (kept in order) over gaps[lo:hi] so that every gap carries about the same estimated issue cost on top of what it holds"""
    if hi <= lo or not ops:
        if ops:
            gaps[max(0, min(lo, len(gaps) - 1))].extend(ops)
        return
    load = [sum(op_cost(o) for o in gaps[g]) for g in range(lo, hi)]
    total = sum(load) + sum(op_cost(o) for o in ops)
    k, g = 0, 0
    while k < len(ops):
        left = hi - lo - g
        target = (total - sum(load[:g])) / left if left > 0 else 1e9      # what the remaining gaps must carry on average
        c = op_cost(ops[k])
        if g < hi - lo - 1 and load[g] > 0 and load[g] + c > target + 2:
            g += 1
            continue
        gaps[lo + g].append(ops[k])
        load[g] += c
        k += 1


def softmax_split(buf, blk):
    """softmax_ops(buf, blk) in three groups: exponentials + row sum (what the reference check needs), bf16 packing,
    and the update of the running row sum (which must wait for the check)"""
    ops = softmax_ops(buf, blk)
    pack = [op for op in ops if op.startswith("v_cvt_pk")]
    lsum = [ops[-1]]
    head = [op for op in ops[:-1] if not op.startswith("v_cvt_pk")]
    return head, pack, lsum


def pv_mfmas(blocks, first=False):
    """first: the item's first tile -- O^T = V^T P^T starts from zero (nobody has to clear the accumulators)"""
    return [mfma_pv(blk, nbd, j, c0=(first and j == 0)) for blk in blocks for j in range(2) for nbd in range(4)]


def emit_step(a, i_par, has_prev, has_next, dma, book, first_pv=False):
    """Key block i (parity i_par: its scores sit in buffer i_par; the next tile's go to the other one).  Two-stage
    pipeline: the step issues  [O^T += V^T(i-1) P^T(i-1)]  (has_prev)  then  [S^T(i+1) = K(i+1) Q^T]  (has_next)  while the
    VALU works through tile i: exponentials, row sums, bf16 packing -- nothing the MFMAs of this step wait for.
      first half   beside the PV MFMAs : K(i+1) fragment reads, exponentials of both blocks, row sum of block A
      second half  beside the S^T MFMAs: V^T(i) fragment reads, row sum of block B, packing (P(i-1) has been consumed),
                                         DMA pieces, stream advance, address rotation, the reference check"""
    cur, nxt = i_par, 1 - i_par
    blocks = ("A", "B")
    mfP = pv_mfmas(blocks, first=first_pv) if has_prev else []
    mfS = []
    if has_next:
        for ks in range(8):
            for blk in blocks:
                mfS.append(mfma_s(nxt, blk, ks))
    mf = mfP + mfS
    nP, N = len(mfP), len(mfP) + len(mfS)
    gaps = [[] for _ in range(max(N, 1))]
    # LDS reads: K(i+1) = second block of the compute stage (i even) or first block of the stage behind it (i odd)
    kr = [kread(f, V_ADDR_V, BLK) if i_par == 0 else kread(f, V_ADDR_K, 0) for f in range(8)] if has_next else []
    vblk = BLK * i_par
    vr_ops = [vread(f, vblk) for f in (0, 2, 4, 6, 1, 3, 5, 7)]
    if ABLATE & 16:
        kr, vr_ops = [], []
    head, pack, lsum = {}, {}, {}
    for blk in blocks:
        head[blk], pack[blk], lsum[blk] = softmax_split(cur, blk)
        if ABLATE & 8:
            head[blk], pack[blk], lsum[blk] = [], [], []
    exps = {blk: head[blk][:16] for blk in blocks}
    adds = {blk: head[blk][16:] for blk in blocks}
    one, two = [], []       # VALU work of the first / second half, in order
    if len(blocks) == 2:
        for k in range(len(exps["A"])):
            one += [exps["A"][k], exps["B"][k]]
        if BALANCE >= 3:     # both row sums beside the S^T MFMAs: the two halves then carry about the same issue cost
            two += [x for pair in zip(adds["A"], adds["B"]) for x in pair]
        else:
            one += adds["A"]
            two += adds["B"]
    else:
        one += exps["A"]
        two += adds["A"]
    for blk in blocks:
        two += pack[blk]
    check = []
    if not ABLATE & 4 and head["A"]:
        if len(blocks) == 2:
            check.append(f"v_max_f32 {vr(V_T1)}, {vr(V_RS['A'])}, {vr(V_RS['B'])}")
            check.append(f"v_cmp_nge_f32 vcc, {CHECK_BITS}, {vr(V_T1)}")      # !(threshold >= largest row sum)
        else:
            check.append(f"v_cmp_nge_f32 vcc, {CHECK_BITS}, {vr(V_RS['A'])}")
    two += check
    if ABLATE & 1:
        dma = [op for op in dma if not (op.startswith("global_load_lds") or "m0" in op)]
    a.i("s_waitcnt lgkmcnt(0)")       # V^T(i-1) fragments (requested in the previous step)
    pre = []
    if nP >= 4 and BALANCE:
        # K reads two per gap while the exponentials may not start yet (the last MFMA of S^T(i) was issued at the very end of the
        # previous step: its readers start 2 MFMAs in), the others between the exponentials; every gap the same issue cost
        gaps[0].extend(kr[0:2])
        gaps[1].extend(kr[2:4])
        rest, krl, n_e = [], kr[4:], 0
        for op in one:
            rest.append(op)
            if op.startswith("v_exp"):
                n_e += 1
                if n_e % 6 == 0 and krl:
                    rest.append(krl.pop(0))
        rest += krl
        spread_balanced(gaps, rest, 2, nP)
    elif nP >= 4:
        # the last MFMA of S^T(i) was issued at the very end of the previous step: its readers start 2 MFMAs in
        spread(gaps, kr, 0, max(1, nP // 2))
        spread(gaps, one, 2, nP)
    else:
        pre = kr + one
    if N - nP > 0 and BALANCE >= 2:
        spread(gaps, vr_ops, nP, nP + max(1, (N - nP) // 2))
        spread(gaps, dma, nP, N)
        spread(gaps, book, nP + (N - nP) // 2, N)
        spread_balanced(gaps, two, nP, N)
        post = []
    elif N - nP > 0:
        spread(gaps, vr_ops, nP, nP + max(1, (N - nP) // 2))
        spread(gaps, two, nP, N)
        spread(gaps, dma, nP, N)
        spread(gaps, book, nP + (N - nP) // 2, N)
        post = []
    else:
        post = vr_ops + two + dma + book
    for n in range(N):
        if n == 0:
            for op in pre:
                a.i(op)
        if n == nP and has_next:
            a.i("s_waitcnt lgkmcnt(0)")  # K(i+1) fragments
        a.i(mf[n])
        for op in gaps[n]:
            a.i(op)
    if N == 0:
        for op in pre:
            a.i(op)
    for op in post:
        a.i(op)
    if check:
        a.i("s_nop 4")   # (VALU wrote VCC just before: nothing documents that the SALU read waits for it)
        a.i("s_cmp_lg_u64 vcc, 0" if not ABLATE & 256 else "s_cmp_eq_u32 0, 0")   # (256: always take the cold path)
        a.ool_call(f".Lpw_coldmid_{cur}_{int(has_next)}_{len(blocks)}")
    for blk in blocks:
        for op in lsum[blk]:
            a.i(op)


def emit_mask(a, buf, blocks):
    """ragged last key block (T % 32 != 0): scores of keys that do not exist -> -1e30 (lane (m, h), register r <-> key
    8 (r >> 2) + 4 h + (r & 3)); the MFMAs that wrote the tile are at least 16 MFMAs behind"""
    l_skip = a.uniq("nomask")
    a.i(f"s_cmp_eq_u32 {sr(S_RAG)}, 0")
    a.i(f"s_cbranch_scc1 {l_skip}")
    a.i("s_nop 7")   # the MFMAs that wrote the tile were the last instructions of the previous step
    a.i("s_nop 7")
    a.i(f"v_sub_u32 {vr(V_LIM)}, {sr(S_RAG)}, {vr(V_H4)}")
    for blk in blocks:
        S = V_S[(buf, blk)]
        for r in range(16):
            a.i(f"v_cmp_lt_i32 vcc, {8 * (r >> 2) + (r & 3)}, {vr(V_LIM)}")
            a.i(f"v_cndmask_b32 {vr(S + r)}, {vr(V_NEGB)}, {vr(S + r)}, vcc")
    a.label(l_skip)


def emit_stage(a, kind):
    """kind: 'fmid' (the item's first stage, steps 0 and 1, tile 2 exists), 'mid' (two steps in the middle), 'last2' (the
    item's last two tiles), 'last1' (its last tile alone), 'flast2' (an item of two tiles).  Every stage: one barrier,
    8 DMA pieces, one stream advance, one rotation of the ring addresses."""
    stamp(a, 11)
    emit_stage_top(a)
    stamp(a, 2)
    kh, vh = dma_half_ops(0), dma_half_ops(1)
    c_even, c_odd, c_last = 4, 5, 6
    blocks = ("A", "B")
    odd_book = rotate_ops()
    if kind in ("mid", "fmid"):
        emit_step(a, 0, kind == "mid", True, kh, [])
        stamp(a, c_even)
        emit_step(a, 1, True, True, vh + dma_advance_ops(a),
                  odd_book + ([f"s_sub_u32 {sr(S_CNT)}, {sr(S_CNT)}, 1"] if kind == "mid" else []), first_pv=(kind == "fmid"))
        stamp(a, c_odd)
    elif kind in ("last2", "flast2"):
        emit_step(a, 0, kind == "last2", True, kh, [])
        stamp(a, c_even)
        emit_mask(a, 1, blocks)
        emit_step(a, 1, True, False, vh + dma_advance_ops(a), odd_book, first_pv=(kind == "flast2"))
        stamp(a, c_last)
    else:
        emit_mask(a, 0, blocks)
        emit_step(a, 0, True, False, kh + vh + dma_advance_ops(a), odd_book)
        stamp(a, c_last)


# ------------------------------------------------------------------------------------------------ cold paths
def emit_move(a, buf, blk, first, also_next, l_skip):
    """online_softmax_shifted() of savad_kernels_bf16.h for one query block's tile in buffer `buf`: row maximum (both
    halves); rows whose maximum exceeds the reference by 2^16 (first tile: is more than 2^16 away from 0) move their
    reference there: d = move ? mx : 0; l and O scaled by 2^-d (not on the first tile: both are still zero), scores and
    -reference shifted by d -- and the scores of the NEXT tile too when its MFMAs already ran against the old reference"""
    S = V_S[(buf, blk)]
    for op in max_ops(buf, blk, V_MX[blk]):
        a.i(op)
    for op in half_exchange(V_MX[blk], V_T0, "v_max_f32"):
        a.i(op)
    if first:
        a.i(f"v_cmp_gt_f32 vcc, 0xc1800000, {vr(V_MX[blk])}")       # (first && mx < -16) ...
        a.i(f"s_mov_b64 {sr(S_T4, 2)}, vcc")
    a.i(f"v_cmp_lt_f32 vcc, 0x41800000, {vr(V_MX[blk])}")           # move = mx > 16 ...
    if first:
        a.i(f"s_or_b64 vcc, vcc, {sr(S_T4, 2)}")
    a.i("s_cmp_eq_u64 vcc, 0")
    a.i(f"s_cbranch_scc1 {l_skip}")
    if TIMING:   # lane 15: blocks that really moved
        a.i(f"v_readlane_b32 {sr(S_TM)}, {vr(V_ACC)}, 15")
        a.i(f"s_add_u32 {sr(S_TM)}, {sr(S_TM)}, 1")
        a.i(f"v_writelane_b32 {vr(V_ACC)}, {sr(S_TM)}, 15")
    a.i(f"v_cndmask_b32 {vr(V_D)}, 0, {vr(V_MX[blk])}, vcc")       # d = move ? mx : 0
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
    if also_next:   # the next tile's scores ran against the old reference: again, as the reference kernel would have
        a.i("s_nop 1")  # computed them (K(i+1) is still in a[192:223]); C = the new -reference
        for ks in range(8):
            a.i(mfma_s(1 - buf, blk, ks))


def emit_count(a, lane):
    if TIMING:
        a.i(f"v_readlane_b32 {sr(S_TM)}, {vr(V_ACC)}, {lane}")
        a.i(f"s_add_u32 {sr(S_TM)}, {sr(S_TM)}, 1")
        a.i(f"v_writelane_b32 {vr(V_ACC)}, {sr(S_TM)}, {lane}")


def emit_cold_first(a):
    """the item's first tile (buffer 0, both blocks): the reference is SET (nothing to rescale)"""
    a.label(".Lpw_coldfirst_0")
    emit_count(a, 14)
    a.i("s_nop 7")
    a.i("s_nop 7")
    a.i("s_nop 7")
    for blk in ("A", "B"):
        l_skip = a.uniq("cfskip")
        emit_move(a, 0, blk, True, False, l_skip)
        a.label(l_skip)
    a.i("s_nop 1")
    a.i(f"s_setpc_b64 {sr(S_RET, 2)}")


def emit_cold_mid(a, cur, has_next, nblocks):
    """A row sum of tile i (buffer cur) left 2^16 or is not a number: entered between the score MFMAs of tile i+1 and the
    PV MFMAs of tile i, with the exponentials, row sums and block A's probabilities of tile i computed against the
    standing reference and the running row sums not yet updated.  Per block: the exact test of online_softmax_shifted();
    when a reference really moves, O / l / scores / -reference are brought to it (the next tile's scores too: they were
    computed against the old one) and the block's exponentials, row sum and probabilities are redone."""
    a.label(f".Lpw_coldmid_{cur}_{int(has_next)}_{nblocks}")
    emit_count(a, 12 + cur)
    a.i("s_nop 7")   # every MFMA of the next tile's scores (and every PV MFMA of the previous tile) has retired
    a.i("s_nop 7")
    a.i("s_nop 7")
    for blk in ("A", "B")[:nblocks]:
        l_skip = a.uniq("cmskip")
        emit_move(a, cur, blk, False, has_next, l_skip)
        head, pack, _ = softmax_split(cur, blk)
        for op in head + pack:   # (block B's packing runs again behind the check: same values)
            a.i(op)
        a.label(l_skip)
    a.i("s_nop 1")
    a.i(f"s_setpc_b64 {sr(S_RET, 2)}")


# ------------------------------------------------------------------------------------------------ item prologue / epilogue
def q_move_ops():
    return [f"v_accvgpr_write_b32 {ar(A_Q['A'] + r)}, {vr(V_QS + r)}" for r in range(64)]


def s0_mfmas():
    """S^T(0) = K(0) Q^T of both query blocks against the reference 0 (C = 0)"""
    return [mfma_s(0, blk, ks, c0=(ks == 0)) for ks in range(8) for blk in ("A", "B")]


def emit_item_prologue(a):
    """Q of this item from staging into a[128:191] and S^T(0) -- unless the previous item's seam has done both (S_PRE);
    reference of tile 0.  (O needs no clearing: the first tile's PV MFMAs start from C = 0.)"""
    l_pre = a.uniq("predone")
    if SEAM:
        a.i(f"s_cmp_eq_u32 {sr(S_PRE)}, 1")
        a.i(f"s_cbranch_scc1 {l_pre}")
    a.i("s_waitcnt vmcnt(8)")   # the staged Q is older than the newest stage of DMA pieces
    stamp(a, 26)
    for op in q_move_ops():
        a.i(op)
    stamp(a, 27)
    for f in range(8):
        a.i(kread(f, V_ADDR_V, 0))
    a.i("s_waitcnt lgkmcnt(0)")
    stamp(a, 28)
    for op in s0_mfmas():
        a.i(op)
    a.label(l_pre)
    a.i(f"s_mov_b32 {sr(S_PRE)}, 0")
    for blk in ("A", "B"):
        a.i(f"v_mov_b32 {vr(V_L[blk])}, 0")
        for r in range(16):
            a.i(f"v_mov_b32 {vr(V_NEGM[blk] + r)}, 0")
    stamp(a, 29)
    # tile 0 is the last tile only when QB == 1, which this kernel never sees (T > 32)
    a.i(f"s_call_b64 {sr(S_RET, 2)}, .Lpw_coldfirst_0")
    a.i(f"s_sub_u32 {sr(S_CNT)}, {sr(S_QB)}, 3")          # 'mid' stages between the first and the last: (QB - 3) / 2
    a.i(f"s_max_i32 {sr(S_CNT)}, {sr(S_CNT)}, 0")
    a.i(f"s_lshr_b32 {sr(S_CNT)}, {sr(S_CNT)}, 1")


def emit_recip(a, x, out, t0, t1, t2, t3):
    """out = 1.0f / x as hipcc emits it (IEEE): v_div_scale / v_rcp / Newton / v_div_fmas / v_div_fixup (clobbers vcc, s[S_T4:S_T5])"""
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


def emit_normalise(a, blk, nregs, dst):
    """context block(s) of query block blk: a[A_O[blk] .. + nregs) / row sum -> bf16 pairs in v[dst ..]; rows of a ragged
    last query block that do not exist store exact zeros (as store_ctx of savad_kernels_bf16.h)"""
    a.i(f"v_mov_b32 {vr(V_T0)}, {vr(V_L[blk])}")
    a.i("s_nop 1")
    a.i(f"v_permlane32_swap_b32 {vr(V_L[blk])}, {vr(V_T0)}")
    a.i(f"v_add_f32 {vr(V_T0)}, {vr(V_L[blk])}, {vr(V_T0)}")
    out = V_INV[blk]
    emit_recip(a, V_T0, out, V_T1, V_T2, V_T3, V_T4)
    a.i(f"v_cmp_gt_i32 vcc, {sr(S_VALID[blk])}, {vr(V_M)}")
    a.i("s_nop 1")
    a.i(f"v_cndmask_b32 {vr(out)}, 0, {vr(out)}, vcc")   # (rows that do not exist: factor 0, cleaned up below)
    l_rag, l_end = a.uniq("nrag"), a.uniq("nend")
    a.i(f"s_cmp_lt_i32 {sr(S_VALID[blk])}, 32")
    a.i(f"s_cbranch_scc1 {l_rag}")
    for ragged in (False, True):
        if ragged:
            a.label(l_rag)
        for e in range(nregs // 2):
            a.i(f"v_accvgpr_read_b32 {vr(V_T1)}, {ar(A_O[blk] + 2 * e)}")
            a.i(f"v_accvgpr_read_b32 {vr(V_T2)}, {ar(A_O[blk] + 2 * e + 1)}")
            a.i(f"v_mul_f32 {vr(V_T1)}, {vr(V_T1)}, {vr(out)}")
            a.i(f"v_mul_f32 {vr(V_T2)}, {vr(V_T2)}, {vr(out)}")
            if ragged:   # 0 * inf / NaN of a row that does not exist must still store 0
                a.i(f"v_cvt_pk_bf16_f32 {vr(V_T1)}, {vr(V_T1)}, {vr(V_T2)}")
                a.i(f"v_cndmask_b32 {vr(dst + e)}, 0, {vr(V_T1)}, vcc")
            else:
                a.i(f"v_cvt_pk_bf16_f32 {vr(dst + e)}, {vr(V_T1)}, {vr(V_T2)}")
        if not ragged:
            a.i(f"s_branch {l_end}")
    a.label(l_end)


def normalise_ops(blk, dst):
    """emit_normalise for a query block whose 32 rows all exist, as a straight list (the fast seam spreads it over MFMA gaps)"""
    rec = Asm()
    rec.i(f"v_mov_b32 {vr(V_T0)}, {vr(V_L[blk])}")
    rec.i("s_nop 1")
    rec.i(f"v_permlane32_swap_b32 {vr(V_L[blk])}, {vr(V_T0)}")
    rec.i(f"v_add_f32 {vr(V_T0)}, {vr(V_L[blk])}, {vr(V_T0)}")
    emit_recip(rec, V_T0, V_INV[blk], V_T1, V_T2, V_T3, V_T4)
    head = [ln.strip() for ln in rec.lines]
    body = []
    for e in range(32):
        body += [f"v_accvgpr_read_b32 {vr(V_T1)}, {ar(A_O[blk] + 2 * e)}", f"v_accvgpr_read_b32 {vr(V_T2)}, {ar(A_O[blk] + 2 * e + 1)}",
                 f"v_mul_f32 {vr(V_T1)}, {vr(V_T1)}, {vr(V_INV[blk])}", f"v_mul_f32 {vr(V_T2)}, {vr(V_T2)}, {vr(V_INV[blk])}",
                 f"v_cvt_pk_bf16_f32 {vr(dst + e)}, {vr(V_T1)}, {vr(V_T2)}"]
    return head, body


def emit_seam(a):
    """The end of an ordinary item.  When this wave's NEXT item is an ordinary one too and both of its query blocks here are
    whole, the last tile's PV MFMAs carry the next item's Q move in their gaps, and that item's S^T(0) MFMAs carry the first
    third of this item's epilogue -- the matrix pipe never waits for the seam's VALU work, and the next prologue finds its
    operands in place (S_PRE).  Otherwise: the plain PV MFMAs and the epilogue."""
    l_plain, l_done = a.uniq("seamplain"), a.uniq("seamdone")
    a.i("s_waitcnt lgkmcnt(0)")      # V^T fragments of the last tile
    if SEAM:
        a.i(f"s_and_b32 {sr(S_T0)}, {sr(S_NFLAGS)}, 5")
        a.i(f"s_cmp_eq_u32 {sr(S_T0)}, 1")            # the next item: this wave has a block, and it is not a key-split item
        a.i(f"s_cbranch_scc0 {l_plain}")
        a.i(f"s_cmp_ge_i32 {sr(S_VALID['B'])}, 32")   # both query blocks of THIS item are whole (implies block A's 32 rows)
        a.i(f"s_cbranch_scc0 {l_plain}")
        a.i("s_waitcnt vmcnt(8)")    # the staged Q of the next item is older than the newest stage of DMA pieces
        for f in range(8):           # K(0) of the next item: its stage 0 sits behind the barrier of this item's last stage
            a.i(kread(f, V_ADDR_V, 0))
        emit_with_gaps(a, pv_mfmas(("A", "B")), q_move_ops())
        stamp(a, 11)
        a.i("s_waitcnt lgkmcnt(0)")
        a.i("s_nop 1")
        (ha, ba), (hb, bb) = normalise_ops("A", V_CTX["A"]), normalise_ops("B", V_CTX["B"])
        # the row sums and reciprocals first (no O access: the PV MFMAs are still in flight), then block A's context from gap 5 on
        fill = ha + hb + ba + bb
        k = 7 * 16                   # ~7 instructions per gap hide; the rest follows the last MFMA
        emit_with_gaps(a, s0_mfmas(), fill[:k])
        for op in fill[k:]:
            a.i(op)
        emit_ctx_dest(a, False)
        a.i(f"s_mov_b32 {sr(S_PRE)}, 1")
        stamp(a, 8)
        a.i(f"s_branch {l_done}")
        a.label(l_plain)
    for op in pv_mfmas(("A", "B")):
        a.i(op)
    stamp(a, 11)
    emit_item_epilogue(a)
    stamp(a, 8)
    a.label(l_done)


def emit_ctx_dest(a, split):
    """where the finished item's context goes: blocks qa, qa + 1 of the sequence (split: this wave's fragments 2w, 2w + 1)"""
    a.i(f"s_add_u32 {sr(S_T3)}, {sr(S_SEQBLK)}, {sr(S_QA)}")
    emit_block_addr(a, S_CDST, S_CTXF, S_T3, S_T2, S_T1)
    a.i(f"s_add_u32 {sr(S_T3)}, {sr(S_T3)}, 1")
    emit_block_addr(a, S_CDSTB, S_CTXF, S_T3, S_T2, S_T1)
    if split:
        a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_W)}, 11")
        for d in (S_CDST, S_CDSTB):
            a.i(f"s_add_u32 {sr(d)}, {sr(d)}, {sr(S_T0)}")
            a.i(f"s_addc_u32 {sr(d + 1)}, {sr(d + 1)}, 0")
    a.i(f"s_mov_b32 {sr(S_PEND)}, {sr(S_FLAGS)}")


def emit_item_epilogue(a):
    """normalise O by the row sums, pack to bf16 fragments into the staging registers of the stores, describe them"""
    a.i("s_nop 7")
    a.i("s_nop 7")
    a.i("s_nop 7")
    for blk in ("A", "B"):
        emit_normalise(a, blk, 64, V_CTX[blk])
    emit_ctx_dest(a, False)


def emit_item_body(a, tag):
    """the stages of an item whose prologue has run: QB == 2: 'flast2'; else 'fmid', (QB - 3) / 2 x 'mid', then 'last2' (QB
    even) or 'last1' (QB odd); the PV MFMAs of the last tile; the epilogue"""
    l_mid, l_last, l_l1, l_done, l_two = (f".Lpw_{tag}_{x}" for x in ("mid", "last", "last1", "done", "two"))
    a.i(f"s_cmp_eq_u32 {sr(S_QB)}, 2")
    a.i(f"s_cbranch_scc1 {l_two}")
    emit_stage(a, "fmid")
    a.i(f"s_cmp_eq_u32 {sr(S_CNT)}, 0")
    a.i(f"s_cbranch_scc1 {l_last}")
    a.label(l_mid)
    emit_stage(a, "mid")
    a.i(f"s_cmp_lg_u32 {sr(S_CNT)}, 0")
    a.i(f"s_cbranch_scc1 {l_mid}")
    a.label(l_last)
    a.i(f"s_bitcmp1_b32 {sr(S_QB)}, 0")
    a.i(f"s_cbranch_scc1 {l_l1}")
    emit_stage(a, "last2")
    a.i(f"s_branch {l_done}")
    a.label(l_l1)
    emit_stage(a, "last1")
    a.i(f"s_branch {l_done}")
    a.label(l_two)
    emit_stage(a, "flast2")
    a.label(l_done)
    emit_seam(a)
    a.i("s_branch .Lpw_next_item")


# ------------------------------------------------------------------------------------------------ key-split tail items
def emit_with_gaps(a, mfmas, fillers):
    """the MFMAs with the filler instructions (kept in order) spread evenly over the gaps behind them"""
    gaps = [[] for _ in mfmas]
    spread(gaps, fillers, 0, len(mfmas))
    for mf, g in zip(mfmas, gaps):
        a.i(mf)
        for op in g:
            a.i(op)


def emit_ks_scores(a, nb, dma):
    """Key-split item, first stage of a pair: this wave's key block S_KB = 4 p + w of the pair of stages (c, c + 1) that
    is resident behind this stage's barrier (waves 0 / 1: blocks 0 / 1 of the compute stage, waves 2 / 3: of the stage
    behind it) against the item's nb query blocks with all 128 features: K and V^T fragments into registers (the V^T
    image's slot may be refilled from the next barrier on), scores, softmax against this wave's OWN reference.  The
    stage's DMA pieces ride in the gaps of the score MFMAs (beside VALU work each would stall the wave 50 - 110 cycles)."""
    blocks = ("A", "B")[:nb]
    hi, join, nomask, notfirst = (a.uniq("ks" + x) for x in ("hi", "join", "nomask", "nf"))
    a.i(f"s_cmp_lt_u32 {sr(S_W)}, 2")
    a.i(f"s_cbranch_scc0 {hi}")
    a.i(f"v_mov_b32 {vr(V_KSADDR)}, {vr(V_ADDR_V)}")
    a.i(f"s_branch {join}")
    a.label(hi)
    a.i(f"v_mov_b32 {vr(V_KSADDR)}, {vr(V_ADDR_K)}")
    a.label(join)
    a.i(f"s_and_b32 {sr(S_T0)}, {sr(S_W)}, 1")
    a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_T0)}, 13")
    a.i(f"v_add_u32 {vr(V_KSADDR)}, {sr(S_T0)}, {vr(V_KSADDR)}")
    for f in range(8):
        a.i(f"ds_read_b128 {ar(A_K + 4 * f, 4)}, {vr(V_KSADDR)} offset:{f * FRAG}")
    for f in range(8):
        a.i(f"ds_read_b128 {ar(A_V + 4 * f, 4)}, {vr(V_KSADDR)} offset:{16384 + f * FRAG}")
    a.i("s_waitcnt lgkmcnt(8)")      # the K fragments (LDS returns in order)
    emit_with_gaps(a, [mfma_s(0, blk, ks) for ks in range(8) for blk in blocks], dma)
    a.i("s_nop 7")                   # MFMA results -> VALU
    a.i("s_nop 7")
    # the sequence's last key block may be ragged
    a.i(f"s_sub_u32 {sr(S_T0)}, {sr(S_QB)}, 1")
    a.i(f"s_cmp_eq_u32 {sr(S_KB)}, {sr(S_T0)}")
    a.i(f"s_cbranch_scc0 {nomask}")
    emit_mask(a, 0, blocks)
    a.label(nomask)
    # this wave's first tile sets its reference
    a.i(f"s_cmp_eq_u32 {sr(S_KFIRST)}, 0")
    a.i(f"s_cbranch_scc1 {notfirst}")
    a.i(f"s_call_b64 {sr(S_RET, 2)}, .Lpw_ksfirst_{nb}")
    a.i(f"s_mov_b32 {sr(S_KFIRST)}, 0")
    a.label(notfirst)
    for blk in blocks:               # the exponentials here, the rest of the softmax behind the next barrier: the two waves that
        head, _, _ = softmax_split(0, blk)   # work on this stage and the two that finish the previous pair's tile take about as long
        for op in head[:16]:
            a.i(op)
    a.i("s_waitcnt lgkmcnt(0)")      # the V^T fragments are in registers before the barrier that frees their slot
    a.i(f"s_mov_b32 {sr(S_KPEND)}, 1")


def emit_ks_pv(a, nb, dma):
    """Key-split item, second stage of a pair (or behind the item's last stage): row sums, reference check and bf16 packing of
    the pending tile, then O^T += V^T P^T, operands in registers since the first stage; the stage's DMA pieces in the gaps"""
    blocks = ("A", "B")[:nb]
    lsum = []
    for blk in blocks:
        head, pack, ls = softmax_split(0, blk)
        for op in head[16:] + pack:
            a.i(op)
        lsum += ls
    if nb == 2:
        a.i(f"v_max_f32 {vr(V_T1)}, {vr(V_RS['A'])}, {vr(V_RS['B'])}")
        a.i(f"v_cmp_nge_f32 vcc, {CHECK_BITS}, {vr(V_T1)}")
    else:
        a.i(f"v_cmp_nge_f32 vcc, {CHECK_BITS}, {vr(V_RS['A'])}")
    a.i("s_nop 4")
    a.i("s_cmp_lg_u64 vcc, 0")
    a.ool_call(f".Lpw_coldmid_0_0_{nb}")
    for op in lsum:
        a.i(op)
    a.i("s_nop 1")
    emit_with_gaps(a, pv_mfmas(blocks), dma)
    a.i(f"s_mov_b32 {sr(S_KPEND)}, 0")


def emit_ks_first(a, nb):
    """first tile of a key-split wave (buffer 0): the reference is SET (nothing to rescale)"""
    a.label(f".Lpw_ksfirst_{nb}")
    for blk in ("A", "B")[:nb]:
        l_skip = a.uniq("kfskip")
        emit_move(a, 0, blk, True, False, l_skip)
        a.label(l_skip)
    a.i("s_nop 1")
    a.i(f"s_setpc_b64 {sr(S_RET, 2)}")


def emit_ks_combine(a, nb):
    """The four waves hold partial (O, l) of the same query block(s) against their own references r_w.  Through the LDS
    slot of the item's last stage (free until the next stage's barrier): everybody publishes (-r_w, l_w) per row; the
    common reference is the largest; wave w scales its O by f_w = 2^(r_w - r) / sum_u l_u 2^(r_u - r); then, feature block
    d = 0 .. 3 in turn, the other three waves hand their scaled block d to wave d (two alternating 12 KiB buffers, one
    barrier per round), which adds them to its own in a fixed order, packs to bf16 and keeps the two fragments for the
    item's stores (as the stores of a split item expect them: V_CTX[blk] + 0 .. 7)."""
    blocks = ("A", "B")[:nb]
    TS, TR, TO = 0, (16, 48, 64), 80     # temporaries: source block, the three received blocks, own block (all dead by now)
    ML = {"A": 0, "B": 16}               # (-reference, row sum) of the four waves: 8 registers per query block
    a.i("s_nop 7")                       # the last PV MFMAs have retired
    a.i("s_nop 7")
    a.i("s_nop 7")
    a.i(f"s_sub_i32 {sr(S_SCRB)}, {sr(S_KS)}, {2 * STAGE}")     # the slot two behind the stage S_KS names = the item's last stage
    a.i(f"s_cmp_lt_i32 {sr(S_SCRB)}, 0")
    a.i(f"s_cselect_b32 {sr(S_T0)}, {NRING * STAGE}, 0")
    a.i(f"s_add_u32 {sr(S_SCRB)}, {sr(S_SCRB)}, {sr(S_T0)}")
    a.i(f"s_add_u32 {sr(S_SCRB)}, {sr(S_SCRB)}, {sr(S_LDS)}")
    a.i(f"v_add_u32 {vr(V_SCR16)}, {sr(S_SCRB)}, {vr(V_LANE16)}")
    a.i(f"v_lshrrev_b32 {vr(V_SCR8)}, 1, {vr(V_LANE16)}")
    a.i(f"v_add_u32 {vr(V_SCR8)}, {sr(S_SCRB)}, {vr(V_SCR8)}")
    a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_W)}, 9")
    a.i(f"v_add_u32 {vr(V_T5)}, {sr(S_T0)}, {vr(V_SCR8)}")      # this wave's (-reference, row sum) slot
    l_has = a.uniq("kshas")
    a.i(f"s_cmp_eq_u32 {sr(S_KFIRST)}, 0")
    a.i(f"s_cbranch_scc1 {l_has}")
    for blk in blocks:   # a wave that saw no key block: reference -FLT_MAX, so that its zero sums do not set the common one
        a.i(f"v_mov_b32 {vr(V_NEGM[blk])}, {FLT_MAX_BITS}")
    a.label(l_has)
    a.i("s_barrier")                     # everybody has finished reading the ring slot that becomes the scratch
    for bi, blk in enumerate(blocks):
        t = V_T0 if bi == 0 else V_T2    # (V_T0, V_T1) / (V_T2, V_T3): consecutive registers for the 8-byte write
        a.i(f"v_mov_b32 {vr(t + 1)}, {vr(V_L[blk])}")
        a.i("s_nop 1")
        a.i(f"v_permlane32_swap_b32 {vr(V_L[blk])}, {vr(t + 1)}")
        a.i(f"v_add_f32 {vr(t + 1)}, {vr(V_L[blk])}, {vr(t + 1)}")        # the whole row sum
        a.i(f"v_mov_b32 {vr(t)}, {vr(V_NEGM[blk])}")
        a.i(f"ds_write_b64 {vr(V_T5)}, {vr(t, 2)} offset:{bi * 2048}")
    a.i("s_waitcnt lgkmcnt(0)")
    a.i("s_barrier")
    for bi, blk in enumerate(blocks):
        for u in range(4):
            a.i(f"ds_read_b64 {vr(ML[blk] + 2 * u, 2)}, {vr(V_SCR8)} offset:{bi * 2048 + u * 512}")
    a.i("s_waitcnt lgkmcnt(0)")
    for blk in blocks:
        n = [ML[blk] + 2 * u for u in range(4)]
        l = [ML[blk] + 2 * u + 1 for u in range(4)]
        a.i(f"v_min_f32 {vr(V_T0)}, {vr(n[0])}, {vr(n[1])}")
        a.i(f"v_min_f32 {vr(V_T1)}, {vr(n[2])}, {vr(n[3])}")
        a.i(f"v_min_f32 {vr(V_T0)}, {vr(V_T0)}, {vr(V_T1)}")              # -(common reference)
        for u in range(4):
            a.i(f"v_sub_f32 {vr(n[u])}, {vr(V_T0)}, {vr(n[u])}")          # r_u - r
        a.i(f"v_sub_f32 {vr(V_T2)}, {vr(V_T0)}, {vr(V_NEGM[blk])}")       # r_w - r
        for u in range(4):
            a.i(f"v_exp_f32 {vr(n[u])}, {vr(n[u])}")
        a.i(f"v_exp_f32 {vr(V_T2)}, {vr(V_T2)}")
        a.i("s_nop 0")
        a.i(f"v_mul_f32 {vr(V_T1)}, {vr(l[0])}, {vr(n[0])}")
        for u in range(1, 4):
            a.i(f"v_fmac_f32 {vr(V_T1)}, {vr(l[u])}, {vr(n[u])}")         # the row sum against the common reference
        emit_recip(a, V_T1, V_INV[blk], V_T3, V_T4, V_T5, V_T0)
        a.i(f"v_mul_f32 {vr(V_INV[blk])}, {vr(V_INV[blk])}, {vr(V_T2)}")
        a.i(f"v_cmp_gt_i32 vcc, {sr(S_VALID[blk])}, {vr(V_M)}")
        a.i(f"v_cndmask_b32 {vr(V_INV[blk])}, 0, {vr(V_INV[blk])}, vcc")  # (rows that do not exist: cleaned up at the packing)
    rnd = 0
    for blk in blocks:
        for d in range(4):
            buf = 4096 + (rnd & 1) * 12288
            l_meet, l_skip = a.uniq("ksmeet"), a.uniq("ksskip")
            a.i(f"s_cmp_eq_u32 {sr(S_W)}, {d}")
            a.i(f"s_cbranch_scc1 {l_meet}")
            a.i(f"s_cmp_gt_u32 {sr(S_W)}, {d}")          # rank of this wave among the three sources
            a.i(f"s_cselect_b32 {sr(S_T0)}, 1, 0")
            a.i(f"s_sub_u32 {sr(S_T0)}, {sr(S_W)}, {sr(S_T0)}")
            a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_T0)}, 12")
            a.i(f"v_add_u32 {vr(V_T5)}, {sr(S_T0)}, {vr(V_SCR16)}")
            for r in range(16):
                a.i(f"v_accvgpr_read_b32 {vr(TS + r)}, {ar(A_O[blk] + 16 * d + r)}")
            for r in range(16):
                a.i(f"v_mul_f32 {vr(TS + r)}, {vr(TS + r)}, {vr(V_INV[blk])}")
            for q in range(4):
                a.i(f"ds_write_b128 {vr(V_T5)}, {vr(TS + 4 * q, 4)} offset:{buf + q * 1024}")
            a.label(l_meet)
            a.i("s_waitcnt lgkmcnt(0)")
            a.i("s_barrier")
            a.i(f"s_cmp_eq_u32 {sr(S_W)}, {d}")
            a.i(f"s_cbranch_scc0 {l_skip}")
            for k in range(3):
                for q in range(4):
                    a.i(f"ds_read_b128 {vr(TR[k] + 4 * q, 4)}, {vr(V_SCR16)} offset:{buf + k * 4096 + q * 1024}")
            for r in range(16):
                a.i(f"v_accvgpr_read_b32 {vr(TO + r)}, {ar(A_O[blk] + 16 * d + r)}")
            for r in range(16):
                a.i(f"v_mul_f32 {vr(TO + r)}, {vr(TO + r)}, {vr(V_INV[blk])}")
            a.i("s_waitcnt lgkmcnt(0)")
            for k in range(3):
                for r in range(16):
                    a.i(f"v_add_f32 {vr(TO + r)}, {vr(TO + r)}, {vr(TR[k] + r)}")
            a.i(f"v_cmp_gt_i32 vcc, {sr(S_VALID[blk])}, {vr(V_M)}")
            for e in range(8):   # 0 * inf / NaN of a row that does not exist must still store 0
                a.i(f"v_cvt_pk_bf16_f32 {vr(V_T1)}, {vr(TO + 2 * e)}, {vr(TO + 2 * e + 1)}")
                a.i(f"v_cndmask_b32 {vr(V_CTX[blk] + e)}, 0, {vr(V_T1)}, vcc")
            a.label(l_skip)
            rnd += 1
    emit_ctx_dest(a, True)


def emit_ks_item(a, nb):
    """A tail group of nb = 1 or 2 query blocks as a KEY-SPLIT item: all four waves take the same query block(s) and a
    quarter of the key blocks each (key block kb belongs to wave kb % 4), with all 128 features -- no wave repeats
    another one's scores or exponentials; the partial (O, l, reference) are combined through LDS at the end
    (emit_ks_combine).  The K / V^T stream, its barriers and its counted waits are those of every item (one stage = two
    key blocks per barrier); the arithmetic happens on every second stage, when the pair (c, c + 1) is resident."""
    blocks = ("A", "B")[:nb]
    loop, done = (f".Lpw_ks{nb}_{x}" for x in ("loop", "done"))
    a.i("s_waitcnt vmcnt(8)")   # the staged Q is older than the newest stage of DMA pieces
    for blk in blocks:
        for r in range(32):
            a.i(f"v_accvgpr_write_b32 {ar(A_Q[blk] + r)}, {vr(V_QS + (0 if blk == 'A' else 32) + r)}")
    for blk in blocks:
        for r in range(64):
            a.i(f"v_accvgpr_write_b32 {ar(A_O[blk] + r)}, 0")
        a.i(f"v_mov_b32 {vr(V_L[blk])}, 0")
        for r in range(16):
            a.i(f"v_mov_b32 {vr(V_NEGM[blk] + r)}, 0")
    a.i(f"s_mov_b32 {sr(S_KB)}, {sr(S_W)}")
    a.i(f"s_mov_b32 {sr(S_KFIRST)}, 1")
    a.i(f"s_mov_b32 {sr(S_KPEND)}, 0")
    a.i(f"s_mov_b32 {sr(S_CNT)}, {sr(S_NST)}")
    stamp(a, 22)
    a.label(loop)
    for half in (0, 1):
        plain, joined = a.uniq("ksplain"), a.uniq("ksjoined")
        stamp(a, 11)
        emit_stage_top(a)
        stamp(a, 2)
        dma = [] if ABLATE & 1 else dma_half_ops(0) + dma_half_ops(1)
        if half == 0:   # the wave's key block of this pair, if the sequence has it: scores + softmax
            a.i(f"s_cmp_lt_u32 {sr(S_KB)}, {sr(S_QB)}")
            a.i(f"s_cbranch_scc0 {plain}")
            emit_ks_scores(a, nb, dma)
        else:           # ... and its share of the context
            a.i(f"s_cmp_eq_u32 {sr(S_KPEND)}, 0")
            a.i(f"s_cbranch_scc1 {plain}")
            emit_ks_pv(a, nb, dma)
        a.i(f"s_branch {joined}")
        a.label(plain)
        for op in dma:
            a.i(op)
        a.label(joined)
        emit_dma_advance(a)
        stamp(a, 20)
        if half == 1:
            a.i(f"s_add_u32 {sr(S_KB)}, {sr(S_KB)}, 4")
        for op in rotate_ops():
            a.i(op)
        a.i(f"s_sub_u32 {sr(S_CNT)}, {sr(S_CNT)}, 1")
        if half == 0:
            a.i(f"s_cmp_eq_u32 {sr(S_CNT)}, 0")
            a.i(f"s_cbranch_scc1 {done}")
        else:
            a.i(f"s_cmp_lg_u32 {sr(S_CNT)}, 0")
            a.i(f"s_cbranch_scc1 {loop}")
    a.label(done)
    nopend = a.uniq("ksnopend")
    a.i(f"s_cmp_eq_u32 {sr(S_KPEND)}, 0")     # the item's last stage has no partner: its tile's context here
    a.i(f"s_cbranch_scc1 {nopend}")
    emit_ks_pv(a, nb, [])
    a.label(nopend)
    stamp(a, 11)
    emit_ks_combine(a, nb)
    stamp(a, 21)
    a.i("s_branch .Lpw_next_item")


def emit_all():
    a = Asm()
    a.c("generated by scripts/gen_attn_pw.py -- do not edit")
    # ---- inputs -> fixed homes (operands: see savad_attn_pw_bf16.h)
    a.i(f"s_mov_b64 {sr(S_QF, 2)}, %0")
    a.i(f"s_mov_b64 {sr(S_KF, 2)}, %1")
    a.i(f"s_mov_b64 {sr(S_VTF, 2)}, %2")
    a.i(f"s_mov_b64 {sr(S_CTXF, 2)}, %3")
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
    a.i("s_mov_b64 exec, -1")
    for _ in range(PAD):
        a.i("s_nop 0")
    if WGTIME:
        a.i(f"s_memrealtime {sr(S_TM, 2)}")
        a.i("s_waitcnt lgkmcnt(0)")
        a.i(f"s_and_b32 {sr(81)}, {sr(S_TM)}, 0xffff")        # start, low 16 bits of the 10 ns clock
    if TIMING:
        a.i(f"v_mov_b32 {vr(V_ACC)}, 0")
        a.i(f"s_memtime {sr(S_TM, 2)}")
        a.i("s_waitcnt lgkmcnt(0)")
        a.i(f"v_writelane_b32 {vr(V_ACC)}, {sr(S_TM)}, 31")
        a.i(f"v_writelane_b32 {vr(V_ACC)}, {sr(S_TM)}, 16")       # lanes 16 / 17: shader clock at entry / exit
        a.i(f"s_memrealtime {sr(S_TM, 2)}")
        a.i("s_waitcnt lgkmcnt(0)")
        a.i(f"v_writelane_b32 {vr(V_ACC)}, {sr(S_TM)}, 18")       # lanes 18 / 19: 100 MHz clock at entry / exit
    # ---- derived constants
    a.i(f"s_add_u32 {sr(S_QB)}, {sr(S_T)}, 31")
    a.i(f"s_lshr_b32 {sr(S_QB)}, {sr(S_QB)}, 5")
    a.i(f"s_add_u32 {sr(S_NST)}, {sr(S_QB)}, 1")
    a.i(f"s_lshr_b32 {sr(S_NST)}, {sr(S_NST)}, 1")
    a.i(f"s_lshr_b32 {sr(S_NGF)}, {sr(S_QB)}, 3")
    a.i(f"s_and_b32 {sr(S_TAILQ)}, {sr(S_QB)}, 7")
    a.i(f"s_and_b32 {sr(S_RAG)}, {sr(S_T)}, 31")
    a.i(f"s_add_u32 {sr(S_BX)}, %4, 7")                      # sequences of this XCD: b = 8 bi + xcd < B
    a.i(f"s_sub_u32 {sr(S_BX)}, {sr(S_BX)}, {sr(S_XCD)}")
    a.i(f"s_lshr_b32 {sr(S_BX)}, {sr(S_BX)}, 3")
    # d = the largest power of two that divides NGF and the stride (the stride is one: 32 workgroups per XCD); S_ATTSH = log2(stride / d)
    a.i(f"s_max_u32 {sr(S_T0)}, {sr(S_NGF)}, 1")
    a.i(f"s_ff1_i32_b32 {sr(S_T0)}, {sr(S_T0)}")
    a.i(f"s_ff1_i32_b32 {sr(S_T1)}, {sr(S_STRIDE)}")
    a.i(f"s_min_u32 {sr(S_T0)}, {sr(S_T0)}, {sr(S_T1)}")
    a.i(f"s_sub_u32 {sr(S_ATTSH)}, {sr(S_T1)}, {sr(S_T0)}")
    a.i(f"s_lshl_b32 {sr(S_ATTMASK)}, 1, {sr(S_T0)}")
    a.i(f"s_sub_u32 {sr(S_ATTMASK)}, {sr(S_ATTMASK)}, 1")
    a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_W)}, 12")
    a.i(f"s_add_u32 {sr(S_LDSW)}, {sr(S_LDS)}, {sr(S_T0)}")
    a.i(f"v_add_u32 {vr(V_OFF[1])}, 4096, {vr(V_LANE16)}")
    a.i(f"v_lshrrev_b32 {vr(V_M)}, 4, {vr(V_LANE16)}")        # lane
    a.i(f"v_lshrrev_b32 {vr(V_H4)}, 5, {vr(V_M)}")            # h
    a.i(f"v_lshlrev_b32 {vr(V_H4)}, 2, {vr(V_H4)}")           # 4 h
    a.i(f"v_and_b32 {vr(V_M)}, 31, {vr(V_M)}")
    a.i(f"v_mov_b32 {vr(V_NEGB)}, {hex(NEG_BIG_BITS)}")
    a.i(f"v_add_u32 {vr(V_LDSL)}, {sr(S_LDS)}, {vr(V_LANE16)}")
    a.i(f"v_mov_b32 {vr(V_ADDR_V)}, {vr(V_LDSL)}")            # the stream's stage 0 sits in slot 0, stage 1 in slot 1
    a.i(f"s_mov_b32 {sr(S_KS)}, {STAGE}")
    a.i(f"v_add_u32 {vr(V_ADDR_K)}, {sr(S_KS)}, {vr(V_LDSL)}")
    a.i(f"s_mov_b32 {sr(S_PEND)}, 0")
    a.i(f"s_mov_b32 {sr(S_QPEND)}, 0")
    a.i(f"s_mov_b32 {sr(S_PRE)}, 0")
    a.i(f"s_mov_b32 {sr(S_DSTREAM)}, 0")
    a.i(f"s_mov_b32 {sr(S_DLDS)}, {sr(S_LDSW)}")
    a.i(f"s_mov_b32 {sr(S_DBASE)}, {sr(S_LDSW)}")
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
    for _ in range(AHEAD):
        emit_dma_half(a, 0)
        emit_dma_half(a, 1)
        emit_dma_advance(a)
    # stage 0 has landed for everybody before the first item's prologue reads K(0) from it
    a.i(f"s_waitcnt vmcnt({8 * (AHEAD - 1)})")
    a.i("s_barrier")

    # ================================================================================ item loop
    a.label(".Lpw_item")
    a.i(f"s_mov_b32 {sr(S_FLAGS)}, {sr(S_NFLAGS)}")
    a.i(f"s_mov_b32 {sr(S_QA)}, {sr(S_NQA)}")
    a.i(f"s_mov_b32 {sr(S_SEQBLK)}, {sr(S_NSEQBLK)}")
    a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_QA)}, 5")              # rows that exist in block A / B: T - 32 qb
    a.i(f"s_sub_i32 {sr(S_VALID['A'])}, {sr(S_T)}, {sr(S_T0)}")
    a.i(f"s_sub_i32 {sr(S_VALID['B'])}, {sr(S_VALID['A'])}, 32")
    # advance the compute cursor and describe the next item (its Q is requested behind this item's first barrier)
    stamp(a, 0)
    emit_cursor_next(a, S_CC, "c")
    stamp(a, 24)
    a.i(f"s_mov_b32 {sr(S_NFLAGS)}, 0")
    l_nonext = a.uniq("nonext")
    a.i(f"s_cmp_eq_u32 {sr(S_CC + 3)}, 0")
    a.i(f"s_cbranch_scc1 {l_nonext}")
    emit_item_params(a, S_CC, S_NFLAGS, S_NQA, S_NSEQBLK)
    a.label(l_nonext)
    a.i(f"s_mov_b32 {sr(S_QPEND)}, 1")
    stamp(a, 25)
    a.i(f"s_bitcmp1_b32 {sr(S_FLAGS)}, 0")
    a.i("s_cbranch_scc0 .Lpw_idle_item")
    if SPLIT_MAX >= 1:
        a.i(f"s_bitcmp1_b32 {sr(S_FLAGS)}, 2")
        a.i("s_cbranch_scc1 .Lpw_ks_item")
    emit_item_prologue(a)
    stamp(a, 1)
    emit_item_body(a, "n")

    # ---- a wave without a query block in this item: keeps the stream and the barriers going
    a.label(".Lpw_idle_item")
    a.i(f"s_mov_b32 {sr(S_CNT)}, {sr(S_NST)}")
    a.label(".Lpw_idle_stage")
    emit_stage_top(a)
    if not ABLATE & 1:
        emit_dma_half(a, 0)
        emit_dma_half(a, 1)
    emit_dma_advance(a)
    for op in rotate_ops():
        a.i(op)
    stamp(a, 9)
    a.i(f"s_sub_u32 {sr(S_CNT)}, {sr(S_CNT)}, 1")
    a.i(f"s_cmp_lg_u32 {sr(S_CNT)}, 0")
    a.i("s_cbranch_scc1 .Lpw_idle_stage")

    a.label(".Lpw_next_item")
    a.i(f"s_cmp_eq_u32 {sr(S_CC + 3)}, 0")
    a.i("s_cbranch_scc0 .Lpw_item")
    # ---- the last item's stores
    emit_ctx_stores(a)
    stamp(a, 10)
    if TIMING:  # wave 0 of workgroup 0 publishes its counters
        l_nopub = a.uniq("nopub")
        a.i(f"s_memtime {sr(S_TM, 2)}")
        a.i("s_waitcnt lgkmcnt(0)")
        a.i(f"v_writelane_b32 {vr(V_ACC)}, {sr(S_TM)}, 17")
        a.i(f"s_memrealtime {sr(S_TM, 2)}")
        a.i("s_waitcnt lgkmcnt(0)")
        a.i(f"v_writelane_b32 {vr(V_ACC)}, {sr(S_TM)}, 19")
        a.i(f"s_or_b32 {sr(S_T0)}, {sr(S_XCD)}, {sr(S_J)}")
        a.i(f"s_or_b32 {sr(S_T0)}, {sr(S_T0)}, {sr(S_W)}")
        a.i(f"s_cmp_lg_u32 {sr(S_T0)}, 0")
        a.i(f"s_cbranch_scc1 {l_nopub}")
        a.i("s_mov_b32 exec_hi, 0")
        a.i(f"v_lshlrev_b32 {vr(V_T0)}, 2, {vr(V_M)}")
        a.i(f"global_store_dword {vr(V_T0)}, {vr(V_ACC)}, %16")   # the input operand's own register (s0..s23 are never written)
        a.i("s_mov_b32 exec_hi, -1")
        a.label(l_nopub)
    a.label(".Lpw_end")
    if WGTIME:
        l_nowg = a.uniq("nowg")
        a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
        a.i(f"s_memrealtime {sr(S_TM, 2)}")
        a.i("s_waitcnt lgkmcnt(0)")
        a.i(f"s_cmp_lg_u32 {sr(S_W)}, 0")
        a.i(f"s_cbranch_scc1 {l_nowg}")
        a.i(f"s_cmp_ge_u32 {sr(S_J)}, 16")
        a.i(f"s_cbranch_scc1 {l_nowg}")
        a.i(f"s_lshl_b32 {sr(S_TM)}, {sr(S_TM)}, 16")
        a.i(f"s_or_b32 {sr(S_TM)}, {sr(S_TM)}, {sr(81)}")     # end << 16 | start
        a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_J)}, 3")
        a.i(f"s_add_u32 {sr(S_T0)}, {sr(S_T0)}, {sr(S_XCD)}")
        a.i(f"s_lshl_b32 {sr(S_T0)}, {sr(S_T0)}, 2")
        a.i(f"v_mov_b32 {vr(V_T0)}, {sr(S_T0)}")
        a.i(f"v_mov_b32 {vr(V_T1)}, {sr(S_TM)}")
        a.i("s_mov_b64 exec, 1")
        a.i(f"global_store_dword {vr(V_T0)}, {vr(V_T1)}, %16")
        a.i("s_mov_b64 exec, -1")
        a.label(l_nowg)
    a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a.i("s_branch .Lpw_exit")
    # ---- out-of-line code: stubs, subroutines
    # ---- key-split tail items
    if SPLIT_MAX >= 1:
        a.label(".Lpw_ks_item")
        if SPLIT_MAX >= 2:
            a.i(f"s_bitcmp1_b32 {sr(S_FLAGS)}, 1")
            a.i("s_cbranch_scc0 .Lpw_ks1_item")
            emit_ks_item(a, 2)
        a.label(".Lpw_ks1_item")
        emit_ks_item(a, 1)
    a.lines += a.tail
    a.tail = []
    for cur in (0, 1):
        for hn in (True, False):
            emit_cold_mid(a, cur, hn, 2)
    if SPLIT_MAX >= 1:
        emit_cold_mid(a, 0, False, 1)
        for nb in range(1, min(SPLIT_MAX, 2) + 1):
            emit_ks_first(a, nb)
    emit_cold_first(a)
    emit_post_barrier_sub(a)
    emit_dma_next_item_sub(a)
    a.lines += a.tail
    a.label(".Lpw_exit")
    return a


def render(a):
    body = "\n".join(a.lines)
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
    global TIMING, ABLATE, SPLIT_MAX
    if "--out" in sys.argv:  # experiment variant: [--ablate MASK] [--timing] --out FILE
        ABLATE = int(sys.argv[sys.argv.index("--ablate") + 1]) if "--ablate" in sys.argv else 0
        TIMING = "--timing" in sys.argv or "--count" in sys.argv
        global WGTIME
        WGTIME = "--wgtime" in sys.argv
        global COUNT_ONLY
        COUNT_ONLY = "--count" in sys.argv
        if "--pad" in sys.argv:
            global PAD
            PAD = int(sys.argv[sys.argv.index("--pad") + 1])
        global QPOL, CPOL
        if "--qpol" in sys.argv:
            QPOL = POLICY[int(sys.argv[sys.argv.index("--qpol") + 1])]
        if "--cpol" in sys.argv:
            CPOL = POLICY[int(sys.argv[sys.argv.index("--cpol") + 1])]
        if "--balance" in sys.argv:
            global BALANCE
            BALANCE = int(sys.argv[sys.argv.index("--balance") + 1])
        if "--nt" in sys.argv:
            global NT
            NT = int(sys.argv[sys.argv.index("--nt") + 1])
        if "--no-seam" in sys.argv:
            global SEAM
            SEAM = False
        if "--no-attach" in sys.argv:
            global ATTACH
            ATTACH = False
        if "--rowsum" in sys.argv:
            global ROWSUM
            ROWSUM = sys.argv[sys.argv.index("--rowsum") + 1]
            assert ROWSUM in ("seq", "pk")
        if "--split-max" in sys.argv:
            SPLIT_MAX = int(sys.argv[sys.argv.index("--split-max") + 1])
        Path(sys.argv[sys.argv.index("--out") + 1]).write_text(render(emit_all()))
        return
    a = emit_all()
    text, clob = render(a), render_clobbers()
    keep = SPLIT_MAX
    SPLIT_MAX = 0
    a0 = emit_all()
    text0 = render(a0).replace(".Lpw_", ".Lpwn_")   # (both streams live in one translation unit: assembler-local labels must differ)
    assert render_clobbers() == clob   # one clobber list serves both streams
    SPLIT_MAX = keep
    if "--check" in sys.argv:
        stale = [f for f, t in ((OUT, text), (OUT_CLOB, clob), (OUT_NOSPLIT, text0)) if not f.exists() or f.read_text() != t]
        if stale:
            print(f"stale: {[str(f) for f in stale]}: run python scripts/gen_attn_pw.py", file=sys.stderr)
            sys.exit(1)
        return
    OUT.write_text(text)
    OUT_CLOB.write_text(clob)
    OUT_NOSPLIT.write_text(text0)
    print(f"{OUT}: {len(a.lines)} lines; {OUT_NOSPLIT}: {len(a0.lines)} lines", file=sys.stderr)


if __name__ == "__main__":
    main()
