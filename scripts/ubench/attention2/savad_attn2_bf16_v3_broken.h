// savad_attn2_bf16.h -- bf16 flash attention, second generation: 64 query rows per wave, one wave per SIMD.
//
// Same data layout and arithmetic as attention_kernel_bf16 (savad_kernels_bf16.h): fragment-major Q / K / V^T in,
// normalised context fragments out, scores in the base-2 exponent domain relative to a per-row reference.
// What changes is the schedule (vad/modeling/transformer.py:305-346,351-363 is still what is computed):
//   * a wave owns a PAIR of query blocks (64 rows): every K / V^T fragment it reads from LDS feeds two MFMAs, and
//     every DMA instruction, barrier and loop instruction is shared by twice the matrix work -- the first-generation
//     kernel issued ~10 non-MFMA instructions per MFMA (rocprofv3: 5.5 VALU + 3.7 SALU + 1 LDS) and was issue bound
//     at 39 % MFMA-busy;
//   * a workgroup is 4 such waves, ONE PER SIMD, and walks its query pairs in rounds; the K / V^T stream of the
//     sequence runs continuously through a 4-stage LDS ring (64 keys = 32 KiB per stage, 128 KiB), fed by
//     asynchronous global->LDS DMA with counted vmcnt waits;
//   * register file by hand: the accumulator file holds O (a[0:127]) and Q (a[128:191]), owned by inline asm and
//     invisible to the compiler's allocator -- left to itself, hipcc selects the AGPR form for every MFMA of a
//     one-wave-per-SIMD kernel and then moves every score tile AGPR -> VGPR for the softmax and shuffles the O
//     accumulators between the files at the loop back-edge (150-400 v_accvgpr moves per key tile, measured in the
//     ISA); the score tiles, the reference, K / V^T fragments and the probabilities are ordinary variables in the
//     256 architectural VGPRs, and the S^T MFMAs are issued in VGPR form (D = scores, C = -reference);
//   * the instruction stream of a key tile is laid out by hand (every instruction of the hot loop is a volatile asm
//     statement, so program order IS issue order): S(j+1) = K(j+1) Q^T rides in front of the softmax of tile j --
//     per MFMA slot 3-4 VALU instructions (exponentials, row sums, bf16 packing) -- and the row maxima of tile j+1
//     sit between the MFMAs of O += V^T(j) P(j); consecutive MFMAs never share an accumulator;
//   * the reference of a row only moves when a score exceeds it by 2^40 (or on the first tile of a round, where it
//     is set to the row maximum): p <= 2^40 (bf16 keeps relative precision at any scale), sums in fp32 stay far
//     inside range.  One rarely taken branch per tile.
// Hazards the compiler cannot see into asm (CDNA3/4 ISA, "manually inserted wait states"): an MFMA result in VGPRs
// must not be read by a VALU instruction for passes + 3 wait states -- by construction every consumer of a score
// tile sits at least 8 MFMAs behind its producer (s_nop padding on the two cold paths); a transcendental result is
// never consumed by the next instruction.
#pragma once
#include <type_traits>

#include "savad_kernels_bf16.h"

namespace savad {
namespace bf {

constexpr int A2_NRING = 4;                    // LDS stages (2 key blocks of K + 2 of V^T each)
constexpr int A2_STAGE_BYTES = 4 * BLK_BYTES;  // 32 KiB
constexpr float A2_MOVE_LOG2 = 40.0f;
// accumulator-file map (asm-owned)
constexpr int A2_OA = 0, A2_OB = 64, A2_QA = 128, A2_QB = 160;  // K fragments a[192:223], V^T fragments a[224:255]

#define A2_CLOB10(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
// tells the compiler that a0..a255 are in use (kernel descriptor's AGPR count; never picked as spill slots)
__device__ __forceinline__ void a2_reserve_acc() {
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", A2_CLOB10(1), A2_CLOB10(2), A2_CLOB10(3), A2_CLOB10(4),
                 A2_CLOB10(5), A2_CLOB10(6), A2_CLOB10(7), A2_CLOB10(8), A2_CLOB10(9), A2_CLOB10(10), A2_CLOB10(11), A2_CLOB10(12),
                 A2_CLOB10(13), A2_CLOB10(14), A2_CLOB10(15), A2_CLOB10(16), A2_CLOB10(17), A2_CLOB10(18), A2_CLOB10(19), A2_CLOB10(20),
                 A2_CLOB10(21), A2_CLOB10(22), A2_CLOB10(23), A2_CLOB10(24), "a250", "a251", "a252", "a253", "a254", "a255");
}

// ---- the instruction set of the hot loop (volatile: program order is issue order)
template <int Q0>  // first MFMA of a score chain: D = K-fragment x Q-fragment + (-reference)
__device__ __forceinline__ void a2_mfma_s0(f32x16& d, const bf16x8& k, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%3:%4], %2" : "=&v"(d) : "v"(k), "v"(c), "n"(Q0), "n"(Q0 + 3));
}
template <int Q0>
__device__ __forceinline__ void a2_mfma_s(f32x16& d, const bf16x8& k) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(d) : "v"(k), "n"(Q0), "n"(Q0 + 3));
}
template <int O0>  // O^T block += V^T fragment x P fragment
__device__ __forceinline__ void a2_mfma_o(const bf16x8& v, const bf16x8& p) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(v), "v"(p), "n"(O0), "n"(O0 + 15));
}
__device__ __forceinline__ float a2_exp2(float x) {
    float r;
    asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float a2_add(float a, float b) {
    float r;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned a2_cvt2(float lo, float hi) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float a2_max3(float a, float b, float c) {
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ bf16x8 a2_frag(unsigned a, unsigned b, unsigned c, unsigned d) { return __builtin_bit_cast(bf16x8, u32x4{a, b, c, d}); }

template <int A0, int N>  // a[A0 .. A0+N) = 0
__device__ __forceinline__ void a2_acc_zero() {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_accvgpr_write_b32 a[%0], 0" : : "n"(A0 + i));
}
template <int A0>  // a[A0 .. A0+4) = one fragment
__device__ __forceinline__ void a2_acc_put4(const bf16x8& f) {
    const u32x4 u = __builtin_bit_cast(u32x4, f);
    asm volatile("v_accvgpr_write_b32 a[%4], %0\n\tv_accvgpr_write_b32 a[%5], %1\n\tv_accvgpr_write_b32 a[%6], %2\n\tv_accvgpr_write_b32 a[%7], %3"
                 : : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "n"(A0), "n"(A0 + 1), "n"(A0 + 2), "n"(A0 + 3));
}
template <int A0>
__device__ __forceinline__ void a2_acc_get16(f32x16& v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float t;
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(t) : "n"(A0 + i));
        v[i] = t;
    }
}
template <int A0>  // a[A0 .. A0+64) *= alpha (cold path)
__device__ __forceinline__ void a2_acc_scale64(float alpha) {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        float t;
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(t) : "n"(A0 + i));
        t *= alpha;
        asm volatile("v_accvgpr_write_b32 a[%0], %1" : : "n"(A0 + i), "v"(t));
    }
}
__device__ __forceinline__ void a2_nops24() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7"); }

// One DMA instruction: 64 lanes x 16 B from src (wave-uniform) + lane * 16 to LDS byte address lds_addr + lane * 16.
// M0 is not saved / restored: nothing else in this kernel uses it (gfx9 DS instructions do not; checked in the ISA).
__device__ __forceinline__ void a2_dma1k(const char* src, unsigned lds_addr, unsigned lane_off) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(lane_off), "s"(src), "s"(lds_addr)
        : "memory");
}

// stage gs (global stage counter) = key blocks 2s, 2s+1 of the sequence (s = gs % NST): wave w moves 8 of the 32 KiB
__device__ __forceinline__ void a2_issue_stage(char* smem, int gs, int NST, const char* kseq, const char* vtseq, int w, int lane) {
    if (SAVAD_ABLATE & 1) return;
    const int s = gs % NST;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem) + (unsigned)(gs & (A2_NRING - 1)) * A2_STAGE_BYTES;
    const unsigned off = (unsigned)lane * 16u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = w + 4 * k;  // KiB 0..31 of the stage: [K blk 2s | K blk 2s+1 | V^T blk 2s | V^T blk 2s+1]
        const char* base = (k < 4 ? kseq : vtseq) + (size_t)(2 * s) * BLK_BYTES + (size_t)(i & 15) * FRAG_BYTES;
        a2_dma1k(base, lds0 + (unsigned)i * FRAG_BYTES, off);
    }
}

// Wait until this wave's share of a stage has landed, then barrier.  younger (wave-uniform): at least one stage was
// issued after it -- then "at most 8 DMA instructions outstanding" implies it has landed (loads return in order; other
// vector-memory operations in flight can only make the wait longer) -- else everything is drained.
__device__ __forceinline__ void a2_acquire(bool younger) {
    if (SAVAD_ABLATE & 2) return;
    if (younger)
        __builtin_amdgcn_s_waitcnt(0x0F70 | 8);  // vmcnt(8)
    else
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    asm volatile("" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ float a2_max16(const f32x16& v) {
    float m = a2_max3(v[0], v[1], v[2]);
#pragma unroll
    for (int r = 3; r + 1 < 16; r += 2) m = a2_max3(m, v[r], v[r + 1]);
    return fmaxf(m, v[15]);
}

// keys that do not exist (ragged last tile) -> probability 0.  lane (m,h), register r <-> key 8(r>>2)+4h+(r&3)
__device__ __forceinline__ void a2_mask(f32x16& sc, int lim /* T - 32*jt - 4*h */) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = (8 * (r >> 2) + (r & 3) < lim) ? sc[r] : NEG_BIG;
}

// Reference move of a freshly computed score tile (already relative to the current reference), cold path.
// first: the reference is SET to the row maximum (O and l are still zero); otherwise l and O (a[O0..O0+64)) are rescaled.
template <int O0>
__device__ __forceinline__ void a2_move(f32x16& sc, f32x16& negm, float& l, float mx /* row maximum, both halves */, bool first) {
    const bool move = first || (mx > A2_MOVE_LOG2);
    const float d = move ? mx : 0.0f;  // new reference = old + d
    if (!first) {
        const float alpha = __builtin_amdgcn_exp2f(-d);
        l *= alpha;
        a2_nops24();  // the last O MFMAs must have retired before their accumulators are read
        a2_acc_scale64<O0>(alpha);
        asm volatile("s_nop 3");
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        sc[r] -= d;
        negm[r] -= d;
    }
}

__device__ __forceinline__ void a2_load_frags8(bf16x8 (&f)[8], const char* p, int lane) {
    if (SAVAD_ABLATE & 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane, 0x3c003c00u, (unsigned)i, 0u});
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = ldfrag(p + (i * 64 + lane) * 16);
}

// scores of one tile for both query blocks: the two chains alternate, K fragment ks feeds both
__device__ __forceinline__ void a2_scores(f32x16& da, f32x16& db, const bf16x8 (&k)[8], const f32x16& ca, const f32x16& cb) {
    a2_mfma_s0<A2_QA + 0>(da, k[0], ca);
    a2_mfma_s0<A2_QB + 0>(db, k[0], cb);
    a2_mfma_s<A2_QA + 4>(da, k[1]);
    a2_mfma_s<A2_QB + 4>(db, k[1]);
    a2_mfma_s<A2_QA + 8>(da, k[2]);
    a2_mfma_s<A2_QB + 8>(db, k[2]);
    a2_mfma_s<A2_QA + 12>(da, k[3]);
    a2_mfma_s<A2_QB + 12>(db, k[3]);
    a2_mfma_s<A2_QA + 16>(da, k[4]);
    a2_mfma_s<A2_QB + 16>(db, k[4]);
    a2_mfma_s<A2_QA + 20>(da, k[5]);
    a2_mfma_s<A2_QB + 20>(db, k[5]);
    a2_mfma_s<A2_QA + 24>(da, k[6]);
    a2_mfma_s<A2_QB + 24>(db, k[6]);
    a2_mfma_s<A2_QA + 28>(da, k[7]);
    a2_mfma_s<A2_QB + 28>(db, k[7]);
}

#ifdef SAVAD_TIMING
#define A2_T(i) do { tn_ = __builtin_readcyclecounter(); tacc_[i] += tn_ - tp_; tp_ = tn_; } while (0)
#else
#define A2_T(i) do {} while (0)
#endif

// ---- asm-owned operand fragments: K of the tile whose scores are computed next in a[192:223], V^T of the tile whose
// probabilities are consumed next in a[224:255]; LDS reads go straight into them and are counted by hand
constexpr int A2_KF = 192, A2_VF = 224;
template <int A0, int OFF>  // one fragment: a[A0 .. A0+4) = LDS[addr + OFF .. +1 KiB) (lane-linear)
__device__ __forceinline__ void a2_lds_frag(unsigned addr) {
    if (SAVAD_ABLATE & 8) return;
    asm volatile("ds_read_b128 a[%1:%2], %0 offset:%3" : : "v"(addr), "n"(A0), "n"(A0 + 3), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void a2_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N));
}
template <int KS, int QBASE>  // score chain MFMA: D (VGPR) = K fragment KS x Q fragment KS + C
__device__ __forceinline__ void a2_s_first(f32x16& d, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%2:%3], a[%4:%5], %1" : "=&v"(d) : "v"(c), "n"(A2_KF + 4 * KS), "n"(A2_KF + 4 * KS + 3),
                 "n"(QBASE + 4 * KS), "n"(QBASE + 4 * KS + 3));
}
template <int KS, int QBASE>
__device__ __forceinline__ void a2_s_first0(f32x16& d) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%1:%2], a[%3:%4], 0" : "=v"(d) : "n"(A2_KF + 4 * KS), "n"(A2_KF + 4 * KS + 3),
                 "n"(QBASE + 4 * KS), "n"(QBASE + 4 * KS + 3));
}
template <int KS, int QBASE>
__device__ __forceinline__ void a2_s_acc(f32x16& d) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%1:%2], a[%3:%4], %0" : "+v"(d) : "n"(A2_KF + 4 * KS), "n"(A2_KF + 4 * KS + 3),
                 "n"(QBASE + 4 * KS), "n"(QBASE + 4 * KS + 3));
}
template <int O0, int VI>  // O^T block (a[O0..O0+16)) += V^T fragment VI x P
__device__ __forceinline__ void a2_o_acc(const bf16x8& p) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%1:%2], a[%3:%4], %0, a[%1:%2]" : : "v"(p), "n"(O0), "n"(O0 + 15), "n"(A2_VF + 4 * VI),
                 "n"(A2_VF + 4 * VI + 3));
}
template <int O0, int VI>  // first touch of an accumulator in a round: C = 0
__device__ __forceinline__ void a2_o_first(const bf16x8& p) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%1:%2], a[%3:%4], %0, 0" : : "v"(p), "n"(O0), "n"(O0 + 15), "n"(A2_VF + 4 * VI),
                 "n"(A2_VF + 4 * VI + 3));
}
// DMA piece K of a half stage: m0 = ldsb + IMM
template <int IMM>
__device__ __forceinline__ void a2_dma_piece(const char* src, unsigned ldsb, unsigned voff) {
    if (SAVAD_ABLATE & 1) return;
    asm volatile(
        "s_add_u32 m0, %2, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(src), "s"(ldsb), "n"(IMM)
        : "memory", "scc");
}
struct A2Desc {  // one stage to be DMA'd by this wave: 4 pieces of K, then 4 of V^T (1 KiB each)
    const char* ksrc;
    const char* vsrc;
    unsigned ldsb;
};

constexpr float A2_SUM_LIMIT = 1.152921504606846976e18f;  // 2^60: a tile's row sum beyond it moves the reference

// Recovery path of one query block (cold): the row sum of tile n overflowed the window, i.e. some score outran the
// reference by ~2^56 or more.  Sets the reference to the tile's row maximum, rescales l and O (which hold tiles < n
// only: the O MFMAs of tile n have not been issued), shifts the scores of tile n AND of tile n+1 (already computed
// against the old reference), and redoes the exponentials / row sum / packing of tile n.
template <int O0>
__device__ __forceinline__ void a2_recover(f32x16& c, f32x16& nx, f32x16& negm, float& l, float& rs, bf16x8& p0, bf16x8& p1) {
    a2_nops24();  // nx was just written by MFMAs
    const float mx = half_max(a2_max16(c));
    a2_move<O0>(c, negm, l, mx, false);  // (mx > A2_MOVE_LOG2 is implied by the overflow, else d = 0 and nothing changes)
    const float d = mx > A2_MOVE_LOG2 ? mx : 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) nx[e] -= d;
    float ex[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ex[e] = __builtin_amdgcn_exp2f(c[e]);
    rs = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) rs += ex[e];
    p0 = a2_frag(a2_cvt2(ex[0], ex[1]), a2_cvt2(ex[2], ex[3]), a2_cvt2(ex[4], ex[5]), a2_cvt2(ex[6], ex[7]));
    p1 = a2_frag(a2_cvt2(ex[8], ex[9]), a2_cvt2(ex[10], ex[11]), a2_cvt2(ex[12], ex[13]), a2_cvt2(ex[14], ex[15]));
}

// T > 32.  Workgroup = (sequence b, group g of NG): the group's query PAIRS [p0, p1) are processed in rounds of 4
// (one pair per wave); every round streams the sequence's key blocks through the ring (global stage = round * NST +
// stage: the stream has no seam between rounds).  Per round and wave: NT tile steps; step j turns the scores of tile j
// into probabilities and accumulates O, and computes the scores of tile j+1 -- in the LAST step of a round those of the
// NEXT round's tile 0 (its Q fragments were switched in during step NT-2), so the pipeline never drains.
__global__ __launch_bounds__(256, 1) void attention2_kernel_bf16(const char* __restrict__ qf, const char* __restrict__ kf,
                                                                 const char* __restrict__ vtf, char* __restrict__ ctxf, int B, int T,
                                                                 int NG) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // A2_NRING stages
    a2_reserve_acc();
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int QB = (T + 31) / 32, NT = QB, NP = (QB + 1) / 2, NST = (QB + 1) / 2;
    int b, g;
    if (!xcd_balanced_map(B, NG, b, g)) return;
    const int p0 = (g * NP) / NG, p1 = ((g + 1) * NP) / NG;
    const int NR = (p1 - p0 + 3) / 4;
    const int GS = NR * NST;
    const char* kseq = kf + (size_t)b * QB * BLK_BYTES + (size_t)w * FRAG_BYTES;   // + this wave's KiB within a 4-KiB group
    const char* vtseq = vtf + (size_t)b * QB * BLK_BYTES + (size_t)w * FRAG_BYTES;
    const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned voff0 = lane16, voff1 = lane16 + 4096u, voff2 = lane16 + 8192u, voff3 = lane16 + 12288u;
    auto issue_stage = [&](int gs) {  // stage gs = key blocks 2s, 2s+1 (s = gs % NST): this wave moves KiB w, w+4, ... of its 32
        const int s = gs % NST;
        const char* ksrc = kseq + (size_t)(2 * s) * BLK_BYTES;
        const char* vsrc = vtseq + (size_t)(2 * s) * BLK_BYTES;
        const unsigned ldsb = ring0 + (unsigned)(gs & (A2_NRING - 1)) * A2_STAGE_BYTES + (unsigned)w * FRAG_BYTES;
        a2_dma_piece<0>(ksrc, ldsb, voff0);
        a2_dma_piece<4096>(ksrc, ldsb, voff1);
        a2_dma_piece<8192>(ksrc, ldsb, voff2);
        a2_dma_piece<12288>(ksrc, ldsb, voff3);
        a2_dma_piece<16384>(vsrc, ldsb, voff0);
        a2_dma_piece<20480>(vsrc, ldsb, voff1);
        a2_dma_piece<24576>(vsrc, ldsb, voff2);
        a2_dma_piece<28672>(vsrc, ldsb, voff3);
    };
    int issued = 0;
    for (; issued < A2_NRING && issued < GS; ++issued) issue_stage(issued);
#ifdef SAVAD_TIMING
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = __builtin_readcyclecounter(), tn_;
#endif

    auto round_blocks = [&](int r, int& qbA, int& qbB, bool& storeA, bool& storeB) {
        const int pair = p0 + 4 * r + w;
        const bool live = pair < p1;  // a wave without a pair computes on the group's first one and stores nothing
        qbA = 2 * (live ? pair : p0);
        const bool hasB = qbA + 1 < QB;  // the last pair of an odd QB is a single block (computed twice, stored once)
        qbB = hasB ? qbA + 1 : qbA;
        storeA = live;
        storeB = live && hasB;
    };
    int qbA, qbB;
    bool storeA, storeB;
    round_blocks(0, qbA, qbB, storeA, storeB);
    bf16x8 qn[16];  // next round's Q fragments on their way to a[128:191]
    auto load_q = [&](int qa, int qb) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            qn[i] = ldfrag(qf + ((size_t)b * QB + qa) * BLK_BYTES + (i * 64 + lane) * 16);
            qn[8 + i] = ldfrag(qf + ((size_t)b * QB + qb) * BLK_BYTES + (i * 64 + lane) * 16);
        }
    };
    auto put_q = [&]() {
        a2_acc_put4<A2_QA + 0>(qn[0]); a2_acc_put4<A2_QA + 4>(qn[1]); a2_acc_put4<A2_QA + 8>(qn[2]); a2_acc_put4<A2_QA + 12>(qn[3]);
        a2_acc_put4<A2_QA + 16>(qn[4]); a2_acc_put4<A2_QA + 20>(qn[5]); a2_acc_put4<A2_QA + 24>(qn[6]); a2_acc_put4<A2_QA + 28>(qn[7]);
        a2_acc_put4<A2_QB + 0>(qn[8]); a2_acc_put4<A2_QB + 4>(qn[9]); a2_acc_put4<A2_QB + 8>(qn[10]); a2_acc_put4<A2_QB + 12>(qn[11]);
        a2_acc_put4<A2_QB + 16>(qn[12]); a2_acc_put4<A2_QB + 20>(qn[13]); a2_acc_put4<A2_QB + 24>(qn[14]); a2_acc_put4<A2_QB + 28>(qn[15]);
    };
    load_q(qbA, qbB);
    put_q();

    f32x16 nega = zero16(), negb = zero16();  // -reference of the lane's query row, in every register
    float la = 0.0f, lb = 0.0f;               // this lane's half of the row sums
    f32x16 s0a, s0b, s1a, s1b;                // score tiles: even tiles of a round in s0*, odd tiles in s1*

    auto k_reads = [&](unsigned a) {  // the 8 K fragments of a tile -> a[192:223]
        a2_lds_frag<A2_KF + 0, 0>(a); a2_lds_frag<A2_KF + 4, 1024>(a); a2_lds_frag<A2_KF + 8, 2048>(a); a2_lds_frag<A2_KF + 12, 3072>(a);
        a2_lds_frag<A2_KF + 16, 4096>(a); a2_lds_frag<A2_KF + 20, 5120>(a); a2_lds_frag<A2_KF + 24, 6144>(a); a2_lds_frag<A2_KF + 28, 7168>(a);
    };
    // first tile of a round: reference := row maximum (O and l untouched)
    auto open_round = [&](f32x16& na, f32x16& nb) {
        a2_nops24();
        nega = zero16();
        negb = zero16();
        float dummy = 0.0f;
        a2_move<A2_OA>(na, nega, dummy, half_max(a2_max16(na)), true);
        a2_move<A2_OB>(nb, negb, dummy, half_max(a2_max16(nb)), true);
    };

    // ---- prologue: stage 0 has landed; K(0) -> scores of tile 0 -> reference; K(1) on its way
    a2_acquire(issued - 1 > 0);
    k_reads(ring0 + lane16);
    a2_lgkm<0>();
    a2_s_first0<0, A2_QA>(s0a); a2_s_first0<0, A2_QB>(s0b);
    a2_s_acc<1, A2_QA>(s0a); a2_s_acc<1, A2_QB>(s0b); a2_s_acc<2, A2_QA>(s0a); a2_s_acc<2, A2_QB>(s0b);
    a2_s_acc<3, A2_QA>(s0a); a2_s_acc<3, A2_QB>(s0b); a2_s_acc<4, A2_QA>(s0a); a2_s_acc<4, A2_QB>(s0b);
    a2_s_acc<5, A2_QA>(s0a); a2_s_acc<5, A2_QB>(s0b); a2_s_acc<6, A2_QA>(s0a); a2_s_acc<6, A2_QB>(s0b);
    a2_s_acc<7, A2_QA>(s0a); a2_s_acc<7, A2_QB>(s0b);
    open_round(s0a, s0b);
    k_reads(ring0 + lane16 + BLK_BYTES);  // K of tile 1 (T > 32: it exists)
    s1a = s0a;
    s1b = s0b;
    A2_T(0);

    // The compiler believes an asm MFMA has consumed its VGPR operands the moment it is issued and hands their registers
    // to the next temporary -- two instructions later, while the matrix core is still reading them (seen: block B's
    // packing written over block A's P fragments; context scaled by 0.44, block B inf).  Every P fragment is therefore
    // kept alive (an empty asm that "uses" it) until at least two further MFMAs have been issued.
    bf16x8 pb0 = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u}), pb1 = pb0;
    auto keep = [](const bf16x8& x, const bf16x8& y) { asm volatile("" : : "v"(x), "v"(y)); };

    // One tile step of round r.  FAST (1 <= j <= NT-3): no special case applies -- tile j+1 and j+2 belong to this round,
    // the accumulators are live, nothing to mask; the stage hand-over falls on the even tiles.  Otherwise the wave-
    // uniform special cases are evaluated at run time (first / last two steps of a round).
    auto step = [&](auto fast_tag, auto even_tag, int r, int j, f32x16& ca, f32x16& cb, f32x16& na, f32x16& nb) {
        constexpr bool FAST = decltype(fast_tag)::value, EVEN = decltype(even_tag)::value;
        const int G0 = r * NST;
        const bool first_pv = FAST ? false : j == 0;
        const bool nr1 = FAST ? false : j + 1 == NT;              // tile j+1 is the next round's tile 0
        const bool has1 = FAST ? true : (!nr1 || r + 1 < NR);     // there is a next tile at all
        int r2 = r, j2 = j + 2;
        if (!FAST && j2 >= NT) {
            j2 -= NT;
            r2 = r + 1;
        }
        const bool has2 = FAST ? true : r2 < NR;
        const int stage1 = nr1 ? G0 + NST : G0 + ((j + 1) >> 1);  // stage of tile j+1
        const int stage2 = r2 * NST + (j2 >> 1);                  // stage of tile j+2
        const bool acq = FAST ? EVEN : (has2 && stage2 != stage1);
        const bool qload = FAST ? false : (r + 1 < NR && j == (NT >= 3 ? NT - 3 : 0));
        const bool qswitch = FAST ? false : (r + 1 < NR && j == NT - 2);
        if (!FAST && 32 * j + 32 > T) {  // ragged last tile of the round: keys that do not exist get probability 0
            const int lim = T - 32 * j - 4 * h;
            a2_mask(ca, lim);
            a2_mask(cb, lim);
        }
        if (FAST ? j == NT - 3 : qload) {
            if (r + 1 < NR) {
                int qa, qb;
                bool sa_, sb_;
                round_blocks(r + 1, qa, qb, sa_, sb_);
                load_q(qa, qb);
            }
        }
        const unsigned vaddr = ring0 + (unsigned)((G0 + (j >> 1)) & (A2_NRING - 1)) * A2_STAGE_BYTES + (unsigned)(2 + (j & 1)) * BLK_BYTES + lane16;
        // ---- phase 1: the 16 score MFMAs of tile j+1; behind each of the first eight, four exponentials of tile j; behind
        //      the last eight, the row sum + bf16 packing of block A and one V^T fragment read each
        float ea[16], eb[16];
        a2_lgkm<0>();  // K(j+1) fragments (read during the previous step)
#ifdef A2_PAD
        a2_nops24();
#endif
        if (!FAST && nr1) {  // next round: new Q (already switched in), reference 0
            a2_s_first0<0, A2_QA>(na);
            ea[0] = a2_exp2(ca[0]); ea[1] = a2_exp2(ca[1]); eb[0] = a2_exp2(cb[0]); eb[1] = a2_exp2(cb[1]);
            a2_s_first0<0, A2_QB>(nb);
        } else {
            a2_s_first<0, A2_QA>(na, nega);
            ea[0] = a2_exp2(ca[0]); ea[1] = a2_exp2(ca[1]); eb[0] = a2_exp2(cb[0]); eb[1] = a2_exp2(cb[1]);
            a2_s_first<0, A2_QB>(nb, negb);
        }
        ea[2] = a2_exp2(ca[2]); ea[3] = a2_exp2(ca[3]); eb[2] = a2_exp2(cb[2]); eb[3] = a2_exp2(cb[3]);
        a2_s_acc<1, A2_QA>(na);
        keep(pb0, pb1);  // the previous step's last O MFMAs have read them by now
        ea[4] = a2_exp2(ca[4]); ea[5] = a2_exp2(ca[5]); eb[4] = a2_exp2(cb[4]); eb[5] = a2_exp2(cb[5]);
        a2_s_acc<1, A2_QB>(nb);
        ea[6] = a2_exp2(ca[6]); ea[7] = a2_exp2(ca[7]); eb[6] = a2_exp2(cb[6]); eb[7] = a2_exp2(cb[7]);
        a2_s_acc<2, A2_QA>(na);
        ea[8] = a2_exp2(ca[8]); ea[9] = a2_exp2(ca[9]); eb[8] = a2_exp2(cb[8]); eb[9] = a2_exp2(cb[9]);
        a2_s_acc<2, A2_QB>(nb);
        ea[10] = a2_exp2(ca[10]); ea[11] = a2_exp2(ca[11]); eb[10] = a2_exp2(cb[10]); eb[11] = a2_exp2(cb[11]);
        a2_s_acc<3, A2_QA>(na);
        ea[12] = a2_exp2(ca[12]); ea[13] = a2_exp2(ca[13]); eb[12] = a2_exp2(cb[12]); eb[13] = a2_exp2(cb[13]);
        a2_s_acc<3, A2_QB>(nb);
        ea[14] = a2_exp2(ca[14]); ea[15] = a2_exp2(ca[15]); eb[14] = a2_exp2(cb[14]); eb[15] = a2_exp2(cb[15]);
        float ra, rb;
        unsigned ua[8], ub[8];
        // (dependent instructions are kept two apart: hipcc pads every adjacent dependent pair of asm statements with s_nop)
#ifdef A2_OLD_SLOT
#define A2_SLOT(x, i)                                                                          \
    r##x = (i) == 0 ? a2_add(e##x[0], e##x[1]) : a2_add(a2_add(r##x, e##x[2 * (i)]), e##x[2 * (i) + 1]); \
    u##x[i] = a2_cvt2(e##x[2 * (i)], e##x[2 * (i) + 1]);
#else
#define A2_SLOT(x, i)                                              \
    r##x = (i) == 0 ? a2_add(e##x[0], e##x[1]) : a2_add(r##x, e##x[2 * (i)]); \
    u##x[i] = a2_cvt2(e##x[2 * (i)], e##x[2 * (i) + 1]);         \
    if ((i) != 0) r##x = a2_add(r##x, e##x[2 * (i) + 1]);
#endif
        // V^T fragments are read in the order phase 2 consumes them: 0, 2, 4, 6, 1, 3, 5, 7
        a2_s_acc<4, A2_QA>(na);
        A2_SLOT(a, 0)
        a2_lds_frag<A2_VF + 0, 0>(vaddr);
        a2_s_acc<4, A2_QB>(nb);
        A2_SLOT(a, 1)
        a2_lds_frag<A2_VF + 8, 2048>(vaddr);
        a2_s_acc<5, A2_QA>(na);
        A2_SLOT(a, 2)
        a2_lds_frag<A2_VF + 16, 4096>(vaddr);
        a2_s_acc<5, A2_QB>(nb);
        A2_SLOT(a, 3)
        a2_lds_frag<A2_VF + 24, 6144>(vaddr);
        a2_s_acc<6, A2_QA>(na);
        A2_SLOT(a, 4)
        a2_lds_frag<A2_VF + 4, 1024>(vaddr);
        a2_s_acc<6, A2_QB>(nb);
        A2_SLOT(a, 5)
        a2_lds_frag<A2_VF + 12, 3072>(vaddr);
        a2_s_acc<7, A2_QA>(na);
        A2_SLOT(a, 6)
        a2_lds_frag<A2_VF + 20, 5120>(vaddr);
        a2_s_acc<7, A2_QB>(nb);
        A2_SLOT(a, 7)
        a2_lds_frag<A2_VF + 28, 7168>(vaddr);
        bf16x8 pa0 = a2_frag(ua[0], ua[1], ua[2], ua[3]), pa1 = a2_frag(ua[4], ua[5], ua[6], ua[7]);
#ifndef A2_NO_RECOVER
        if (__any(!(ra < A2_SUM_LIMIT))) {  // rare, wave-uniform
            asm volatile("" ::: "memory");  // (keeps it a real branch)
            a2_recover<A2_OA>(ca, na, nega, la, ra, pa0, pa1);
        }
#endif
        la = a2_add(la, ra);
        A2_T(2);
        // ---- stage hand-over: the stage of tile j+2 must have landed before its K fragments are read below; after the
        //      barrier nobody needs the stage of tile j any more (its V^T fragments were read above) and the stream moves on
        if (acq) {
            a2_acquire(issued - 1 > stage2);
            A2_T(1);
            if (issued < stage1 + A2_NRING && issued < GS) issue_stage(issued++);
            A2_T(3);
        }
        const unsigned kaddr2 = ring0 + (unsigned)(stage2 & (A2_NRING - 1)) * A2_STAGE_BYTES + (unsigned)(j2 & 1) * BLK_BYTES + lane16;
        // ---- phase 2: the 16 O MFMAs of tile j.  Block A's eight first, with the row sum + packing of block B behind
        //      them; then block B's, with the K fragment reads of tile j+2 (and, once per round, the next round's Q)
        a2_lgkm<0>();  // V^T(j) fragments
#ifdef A2_PAD
        a2_nops24();
#endif
        if (!FAST && first_pv) {
            a2_o_first<A2_OA + 0, 0>(pa0);
            A2_SLOT(b, 0)
            a2_o_first<A2_OA + 16, 2>(pa0);
            A2_SLOT(b, 1)
            a2_o_first<A2_OA + 32, 4>(pa0);
            A2_SLOT(b, 2)
            a2_o_first<A2_OA + 48, 6>(pa0);
            A2_SLOT(b, 3)
        } else {
            a2_o_acc<A2_OA + 0, 0>(pa0);
            A2_SLOT(b, 0)
            a2_o_acc<A2_OA + 16, 2>(pa0);
            A2_SLOT(b, 1)
            a2_o_acc<A2_OA + 32, 4>(pa0);
            A2_SLOT(b, 2)
            a2_o_acc<A2_OA + 48, 6>(pa0);
            A2_SLOT(b, 3)
        }
        a2_o_acc<A2_OA + 0, 1>(pa1);
        A2_SLOT(b, 4)
        a2_o_acc<A2_OA + 16, 3>(pa1);
        A2_SLOT(b, 5)
        a2_o_acc<A2_OA + 32, 5>(pa1);
        A2_SLOT(b, 6)
        a2_o_acc<A2_OA + 48, 7>(pa1);
        A2_SLOT(b, 7)
#undef A2_SLOT
        pb0 = a2_frag(ub[0], ub[1], ub[2], ub[3]);
        pb1 = a2_frag(ub[4], ub[5], ub[6], ub[7]);
#ifndef A2_NO_RECOVER
        if (__any(!(rb < A2_SUM_LIMIT))) {
            asm volatile("" ::: "memory");
            a2_recover<A2_OB>(cb, nb, negb, lb, rb, pb0, pb1);
        }
#endif
        lb = a2_add(lb, rb);
        if (!FAST && qswitch) put_q();  // Q of this round was last used by this step's score MFMAs
        if (!FAST && first_pv) {
            a2_o_first<A2_OB + 0, 0>(pb0);
            a2_lds_frag<A2_KF + 0, 0>(kaddr2);
            a2_o_first<A2_OB + 16, 2>(pb0);
            keep(pa0, pa1);
            a2_lds_frag<A2_KF + 4, 1024>(kaddr2);
            a2_o_first<A2_OB + 32, 4>(pb0);
            a2_lds_frag<A2_KF + 8, 2048>(kaddr2);
            a2_o_first<A2_OB + 48, 6>(pb0);
            a2_lds_frag<A2_KF + 12, 3072>(kaddr2);
        } else {
            a2_o_acc<A2_OB + 0, 0>(pb0);
            a2_lds_frag<A2_KF + 0, 0>(kaddr2);
            a2_o_acc<A2_OB + 16, 2>(pb0);
            keep(pa0, pa1);
            a2_lds_frag<A2_KF + 4, 1024>(kaddr2);
            a2_o_acc<A2_OB + 32, 4>(pb0);
            a2_lds_frag<A2_KF + 8, 2048>(kaddr2);
            a2_o_acc<A2_OB + 48, 6>(pb0);
            a2_lds_frag<A2_KF + 12, 3072>(kaddr2);
        }
        a2_o_acc<A2_OB + 0, 1>(pb1);
        a2_lds_frag<A2_KF + 16, 4096>(kaddr2);
        a2_o_acc<A2_OB + 16, 3>(pb1);
        a2_lds_frag<A2_KF + 20, 5120>(kaddr2);
        a2_o_acc<A2_OB + 32, 5>(pb1);
        a2_lds_frag<A2_KF + 24, 6144>(kaddr2);
        a2_o_acc<A2_OB + 48, 7>(pb1);
        a2_lds_frag<A2_KF + 28, 7168>(kaddr2);
        // The score tile of the next step must stay allocated while the MFMAs that write it are in flight -- also when
        // nobody reads it afterwards (last step of the kernel): the compiler would hand its registers to this step's
        // temporaries, and the matrix core would then overwrite those (seen: the last tile of every sequence corrupted).
        asm volatile("" : : "v"(na), "v"(nb));
        A2_T(4);
        if (!FAST && nr1) {
            if (has1) open_round(na, nb);  // the next round's first scores: reference := row maximum
            A2_T(5);
            // ---- the round is complete: normalise and store its context fragments (invalid query slots: zeros)
            a2_nops24();  // the last O MFMAs must have retired before their accumulators are read
            const float ta = half_sum(la), tb = half_sum(lb);
            const bool va = 32 * qbA + (lane & 31) < T, vb = 32 * qbB + (lane & 31) < T;
            const float ia = va ? 1.0f / ta : 0.0f, ib = vb ? 1.0f / tb : 0.0f;
            auto store_block = [&](auto o0_tag, int qblk, float inv, int nbd) {
                f32x16 o;
                a2_acc_get16<decltype(o0_tag)::value>(o);
                o *= inv;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    stfrag(ctxf + (((size_t)b * QB + qblk) * 8 + 2 * nbd + jj) * FRAG_BYTES + lane * 16, pack_half(o, jj));
            };
            if (storeA) {
                store_block(std::integral_constant<int, A2_OA + 0>{}, qbA, ia, 0);
                store_block(std::integral_constant<int, A2_OA + 16>{}, qbA, ia, 1);
                store_block(std::integral_constant<int, A2_OA + 32>{}, qbA, ia, 2);
                store_block(std::integral_constant<int, A2_OA + 48>{}, qbA, ia, 3);
            }
            if (storeB) {
                store_block(std::integral_constant<int, A2_OB + 0>{}, qbB, ib, 0);
                store_block(std::integral_constant<int, A2_OB + 16>{}, qbB, ib, 1);
                store_block(std::integral_constant<int, A2_OB + 32>{}, qbB, ib, 2);
                store_block(std::integral_constant<int, A2_OB + 48>{}, qbB, ib, 3);
            }
            la = 0.0f;
            lb = 0.0f;
            if (r + 1 < NR) round_blocks(r + 1, qbA, qbB, storeA, storeB);
            A2_T(6);
        }
    };
    const std::true_type yes{};
    const std::false_type no{};
    for (int r = 0; r < NR; ++r) {
        int j = 0;
        step(no, yes, r, j++, s0a, s0b, s1a, s1b);                   // j = 0: first touch of the accumulators
#ifndef A2_NO_FAST
        for (; j + 1 <= NT - 3; j += 2) {                           // j odd, j+1 even, both in 1 .. NT-3
            step(yes, no, r, j, s1a, s1b, s0a, s0b);
            step(yes, yes, r, j + 1, s0a, s0b, s1a, s1b);
        }
#endif
        for (; j < NT; ++j) {                                      // the last steps of the round (Q switch, next round's first scores)
            if (j & 1)
                step(no, no, r, j, s1a, s1b, s0a, s0b);
            else
                step(no, yes, r, j, s0a, s0b, s1a, s1b);
        }
        if (NT & 1) {  // the next round's tile-0 scores were produced into the odd buffers: a round starts on the even ones
            s0a = s1a;
            s0b = s1b;
        }
    }
    a2_lgkm<0>();
#ifdef SAVAD_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i2 = 0; i2 < 8; ++i2) g_savad_dbg[i2] = tacc_[i2];
#endif
}

}  // namespace bf
}  // namespace savad
