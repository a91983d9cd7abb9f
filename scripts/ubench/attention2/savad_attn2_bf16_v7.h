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

// T > 32.  Workgroup = (sequence b, group g of NG): the group's query PAIRS [p0, p1) are processed in rounds of 4
// (one pair per wave).  The tile sequence of a wave is FLAT: n = round * NT + tile; the K / V^T stream of the
// sequence restarts with every round and runs through the ring without a seam (global stage = round * NST + stage).
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
    const int GS = NR * NST, NTOT = NR * NT;
    const char* kseq = kf + (size_t)b * QB * BLK_BYTES + (size_t)w * FRAG_BYTES;   // + this wave's KiB within a 4-KiB group
    const char* vtseq = vtf + (size_t)b * QB * BLK_BYTES + (size_t)w * FRAG_BYTES;
    const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned voff0 = lane16, voff1 = lane16 + 4096u, voff2 = lane16 + 8192u, voff3 = lane16 + 12288u;
    auto make_desc = [&](int gs) {  // stage gs = key blocks 2s, 2s+1 of the sequence (s = gs % NST)
        const int s = gs % NST;
        return A2Desc{kseq + (size_t)(2 * s) * BLK_BYTES, vtseq + (size_t)(2 * s) * BLK_BYTES,
                      ring0 + (unsigned)(gs & (A2_NRING - 1)) * A2_STAGE_BYTES + (unsigned)w * FRAG_BYTES};
    };
    auto issue_lo = [&](const A2Desc& d) {  // K half: KiB w, w+4, w+8, w+12 of the stage
        a2_dma_piece<0>(d.ksrc, d.ldsb, voff0);
        a2_dma_piece<4096>(d.ksrc, d.ldsb, voff1);
        a2_dma_piece<8192>(d.ksrc, d.ldsb, voff2);
        a2_dma_piece<12288>(d.ksrc, d.ldsb, voff3);
    };
    auto issue_hi = [&](const A2Desc& d) {  // V^T half: KiB 16+w, ...
        a2_dma_piece<16384>(d.vsrc, d.ldsb, voff0);
        a2_dma_piece<20480>(d.vsrc, d.ldsb, voff1);
        a2_dma_piece<24576>(d.vsrc, d.ldsb, voff2);
        a2_dma_piece<28672>(d.vsrc, d.ldsb, voff3);
    };
    int issued = 0;  // stages whose DMA has been issued or scheduled
    for (; issued < A2_NRING && issued < GS; ++issued) {
        const A2Desc d = make_desc(issued);
        issue_lo(d);
        issue_hi(d);
    }
#ifdef SAVAD_TIMING
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = __builtin_readcyclecounter(), tn_;
#endif

    // ---- per-round quantities
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
    load_q(qbA, qbB);
    a2_acc_put4<A2_QA + 0>(qn[0]); a2_acc_put4<A2_QA + 4>(qn[1]); a2_acc_put4<A2_QA + 8>(qn[2]); a2_acc_put4<A2_QA + 12>(qn[3]);
    a2_acc_put4<A2_QA + 16>(qn[4]); a2_acc_put4<A2_QA + 20>(qn[5]); a2_acc_put4<A2_QA + 24>(qn[6]); a2_acc_put4<A2_QA + 28>(qn[7]);
    a2_acc_put4<A2_QB + 0>(qn[8]); a2_acc_put4<A2_QB + 4>(qn[9]); a2_acc_put4<A2_QB + 8>(qn[10]); a2_acc_put4<A2_QB + 12>(qn[11]);
    a2_acc_put4<A2_QB + 16>(qn[12]); a2_acc_put4<A2_QB + 20>(qn[13]); a2_acc_put4<A2_QB + 24>(qn[14]); a2_acc_put4<A2_QB + 28>(qn[15]);

    f32x16 nega = zero16(), negb = zero16();  // -reference of the lane's query row, in every register
    float la = 0.0f, lb = 0.0f;               // this lane's half of the row sums
    f32x16 s0a, s0b, s1a, s1b;                // score tiles, ping-pong

    // ---- prologue: stage 0 has landed; K(0) -> scores of tile 0 -> reference; K(1) on its way
    int acquired = 0;
    a2_acquire(issued - 1 > 0);
    {
        const unsigned a0 = ring0 + lane16;
        a2_lds_frag<A2_KF + 0, 0>(a0); a2_lds_frag<A2_KF + 4, 1024>(a0); a2_lds_frag<A2_KF + 8, 2048>(a0); a2_lds_frag<A2_KF + 12, 3072>(a0);
        a2_lds_frag<A2_KF + 16, 4096>(a0); a2_lds_frag<A2_KF + 20, 5120>(a0); a2_lds_frag<A2_KF + 24, 6144>(a0); a2_lds_frag<A2_KF + 28, 7168>(a0);
        a2_lgkm<0>();
        a2_s_first0<0, A2_QA>(s0a); a2_s_first0<0, A2_QB>(s0b);
        a2_s_acc<1, A2_QA>(s0a); a2_s_acc<1, A2_QB>(s0b); a2_s_acc<2, A2_QA>(s0a); a2_s_acc<2, A2_QB>(s0b);
        a2_s_acc<3, A2_QA>(s0a); a2_s_acc<3, A2_QB>(s0b); a2_s_acc<4, A2_QA>(s0a); a2_s_acc<4, A2_QB>(s0b);
        a2_s_acc<5, A2_QA>(s0a); a2_s_acc<5, A2_QB>(s0b); a2_s_acc<6, A2_QA>(s0a); a2_s_acc<6, A2_QB>(s0b);
        a2_s_acc<7, A2_QA>(s0a); a2_s_acc<7, A2_QB>(s0b);
        a2_nops24();
        a2_move<A2_OA>(s0a, nega, la, half_max(a2_max16(s0a)), true);
        a2_move<A2_OB>(s0b, negb, lb, half_max(a2_max16(s0b)), true);
        const unsigned a1 = a0 + BLK_BYTES;  // K of tile 1 (T > 32: it exists)
        a2_lds_frag<A2_KF + 0, 0>(a1); a2_lds_frag<A2_KF + 4, 1024>(a1); a2_lds_frag<A2_KF + 8, 2048>(a1); a2_lds_frag<A2_KF + 12, 3072>(a1);
        a2_lds_frag<A2_KF + 16, 4096>(a1); a2_lds_frag<A2_KF + 20, 5120>(a1); a2_lds_frag<A2_KF + 24, 6144>(a1); a2_lds_frag<A2_KF + 28, 7168>(a1);
        s1a = s0a;
        s1b = s0b;
    }
    A2_T(0);

    // ---- flat tile loop state
    int r = 0, j = 0;
    // The stage scheduled at a hand-over is issued by wave w at the w-th of the DMA points that follow (two per phase):
    // the four bursts of eight DMA instructions then run one after the other instead of fighting for the CU's one
    // vector-memory port (all four at once: ~85 cycles per instruction and wave; alone: ~25).
    int mine_gs = -1, mine_wait = 0;
    auto issue_all = [&](int gs) {
        const A2Desc d = make_desc(gs);
        issue_lo(d);
        issue_hi(d);
    };
    auto dma_point = [&]() {
        if (mine_gs >= 0) {
            if (mine_wait == 0) {
                issue_all(mine_gs);
                mine_gs = -1;
            } else {
                --mine_wait;
            }
        }
    };

    // One step: scores of tile n = (r, j) are in (ca, cb); produces those of tile n+1 in (na, nb).
    // FAST (1 <= j <= NT-4): none of the per-round special cases applies -- they are compiled out.
    auto kill_q = [&]() {  // ends the live range of the Q staging registers (they are only needed between load and switch)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "=v"(qn[i]));
    };
    kill_q();
    auto step = [&](auto fast_tag, f32x16& ca, f32x16& cb, f32x16& na, f32x16& nb) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const int s = j >> 1, tt = j & 1;
        const bool nr1 = FAST ? false : j + 1 == NT;          // tile n+1 opens the next round (if it exists)
        const bool last = FAST ? false : (nr1 && r + 1 >= NR);
        int r2 = r, j2 = j + 2;                               // tile n+2
        if (!FAST && j2 >= NT) {
            j2 -= NT;
            r2 = r + 1;
        }
        const bool has2 = FAST ? true : r2 < NR;
        const int stage2 = has2 ? r2 * NST + (j2 >> 1) : acquired;
        const bool first_pv = FAST ? false : j == 0;
        const bool qload = FAST ? false : (r + 1 < NR && j == (NT >= 3 ? NT - 3 : 0));
        const bool qswitch = FAST ? false : (r + 1 < NR && j == NT - 2);
        const unsigned vaddr = ring0 + (unsigned)((r * NST + s) & (A2_NRING - 1)) * A2_STAGE_BYTES + (unsigned)(2 + tt) * BLK_BYTES + lane16;
        const unsigned kaddr2 = ring0 + (unsigned)(stage2 & (A2_NRING - 1)) * A2_STAGE_BYTES + (unsigned)(j2 & 1) * BLK_BYTES + lane16;
        if (qload) {
            int qa, qb;
            bool sa_, sb_;
            round_blocks(r + 1, qa, qb, sa_, sb_);
            load_q(qa, qb);
        }
        // ---- phase 1: the 16 score MFMAs of tile n+1 (K fragment ks lands just in time: counted waits); behind each
        //      of the first eight, four exponentials of tile n; behind the last eight, row sum + packing of block A and
        //      one V^T fragment read of tile n each
        float ea[16], eb[16];
        dma_point();
        a2_lgkm<0>();
        if (nr1) {  // new round: new Q (already switched), reference 0
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
#define A2_SLOT_A(i)                                                                          \
    ra = (i) == 0 ? a2_add(ea[0], ea[1]) : a2_add(a2_add(ra, ea[2 * (i)]), ea[2 * (i) + 1]); \
    ua[i] = a2_cvt2(ea[2 * (i)], ea[2 * (i) + 1]);
        // V^T fragments are read in the order phase 2 consumes them: 0, 2, 4, 6, 1, 3, 5, 7
        dma_point();
        a2_s_acc<4, A2_QA>(na);
        A2_SLOT_A(0)
        a2_lds_frag<A2_VF + 0, 0>(vaddr);
        a2_s_acc<4, A2_QB>(nb);
        A2_SLOT_A(1)
        a2_lds_frag<A2_VF + 8, 2048>(vaddr);
        a2_s_acc<5, A2_QA>(na);
        A2_SLOT_A(2)
        a2_lds_frag<A2_VF + 16, 4096>(vaddr);
        a2_s_acc<5, A2_QB>(nb);
        A2_SLOT_A(3)
        a2_lds_frag<A2_VF + 24, 6144>(vaddr);
        a2_s_acc<6, A2_QA>(na);
        A2_SLOT_A(4)
        a2_lds_frag<A2_VF + 4, 1024>(vaddr);
        a2_s_acc<6, A2_QB>(nb);
        A2_SLOT_A(5)
        a2_lds_frag<A2_VF + 12, 3072>(vaddr);
        a2_s_acc<7, A2_QA>(na);
        A2_SLOT_A(6)
        a2_lds_frag<A2_VF + 20, 5120>(vaddr);
        a2_s_acc<7, A2_QB>(nb);
        A2_SLOT_A(7)
        a2_lds_frag<A2_VF + 28, 7168>(vaddr);
#undef A2_SLOT_A
        la = a2_add(la, ra);
        const bf16x8 pa0 = a2_frag(ua[0], ua[1], ua[2], ua[3]), pa1 = a2_frag(ua[4], ua[5], ua[6], ua[7]);
        A2_T(2);
        // ---- stage hand-over: the stage of tile n+2 must have landed before its K fragments are read in phase 2.
        // After the barrier nobody needs the stage of tile n any more (its V^T fragments were read above), so the next
        // stage of the stream (three beyond tile n+1's) is scheduled: its K half rides in this step, its V^T half in the next.
        if (has2 && stage2 > acquired) {
            a2_acquire(issued - 1 > stage2);
            acquired = stage2;
            A2_T(1);
            const int stage1 = nr1 ? (r + 1) * NST : r * NST + ((j + 1) >> 1);
            if (issued < stage1 + A2_NRING && issued < GS) {
                if (mine_gs >= 0) issue_all(mine_gs);  // (a stage still pending from the previous hand-over: irregular spacing at a round seam)
                mine_gs = issued++;
                mine_wait = w;
            }
        }
        dma_point();
        // ---- phase 2: the 16 O MFMAs of tile n.  Block A's eight first, with row sum + packing of block B behind them;
        //      then block B's, with the row maxima of tile n+1 and one K fragment read of tile n+2 each.  DMA pieces
        //      behind MFMAs 1, 5, 9, 13; the next round's Q fragments (once per round) behind all of them.
#define A2_SLOT_B(i)                                                                          \
    rb = (i) == 0 ? a2_add(eb[0], eb[1]) : a2_add(a2_add(rb, eb[2 * (i)]), eb[2 * (i) + 1]); \
    ub[i] = a2_cvt2(eb[2 * (i)], eb[2 * (i) + 1]);
        a2_lgkm<0>();
        if (first_pv) {
            a2_o_first<A2_OA + 0, 0>(pa0);
            A2_SLOT_B(0)
            a2_o_first<A2_OA + 16, 2>(pa0);
            A2_SLOT_B(1)
            a2_o_first<A2_OA + 32, 4>(pa0);
            A2_SLOT_B(2)
            a2_o_first<A2_OA + 48, 6>(pa0);
            A2_SLOT_B(3)
        } else {
            a2_o_acc<A2_OA + 0, 0>(pa0);
            A2_SLOT_B(0)
            a2_o_acc<A2_OA + 16, 2>(pa0);
            A2_SLOT_B(1)
            a2_o_acc<A2_OA + 32, 4>(pa0);
            A2_SLOT_B(2)
            a2_o_acc<A2_OA + 48, 6>(pa0);
            A2_SLOT_B(3)
        }
        if (qswitch) {
            a2_acc_put4<A2_QA + 0>(qn[0]); a2_acc_put4<A2_QA + 4>(qn[1]); a2_acc_put4<A2_QA + 8>(qn[2]); a2_acc_put4<A2_QA + 12>(qn[3]);
            a2_acc_put4<A2_QA + 16>(qn[4]); a2_acc_put4<A2_QA + 20>(qn[5]); a2_acc_put4<A2_QA + 24>(qn[6]); a2_acc_put4<A2_QA + 28>(qn[7]);
            a2_acc_put4<A2_QB + 0>(qn[8]); a2_acc_put4<A2_QB + 4>(qn[9]); a2_acc_put4<A2_QB + 8>(qn[10]); a2_acc_put4<A2_QB + 12>(qn[11]);
            a2_acc_put4<A2_QB + 16>(qn[12]); a2_acc_put4<A2_QB + 20>(qn[13]); a2_acc_put4<A2_QB + 24>(qn[14]); a2_acc_put4<A2_QB + 28>(qn[15]);
            kill_q();
        }
        dma_point();
        a2_o_acc<A2_OA + 0, 1>(pa1);
        A2_SLOT_B(4)
        a2_o_acc<A2_OA + 16, 3>(pa1);
        A2_SLOT_B(5)
        a2_o_acc<A2_OA + 32, 5>(pa1);
        A2_SLOT_B(6)
        a2_lgkm<0>();
        a2_o_acc<A2_OA + 48, 7>(pa1);
        A2_SLOT_B(7)
#undef A2_SLOT_B
        lb = a2_add(lb, rb);
        const bf16x8 pb0 = a2_frag(ub[0], ub[1], ub[2], ub[3]), pb1 = a2_frag(ub[4], ub[5], ub[6], ub[7]);
        if (first_pv) {
            a2_o_first<A2_OB + 0, 0>(pb0);
            a2_lds_frag<A2_KF + 0, 0>(kaddr2);
            a2_o_first<A2_OB + 16, 2>(pb0);
            a2_lds_frag<A2_KF + 4, 1024>(kaddr2);
            a2_o_first<A2_OB + 32, 4>(pb0);
            a2_lds_frag<A2_KF + 8, 2048>(kaddr2);
            a2_o_first<A2_OB + 48, 6>(pb0);
            a2_lds_frag<A2_KF + 12, 3072>(kaddr2);
        } else {
            a2_o_acc<A2_OB + 0, 0>(pb0);
            a2_lds_frag<A2_KF + 0, 0>(kaddr2);
            a2_o_acc<A2_OB + 16, 2>(pb0);
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
        asm volatile("" : : "v"(na), "v"(nb), "v"(pb0), "v"(pb1));  // async MFMA operands / results stay allocated
        A2_T(4);
        if (!last) {
            // ONE rarely taken branch closes the step: first tile of a round (reference := row maximum), ragged last tile
            // of a round (mask), or a reference that has to move
            float mxa = 0.0f, mxb = 0.0f;
            if (nr1) {
                a2_nops24();
                mxa = half_max(a2_max16(na));
                mxb = half_max(a2_max16(nb));
            }
            const int j1 = nr1 ? 0 : j + 1;
            const bool ragged = 32 * j1 + 32 > T;  // wave-uniform
            if (nr1 || ragged) {
                if (ragged) {
                    const int lim = T - 32 * j1 - 4 * h;
                    a2_mask(na, lim);
                    a2_mask(nb, lim);
                    mxa = half_max(a2_max16(na));
                    mxb = half_max(a2_max16(nb));
                }
                if (nr1) {  // (O and l of the closing round are not touched: they are finalised below)
                    nega = zero16();
                    negb = zero16();
                    float dummy = 0.0f;
                    a2_move<A2_OA>(na, nega, dummy, mxa, true);
                    a2_move<A2_OB>(nb, negb, dummy, mxb, true);
                } else {
                    a2_move<A2_OA>(na, nega, la, mxa, false);
                    a2_move<A2_OB>(nb, negb, lb, mxb, false);
                }
            }
        }
        A2_T(5);
        if (nr1) {  // ---- the round is complete: normalise and store its context fragments (invalid query slots: zeros)
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
            j = 0;
            ++r;
            if (r < NR) round_blocks(r, qbA, qbB, storeA, storeB);
            A2_T(6);
        } else {
            ++j;
        }
    };
    // A round starts with the scores of its tile 0 in s0*; even tiles read s0* and write s1*, odd tiles the other way.
    const std::true_type yes{};
    const std::false_type no{};
    for (int rr = 0; rr < NR; ++rr) {
        step(no, s0a, s0b, s1a, s1b);                      // j = 0: first touch of the accumulators
        int left = NT - 1;
        while (j + 1 <= NT - 4) {                          // j odd, j + 1 even, both in 1 .. NT-4
            step(yes, s1a, s1b, s0a, s0b);
            step(yes, s0a, s0b, s1a, s1b);
            left -= 2;
        }
        // the last (up to five) steps of the round, straight-line: Q load / switch, the next round's first scores
        if (left-- > 0) step(no, s1a, s1b, s0a, s0b);
        if (left-- > 0) step(no, s0a, s0b, s1a, s1b);
        if (left-- > 0) step(no, s1a, s1b, s0a, s0b);
        if (left-- > 0) step(no, s0a, s0b, s1a, s1b);
        if (left-- > 0) step(no, s1a, s1b, s0a, s0b);
        if (NT & 1) {  // the next round's tile-0 scores were produced into the odd buffers
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
