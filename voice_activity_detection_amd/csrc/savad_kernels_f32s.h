// savad_kernels_f32s.h -- "fp32s": the fp32-parity forward pass on the bf16 matrix pipe.
//
// gfx950 has no TF32; its exact-fp32 MFMA runs at 1/16 of the bf16 rate (157 TF against 2.5 PF).  Here every GEMM operand
// is kept as THREE bf16 images
//
//      a = hi + mid + lo,   hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid)      (exact: 3 x 8 significand bits)
//
// and every product a.b is evaluated as SIX v_mfma_f32_32x32x16_bf16 into one fp32 accumulator
//
//      hi.lo + lo.hi + mid.mid + hi.mid + mid.hi + hi.hi                 (the three dropped terms are below 2^-32 |a||b|)
//
// bf16 x bf16 products are exact in fp32 and the accumulation is fp32, so the result carries fp32's own rounding noise
// (measured on MI355X against fp64, scripts/ubench/split6_probe.hip: 0.9 - 6.4e-7 of sum|ab| for K = 128 .. 800, the
// exact-fp32 MFMA 1.1 - 9.1e-7 on the same data; three products instead of six: 10x worse) at 6/16 of the fp32 pipe's time.
// LayerNorm, softmax, residual stream, biases: fp32, exactly as in savad_kernels.h.  Nothing is rounded to bf16 anywhere:
// what is stored between kernels are fp32 values (the residual stream) or their exact three-piece images (Q, K, V^T).
//
// Skeleton = savad_kernels_bf16.h (fragment-major buffers, transposed-form GEMMs whose accumulator registers ARE the next
// GEMM's operand, weights through an LDS ring fed by global->LDS DMA), with a "fragment" replaced by a TRIPLE of fragments:
//
//   triple    = [piece 3 (hi, mid, lo)][lane 64][8 bf16] = 3 KiB: one K-step (16 features) of 32 rows
//   q, k      : [block][ks 8][piece 3][lane][8]                      (24 KiB per 32-row block)
//   vt        : [block][nbd 4][j 2][piece 3][lane][8]                (V^T: lane = feature, 8 keys)
//   h         : [block][nb 4][g 4][lane 64][4 f32]                   (residual stream, fp32, 16 KiB per block)
//   weights   : [n-block][ks][piece 3][lane][8], split once by pack_weight_frags3_kernel
//   ring      : 3 slots of 48 KiB = two n-blocks of a weight matrix (K = 128), or the K and V^T images of ONE key block;
//               the DMA runs two slots ahead; one 4-wave workgroup per CU (one wave per SIMD, up to 512 VGPRs).
//
// Reference being restated: vad/models/self_attention.py:23-28, vad/modeling/transformer.py:24-61,227-238,258-363,366-382.
#pragma once
#include "savad_kernels_bf16.h"
#include <type_traits>

namespace savad {
#ifdef SAVAD_TIMING
__device__ long long g_savad_wg[1024][4];   // per workgroup of the fp32s fused launch: s_memtime at start / end, s_memrealtime at start / end
#endif
namespace fs {

using bf::bf16x8;
using bf::u32x4;
using bf::ldfrag;
using bf::stfrag;
using bf::FRAG_BYTES;
using bf::slot_row;
using bf::AttnState;
using bf::attn_state_init;
using bf::online_softmax_shifted;

constexpr int TFRAG_BYTES = 3 * FRAG_BYTES;   // one triple
constexpr int BLK3_BYTES = 8 * TFRAG_BYTES;   // 32 rows x 128 features as triples: 24 KiB
constexpr int SLOT_BYTES = 2 * BLK3_BYTES;    // a ring slot: 48 KiB
constexpr int NRING3 = 3;
constexpr int HBLK_BYTES = 32 * D * 4;        // a residual block (fp32)
constexpr int ROW_LDS_BYTES = NRING3 * SLOT_BYTES + 9 * D * 4;

struct Tri {
    bf16x8 h, m, l;
};

// a = h + m + l exactly (round-to-nearest pieces: every residual is exactly representable in fp32)
__device__ __forceinline__ void split1(float a, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)a;
    if (SAVAD_ABLATE & 128) {  // experiment builds: no residual arithmetic
        m = l = h;
        return;
    }
    const float r1 = a - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}
__device__ __forceinline__ Tri split8(const float (&v)[8]) {
    Tri t;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 h, m, l;
        split1(v[e], h, m, l);
        t.h[e] = h;
        t.m[e] = m;
        t.l[e] = l;
    }
    return t;
}
// registers 8j..8j+7 of a C-layout 32-feature block -> the triple of K-step j of that block
__device__ __forceinline__ Tri split_half(const f32x16& v, int j) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = v[8 * j + e];
    return split8(t);
}
// LayerNorm'ed row (xg[G][s] = feature 8G + 4h + s) -> the 8 K-step triples
__device__ __forceinline__ void split_row(const f32x4 (&xg)[16], Tri (&xp)[8]) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = xg[2 * ks + (e >> 2)][e & 3];
        xp[ks] = split8(t);
    }
}
__device__ __forceinline__ Tri ldtri(const char* p /* lane's 16 bytes of the hi piece */) {
    return Tri{ldfrag(p), ldfrag(p + FRAG_BYTES), ldfrag(p + 2 * FRAG_BYTES)};
}
__device__ __forceinline__ void sttri(char* p, const Tri& t) {
    stfrag(p, t.h);
    stfrag(p + FRAG_BYTES, t.m);
    stfrag(p + 2 * FRAG_BYTES, t.l);
}
__device__ __forceinline__ Tri zero_tri() {
    const bf16x8 z = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
    return Tri{z, z, z};
}

#define SAVAD_MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// acc0 += A0 . B, acc1 += A1 . B: six products each, the two accumulators taking turns (a dependent MFMA issued straight behind
// its producer waits ~6 cycles: 43 against 37.6 cycles per MFMA with one wave per SIMD, scripts/ubench/split6_probe.hip).
// SWAP: operands exchanged (acc = B . A^T form: the V^T projection)
template <bool SWAP>
__device__ __forceinline__ void mfma6x2_lo(f32x16& acc0, f32x16& acc1, const Tri& a0, const Tri& a1, const Tri& b) {
#define SAVAD_MF2(pa, pb)                                                           \
    acc0 = SWAP ? SAVAD_MF(b.pb, a0.pa, acc0) : SAVAD_MF(a0.pa, b.pb, acc0);        \
    acc1 = SWAP ? SAVAD_MF(b.pb, a1.pa, acc1) : SAVAD_MF(a1.pa, b.pb, acc1);
    SAVAD_MF2(h, l) SAVAD_MF2(l, h) SAVAD_MF2(m, m)
}
template <bool SWAP>
__device__ __forceinline__ void mfma6x2_hi(f32x16& acc0, f32x16& acc1, const Tri& a0, const Tri& a1, const Tri& b) {
    SAVAD_MF2(h, m) SAVAD_MF2(m, h) SAVAD_MF2(h, h)
#undef SAVAD_MF2
}
template <bool SWAP>
__device__ __forceinline__ void mfma6x2(f32x16& acc0, f32x16& acc1, const Tri& a0, const Tri& a1, const Tri& b) {
    mfma6x2_lo<SWAP>(acc0, acc1, a0, a1, b);
    mfma6x2_hi<SWAP>(acc0, acc1, a0, a1, b);
}

// ---- residual stream blocks (fp32, fragment-major: every access a contiguous 1 KiB wave access)
__device__ __forceinline__ void load_hblock32(f32x16 (&x)[4], const float* hb, int lane) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = ld4(hb + ((nb * 4 + g) * 64 + lane) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) x[nb][4 * g + s] += t[s];
        }
}
__device__ __forceinline__ void store_hblock32(float* hb, const f32x16 (&x)[4], int lane) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 t;
#pragma unroll
            for (int s = 0; s < 4; ++s) t[s] = x[nb][4 * g + s];
            st4(hb + ((nb * 4 + g) * 64 + lane) * 4, t);
        }
}

// ---- the ring: 3 slots of 48 KiB; a slot = two 24-KiB segments, each contiguous in global memory.  Wave w moves 12 KiB of a slot:
// half (w & 1) of segment (w >> 1), as twelve 1-KiB DMA instructions (global_load_lds_dwordx4).  The stream runs TWO slots ahead:
//
//      step t:   s_waitcnt vmcnt(12)   the newest slot's twelve pieces may stay in flight: slot t has landed (in-order retirement;
//                                      vector stores / loads the wave issued since only make the wait longer, never shorter)
//                s_barrier             slot t is published -- and every wave is done with slot t - 1, whose place
//                DMA of slot t + 2     takes the pieces, ONE AT A TIME between the MFMAs of step t: issued back to back a DMA
//                                      instruction stalls the wave for 60 - 180 cycles with the matrix pipe idle behind it
//                                      (scripts/ubench/f32s_ablate.sh: 53 of 400 us per [32,800,80] forward before the interleave).
//
// Every wait is the same unconditional instruction (no run-time choice of the count: scripts/check_async_loads.py follows it
// statically); the two waits at the END of a stream, where nothing younger is in flight, are vmcnt(0) at compile-time positions.
#ifndef SAVAD_PIECE_FENCE
#define SAVAD_PIECE_FENCE 0x6   // __builtin_amdgcn_sched_barrier mask around a DMA piece: only VALU / SALU may move across it
#endif
struct DmaJob {
    const char* src;  // wave-uniform: this wave's 12 KiB of the slot in global memory
    unsigned ldsb;    // ... and their place in LDS (byte address)
};
struct Ring3 {
    static constexpr int PER = 12;
    char* base;
    int w, lane;
    unsigned voff[3];  // lane * 16 + g * 4096

    __device__ __forceinline__ Ring3(char* smem, int w_, int lane_) : base(smem), w(w_), lane(lane_) {
#pragma unroll
        for (int g = 0; g < 3; ++g) voff[g] = (unsigned)lane_ * 16u + 4096u * g;
    }
    __device__ __forceinline__ char* slot(int t) const { return base + (t % NRING3) * SLOT_BYTES; }
    // this wave's share of the slot whose two segments start at seg0 / seg1 (wave-uniform), landing in ring slot `t`
    __device__ __forceinline__ DmaJob job(int t, const char* seg0, const char* seg1) const {
        const unsigned slot0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)slot(t));
        return DmaJob{((w >> 1) ? seg1 : seg0) + (size_t)(w & 1) * (BLK3_BYTES / 2), slot0 + (unsigned)w * (BLK3_BYTES / 2)};
    }
    // piece I of the job's twelve: the instruction's immediate offset moves the global source and the LDS destination together
    template <int I>
    __device__ __forceinline__ void piece(const DmaJob& j) const {
        if (SAVAD_ABLATE & 1) return;
        __builtin_amdgcn_sched_barrier(SAVAD_PIECE_FENCE);   // the piece stays where the source puts it: ONE between two groups of MFMAs
        piece_raw<I>(j);
        __builtin_amdgcn_sched_barrier(SAVAD_PIECE_FENCE);
    }
    template <int I>
    __device__ __forceinline__ void piece_raw(const DmaJob& j) const {
        asm volatile(
            "s_add_u32 m0, %2, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:%4"
            :
            : "v"(voff[I / 4]), "s"(j.src), "s"(j.ldsb), "n"((I / 4) * 4096), "n"((I % 4) * 1024)
            : "memory", "scc");
    }
    __device__ __forceinline__ void issue_all(const DmaJob& j) const {  // prologues: nothing to hide the pieces behind yet
        if (SAVAD_ABLATE & 1) return;
        piece_raw<0>(j); piece_raw<1>(j); piece_raw<2>(j); piece_raw<3>(j); piece_raw<4>(j); piece_raw<5>(j);
        piece_raw<6>(j); piece_raw<7>(j); piece_raw<8>(j); piece_raw<9>(j); piece_raw<10>(j); piece_raw<11>(j);
    }
    // slot t has landed for every wave.  NEWER = slots issued after slot t (1 in the stream, 2 behind a three-slot prologue, 0 for the
    // stream's last slot)
    template <int NEWER>
    __device__ __forceinline__ void acquire() const {
        if (SAVAD_ABLATE & 2) return;
        __builtin_amdgcn_s_waitcnt(bf::Ring<4>::vmcnt_imm(NEWER * PER));
        asm volatile("" ::: "memory");
        __syncthreads();
    }
};

// acc0 / acc1 += W[the slot's two n-blocks] . x   (transposed form: lane = data row, registers = output features; SWAP: the V^T form).
// The sixteen weight triples are read from LDS by HAND-issued ds_read_b128, one K-step (six fragments) ahead of the twelve MFMAs that
// consume them, with counted lgkmcnt waits (the compiler keeps two fragments in flight and waits out an LDS round trip every few
// MFMAs; LDS data returns in order and scalar-memory returns can only add to what a counted wait has seen complete, so "at most six
// outstanding" means the older six have landed).  DMA: the twelve pieces of `job` go out between the K-steps.
template <bool SWAP, bool DMA>
__device__ __forceinline__ void gemm_slot(f32x16& acc0, f32x16& acc1, const char* slot, const Tri (&xp)[8], const Ring3& ring,
                                          const DmaJob& job) {
    if (SAVAD_ABLATE & 8) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) mfma6x2<SWAP>(acc0, acc1, xp[7 - ks], xp[ks ^ 1], xp[ks]);
        return;
    }
    const unsigned a = (unsigned)(size_t)slot + (unsigned)ring.lane * 16u;  // LDS byte address (low half of the flat address)
    u32x4 f[2][6];
#define SAVAD_G_LD(ks, i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[(ks) & 1][i]) : "v"(a), "n"(((i) / 3) * BLK3_BYTES + (ks) * TFRAG_BYTES + ((i) % 3) * FRAG_BYTES))
#define SAVAD_G_LD6(ks) SAVAD_G_LD(ks, 0); SAVAD_G_LD(ks, 1); SAVAD_G_LD(ks, 2); SAVAD_G_LD(ks, 3); SAVAD_G_LD(ks, 4); SAVAD_G_LD(ks, 5)
#define SAVAD_G_STEP(ks, P0, P1)                                                                                              \
    {                                                                                                                         \
        if constexpr ((ks) + 1 < 8) { SAVAD_G_LD6((ks) + 1); }                                                                \
        asm volatile("s_waitcnt lgkmcnt(%6)"                                                                                  \
                     : "+v"(f[(ks) & 1][0]), "+v"(f[(ks) & 1][1]), "+v"(f[(ks) & 1][2]), "+v"(f[(ks) & 1][3]), "+v"(f[(ks) & 1][4]), \
                       "+v"(f[(ks) & 1][5])                                                                                   \
                     : "n"((ks) + 1 < 8 ? 6 : 0));                                                                            \
        const Tri w0_{__builtin_bit_cast(bf16x8, f[(ks) & 1][0]), __builtin_bit_cast(bf16x8, f[(ks) & 1][1]),                  \
                      __builtin_bit_cast(bf16x8, f[(ks) & 1][2])};                                                            \
        const Tri w1_{__builtin_bit_cast(bf16x8, f[(ks) & 1][3]), __builtin_bit_cast(bf16x8, f[(ks) & 1][4]),                  \
                      __builtin_bit_cast(bf16x8, f[(ks) & 1][5])};                                                            \
        mfma6x2_lo<SWAP>(acc0, acc1, w0_, w1_, xp[ks]);                                                                       \
        if constexpr (DMA && (P1) >= 0) ring.template piece<((P1) >= 0 ? (P1) : 0)>(job);                                     \
        mfma6x2_hi<SWAP>(acc0, acc1, w0_, w1_, xp[ks]);                                                                       \
        if constexpr (DMA) ring.template piece<P0>(job);                                                                      \
    }
    SAVAD_G_LD6(0);
    // (P0 behind the K-step's twelve MFMAs, P1 -- every other K-step -- in their middle: one piece per six or twelve MFMAs)
    SAVAD_G_STEP(0, 1, 0) SAVAD_G_STEP(1, 2, -1) SAVAD_G_STEP(2, 4, 3) SAVAD_G_STEP(3, 5, -1)
    SAVAD_G_STEP(4, 7, 6) SAVAD_G_STEP(5, 8, -1) SAVAD_G_STEP(6, 10, 9) SAVAD_G_STEP(7, 11, -1)
#undef SAVAD_G_STEP
#undef SAVAD_G_LD6
#undef SAVAD_G_LD
}

// ---------------------------------------------------------------------------------------------
// Weight packing: fp32 [N][K] (LayerNorm already folded) -> triples [N/32][K/16][3][64][8]
// ---------------------------------------------------------------------------------------------
__global__ void pack_weight_frags3_kernel(const float* __restrict__ W, int N, int K, __bf16* __restrict__ out) {
    const int KS = K / 16;
    const size_t total = (size_t)(N / 32) * KS * 64 * 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const size_t f = i >> 9;  // (n-block, ks)
        const int ks = (int)(f % KS);
        const int nblk = (int)(f / KS);
        const int n = lane & 31, h = lane >> 5;
        const int k = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (e >> 2) + 4 * h + (e & 3);
        __bf16 ph, pm, pl;
        split1(W[(size_t)(32 * nblk + n) * K + k], ph, pm, pl);
        const size_t o = f * 3 * 512 + (size_t)lane * 8 + e;
        out[o] = ph;
        out[o + 512] = pm;
        out[o + 1024] = pl;
    }
}

// One of the six QKV slots: slot S covers n-blocks 2 (S & 1), 2 (S & 1) + 1 of projection rb = S >> 1 (0 query, 1 key:
// transposed form; 2 value: swapped form -> V^T).  Q is stored PRE-SCALED by qscale = log2(e) / sqrt(D).
template <int S, bool DMA>
__device__ __forceinline__ void qkv_slot(const char* slot, const Tri (&xp)[8], const float* lbq, char* __restrict__ qf, char* __restrict__ kf,
                                         char* __restrict__ vtf, int blk, const Ring3& ring, const DmaJob& job, float qscale, bool live) {
    const int lane = ring.lane, n = lane & 31, h = lane >> 5;
    constexpr int rb = S >> 1, nb0 = 2 * (S & 1);
    f32x16 acc[2];
    if constexpr (rb < 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = bias_block(lbq + D * rb + 32 * (nb0 + i), h);
        gemm_slot<false, DMA>(acc[0], acc[1], slot, xp, ring, job);
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float bv = lbq[2 * D + 32 * (nb0 + i) + n];  // lane = output feature
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = bv;
        }
        gemm_slot<true, DMA>(acc[0], acc[1], slot, xp, ring, job);
    }
    if constexpr (rb == 0) {
        acc[0] *= qscale;
        acc[1] *= qscale;
    }
    if (!live) return;
    char* dst = rb == 0 ? qf : (rb == 1 ? kf : vtf);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            sttri(dst + (size_t)blk * BLK3_BYTES + (2 * (nb0 + i) + j) * TFRAG_BYTES + lane * 16, split_half(acc[i], j));
}

// features f0..f0+3 and f0+8..f0+11 of one input row -> the triple of an input K-step
// (non-temporal: the features are read once; the 3.5 MB of weight triples every workgroup streams fill an XCD's 4 MB L2 almost alone,
// and a feature stream allocated beside them kept evicting them -- [65536,7,80]: 558 MB fetched per launch against 151 MB algorithmic)
__device__ __forceinline__ Tri load_x_tri(const float* p, bool valid) {
    f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 8));
    if (!valid) a = b = f32x4{0.f, 0.f, 0.f, 0.f};
    float t[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        t[e] = a[e];
        t[4 + e] = b[e];
    }
    return split8(t);
}

// h0[nb] += Win[n-block nb] . x^T for one 32-row block: the input Linear (vad/models/self_attention.py:12-16) of the wave-per-block kernels.
// Every feature piece of the row is requested up front and the four weight triples of a K-step one K-step ahead of their MFMAs: read
// where they are used (round 6's first version) each of the F / 16 K-steps paid its own L2 round trip before a single MFMA could issue
// -- [32,800,80] input stage 38 -> 34 us.  (What bounds the stage now is not this GEMM: it moves 800 KiB per workgroup -- ring 288, input
// weights 60 per wave-copy, positional rows 64, features 40, 352 of Q / K / V^T triples and residual written -- 160 MB per launch in
// 27 us = 5.9 TB/s, the rate a copy kernel reaches here.  One copy of the input weights per workgroup through LDS instead of one per
// wave, and all of a wave's weight triples requested up front, both measured: 34.2 / 34.1 us, no change; scripts/ubench/phase_timing_input_f32s.py.)
__device__ __forceinline__ void input_gemm_f32s(f32x16 (&h0)[4], const float* xr, bool valid, const char* win, int KS, int lane, int h) {
    constexpr int KSMAX = 8;   // feature pieces requested up front (F <= 128; the tail loop takes what is beyond)
    f32x4 xa[KSMAX], xc[KSMAX];
#pragma unroll
    for (int ks = 0; ks < KSMAX; ++ks) {
        const int kc = ks < KS ? ks : KS - 1;
        const float* px = xr + 32 * (kc >> 1) + 16 * (kc & 1) + 4 * h;
        xa[ks] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(px));
        xc[ks] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(px + 8));
    }
    auto wtri = [&](int nb, int ks) { return ldtri(win + (size_t)(nb * KS + (ks < KS ? ks : KS - 1)) * TFRAG_BYTES + lane * 16); };
    Tri wn[4] = {wtri(0, 0), wtri(1, 0), wtri(2, 0), wtri(3, 0)};
#pragma unroll
    for (int ks = 0; ks < KSMAX; ++ks)
        if (ks < KS) {
            const Tri w0 = wn[0], w1 = wn[1], w2 = wn[2], w3 = wn[3];
            if (ks + 1 < KSMAX) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) wn[nb] = wtri(nb, ks + 1);
            }
            float t[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t[e] = valid ? xa[ks][e] : 0.0f;
                t[4 + e] = valid ? xc[ks][e] : 0.0f;
            }
            const Tri xf = split8(t);
            mfma6x2<false>(h0[0], h0[1], w0, w1, xf);
            mfma6x2<false>(h0[2], h0[3], w2, w3, xf);
        }
    for (int ks = KSMAX; ks < KS; ++ks) {
        const int f0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
        const Tri xf = load_x_tri(xr + f0, valid);
#pragma unroll
        for (int nb = 0; nb < 4; nb += 2)
            mfma6x2<false>(h0[nb], h0[nb + 1], ldtri(win + (size_t)(nb * KS + ks) * TFRAG_BYTES + lane * 16),
                           ldtri(win + (size_t)((nb + 1) * KS + ks) * TFRAG_BYTES + lane * 16), xf);
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel 1 (fp32s): input Linear + PE -> h -> LN -> Q, K, V^T triples.  4 waves = 4 blocks.
// (vad/models/self_attention.py:12-16,24; vad/modeling/transformer.py:281-284,392-401)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void input_qkv_kernel_f32s(const float* __restrict__ x, long xbs, int B, int T, int F, int nblk,
                                                                const char* __restrict__ win_frag, const float* __restrict__ bin,
                                                                const float* __restrict__ pe, const char* __restrict__ wqkv_frag,
                                                                const float* __restrict__ bqkv, float* __restrict__ hbuf,
                                                                char* __restrict__ qf, char* __restrict__ kf, char* __restrict__ vtf,
                                                                float qscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lbq = reinterpret_cast<float*>(smem + NRING3 * SLOT_BYTES);
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * 4 + w;
    const Ring3 ring(smem, w, lane);
    auto job = [&](int t) { return ring.job(t, wqkv_frag + (size_t)(2 * t) * BLK3_BYTES, wqkv_frag + (size_t)(2 * t + 1) * BLK3_BYTES); };
    SAVAD_STAMP(57);
    ring.issue_all(job(0));
    ring.issue_all(job(1));
    stage_bias(lbq, bqkv, 3 * D);
    SAVAD_STAMP(58);
    size_t row;
    int t_frame;
    const bool valid = (blk < nblk) && slot_row(B, T, blk, m, row, t_frame);
    if (!valid) {
        row = 0;
        t_frame = 0;
    }
    f32x16 h0[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        h0[nb] = zero16();
        add_bias(h0[nb], bin + 32 * nb, h);
        add_block(h0[nb], pe + (size_t)t_frame * D + 32 * nb, h);
    }
    input_gemm_f32s(h0, x + x_row_offset(row, T, F, xbs), valid, win_frag, F / 16, lane, h);
    SAVAD_STAMP(59);
    const bool live = blk < nblk;
    if (live) store_hblock32(hbuf + (size_t)blk * (32 * D), h0, lane);
    f32x4 xg[16];
    layernorm_regs(h0, xg);
    Tri xp[8];
    split_row(xg, xp);
    SAVAD_STAMP(60);
    const DmaJob none{nullptr, 0u};
#define SAVAD_IQ_STEP(T_)                                                                                                      \
    ring.acquire<((T_) + 1 < 6) ? 1 : 0>();                                                                                    \
    qkv_slot<T_, ((T_) + 2 < 6)>(ring.slot(T_), xp, lbq, qf, kf, vtf, blk, ring, (T_) + 2 < 6 ? job((T_) + 2) : none, qscale, live);
    SAVAD_IQ_STEP(0) SAVAD_IQ_STEP(1)
    SAVAD_STAMP(61);
    SAVAD_IQ_STEP(2) SAVAD_IQ_STEP(3)
    SAVAD_STAMP(62);
    SAVAD_IQ_STEP(4) SAVAD_IQ_STEP(5)
    SAVAD_STAMP(63);
#undef SAVAD_IQ_STEP
}

// ---------------------------------------------------------------------------------------------
// Attention on triples (vad/modeling/transformer.py:305-346,351-363), SOFTWARE-PIPELINED over the key tiles: with one wave per SIMD
// nothing else covers the ~190 VALU instructions a tile's softmax and the split of its probabilities cost, so step j of a wave runs
//
//      region A:   S(j+1)^T = K(j+1) Q^T   48 MFMAs on two accumulators (the standing reference -negm rides in as C of one of them)
//                  || exponentials, row sums and the three-piece split of tile j's probabilities (VALU, two scores per K-step)
//      region B:   O^T += V(j)^T P(j)^T    48 MFMAs on the four context accumulators
//                  || row maxima of S(j+1) against the reference, the DMA pieces of ring slot j + 2
//      (rare)      a row maximum left the 2^16 window: move the reference -- rescale O and l, shift S(j+1)
//
// in ONE ring slot: slot j of the stream holds V(j)^T and K(j+1) (K(0) travels in a slot of its own in front).  LDS: K triples are
// read by hand two K-steps ahead of their MFMAs, V^T triples one group (two triples) ahead, the first group before region A ends.
// The arithmetic is attn_tile's of savad_kernels_bf16.h, value for value (the same reference rule, the same order of sums).
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> their three bf16 pieces, packed pairwise (the dwords of a fragment)
__device__ __forceinline__ void split_pair(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    __bf16 h0, m0, l0, h1, m1, l1;
    split1(a0, h0, m0, l0);
    split1(a1, h1, m1, l1);
    h = __builtin_bit_cast(unsigned, bf16x2{h0, h1});
    m = __builtin_bit_cast(unsigned, bf16x2{m0, m1});
    l = __builtin_bit_cast(unsigned, bf16x2{l0, l1});
}
// row maxima of a fresh score tile against the standing reference; the reference moves when a maximum drifts more than
// 2^RESCALE_LOG2 above it (first tile: away from 0 in either direction) -- online_softmax_shifted's first half
__device__ __forceinline__ void settle_reference(f32x16& sc, AttnState& st, bool first /* wave-uniform */) {
    float mx = sc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
    mx = half_max(mx);
    const bool move = (mx > RESCALE_LOG2) || (first && mx < -RESCALE_LOG2);
    if (__any(move)) {
        const float d = move ? mx : 0.0f;  // new reference = old + d: the row maximum becomes 0
        if (!first) {                      // on the first tile O and l are still zero (and 2^-d may overflow)
            const float alpha = __builtin_amdgcn_exp2f(-d);
            st.l_run *= alpha;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) st.O[nb] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] -= d;
            st.negm[r] -= d;
        }
    }
}
// scores of key tile 0 (prologue: nothing to overlap with yet); kblk = LDS address of the K triples
__device__ __forceinline__ f32x16 qk_tile3(const AttnState& st, const Tri (&qp)[8], const char* kblk, int lane) {
    f32x16 sa = st.negm, sb = zero16();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const Tri k = ldtri(kblk + ks * TFRAG_BYTES + lane * 16);
        sa = SAVAD_MF(k.h, qp[ks].l, sa);
        sb = SAVAD_MF(k.l, qp[ks].h, sb);
        sa = SAVAD_MF(k.m, qp[ks].m, sa);
        sb = SAVAD_MF(k.h, qp[ks].m, sb);
        sa = SAVAD_MF(k.m, qp[ks].h, sa);
        sb = SAVAD_MF(k.h, qp[ks].h, sb);
    }
    return sa + sb;
}
// One pipelined step on ring slot `slot` = [V(j)^T | K(j+1)].  sc: in = tile j's scores relative to the settled reference, out = tile
// j+1's (HAVE_NEXT), masked by `mask_next` and settled.  DMA: the pieces of `job` go out in region B.
template <bool HAVE_NEXT, bool DMA, class MaskNext>
__device__ __forceinline__ void attn_step3(AttnState& st, const Tri (&qp)[8], f32x16& sc, const char* slot, MaskNext mask_next,
                                           const Ring3& ring, const DmaJob& job) {
    const unsigned av = (unsigned)(size_t)slot + (unsigned)ring.lane * 16u, ak = av + BLK3_BYTES;
    f32x16 sa = st.negm, sb = zero16();
    unsigned ph[8], pm[8], pl[8];   // the dwords of p0 (0..3) and p1 (4..7), piece by piece
    float rs = 0.0f;
    u32x4 fk[3][3];
    u32x4 fv[2][6];
#define SAVAD_A_LDK(ks, i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fk[(ks) % 3][i]) : "v"(ak), "n"((ks) * TFRAG_BYTES + (i) * FRAG_BYTES))
#define SAVAD_A_LDK3(ks) SAVAD_A_LDK(ks, 0); SAVAD_A_LDK(ks, 1); SAVAD_A_LDK(ks, 2)
#define SAVAD_A_LDV(g, i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fv[(g) & 1][i]) : "v"(av), "n"((((2 * ((g) >> 1) + (i) / 3) * 2 + ((g) & 1)) * 3 + (i) % 3) * FRAG_BYTES))
#define SAVAD_A_LDV6(g) SAVAD_A_LDV(g, 0); SAVAD_A_LDV(g, 1); SAVAD_A_LDV(g, 2); SAVAD_A_LDV(g, 3); SAVAD_A_LDV(g, 4); SAVAD_A_LDV(g, 5)
    // the VALU slice of K-step ks: scores 2 ks, 2 ks + 1 of tile j -> probabilities, row sum, pieces
#define SAVAD_A_SOFT(ks)                                                                                                       \
    {                                                                                                                          \
        const float e0_ = (SAVAD_ABLATE & 4) ? sc[2 * (ks)] : __builtin_amdgcn_exp2f(sc[2 * (ks)]);                            \
        const float e1_ = (SAVAD_ABLATE & 4) ? sc[2 * (ks) + 1] : __builtin_amdgcn_exp2f(sc[2 * (ks) + 1]);                    \
        rs += e0_;                                                                                                             \
        rs += e1_;                                                                                                             \
        split_pair(e0_, e1_, ph[ks], pm[ks], pl[ks]);                                                                          \
    }
#define SAVAD_A_QK(ks)                                                                                                         \
    {                                                                                                                          \
        if constexpr ((ks) + 2 < 8) { SAVAD_A_LDK3((ks) + 2); }                                                                \
        if constexpr ((ks) == 6) { SAVAD_A_LDV6(0); }                                                                          \
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fk[(ks) % 3][0]), "+v"(fk[(ks) % 3][1]), "+v"(fk[(ks) % 3][2])             \
                     : "n"((ks) + 2 < 8 ? 6 : ((ks) == 6 ? 9 : 6)));                                                           \
        const bf16x8 kh_ = __builtin_bit_cast(bf16x8, fk[(ks) % 3][0]), km_ = __builtin_bit_cast(bf16x8, fk[(ks) % 3][1]),     \
                     kl_ = __builtin_bit_cast(bf16x8, fk[(ks) % 3][2]);                                                        \
        sa = SAVAD_MF(kh_, qp[ks].l, sa);                                                                                      \
        sb = SAVAD_MF(kl_, qp[ks].h, sb);                                                                                      \
        sa = SAVAD_MF(km_, qp[ks].m, sa);                                                                                      \
        sb = SAVAD_MF(kh_, qp[ks].m, sb);                                                                                      \
        sa = SAVAD_MF(km_, qp[ks].h, sa);                                                                                      \
        sb = SAVAD_MF(kh_, qp[ks].h, sb);                                                                                      \
        SAVAD_A_SOFT(ks)                                                                                                       \
    }
    // ---- region A
    if constexpr (HAVE_NEXT) {
        SAVAD_A_LDK3(0);
        SAVAD_A_LDK3(1);
        // waits: K-steps 0..5 leave the two younger K-steps' six reads in flight; K-step 6 the last K-step's three and the six of
        // V group 0 (requested just before it); K-step 7 those six
        SAVAD_A_QK(0) SAVAD_A_QK(1) SAVAD_A_QK(2) SAVAD_A_QK(3) SAVAD_A_QK(4) SAVAD_A_QK(5) SAVAD_A_QK(6) SAVAD_A_QK(7)
    } else {
        SAVAD_A_LDV6(0);
        SAVAD_A_SOFT(0) SAVAD_A_SOFT(1) SAVAD_A_SOFT(2) SAVAD_A_SOFT(3) SAVAD_A_SOFT(4) SAVAD_A_SOFT(5) SAVAD_A_SOFT(6) SAVAD_A_SOFT(7)
    }
#undef SAVAD_A_QK
#undef SAVAD_A_SOFT
#undef SAVAD_A_LDK3
#undef SAVAD_A_LDK
    st.l_run += rs;  // this lane's half of the row sum: the two halves meet once, when the context is normalised
    const Tri p0{__builtin_bit_cast(bf16x8, u32x4{ph[0], ph[1], ph[2], ph[3]}), __builtin_bit_cast(bf16x8, u32x4{pm[0], pm[1], pm[2], pm[3]}),
                 __builtin_bit_cast(bf16x8, u32x4{pl[0], pl[1], pl[2], pl[3]})};
    const Tri p1{__builtin_bit_cast(bf16x8, u32x4{ph[4], ph[5], ph[6], ph[7]}), __builtin_bit_cast(bf16x8, u32x4{pm[4], pm[5], pm[6], pm[7]}),
                 __builtin_bit_cast(bf16x8, u32x4{pl[4], pl[5], pl[6], pl[7]})};
    // ---- region B: PV group g = the triples of (nbd, j) and (nbd + 1, j), nbd = 2 (g >> 1), j = g & 1
#define SAVAD_A_PV(g, PJ)                                                                                                      \
    {                                                                                                                          \
        if constexpr ((g) + 1 < 4) { SAVAD_A_LDV6((g) + 1); }                                                                  \
        asm volatile("s_waitcnt lgkmcnt(%6)"                                                                                   \
                     : "+v"(fv[(g) & 1][0]), "+v"(fv[(g) & 1][1]), "+v"(fv[(g) & 1][2]), "+v"(fv[(g) & 1][3]), "+v"(fv[(g) & 1][4]), \
                       "+v"(fv[(g) & 1][5])                                                                                    \
                     : "n"((g) + 1 < 4 ? 6 : 0));                                                                              \
        const Tri v0_{__builtin_bit_cast(bf16x8, fv[(g) & 1][0]), __builtin_bit_cast(bf16x8, fv[(g) & 1][1]),                   \
                      __builtin_bit_cast(bf16x8, fv[(g) & 1][2])};                                                             \
        const Tri v1_{__builtin_bit_cast(bf16x8, fv[(g) & 1][3]), __builtin_bit_cast(bf16x8, fv[(g) & 1][4]),                   \
                      __builtin_bit_cast(bf16x8, fv[(g) & 1][5])};                                                             \
        mfma6x2<false>(st.O[2 * ((g) >> 1)], st.O[2 * ((g) >> 1) + 1], v0_, v1_, PJ);                                          \
        if constexpr (DMA) {                                                                                                   \
            ring.template piece<3 * (g)>(job);                                                                                 \
            ring.template piece<3 * (g) + 1>(job);                                                                             \
            ring.template piece<3 * (g) + 2>(job);                                                                             \
        }                                                                                                                      \
    }
    SAVAD_A_PV(0, p0) SAVAD_A_PV(1, p1)
    if constexpr (HAVE_NEXT) {   // tile j + 1's scores leave the accumulators while the last PV groups run
        sc = sa + sb;
        mask_next(sc);
    }
    SAVAD_A_PV(2, p0) SAVAD_A_PV(3, p1)
#undef SAVAD_A_PV
#undef SAVAD_A_LDV6
#undef SAVAD_A_LDV
    if constexpr (HAVE_NEXT) settle_reference(sc, st, false);
}
// the same tile with K / V^T in global memory (T <= 32: a block attends to itself; nothing to share, no ring)
template <class Mask>
__device__ __forceinline__ void attn_tile3_global(AttnState& st, const Tri (&qp)[8], const char* kblk, const char* vtblk, Mask mask, int lane) {
    f32x16 sa = st.negm, sb = zero16();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const Tri k = ldtri(kblk + ks * TFRAG_BYTES + lane * 16);
        sa = SAVAD_MF(k.h, qp[ks].l, sa);
        sb = SAVAD_MF(k.l, qp[ks].h, sb);
        sa = SAVAD_MF(k.m, qp[ks].m, sa);
        sb = SAVAD_MF(k.h, qp[ks].m, sb);
        sa = SAVAD_MF(k.m, qp[ks].h, sa);
        sb = SAVAD_MF(k.h, qp[ks].h, sb);
    }
    f32x16 sc = sa + sb;
    mask(sc);
    online_softmax_shifted(sc, st, true);
    const Tri p0 = split_half(sc, 0), p1 = split_half(sc, 1);
#pragma unroll
    for (int nbd = 0; nbd < 4; nbd += 2) {
        mfma6x2<false>(st.O[nbd], st.O[nbd + 1], ldtri(vtblk + ((nbd * 2 + 0) * TFRAG_BYTES) + lane * 16),
                       ldtri(vtblk + (((nbd + 1) * 2 + 0) * TFRAG_BYTES) + lane * 16), p0);
        mfma6x2<false>(st.O[nbd], st.O[nbd + 1], ldtri(vtblk + ((nbd * 2 + 1) * TFRAG_BYTES) + lane * 16),
                       ldtri(vtblk + (((nbd + 1) * 2 + 1) * TFRAG_BYTES) + lane * 16), p1);
    }
}

// ---------------------------------------------------------------------------------------------
// Row chain of one block per wave (vad/modeling/transformer.py:347,234-238,366-382; LAST: :33 and
// vad/models/self_attention.py:26-27): out-projection + residual -> LN -> FFN + residual -> next layer's LN + Q/K/V^T, or
// the encoder LayerNorm + classifier + LogSoftmax.  Weight stream = slots
//   0,1: Wo | 2+4c, 3+4c: W1 chunk c | 4+4c, 5+4c: W2 chunk c (c = 0..3) | 18..23: Wq, Wk, Wv  (two n-blocks per slot)
// at ring positions base + t.  PREFETCHED: slots 0 and 1 are already in flight (the fused launch requests them under its last two
// key tiles, so that the chain does not start with an exposed DMA round trip).
// ---------------------------------------------------------------------------------------------
struct RowArgs3 {
    int B, T, nblk;
    float* hbuf;
    const char* wo_frag;
    const float* bo;
    const char* w1_frag;
    const float* b1;
    const char* w2_frag;
    const float* b2;
    const char* wn_frag;  // !LAST: next layer's Wqkv' triples
    const float* wc;      // LAST: Wc' fp32 [2][D]
    const float* bn;      // !LAST: bqkv' [384]; LAST: bc' [2]
    char *qf, *kf, *vtf;  // !LAST: written (the NEXT layer's buffers)
    float* out;           // LAST
    float qscale;
};
// the two segments of row-chain slot t (compile-time t in the chain itself; 0 / 1 at run time for the fused launch's prefetch)
__device__ __forceinline__ const char* row_seg(const RowArgs3& A, int t, int sgm) {
    if (t < 2) return A.wo_frag + (size_t)(2 * t + sgm) * BLK3_BYTES;
    if (t < 18) {
        const int c = (t - 2) >> 2, r = (t - 2) & 3;
        return r < 2 ? A.w1_frag + (size_t)(4 * c + 2 * r + sgm) * BLK3_BYTES
                     : A.w2_frag + (size_t)((2 * (r - 2) + sgm) * 32 + 8 * c) * TFRAG_BYTES;  // n-block, K-steps 8c..8c+7
    }
    return A.wn_frag + (size_t)(2 * (t - 18) + sgm) * BLK3_BYTES;
}

// the row chain's biases (LAST: the classifier's folded weights and bias in the Q/K/V slot) -> LDS behind the ring; called at the head
// of the kernel, so that their round trip to memory is long over when the chain starts; published by any workgroup barrier
template <bool LAST>
struct RowBiases {
    BiasPiece pieces[4];
    BiasRegs<4> regs;
    float bc;
    float* lbn;
    __device__ __forceinline__ RowBiases(const RowArgs3& A, char* smem) {   // requests the loads
        float* lbo = reinterpret_cast<float*>(smem + NRING3 * SLOT_BYTES);
        float* lb1 = lbo + D;
        float* lb2 = lb1 + DFF;
        lbn = lb2 + D;
        pieces[0] = BiasPiece{lbo, A.bo, D};
        pieces[1] = BiasPiece{lb1, A.b1, DFF};
        pieces[2] = BiasPiece{lb2, A.b2, D};
        pieces[3] = BiasPiece{lbn, LAST ? A.wc : A.bn, LAST ? 2 * D : 3 * D};
        regs = request_bias_pieces(pieces);
        bc = LAST ? A.bn[threadIdx.x & 1] : 0.0f;
    }
    __device__ __forceinline__ void commit() const {   // ... and writes them to LDS
        commit_bias_pieces(pieces, regs);
        if (LAST && threadIdx.x < 2) lbn[2 * D + threadIdx.x] = bc;
    }
};

template <bool LAST, bool PREFETCHED>
__device__ __forceinline__ void row_stage_f32s(const RowArgs3& A, char* smem, Tri (&xp)[8], int blk, bool live, const Ring3& ring, int base) {
    float* lbo = reinterpret_cast<float*>(smem + NRING3 * SLOT_BYTES);
    float* lb1 = lbo + D;
    float* lb2 = lb1 + DFF;
    float* lbn = lb2 + D;
    const int lane = ring.lane, m = lane & 31, h = lane >> 5;
    constexpr int NSLOT = LAST ? 18 : 24;
    auto job = [&](int t) { return ring.job(base + t, row_seg(A, t, 0), row_seg(A, t, 1)); };
    const DmaJob none{nullptr, 0u};
    if (!PREFETCHED) {
        ring.issue_all(job(0));
        ring.issue_all(job(1));
    }
    // acquire slot T_, then acc0 / acc1 += slot . operand with the DMA of slot T_ + 2 between the MFMAs
#define SAVAD_ROW_GEMM(T_, acc0, acc1, operand)                                                                               \
    if ((T_) < 8) SAVAD_STAMP(24 + 3 * (T_));                                                                                 \
    ring.acquire<((T_) + 1 < NSLOT) ? 1 : 0>();                                                                               \
    if ((T_) < 8) SAVAD_STAMP(25 + 3 * (T_));                                                                                 \
    gemm_slot<false, ((T_) + 2 < NSLOT)>(acc0, acc1, ring.slot(base + (T_)), operand, ring, (T_) + 2 < NSLOT ? job((T_) + 2) : none); \
    if ((T_) < 8) SAVAD_STAMP(26 + 3 * (T_));
    float* hb = A.hbuf + (size_t)blk * (32 * D);
    // the residual block is requested here and added BEHIND the out-projection's MFMAs: its round trip to memory hides under them
    // ---- h1 = h + (bo + ctx Wo^T)   (the biases were staged by stage_row_biases at the head of the kernel; any barrier since publishes them)
    ring.acquire<1>();
    f32x16 hres[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) hres[nb] = zero16();
    if (live) load_hblock32(hres, hb, lane);
    f32x16 h1[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) h1[nb] = bias_block(lbo + 32 * nb, h);
    gemm_slot<false, true>(h1[0], h1[1], ring.slot(base), xp, ring, job(2));
    SAVAD_ROW_GEMM(1, h1[2], h1[3], xp)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) h1[nb] += hres[nb];
    SAVAD_STAMP(5);
    f32x4 xg[16];
    layernorm_regs(h1, xg);
    split_row(xg, xp);
    SAVAD_STAMP(6);
    // ---- FFN; its accumulators start from the residual stream (h1 + b2)
    f32x16(&o)[4] = h1;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) o[nb] += bias_block(lb2 + 32 * nb, h);
#define SAVAD_ROW_FFN(ch)                                                                                                     \
    {                                                                                                                         \
        f32x16 a[4];                                                                                                          \
        _Pragma("unroll") for (int nbl = 0; nbl < 4; ++nbl) a[nbl] = bias_block(lb1 + 128 * (ch) + 32 * nbl, h);              \
        SAVAD_ROW_GEMM(2 + 4 * (ch), a[0], a[1], xp)                                                                          \
        SAVAD_ROW_GEMM(3 + 4 * (ch), a[2], a[3], xp)                                                                          \
        if ((ch) == 0) SAVAD_STAMP(20);                                                                                       \
        Tri ap[8];                                                                                                            \
        _Pragma("unroll") for (int nbl = 0; nbl < 4; ++nbl) {                                                                 \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) a[nbl][r] = fmaxf(a[nbl][r], 0.0f);                                \
            ap[2 * nbl] = split_half(a[nbl], 0);                                                                              \
            ap[2 * nbl + 1] = split_half(a[nbl], 1);                                                                          \
        }                                                                                                                     \
        if ((ch) == 0) SAVAD_STAMP(21);                                                                                       \
        SAVAD_ROW_GEMM(4 + 4 * (ch), o[0], o[1], ap)                                                                          \
        SAVAD_ROW_GEMM(5 + 4 * (ch), o[2], o[3], ap)                                                                          \
        SAVAD_STAMP(7 + (ch));                                                                                                \
    }
    SAVAD_ROW_FFN(0) SAVAD_ROW_FFN(1) SAVAD_ROW_FFN(2) SAVAD_ROW_FFN(3)
#undef SAVAD_ROW_FFN
#undef SAVAD_ROW_GEMM
    if (!LAST && live) store_hblock32(hb, o, lane);
    layernorm_regs(o, xg);
    if constexpr (!LAST) {
        split_row(xg, xp);
        SAVAD_STAMP(11);
#define SAVAD_ROW_QKV(S_)                                                                                                     \
    ring.acquire<(18 + (S_) + 1 < NSLOT) ? 1 : 0>();                                                                          \
    qkv_slot<S_, (18 + (S_) + 2 < NSLOT)>(ring.slot(base + 18 + (S_)), xp, lbn, A.qf, A.kf, A.vtf, blk, ring,                 \
                                          18 + (S_) + 2 < NSLOT ? job(18 + (S_) + 2) : none, A.qscale, live);                 \
    SAVAD_STAMP(12 + (S_));
        SAVAD_ROW_QKV(0) SAVAD_ROW_QKV(1) SAVAD_ROW_QKV(2) SAVAD_ROW_QKV(3) SAVAD_ROW_QKV(4) SAVAD_ROW_QKV(5)
#undef SAVAD_ROW_QKV
    } else {
        float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
        for (int G = 0; G < 16; ++G) {
            const f32x4 c0 = ld4(lbn + 8 * G + 4 * h), c1 = ld4(lbn + D + 8 * G + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z0 = __builtin_fmaf(xg[G][e], c0[e], z0);
                z1 = __builtin_fmaf(xg[G][e], c1[e], z1);
            }
        }
        z0 = half_sum(z0) + lbn[2 * D];
        z1 = half_sum(z1) + lbn[2 * D + 1];
        const float mx = fmaxf(z0, z1);
        const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
        size_t row;
        int t_frame;
        const bool valid = live && (blk < A.nblk) && slot_row(A.B, A.T, blk, m, row, t_frame);
        if (h == 0 && valid) *reinterpret_cast<f32x2*>(A.out + row * 2) = f32x2{z0 - lse, z1 - lse};
    }
}

// ---------------------------------------------------------------------------------------------
// One launch per layer: attention of the wave's query block, immediately followed by its row chain.
//   PACKED = false (T > 32): workgroup = (sequence, group of <= 4 query blocks); the sequence's key blocks go through the
//            ring one per slot (K and V^T images), shared by the four waves, and the row chain's weight slots simply continue
//            the stream (its first two are requested under the last two key tiles); q/k/v^T double-buffered between layers.
//   PACKED = true (T <= 32): a block holds floor(32/T) whole sequences and attends to itself under a block-diagonal mask;
//            its K / V^T triples come straight from global memory (nothing to share), 4 blocks per workgroup.
// ---------------------------------------------------------------------------------------------
template <bool LAST, bool PACKED>
__global__ __launch_bounds__(256, 1) void attention_row_kernel_f32s(const char* __restrict__ qf, const char* __restrict__ kf,
                                                                    const char* __restrict__ vtf, int NG, RowArgs3 A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int B = A.B, T = A.T;
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Ring3 ring(smem, w, lane);
    AttnState st;
    attn_state_init(st);
    Tri qp[8];
    int blk_q, base = 0;
    bool active, qvalid;
    SAVAD_STAMP(0);
#ifdef SAVAD_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        g_savad_wg[blockIdx.x][0] = __builtin_readcyclecounter();
        g_savad_wg[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    const RowBiases<LAST> biases(A, smem);   // requested first: their round trip is over long before the row chain wants them
    if constexpr (PACKED) {
        biases.commit();
        blk_q = blockIdx.x * 4 + w;
        active = blk_q < A.nblk;
        const int G = 32 / T;
        qvalid = active && (m < G * T) && (blk_q * G + m / T < B);
        if (active) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qp[ks] = ldtri(qf + (size_t)blk_q * BLK3_BYTES + ks * TFRAG_BYTES + lane * 16);
            bool keyok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jk = 8 * (r >> 2) + 4 * h + (r & 3);
                keyok[r] = (jk < G * T) && (jk / T == m / T) && (blk_q * G + jk / T < B);
            }
            auto mask = [&](f32x16& sc) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = keyok[r] ? sc[r] : NEG_BIG;
            };
            attn_tile3_global(st, qp, kf + (size_t)blk_q * BLK3_BYTES, vtf + (size_t)blk_q * BLK3_BYTES, mask, lane);
        }
    } else {
        const int QB = (T + 31) / 32;  // >= 2
        int b, g;
        if (!xcd_balanced_map(B, NG, b, g)) return;
        const int qb0 = (g * QB) / NG, qb1 = ((g + 1) * QB) / NG;
        const int qb = qb0 + w;
        active = qb < qb1;
        blk_q = b * QB + (active ? qb : qb0);
        qvalid = active && 32 * qb + m < T;
        base = QB;
        // stream index u: slot u = [V(u)^T | K(u+1)] for the key tiles 0 .. QB-1 (K(0) in a slot of its own in front: u = -1, at ring
        // position 2), then the row chain's slots; behind the last key tile K is a re-read of that tile (never used)
        auto job = [&](int u) {
            if (u < QB) {
                const size_t kb = (size_t)b * QB;
                const int uv = u < 0 ? 0 : u, uk = u + 1 < QB ? u + 1 : QB - 1;
                return ring.job(u + 3, vtf + (kb + uv) * BLK3_BYTES, kf + (kb + uk) * BLK3_BYTES);
            }
            return ring.job(u, row_seg(A, u - QB, 0), row_seg(A, u - QB, 1));
        };
        ring.issue_all(job(-1));   // K(0) first: the prologue's scores wait for it
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qp[ks] = ldtri(qf + (size_t)blk_q * BLK3_BYTES + ks * TFRAG_BYTES + lane * 16);
        ring.issue_all(job(0));
        ring.issue_all(job(1));
        biases.commit();
        const int lim = T - 32 * (QB - 1) - 4 * h;   // keys of the last tile that exist, from this lane's first one
        auto no_mask = [](f32x16&) {};
        auto tail_mask = [&](f32x16& sc) {   // (only the last tile of a ragged sequence has missing keys)
            if (32 * QB > T) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = (8 * (r >> 2) + (r & 3) < lim) ? sc[r] : NEG_BIG;
            }
        };
        ring.acquire<2>();   // K(0) has landed; slots 0 and 1 may still be in flight
        SAVAD_STAMP(1);
        if (active) {
            f32x16 sc = qk_tile3(st, qp, ring.slot(2) + BLK3_BYTES, lane);
            settle_reference(sc, st, true);
            SAVAD_STAMP(2);
#pragma unroll 1
            for (int jt = 0; jt + 2 < QB; ++jt) {
                ring.acquire<1>();
                attn_step3<true, true>(st, qp, sc, ring.slot(jt), no_mask, ring, job(jt + 2));
            }
            ring.acquire<1>();
            attn_step3<true, true>(st, qp, sc, ring.slot(QB - 2), tail_mask, ring, job(QB));        // -> the last tile's scores, masked
            ring.acquire<1>();
            attn_step3<false, true>(st, qp, sc, ring.slot(QB - 1), no_mask, ring, job(QB + 1));
            SAVAD_STAMP(3);
        } else {  // a wave without a query block still moves its share of the stream and meets the others at every barrier
#pragma unroll 1
            for (int jt = 0; jt < QB; ++jt) {
                ring.acquire<1>();
                ring.issue_all(job(jt + 2));
            }
        }
    }
    // normalised context -> B-operand triples, in registers (invalid slots and waves without a block: exact zeros)
    Tri xp[8];
    {
        const float inv = qvalid ? 1.0f / half_sum(st.l_run) : 0.0f;
#pragma unroll
        for (int nbd = 0; nbd < 4; ++nbd) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st.O[nbd][r] = qvalid ? st.O[nbd][r] * inv : 0.0f;
            xp[2 * nbd] = split_half(st.O[nbd], 0);
            xp[2 * nbd + 1] = split_half(st.O[nbd], 1);
        }
    }
    SAVAD_STAMP(4);
    row_stage_f32s<LAST, !PACKED>(A, smem, xp, blk_q, active, ring, base);
    SAVAD_STAMP(18);
#ifdef SAVAD_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        g_savad_wg[blockIdx.x][1] = __builtin_readcyclecounter();
        g_savad_wg[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// T <= 32: the WHOLE forward in one launch (the fp32s edition of savad_packed_bf16.h's wave-per-block kernel).  The reference
// pipeline only ever runs 7-frame windows (vad/predictor.py:180-224).  A wave owns one packed block (floor(32/T) whole sequences)
// for ALL layers: Q, K and V^T are produced by the wave that consumes them, so the attention never leaves its registers; the fp32
// residual stream waits in registers; nothing but x, the weight stream and the log-probabilities crosses the CU boundary.  The four
// waves of a workgroup share the weight stream: 24 ring slots per layer -- Wq Wk Wv Wo (two slots each), then W1 / W2 chunks
// alternating (two slots each); behind the last layer the stream re-requests that layer's first two slots (never read), so that every
// step of the stream is the same wait + barrier + request.  Windowed mode (wo.w == T > 0): x is the predictor's feature MATRIX
// [N][F] and sequence s is its window feature[win_base + s + wo.off[0..T-1]] (vad/predictor.py:180-220): the gather is an address
// computation.
// ---------------------------------------------------------------------------------------------
constexpr int PACKED_F32S_MAX_LAYERS = 3;  // ring (144 KiB) + 3 x 4.5 KiB of biases + the classifier fit the 160 KiB of LDS
struct PackedF32sLayer {
    const char *wqkv, *wo, *w1, *w2;  // triples (pack_weight_frags3_kernel), LayerNorm affine folded in
};
struct PackedF32sModel {
    PackedF32sLayer layer[PACKED_F32S_MAX_LAYERS];
    const char* win;    // input Linear triples [4][F/16]
    const float* bin;   // input bias
    const float* pe;    // positional encoding / sqrt(D)
    const float* bias;  // [L][LBIAS]: b1' | b2 | bqkv' | bo of every layer
    const float *wc, *bc;
    int L;
};
inline constexpr int packed_f32s_lds_bytes(int L) { return NRING3 * SLOT_BYTES + (L * LBIAS + 2 * D + 4) * 4; }
// the two segments of slot i (0 .. 23) of a layer's weight stream
__device__ __forceinline__ const char* packed_seg(const PackedF32sLayer& Lw, int i, int sgm) {
    if (i < 6) return Lw.wqkv + (size_t)(2 * i + sgm) * BLK3_BYTES;
    if (i < 8) return Lw.wo + (size_t)(2 * (i - 6) + sgm) * BLK3_BYTES;
    const int c = (i - 8) >> 2, r = (i - 8) & 3;
    return r < 2 ? Lw.w1 + (size_t)(4 * c + 2 * r + sgm) * BLK3_BYTES : Lw.w2 + (size_t)((2 * (r - 2) + sgm) * 32 + 8 * c) * TFRAG_BYTES;
}

__global__ __launch_bounds__(256, 1) void packed_forward_kernel_f32s(const float* __restrict__ x, int B, int T, int F, int nblk,
                                                                     PackedF32sModel M, float qscale, float* __restrict__ out,
                                                                     WindowOffsets wo, int win_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lbias = reinterpret_cast<float*>(smem + NRING3 * SLOT_BYTES);
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * 4 + w;
    const bool live = blk < nblk;  // wave-uniform; a wave without a block still moves its share of the weight stream
    const Ring3 ring(smem, w, lane);
    const int L = M.L;
    // slot i of layer l (24 % 3 == 0: slot i of any layer sits at ring position i % 3); i >= 24: the next layer's (behind the last: its own)
    auto job = [&](int l, int i) {
        const int ll = i < 24 ? l : (l + 1 < L ? l + 1 : l), ii = i < 24 ? i : i - 24;
        const PackedF32sLayer Lw = M.layer[ll];
        return ring.job(ii, packed_seg(Lw, ii, 0), packed_seg(Lw, ii, 1));
    };
    ring.issue_all(job(0, 0));
    ring.issue_all(job(0, 1));
    for (int i = threadIdx.x * 4; i < L * LBIAS; i += 1024) st4(lbias + i, ld4(M.bias + i));  // published by the first ring barrier
    float* lwc = lbias + L * LBIAS;  // the classifier's folded weights [2][D] + bias [2] behind the biases
    for (int i = threadIdx.x * 4; i < 2 * D; i += 1024) st4(lwc + i, ld4(M.wc + i));
    if (threadIdx.x < 2) lwc[2 * D + threadIdx.x] = M.bc[threadIdx.x];

    // ---- slots of the block: sequence blk * G + m / T, frame m % T
    const int G = 32 / T, seq = blk * G + m / T, t_frame = m % T;
    const bool valid = live && m < G * T && seq < B;
    const size_t row = valid ? (size_t)seq * T + t_frame : 0;
    bool keyok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int jk = 8 * (r >> 2) + 4 * h + (r & 3);
        keyok[r] = (jk < G * T) && (jk / T == m / T) && (blk * G + jk / T < B);
    }
    // ---- input Linear + positional encoding (vad/models/self_attention.py:12-16,24)
    f32x16 hres[4], acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        hres[nb] = zero16();
        add_bias(hres[nb], M.bin + 32 * nb, h);
        add_block(hres[nb], M.pe + (size_t)(valid ? t_frame : 0) * D + 32 * nb, h);
    }
    {
        const size_t src_row = wo.w > 0 ? (size_t)win_base + (valid ? seq : 0) + wo.off[valid ? t_frame : 0] : row;
        input_gemm_f32s(hres, x + src_row * (size_t)F, valid, M.win, F / 16, lane, h);
    }
    f32x4 xg[16];
    layernorm_regs(hres, xg);
    Tri xp[8];
    split_row(xg, xp);

    // acquire slot I_ of the layer, then acc0 / acc1 += slot . operand with the DMA of slot I_ + 2 between the MFMAs
#define SAVAD_PK_GEMM(I_, SWAP_, acc0, acc1, operand)                                                                         \
    ring.acquire<1>();                                                                                                        \
    gemm_slot<SWAP_, true>(acc0, acc1, ring.slot(I_), operand, ring, job(l, (I_) + 2));
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const float* lb = lbias + l * LBIAS;
        const float *lb1 = lb, *lb2 = lb + DFF, *lbn = lb + DFF + D, *lbo = lb + DFF + 4 * D;
        // ---- Q (pre-scaled by log2(e)/sqrt(D)) and K in row layout -> triples
        Tri qp[8], kp[8];
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) acc[nbl] = bias_block(lbn + 32 * nbl, h);
        SAVAD_PK_GEMM(0, false, acc[0], acc[1], xp)
        SAVAD_PK_GEMM(1, false, acc[2], acc[3], xp)
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
            acc[nbl] *= qscale;
            qp[2 * nbl] = split_half(acc[nbl], 0);
            qp[2 * nbl + 1] = split_half(acc[nbl], 1);
        }
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) acc[nbl] = bias_block(lbn + D + 32 * nbl, h);
        SAVAD_PK_GEMM(2, false, acc[0], acc[1], xp)
        SAVAD_PK_GEMM(3, false, acc[2], acc[3], xp)
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
            kp[2 * nbl] = split_half(acc[nbl], 0);
            kp[2 * nbl + 1] = split_half(acc[nbl], 1);
        }
        // ---- scores and softmax of the single key tile (vad/modeling/transformer.py:351-363,333)
        f32x16 sa = zero16(), sb = zero16();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            sa = SAVAD_MF(kp[ks].h, qp[ks].l, sa);
            sb = SAVAD_MF(kp[ks].l, qp[ks].h, sb);
            sa = SAVAD_MF(kp[ks].m, qp[ks].m, sa);
            sb = SAVAD_MF(kp[ks].h, qp[ks].m, sb);
            sa = SAVAD_MF(kp[ks].m, qp[ks].h, sa);
            sb = SAVAD_MF(kp[ks].h, qp[ks].h, sb);
        }
        f32x16 sc = sa + sb;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = keyok[r] ? sc[r] : NEG_BIG;
        float l_run;
        {
            float mx = sc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
            mx = half_max(mx);
            float rs = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = __builtin_amdgcn_exp2f(sc[r] - mx);
                rs += sc[r];
            }
            l_run = rs;
        }
        const Tri p0 = split_half(sc, 0), p1 = split_half(sc, 1);
        // ---- V^T (operands swapped: lane = feature, registers = keys) and O^T = V^T P^T, normalised -> the context triples
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
            const float bv = lbn[2 * D + 32 * nbl + m];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nbl][r] = bv;
        }
        SAVAD_PK_GEMM(4, true, acc[0], acc[1], xp)
        SAVAD_PK_GEMM(5, true, acc[2], acc[3], xp)
        {
            const float inv = 1.0f / half_sum(l_run);
#pragma unroll
            for (int nbd = 0; nbd < 4; nbd += 2) {
                f32x16 O0 = zero16(), O1 = zero16();
                mfma6x2<false>(O0, O1, split_half(acc[nbd], 0), split_half(acc[nbd + 1], 0), p0);
                mfma6x2<false>(O0, O1, split_half(acc[nbd], 1), split_half(acc[nbd + 1], 1), p1);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    O0[r] = valid ? O0[r] * inv : 0.0f;
                    O1[r] = valid ? O1[r] * inv : 0.0f;
                }
                xp[2 * nbd] = split_half(O0, 0);
                xp[2 * nbd + 1] = split_half(O0, 1);
                xp[2 * nbd + 2] = split_half(O1, 0);
                xp[2 * nbd + 3] = split_half(O1, 1);
            }
        }
        // ---- h1 = h + bo + ctx Wo^T; LN; FFN on top of the residual stream
        f32x16(&h1)[4] = hres;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) h1[nb] += bias_block(lbo + 32 * nb, h);
        SAVAD_PK_GEMM(6, false, h1[0], h1[1], xp)
        SAVAD_PK_GEMM(7, false, h1[2], h1[3], xp)
        layernorm_regs(h1, xg);
        split_row(xg, xp);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) h1[nb] += bias_block(lb2 + 32 * nb, h);
#define SAVAD_PK_FFN(ch)                                                                                                      \
    {                                                                                                                         \
        _Pragma("unroll") for (int nbl = 0; nbl < 4; ++nbl) acc[nbl] = bias_block(lb1 + 128 * (ch) + 32 * nbl, h);            \
        SAVAD_PK_GEMM(8 + 4 * (ch), false, acc[0], acc[1], xp)                                                                \
        SAVAD_PK_GEMM(9 + 4 * (ch), false, acc[2], acc[3], xp)                                                                \
        Tri ap[8];                                                                                                            \
        _Pragma("unroll") for (int nbl = 0; nbl < 4; ++nbl) {                                                                 \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[nbl][r] = fmaxf(acc[nbl][r], 0.0f);                            \
            ap[2 * nbl] = split_half(acc[nbl], 0);                                                                            \
            ap[2 * nbl + 1] = split_half(acc[nbl], 1);                                                                        \
        }                                                                                                                     \
        SAVAD_PK_GEMM(10 + 4 * (ch), false, h1[0], h1[1], ap)                                                                 \
        SAVAD_PK_GEMM(11 + 4 * (ch), false, h1[2], h1[3], ap)                                                                 \
    }
        SAVAD_PK_FFN(0) SAVAD_PK_FFN(1) SAVAD_PK_FFN(2) SAVAD_PK_FFN(3)
#undef SAVAD_PK_FFN
        layernorm_regs(hres, xg);
        if (l + 1 < L) split_row(xg, xp);
    }
#undef SAVAD_PK_GEMM
    // ---- final LayerNorm (folded into the classifier) + Linear(D, 2) + log-softmax (vad/models/self_attention.py:26-28)
    float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
    for (int Gq = 0; Gq < 16; ++Gq) {
        const f32x4 c0 = ld4(lwc + 8 * Gq + 4 * h), c1 = ld4(lwc + D + 8 * Gq + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            z0 = __builtin_fmaf(xg[Gq][e], c0[e], z0);
            z1 = __builtin_fmaf(xg[Gq][e], c1[e], z1);
        }
    }
    z0 = half_sum(z0) + lwc[2 * D];
    z1 = half_sum(z1) + lwc[2 * D + 1];
    const float mx = fmaxf(z0, z1);
    const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
    if (h == 0 && valid) *reinterpret_cast<f32x2*>(out + row * 2) = f32x2{z0 - lse, z1 - lse};
    wait_vmem_all();  // the two slots requested behind the last layer are never read, but must have landed before the LDS is released
}


// ---------------------------------------------------------------------------------------------
// T <= 32, the LATENCY variant (round 6; the fp32s edition of packed_forward_kernel_bf16_ns): ONE packed block per workgroup, its
// four waves split every GEMM's OUTPUT features -- wave w: features [32w, 32w + 32) of Q, K, V^T, the out-projection and FFN2,
// hidden units [128w, 128w + 128) of FFN1 -- so a block's chain is 636 MFMAs per wave and layer instead of 2 400 and the 250 blocks
// of the reference's 1000-window chunk (vad/predictor.py:180) run on 250 CUs instead of 63.  What a wave does not own it gets
// through LDS: full fp32 rows for the LayerNorms, the Q / K triples for the scores (every wave computes the whole 32 x 32 score
// tile: 48 MFMAs, redundant but identical), the context triples, the ReLU'd hidden triples -- five barriers per layer.
// Weights: a wave reads only ITS triples, straight from L2 into AGPRs in chunks of two K-steps (six fragments, 6 KiB), requested
// by hand FIVE chunks (~60 MFMAs) ahead through an eight-chunk register ring behind counted waits: the chain has no other vector
// memory traffic, loads retire in order, so "30 younger loads may be in flight" means the chunk has landed.  A layer's stream starts
// at the top of the layer (under its first LayerNorm) and is drained at its end.
// Two independent accumulators take turns wherever the chain offers them (Q / K / V^T; two FFN1 blocks; FFN2's K range in two
// halves that are summed at the end -- the one place where the order of an fp32 sum differs from the wave-per-block kernel).
// ---------------------------------------------------------------------------------------------
struct WC6 {
    u32x4 v[6];
};
__device__ __forceinline__ void wc_load(WC6& c, const char* __restrict__ base /* wave-uniform: 6 consecutive fragments */, unsigned voff /* lane * 16 */) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(c.v[i]) : "v"(voff), "s"(base + (i >> 2) * 4096), "n"((i & 3) * 1024));
}
template <int PENDING>  // younger requests allowed to stay in flight
__device__ __forceinline__ void wc_wait(WC6& c) {
    asm volatile("s_waitcnt vmcnt(%6)" : "+a"(c.v[0]), "+a"(c.v[1]), "+a"(c.v[2]), "+a"(c.v[3]), "+a"(c.v[4]), "+a"(c.v[5]) : "n"(PENDING));
}
__device__ __forceinline__ Tri wc_tri(const WC6& c, int j) {
    return Tri{__builtin_bit_cast(bf16x8, c.v[3 * j]), __builtin_bit_cast(bf16x8, c.v[3 * j + 1]), __builtin_bit_cast(bf16x8, c.v[3 * j + 2])};
}
template <int B_, int E_, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B_ < E_) {
        f(std::integral_constant<int, B_>{});
        static_for<B_ + 1, E_>(f);
    }
}
// acc += a . b: the six products in mfma6x2's order (hl, lh, mm, hm, mh, hh)
#define SAVAD_MF6_STEP(acc, a, b, pa, pb) acc = SAVAD_MF((a).pa, (b).pb, acc)
__device__ __forceinline__ void mf6(f32x16& acc, const Tri& a, const Tri& b) {
    SAVAD_MF6_STEP(acc, a, b, h, l); SAVAD_MF6_STEP(acc, a, b, l, h); SAVAD_MF6_STEP(acc, a, b, m, m);
    SAVAD_MF6_STEP(acc, a, b, h, m); SAVAD_MF6_STEP(acc, a, b, m, h); SAVAD_MF6_STEP(acc, a, b, h, h);
}
// acc0 += a0 . b0, acc1 += a1 . b1, taking turns
__device__ __forceinline__ void mf6x2(f32x16& acc0, f32x16& acc1, const Tri& a0, const Tri& b0, const Tri& a1, const Tri& b1) {
#define SAVAD_MF6_2(pa, pb) SAVAD_MF6_STEP(acc0, a0, b0, pa, pb); SAVAD_MF6_STEP(acc1, a1, b1, pa, pb);
    SAVAD_MF6_2(h, l) SAVAD_MF6_2(l, h) SAVAD_MF6_2(m, m) SAVAD_MF6_2(h, m) SAVAD_MF6_2(m, h) SAVAD_MF6_2(h, h)
#undef SAVAD_MF6_2
}

#ifndef SAVAD_NSF_PENDING   // younger loads behind chunk c of a layer (experiment builds: 0 = every chunk waits out the whole stream)
#define SAVAD_NSF_PENDING(c) (6 * ((c) + NSF_AHEAD < NSF_CHUNKS ? NSF_AHEAD : NSF_CHUNKS - 1 - (c)))
#endif
constexpr int NSF_XB_FLOATS = TILE * XLD;          // one fp32 row-exchange buffer
constexpr int NSF_TRI_BYTES = 32 * TFRAG_BYTES;    // 32 triples: the hidden activations; Q / K (16) and the context (8) alias them
inline constexpr int nsf_lds_bytes(int L) { return 2 * NSF_XB_FLOATS * 4 + NSF_TRI_BYTES + (L * LBIAS + 2 * D + 4) * 4; }
constexpr int NSF_CHUNKS = 48;   // chunks per layer and wave: 12 Q/K/V (interleaved), 4 Wo, 16 W1, 16 W2
// (up to three chunks are live at once -- Q, K, V of a K-step pair -- and the request that follows the third one's wait must not land in
// the first one's registers: AHEAD + 3 <= RING)
constexpr int NSF_AHEAD = 3, NSF_RING = 6;
static_assert(NSF_AHEAD + 3 <= NSF_RING && NSF_CHUNKS % NSF_RING == 0, "weight chunk ring");

__global__ __launch_bounds__(256, 1) void packed_forward_kernel_f32s_ns(const float* __restrict__ x, int B, int T, int F, int nblk,
                                                                        PackedF32sModel M, float qscale, float* __restrict__ out,
                                                                        WindowOffsets wo, int win_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xtri = smem;                                           // the eight triples of a LayerNorm's output (24 KiB) ...
    float* xb0 = reinterpret_cast<float*>(smem);                 // ... and, behind the last layer, the final LayerNorm's fp32 rows
    f32x2* stat2 = reinterpret_cast<f32x2*>(smem + 8 * TFRAG_BYTES);   // [wave 4][row 32] (mean, M2) of a wave's 32 features
    char* tbuf = smem + 2 * NSF_XB_FLOATS * 4;                   // hidden triples [0, 96 K); partial scores [0, 16 K); context triples [16 K, 40 K)
    char* ctri = tbuf + 16384;
    float* lbias = reinterpret_cast<float*>(tbuf + NSF_TRI_BYTES);
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned voff = (unsigned)lane * 16u;
    const int blk = blockIdx.x, L = M.L;
    float* lwc = lbias + L * LBIAS;  // the classifier's folded weights [2][D] + bias [2], staged with the biases
    SAVAD_STAMP(40);

    // slot j of the block = sequence blk * G + j / T, frame j % T.  j / T for j < 32, T <= 32 through a float reciprocal: (j + 0.5) / T lies
    // at least 1 / 64 away from an integer, the rounding error of the product is < 1e-5 -- seventeen integer divisions cost the
    // prologue ~1 500 cycles
    const float inv_t = 1.0f / (float)T;
    auto div_t = [&](int j) { return (int)(((float)j + 0.5f) * inv_t); };
    const int G = 32 / T, sq = div_t(m), seq = blk * G + sq, t_frame = m - sq * T;
    const bool valid = blk < nblk && m < G * T && seq < B;
    const size_t row = valid ? (size_t)seq * T + t_frame : 0;
    bool keyok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int jk = 8 * (r >> 2) + 4 * h + (r & 3), jq = div_t(jk);
        keyok[r] = (jk < G * T) && (jq == sq) && (blk * G + jq < B);
    }
    // ---- this wave's weight stream: chunk c of layer l
    WC6 ring[NSF_RING];
    auto chunk_addr = [&](int l, int cc) -> const char* {
        const PackedF32sLayer Lw = M.layer[l];
        if (cc < 12) return Lw.wqkv + (size_t)(((cc % 3) * 4 + w) * 8 + 2 * (cc / 3)) * TFRAG_BYTES;
        if (cc < 16) return Lw.wo + (size_t)(w * 8 + 2 * (cc - 12)) * TFRAG_BYTES;
        if (cc < 32) {
            const int i = cc - 16, b = 2 * (i >> 3) + (i & 1), s2 = (i & 7) >> 1;
            return Lw.w1 + (size_t)((4 * w + b) * 8 + 2 * s2) * TFRAG_BYTES;
        }
        const int i = cc - 32;
        return Lw.w2 + (size_t)(w * 32 + 16 * (i & 1) + 2 * (i >> 1)) * TFRAG_BYTES;
    };
#define SAVAD_NSF_REQ(l_, c_) wc_load(ring[(c_) % NSF_RING], chunk_addr(l_, c_), voff)
    // chunk c_ of the layer: request the chunk NSF_AHEAD further down the layer's stream, then wait for this one.  The stream does NOT
    // run across the layer loop's back edge: registers with a load in flight at a loop header make hipcc copy them between the
    // prologue's and the loop's allocations while the load is still out (scripts/check_async_loads.py flags exactly that)
#define SAVAD_NSF_GET(l_, c_)                                                                            \
    if constexpr ((c_) + NSF_AHEAD < NSF_CHUNKS) { SAVAD_NSF_REQ(l_, (c_) + NSF_AHEAD); }                \
    wc_wait<SAVAD_NSF_PENDING((c_))>(ring[(c_) % NSF_RING])
    // ---- input Linear + positional encoding: this wave's 32 features
    f32x16 own = zero16();
    {
        const size_t src_row = wo.w > 0 ? (size_t)win_base + (valid ? seq : 0) + wo.off[valid ? t_frame : 0] : row;
        const float* xr = x + src_row * (size_t)F;
        const int KS = F / 16;
        add_bias(own, M.bin + 32 * w, h);
        add_block(own, M.pe + (size_t)(valid ? t_frame : 0) * D + 32 * w, h);
        constexpr int NBT = (PACKED_F32S_MAX_LAYERS * LBIAS / 4 + 255) / 256 + 1;
        const int nb4 = (L * LBIAS + 2 * D + 4) / 4;   // biases, then wc [2][D], then bc (+ 2 floats of padding)
        f32x4 bt[NBT];
#pragma unroll
        for (int j = 0; j < NBT; ++j) {
            const int i4 = (int)threadIdx.x + 256 * j;
            if (i4 < nb4) {
                const int i = 4 * i4;
                bt[j] = i < L * LBIAS ? ld4(M.bias + i) : (i < L * LBIAS + 2 * D ? ld4(M.wc + (i - L * LBIAS)) : f32x4{M.bc[0], M.bc[1], 0.0f, 0.0f});
            }
        }
        // every global load of the input GEMM is requested before the first use of any (F <= 128: the loop below takes what is beyond)
        constexpr int KSMAX = 8;
        f32x4 xa[KSMAX], xc[KSMAX];
        Tri wf[KSMAX];
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            const int kc = ks < KS ? ks : KS - 1;
            const float* px = xr + 32 * (kc >> 1) + 16 * (kc & 1) + 4 * h;
            xa[ks] = ld4(px);
            xc[ks] = ld4(px + 8);
            wf[ks] = ldtri(M.win + (size_t)(w * KS + kc) * TFRAG_BYTES + lane * 16);
        }
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks)
            if (ks < KS) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    t[e] = valid ? xa[ks][e] : 0.0f;
                    t[4 + e] = valid ? xc[ks][e] : 0.0f;
                }
                mf6(own, wf[ks], split8(t));
            }
        for (int ks = KSMAX; ks < KS; ++ks) {
            const int f0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
            mf6(own, ldtri(M.win + (size_t)(w * KS + ks) * TFRAG_BYTES + lane * 16), load_x_tri(xr + f0, valid));
        }
#pragma unroll
        for (int j = 0; j < NBT; ++j) {
            const int i4 = (int)threadIdx.x + 256 * j;
            if (i4 < nb4) st4(lbias + 4 * i4, bt[j]);   // published by the first barrier
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue's loads are out of the way: from here on only the weight stream counts
    SAVAD_STAMP(41);
    // full rows through LDS -> layernorm_regs on the wave-per-block kernel's register image -> the 8 K-step triples
    f32x4 xg[16];
    Tri xp[8];
    auto rows_ln = [&](float* xb, const f32x16& mine, bool split) {
        store_block(xb + m * XLD + 32 * w, mine, h);
        __syncthreads();
        f32x16 full[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 t = ld4(xb + m * XLD + 32 * nb + 8 * g + 4 * h);
#pragma unroll
                for (int s = 0; s < 4; ++s) full[nb][4 * g + s] = t[s];
            }
        layernorm_regs(full, xg);
        if (split) split_row(xg, xp);
    };
    // LayerNorm of rows whose features are spread over the four waves WITHOUT every wave redoing all of it: a wave's two-pass (mean, M2)
    // over its own 32 features, the four partials of a row combined exactly (equal counts: mean = average of the means, M2 = sum of
    // the M2s + 32 sum (mean_w - mean)^2), the wave's 32 normalised features split into their two triples, all eight read back.
    // (Two barriers instead of one, and the phase itself is no shorter -- 2.1 k against 1.5 k cycles -- but 470 VALU instructions per
    // LayerNorm leave the stream: [1000,7,80] 73 -> 70 us, same box.)
    auto ln_shared = [&](const f32x16& mine) {
        float s1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s1 += mine[r];
        const float mw = half_sum(s1) * (1.0f / 32.0f);
        float q2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = mine[r] - mw;
            q2 = __builtin_fmaf(d, d, q2);
        }
        q2 = half_sum(q2);
        if (h == 0) stat2[w * 32 + m] = f32x2{mw, q2};
        __syncthreads();
        const f32x2 p0 = stat2[m], p1 = stat2[32 + m], p2 = stat2[64 + m], p3 = stat2[96 + m];
        const float mean = 0.25f * ((p0[0] + p1[0]) + (p2[0] + p3[0]));
        const float d0 = p0[0] - mean, d1 = p1[0] - mean, d2 = p2[0] - mean, d3 = p3[0] - mean;
        const float M2 = ((p0[1] + p1[1]) + (p2[1] + p3[1])) + 32.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        const float rstd = 1.0f / sqrtf(M2 * (1.0f / D) + LN_EPS);
        f32x16 xn;
#pragma unroll
        for (int r = 0; r < 16; ++r) xn[r] = (mine[r] - mean) * rstd;
        sttri(xtri + (2 * w + 0) * TFRAG_BYTES + lane * 16, split_half(xn, 0));
        sttri(xtri + (2 * w + 1) * TFRAG_BYTES + lane * 16, split_half(xn, 1));
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xp[ks] = ldtri(xtri + ks * TFRAG_BYTES + lane * 16);
    };
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        static_for<0, NSF_AHEAD>([&](auto c) { SAVAD_NSF_REQ(l, decltype(c)::value); });   // (their round trip to L2 runs under the LayerNorm)
        ln_shared(own);   // (its barriers also retire the previous layer's hidden triples' readers)
        SAVAD_STAMP(42);
        const float* lb = lbias + l * LBIAS;
        const float *lb1 = lb, *lb2 = lb + DFF, *lbn = lb + DFF + D, *lbo = lb + DFF + 4 * D;
        // ---- Q (pre-scaled), K, V^T of this wave's 32 features, the three accumulators taking turns
        Tri vt0, vt1;
        {
            f32x16 aq = bias_block(lbn + 32 * w, h), ak = bias_block(lbn + D + 32 * w, h), av;
            const float bv = lbn[2 * D + 32 * w + m];
#pragma unroll
            for (int r = 0; r < 16; ++r) av[r] = bv;
            static_for<0, 4>([&](auto sc_) {
                constexpr int s2 = decltype(sc_)::value;
                SAVAD_NSF_GET(l, 3 * s2);
                SAVAD_NSF_GET(l, 3 * s2 + 1);
                SAVAD_NSF_GET(l, 3 * s2 + 2);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const Tri wq = wc_tri(ring[(3 * s2) % NSF_RING], j), wk = wc_tri(ring[(3 * s2 + 1) % NSF_RING], j),
                              wv = wc_tri(ring[(3 * s2 + 2) % NSF_RING], j);
                    const Tri& xo = xp[2 * s2 + j];
#define SAVAD_QKV3(pa, pb)                       \
    aq = SAVAD_MF(wq.pa, xo.pb, aq);             \
    ak = SAVAD_MF(wk.pa, xo.pb, ak);             \
    av = SAVAD_MF(xo.pb, wv.pa, av);
                    SAVAD_QKV3(h, l) SAVAD_QKV3(l, h) SAVAD_QKV3(m, m) SAVAD_QKV3(h, m) SAVAD_QKV3(m, h) SAVAD_QKV3(h, h)
#undef SAVAD_QKV3
                }
            });
            aq *= qscale;
            const Tri q0 = split_half(aq, 0), q1 = split_half(aq, 1), k0 = split_half(ak, 0), k1 = split_half(ak, 1);
            vt0 = split_half(av, 0);
            vt1 = split_half(av, 1);
            // ---- this wave's 32-feature share of the score tile (12 MFMAs), the four shares summed through LDS
            f32x16 sa = zero16(), sb = zero16();
#define SAVAD_NSF_QK(kk, qq)                  \
    sa = SAVAD_MF((kk).h, (qq).l, sa);        \
    sb = SAVAD_MF((kk).l, (qq).h, sb);        \
    sa = SAVAD_MF((kk).m, (qq).m, sa);        \
    sb = SAVAD_MF((kk).h, (qq).m, sb);        \
    sa = SAVAD_MF((kk).m, (qq).h, sa);        \
    sb = SAVAD_MF((kk).h, (qq).h, sb);
            SAVAD_NSF_QK(k0, q0)
            SAVAD_NSF_QK(k1, q1)
#undef SAVAD_NSF_QK
            const f32x16 sp = sa + sb;
            float* pb = reinterpret_cast<float*>(tbuf);
#pragma unroll
            for (int g = 0; g < 4; ++g) st4(pb + ((w * 4 + g) * 64 + lane) * 4, f32x4{sp[4 * g], sp[4 * g + 1], sp[4 * g + 2], sp[4 * g + 3]});
        }
        SAVAD_STAMP(43);
        __syncthreads();  // the four partial score tiles
        SAVAD_STAMP(44);
        f32x16 sc;
        {
            const float* pb = reinterpret_cast<const float*>(tbuf);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 s4 = (ld4(pb + ((0 * 4 + g) * 64 + lane) * 4) + ld4(pb + ((1 * 4 + g) * 64 + lane) * 4)) +
                                 (ld4(pb + ((2 * 4 + g) * 64 + lane) * 4) + ld4(pb + ((3 * 4 + g) * 64 + lane) * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) sc[4 * g + e] = s4[e];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = keyok[r] ? sc[r] : NEG_BIG;
        float l_run;
        {
            float mx = sc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
            mx = half_max(mx);
            float rs = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = __builtin_amdgcn_exp2f(sc[r] - mx);
                rs += sc[r];
            }
            l_run = rs;
        }
        {
            const Tri p0 = split_half(sc, 0), p1 = split_half(sc, 1);
            const float inv = 1.0f / half_sum(l_run);
            f32x16 O = zero16();
            mf6(O, vt0, p0);
            mf6(O, vt1, p1);
#pragma unroll
            for (int r = 0; r < 16; ++r) O[r] = valid ? O[r] * inv : 0.0f;
            sttri(ctri + (2 * w + 0) * TFRAG_BYTES + lane * 16, split_half(O, 0));   // (a region of its own: no barrier for the partial scores' readers)
            sttri(ctri + (2 * w + 1) * TFRAG_BYTES + lane * 16, split_half(O, 1));
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xp[ks] = ldtri(ctri + ks * TFRAG_BYTES + lane * 16);
        SAVAD_STAMP(45);
        // ---- h1 = h + bo + ctx Wo^T (this wave's block), LN2
        f32x16 h1 = own;
        h1 += bias_block(lbo + 32 * w, h);
        static_for<0, 4>([&](auto sc_) {
            constexpr int s2 = decltype(sc_)::value;
            SAVAD_NSF_GET(l, 12 + s2);
#pragma unroll
            for (int j = 0; j < 2; ++j) mf6(h1, wc_tri(ring[(12 + s2) % NSF_RING], j), xp[2 * s2 + j]);
        });
        SAVAD_STAMP(46);
        ln_shared(h1);  // (its barriers also retire the context triples' readers)
        SAVAD_STAMP(47);
        // ---- FFN1: hidden units [128 w, 128 w + 128) as two pairs of blocks, ReLU'd triples 8 w .. 8 w + 7 of the exchange; the first
        // pair's ReLU / split / stores are written behind the second pair's first waits (hipcc still emits them in one piece, 223
        // instructions between two MFMAs -- __builtin_amdgcn_sched_group_barrier patterns did not move it either; -0.8 %, same box)
        {
            auto hidden_out = [&](f32x16& a0, f32x16& a1, int pr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    a0[r] = fmaxf(a0[r], 0.0f);
                    a1[r] = fmaxf(a1[r], 0.0f);
                }
                sttri(tbuf + (8 * w + 4 * pr + 0) * TFRAG_BYTES + lane * 16, split_half(a0, 0));
                sttri(tbuf + (8 * w + 4 * pr + 1) * TFRAG_BYTES + lane * 16, split_half(a0, 1));
                sttri(tbuf + (8 * w + 4 * pr + 2) * TFRAG_BYTES + lane * 16, split_half(a1, 0));
                sttri(tbuf + (8 * w + 4 * pr + 3) * TFRAG_BYTES + lane * 16, split_half(a1, 1));
            };
            f32x16 a0 = bias_block(lb1 + 128 * w, h), a1 = bias_block(lb1 + 128 * w + 32, h);
            static_for<0, 4>([&](auto sc_) {
                constexpr int s2 = decltype(sc_)::value, c0 = 16 + 2 * s2;
                SAVAD_NSF_GET(l, c0);
                SAVAD_NSF_GET(l, c0 + 1);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    mf6x2(a0, a1, wc_tri(ring[c0 % NSF_RING], j), xp[2 * s2 + j], wc_tri(ring[(c0 + 1) % NSF_RING], j), xp[2 * s2 + j]);
            });
            f32x16 b0 = bias_block(lb1 + 128 * w + 64, h), b1 = bias_block(lb1 + 128 * w + 96, h);
            static_for<0, 4>([&](auto sc_) {
                constexpr int s2 = decltype(sc_)::value, c0 = 24 + 2 * s2;
                SAVAD_NSF_GET(l, c0);
                SAVAD_NSF_GET(l, c0 + 1);
                if constexpr (s2 == 0) hidden_out(a0, a1, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    mf6x2(b0, b1, wc_tri(ring[c0 % NSF_RING], j), xp[2 * s2 + j], wc_tri(ring[(c0 + 1) % NSF_RING], j), xp[2 * s2 + j]);
            });
            hidden_out(b0, b1, 1);
        }
        SAVAD_STAMP(48);
        __syncthreads();
        SAVAD_STAMP(49);
        // ---- FFN2 on top of the residual stream: K-steps 0..15 into o, 16..31 into o2, taking turns
        f32x16 o = h1, o2 = zero16();
        o += bias_block(lb2 + 32 * w, h);
        static_for<0, 8>([&](auto sc_) {
            constexpr int s2 = decltype(sc_)::value, c0 = 32 + 2 * s2;
            SAVAD_NSF_GET(l, c0);
            SAVAD_NSF_GET(l, c0 + 1);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                mf6x2(o, o2, wc_tri(ring[c0 % NSF_RING], j), ldtri(tbuf + (2 * s2 + j) * TFRAG_BYTES + lane * 16),
                      wc_tri(ring[(c0 + 1) % NSF_RING], j), ldtri(tbuf + (16 + 2 * s2 + j) * TFRAG_BYTES + lane * 16));
        });
        own = o + o2;
        SAVAD_STAMP(50);
    }
    SAVAD_STAMP(51);
#undef SAVAD_NSF_REQ
#undef SAVAD_NSF_GET
    // ---- final LayerNorm (folded into the classifier) + Linear(D, 2) + log-softmax (vad/models/self_attention.py:26-28)
    rows_ln(xb0, own, false);
    if (w == 0) {
        float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
        for (int Gq = 0; Gq < 16; ++Gq) {
            const f32x4 c0 = ld4(lwc + 8 * Gq + 4 * h), c1 = ld4(lwc + D + 8 * Gq + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z0 = __builtin_fmaf(xg[Gq][e], c0[e], z0);
                z1 = __builtin_fmaf(xg[Gq][e], c1[e], z1);
            }
        }
        z0 = half_sum(z0) + lwc[2 * D];
        z1 = half_sum(z1) + lwc[2 * D + 1];
        const float mx = fmaxf(z0, z1);
        const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
        if (h == 0 && valid) *reinterpret_cast<f32x2*>(out + row * 2) = f32x2{z0 - lse, z1 - lse};
    }
    SAVAD_STAMP(52);
}
#undef SAVAD_MF6_STEP

}  // namespace fs
}  // namespace savad
