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

namespace savad {
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
__device__ __forceinline__ void mfma6x2(f32x16& acc0, f32x16& acc1, const Tri& a0, const Tri& a1, const Tri& b) {
#define SAVAD_MF2(pa, pb)                                                           \
    acc0 = SWAP ? SAVAD_MF(b.pb, a0.pa, acc0) : SAVAD_MF(a0.pa, b.pb, acc0);        \
    acc1 = SWAP ? SAVAD_MF(b.pb, a1.pa, acc1) : SAVAD_MF(a1.pa, b.pb, acc1);
    SAVAD_MF2(h, l) SAVAD_MF2(l, h) SAVAD_MF2(m, m) SAVAD_MF2(h, m) SAVAD_MF2(m, h) SAVAD_MF2(h, h)
#undef SAVAD_MF2
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

// ---- the ring: 3 slots of 48 KiB; a slot = two 24-KiB segments, each contiguous in global memory.  Wave w moves 12 KiB:
// half (w & 1) of segment (w >> 1), as three groups of four 1-KiB DMA instructions.  Completion by counted vmcnt (vector
// memory operations retire in order): "at most 12 k outstanding" = everything but the k newest slots has landed.
struct Ring3 {
    static constexpr int PER = 12;
    static constexpr int DEPTH = 2;
    char* base;
    int w, lane;
    __device__ __forceinline__ char* slot(int t) const { return base + (t % NRING3) * SLOT_BYTES; }
    template <class SegSrc>
    __device__ __forceinline__ void issue(int t, SegSrc seg_src) const {
        if (SAVAD_ABLATE & 1) return;
        const unsigned slot0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)slot(t));
        const unsigned off = (unsigned)lane * 16u;
        const char* s0 = seg_src(w >> 1) + (size_t)(w & 1) * (BLK3_BYTES / 2);
        const unsigned ldsb = slot0 + (unsigned)w * (BLK3_BYTES / 2);
        bf::Ring<4>::dma4k<0>(s0, ldsb, off);
        bf::Ring<4>::dma4k<4096>(s0, ldsb, off + 4096u);
        bf::Ring<4>::dma4k<8192>(s0, ldsb, off + 8192u);
    }
    // slot t has landed for every wave.  newer = slots issued after slot t (0 .. 2); stores_after = this wave's vector stores
    // issued after its newest DMA (they may stay in flight)
    __device__ __forceinline__ void acquire(int newer, int stores_after = 0) const {
        if (SAVAD_ABLATE & 2) return;
        const int n = PER * newer + stores_after;
        if (n == 0) __builtin_amdgcn_s_waitcnt(bf::Ring<4>::vmcnt_imm(0));
        else if (n == 12) __builtin_amdgcn_s_waitcnt(bf::Ring<4>::vmcnt_imm(12));
        else if (n == 24) __builtin_amdgcn_s_waitcnt(bf::Ring<4>::vmcnt_imm(24));
        else __builtin_amdgcn_s_waitcnt(bf::Ring<4>::vmcnt_imm(0));
        asm volatile("" ::: "memory");
        __syncthreads();
    }
};

// acc[0..1] += W[the slot's two n-blocks] . x   (transposed form: lane = data row, registers = output features)
template <bool SWAP>
__device__ __forceinline__ void gemm_slot(f32x16& acc0, f32x16& acc1, const char* slot, const Tri (&xp)[8], int lane) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const Tri w0 = (SAVAD_ABLATE & 8) ? xp[7 - ks] : ldtri(slot + ks * TFRAG_BYTES + lane * 16);
        const Tri w1 = (SAVAD_ABLATE & 8) ? xp[ks ^ 1] : ldtri(slot + BLK3_BYTES + ks * TFRAG_BYTES + lane * 16);
        mfma6x2<SWAP>(acc0, acc1, w0, w1, xp[ks]);
    }
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

// One of the six QKV slots: slot s covers n-blocks 2 (s & 1), 2 (s & 1) + 1 of projection rb = s >> 1 (0 query, 1 key:
// transposed form; 2 value: swapped form -> V^T).  Q is stored PRE-SCALED by qscale = log2(e) / sqrt(D).
__device__ __forceinline__ void qkv_slot(int s, const char* slot, const Tri (&xp)[8], const float* lbq, char* __restrict__ qf,
                                         char* __restrict__ kf, char* __restrict__ vtf, int blk, int lane, float qscale, bool live) {
    const int n = lane & 31, h = lane >> 5;
    const int rb = s >> 1, nb0 = 2 * (s & 1);
    f32x16 acc[2];
    if (rb < 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = bias_block(lbq + D * rb + 32 * (nb0 + i), h);
        gemm_slot<false>(acc[0], acc[1], slot, xp, lane);
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float bv = lbq[2 * D + 32 * (nb0 + i) + n];  // lane = output feature
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = bv;
        }
        gemm_slot<true>(acc[0], acc[1], slot, xp, lane);
    }
    if (rb == 0) {
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
__device__ __forceinline__ Tri load_x_tri(const float* p, bool valid) {
    f32x4 a = ld4(p), b = ld4(p + 8);
    if (!valid) a = b = f32x4{0.f, 0.f, 0.f, 0.f};
    float t[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        t[e] = a[e];
        t[4 + e] = b[e];
    }
    return split8(t);
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
    const Ring3 ring{smem, w, lane};
    constexpr int NSLOT = 6;
    auto issue = [&](int t) { ring.issue(t, [&](int sgm) { return wqkv_frag + (size_t)(2 * t + sgm) * BLK3_BYTES; }); };
    issue(0);
    issue(1);
    stage_bias(lbq, bqkv, 3 * D);
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
    const int KS = F / 16;
    const float* xr = x + x_row_offset(row, T, F, xbs);
    for (int ks = 0; ks < KS; ++ks) {
        const int f0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
        const Tri xf = load_x_tri(xr + f0, valid);
#pragma unroll
        for (int nb = 0; nb < 4; nb += 2) {
            const Tri w0 = ldtri(win_frag + (size_t)(nb * KS + ks) * TFRAG_BYTES + lane * 16);
            const Tri w1 = ldtri(win_frag + (size_t)((nb + 1) * KS + ks) * TFRAG_BYTES + lane * 16);
            mfma6x2<false>(h0[nb], h0[nb + 1], w0, w1, xf);
        }
    }
    if (blk < nblk) store_hblock32(hbuf + (size_t)blk * (32 * D), h0, lane);
    f32x4 xg[16];
    layernorm_regs(h0, xg);
    Tri xp[8];
    split_row(xg, xp);
#pragma unroll 1
    for (int t = 0; t < NSLOT; ++t) {
        ring.acquire(NSLOT - 1 - t < 1 ? NSLOT - 1 - t : 1);
        if (t + 2 < NSLOT) issue(t + 2);
        qkv_slot(t, ring.slot(t), xp, lbq, qf, kf, vtf, blk, lane, qscale, blk < nblk);
    }
}

// ---------------------------------------------------------------------------------------------
// Attention on triples (vad/modeling/transformer.py:305-346,351-363), one 32-key tile:
//   S^T = K Q^T as 8 K-steps x 6 products on two accumulators (negm rides in as C of one of them), the fp32 online softmax
//   of savad_kernels_bf16.h, P split into its three pieces, O^T += V^T P^T as 4 x 2 x 6 products.
// ---------------------------------------------------------------------------------------------
template <class Mask>
__device__ __forceinline__ void attn_tile3(AttnState& st, const Tri (&qp)[8], const char* kblk, const char* vtblk, Mask mask,
                                           bool first, int lane) {
    f32x16 sa = st.negm, sb = zero16();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const Tri k = (SAVAD_ABLATE & 8) ? qp[7 - ks] : ldtri(kblk + ks * TFRAG_BYTES + lane * 16);
        sa = SAVAD_MF(k.h, qp[ks].l, sa);
        sb = SAVAD_MF(k.l, qp[ks].h, sb);
        sa = SAVAD_MF(k.m, qp[ks].m, sa);
        sb = SAVAD_MF(k.h, qp[ks].m, sb);
        sa = SAVAD_MF(k.m, qp[ks].h, sa);
        sb = SAVAD_MF(k.h, qp[ks].h, sb);
    }
    f32x16 sc = sa + sb;
    mask(sc);
    if (!(SAVAD_ABLATE & 4)) online_softmax_shifted(sc, st, first);
    const Tri p0 = split_half(sc, 0), p1 = split_half(sc, 1);
#pragma unroll
    for (int nbd = 0; nbd < 4; nbd += 2) {
        {
            const Tri v0 = (SAVAD_ABLATE & 8) ? qp[nbd] : ldtri(vtblk + ((nbd * 2 + 0) * TFRAG_BYTES) + lane * 16);
            const Tri v1 = (SAVAD_ABLATE & 8) ? qp[nbd + 1] : ldtri(vtblk + (((nbd + 1) * 2 + 0) * TFRAG_BYTES) + lane * 16);
            mfma6x2<false>(st.O[nbd], st.O[nbd + 1], v0, v1, p0);
        }
        {
            const Tri v0 = (SAVAD_ABLATE & 8) ? qp[4 + nbd] : ldtri(vtblk + ((nbd * 2 + 1) * TFRAG_BYTES) + lane * 16);
            const Tri v1 = (SAVAD_ABLATE & 8) ? qp[5 + nbd] : ldtri(vtblk + (((nbd + 1) * 2 + 1) * TFRAG_BYTES) + lane * 16);
            mfma6x2<false>(st.O[nbd], st.O[nbd + 1], v0, v1, p1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row chain of one block per wave (vad/modeling/transformer.py:347,234-238,366-382; LAST: :33 and
// vad/models/self_attention.py:26-27): out-projection + residual -> LN -> FFN + residual -> next layer's LN + Q/K/V^T, or
// the encoder LayerNorm + classifier + LogSoftmax.  Weight stream = ring slots
//   0,1: Wo | 2+4c, 3+4c: W1 chunk c | 4+4c, 5+4c: W2 chunk c (c = 0..3) | 18..23: Wq, Wk, Wv  (two n-blocks per slot)
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

template <bool LAST>
__device__ __forceinline__ void row_stage_f32s(const RowArgs3& A, char* smem, Tri (&xp)[8], int blk, bool live, int lane, int w) {
    float* lbo = reinterpret_cast<float*>(smem + NRING3 * SLOT_BYTES);
    float* lb1 = lbo + D;
    float* lb2 = lb1 + DFF;
    float* lbn = lb2 + D;
    const int m = lane & 31, h = lane >> 5;
    const Ring3 ring{smem, w, lane};
    constexpr int NSLOT = LAST ? 18 : 24;
    auto issue = [&](int t) {
        ring.issue(t, [&](int sgm) -> const char* {
            if (t < 2) return A.wo_frag + (size_t)(2 * t + sgm) * BLK3_BYTES;
            if (t < 18) {
                const int c = (t - 2) >> 2, r = (t - 2) & 3;
                return r < 2 ? A.w1_frag + (size_t)(4 * c + 2 * r + sgm) * BLK3_BYTES
                             : A.w2_frag + (size_t)((2 * (r - 2) + sgm) * 32 + 8 * c) * TFRAG_BYTES;  // n-block, K-steps 8c..8c+7
            }
            return A.wn_frag + (size_t)(2 * (t - 18) + sgm) * BLK3_BYTES;
        });
    };
    auto advance = [&](int t) {
        ring.acquire(NSLOT - 1 - t < 1 ? NSLOT - 1 - t : 1);
        if (t + 2 < NSLOT) issue(t + 2);
    };
    issue(0);
    issue(1);
    const BiasPiece pieces[4] = {{lbo, A.bo, D}, {lb1, A.b1, DFF}, {lb2, A.b2, D}, {lbn, LAST ? A.wc : A.bn, LAST ? 2 * D : 3 * D}};
    const BiasRegs<4> breg = request_bias_pieces(pieces);
    const float bc = LAST ? A.bn[threadIdx.x & 1] : 0.0f;
    float* hb = A.hbuf + (size_t)blk * (32 * D);
    f32x16 h1[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) h1[nb] = zero16();
    if (live) load_hblock32(h1, hb, lane);
    commit_bias_pieces(pieces, breg);
    if (LAST && threadIdx.x < 2) lbn[2 * D + threadIdx.x] = bc;
    // ---- h1 = h + bo + ctx Wo^T
    advance(0);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) h1[nb] += bias_block(lbo + 32 * nb, h);
    gemm_slot<false>(h1[0], h1[1], ring.slot(0), xp, lane);
    advance(1);
    gemm_slot<false>(h1[2], h1[3], ring.slot(1), xp, lane);
    f32x4 xg[16];
    layernorm_regs(h1, xg);
    split_row(xg, xp);
    // ---- FFN; its accumulators start from the residual stream (h1 + b2)
    f32x16(&o)[4] = h1;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) o[nb] += bias_block(lb2 + 32 * nb, h);
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
        const int t0 = 2 + 4 * ch;
        f32x16 a[4];
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) a[nbl] = bias_block(lb1 + 128 * ch + 32 * nbl, h);
        advance(t0);
        gemm_slot<false>(a[0], a[1], ring.slot(t0), xp, lane);
        advance(t0 + 1);
        gemm_slot<false>(a[2], a[3], ring.slot(t0 + 1), xp, lane);
        Tri ap[8];
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[nbl][r] = fmaxf(a[nbl][r], 0.0f);
            ap[2 * nbl] = split_half(a[nbl], 0);
            ap[2 * nbl + 1] = split_half(a[nbl], 1);
        }
        advance(t0 + 2);
        gemm_slot<false>(o[0], o[1], ring.slot(t0 + 2), ap, lane);
        advance(t0 + 3);
        gemm_slot<false>(o[2], o[3], ring.slot(t0 + 3), ap, lane);
    }
    if (!LAST && live) store_hblock32(hb, o, lane);
    layernorm_regs(o, xg);
    if (!LAST) {
        split_row(xg, xp);
#pragma unroll 1
        for (int s = 0; s < 6; ++s) {
            advance(18 + s);
            qkv_slot(s, ring.slot(18 + s), xp, lbn, A.qf, A.kf, A.vtf, blk, lane, A.qscale, live);
        }
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
//            ring one per slot (K and V^T images), shared by the four waves; q/k/v^T double-buffered between layers.
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
    const Ring3 ring{smem, w, lane};
    AttnState st;
    attn_state_init(st);
    Tri qp[8];
    int blk_q;
    bool active, qvalid;
    if constexpr (PACKED) {
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
            attn_tile3(st, qp, kf + (size_t)blk_q * BLK3_BYTES, vtf + (size_t)blk_q * BLK3_BYTES, mask, true, lane);
        }
    } else {
        const int QB = (T + 31) / 32;
        int b, g;
        if (!xcd_balanced_map(B, NG, b, g)) return;
        const int qb0 = (g * QB) / NG, qb1 = ((g + 1) * QB) / NG;
        const int qb = qb0 + w;
        active = qb < qb1;
        blk_q = b * QB + (active ? qb : qb0);
        qvalid = active && 32 * qb + m < T;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qp[ks] = ldtri(qf + (size_t)blk_q * BLK3_BYTES + ks * TFRAG_BYTES + lane * 16);
        auto issue = [&](int stage) {
            const size_t kb = (size_t)b * QB + stage;
            ring.issue(stage, [&](int sgm) { return (sgm == 0 ? kf : vtf) + kb * BLK3_BYTES; });
        };
        issue(0);
        if (QB > 1) issue(1);
#pragma unroll 1
        for (int jt = 0; jt < QB; ++jt) {
            ring.acquire(QB - 1 - jt < 1 ? QB - 1 - jt : 1);
            if (jt + 2 < QB) issue(jt + 2);
            if (!active) continue;
            const char* buf = ring.slot(jt);
            auto mask = [&](f32x16& sc) {  // a REAL (wave-uniform) branch: only the last tile of a ragged sequence has missing keys
                if (32 * jt + 32 > T) {
                    asm volatile("" ::: "memory");
                    const int lim = T - 32 * jt - 4 * h;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] = (8 * (r >> 2) + (r & 3) < lim) ? sc[r] : NEG_BIG;
                }
            };
            attn_tile3(st, qp, buf, buf + BLK3_BYTES, mask, jt == 0, lane);
        }
        __syncthreads();  // everyone is done with the K/V ring: it becomes the weight ring
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
    row_stage_f32s<LAST>(A, smem, xp, blk_q, active, lane, w);
}

// ---------------------------------------------------------------------------------------------
// T <= 32: the WHOLE forward in one launch (the fp32s edition of savad_packed_bf16.h's wave-per-block kernel).  The reference
// pipeline only ever runs 7-frame windows (vad/predictor.py:180-224).  A wave owns one packed block (floor(32/T) whole sequences)
// for ALL layers: Q, K and V^T are produced by the wave that consumes them, so the attention never leaves its registers; the fp32
// residual stream waits in registers; nothing but x, the weight stream and the log-probabilities crosses the CU boundary.  The four
// waves of a workgroup share the weight stream: 24 ring slots per layer -- Wq Wk Wv Wo (two slots each), then W1 / W2 chunks
// alternating (two slots each).  Windowed mode (wo.w == T > 0): x is the predictor's feature MATRIX [N][F] and sequence s is its
// window feature[win_base + s + wo.off[0..T-1]] (vad/predictor.py:180-220): the gather is an address computation.
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

__global__ __launch_bounds__(256, 1) void packed_forward_kernel_f32s(const float* __restrict__ x, int B, int T, int F, int nblk,
                                                                     PackedF32sModel M, float qscale, float* __restrict__ out,
                                                                     WindowOffsets wo, int win_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lbias = reinterpret_cast<float*>(smem + NRING3 * SLOT_BYTES);
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * 4 + w;
    const bool live = blk < nblk;  // wave-uniform; a wave without a block still moves its share of the weight stream
    const Ring3 ring{smem, w, lane};
    const int L = M.L, NS = 24 * L;
    auto issue = [&](int t) {
        const int l = t / 24, i = t - 24 * l;
        const PackedF32sLayer Lw = M.layer[l];
        ring.issue(t, [&](int sgm) -> const char* {
            if (i < 6) return Lw.wqkv + (size_t)(2 * i + sgm) * BLK3_BYTES;
            if (i < 8) return Lw.wo + (size_t)(2 * (i - 6) + sgm) * BLK3_BYTES;
            const int c = (i - 8) >> 2, r = (i - 8) & 3;
            return r < 2 ? Lw.w1 + (size_t)(4 * c + 2 * r + sgm) * BLK3_BYTES
                         : Lw.w2 + (size_t)((2 * (r - 2) + sgm) * 32 + 8 * c) * TFRAG_BYTES;
        });
    };
    auto advance = [&](int t) {
        ring.acquire(NS - 1 - t < 1 ? NS - 1 - t : 1);
        if (t + 2 < NS) issue(t + 2);
    };
    issue(0);
    issue(1);
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
        const float* xr = x + src_row * (size_t)F;
        const int KS = F / 16;
        for (int ks = 0; ks < KS; ++ks) {
            const int f0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
            const Tri xf = load_x_tri(xr + f0, valid);
#pragma unroll
            for (int nb = 0; nb < 4; nb += 2) {
                const Tri w0 = ldtri(M.win + (size_t)(nb * KS + ks) * TFRAG_BYTES + lane * 16);
                const Tri w1 = ldtri(M.win + (size_t)((nb + 1) * KS + ks) * TFRAG_BYTES + lane * 16);
                mfma6x2<false>(hres[nb], hres[nb + 1], w0, w1, xf);
            }
        }
    }
    f32x4 xg[16];
    layernorm_regs(hres, xg);
    Tri xp[8];
    split_row(xg, xp);

#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const int t0 = 24 * l;
        const float* lb = lbias + l * LBIAS;
        const float *lb1 = lb, *lb2 = lb + DFF, *lbn = lb + DFF + D, *lbo = lb + DFF + 4 * D;
        // ---- Q (pre-scaled by log2(e)/sqrt(D)) and K in row layout -> triples
        Tri qp[8], kp[8];
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) acc[nbl] = bias_block(lbn + 32 * nbl, h);
        advance(t0);
        gemm_slot<false>(acc[0], acc[1], ring.slot(t0), xp, lane);
        advance(t0 + 1);
        gemm_slot<false>(acc[2], acc[3], ring.slot(t0 + 1), xp, lane);
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
            acc[nbl] *= qscale;
            qp[2 * nbl] = split_half(acc[nbl], 0);
            qp[2 * nbl + 1] = split_half(acc[nbl], 1);
        }
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) acc[nbl] = bias_block(lbn + D + 32 * nbl, h);
        advance(t0 + 2);
        gemm_slot<false>(acc[0], acc[1], ring.slot(t0 + 2), xp, lane);
        advance(t0 + 3);
        gemm_slot<false>(acc[2], acc[3], ring.slot(t0 + 3), xp, lane);
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
        advance(t0 + 4);
        gemm_slot<true>(acc[0], acc[1], ring.slot(t0 + 4), xp, lane);
        advance(t0 + 5);
        gemm_slot<true>(acc[2], acc[3], ring.slot(t0 + 5), xp, lane);
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
        advance(t0 + 6);
        gemm_slot<false>(h1[0], h1[1], ring.slot(t0 + 6), xp, lane);
        advance(t0 + 7);
        gemm_slot<false>(h1[2], h1[3], ring.slot(t0 + 7), xp, lane);
        layernorm_regs(h1, xg);
        split_row(xg, xp);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) h1[nb] += bias_block(lb2 + 32 * nb, h);
#pragma unroll 1
        for (int ch = 0; ch < 4; ++ch) {
            const int tc = t0 + 8 + 4 * ch;
#pragma unroll
            for (int nbl = 0; nbl < 4; ++nbl) acc[nbl] = bias_block(lb1 + 128 * ch + 32 * nbl, h);
            advance(tc);
            gemm_slot<false>(acc[0], acc[1], ring.slot(tc), xp, lane);
            advance(tc + 1);
            gemm_slot<false>(acc[2], acc[3], ring.slot(tc + 1), xp, lane);
            Tri ap[8];
#pragma unroll
            for (int nbl = 0; nbl < 4; ++nbl) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nbl][r] = fmaxf(acc[nbl][r], 0.0f);
                ap[2 * nbl] = split_half(acc[nbl], 0);
                ap[2 * nbl + 1] = split_half(acc[nbl], 1);
            }
            advance(tc + 2);
            gemm_slot<false>(h1[0], h1[1], ring.slot(tc + 2), ap, lane);
            advance(tc + 3);
            gemm_slot<false>(h1[2], h1[3], ring.slot(tc + 3), ap, lane);
        }
        layernorm_regs(hres, xg);
        if (l + 1 < L) split_row(xg, xp);
    }
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
}

}  // namespace fs
}  // namespace savad
