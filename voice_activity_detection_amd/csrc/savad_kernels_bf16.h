// savad_kernels_bf16.h -- bf16-operand variant of the forward pass (BASELINE.json configs[2..3]:
// bf16 weights / activations, fp32 accumulation, fp32 softmax and LayerNorm statistics, fp32
// residual stream).  Same row-layout idea as savad_kernels.h, on v_mfma_f32_32x32x16_bf16
// (32 cycles, 32768 FLOP: 16x the fp32 MFMA rate), with everything the kernels exchange stored in
// FRAGMENT-MAJOR order so that every load / store / DMA is a contiguous 1 KiB wave access:
//
//   "block"   = 32 data slots (one MFMA tile of rows).  T > 32: block (b, qb) holds frames
//               32qb .. 32qb+31 of sequence b; T <= 32: block holds floor(32/T) whole sequences.
//   K-step    = 16 features.  Lane (m, h) of a fragment holds 8 bf16: features
//               32kb + 16j + 8(e>>2) + 4h + (e&3), e = 0..7, for K-step ks = 2kb + j -- exactly
//               registers 8j..8j+7 of the MFMA C/D layout of a 32-feature block, so an accumulator
//               becomes the next GEMM's operand with 8 v_cvt_pk and no data movement.
//   q, k, ctx : [block][ks 8][lane 64][8 bf16]      (B / A operand fragments, 8 KiB per block)
//   vt        : [block][nbd 4][j 2][lane 64][8 bf16] (V^T fragments: lane = feature, 8 keys; obtained
//               for free by issuing the V projection with the MFMA operands swapped)
//   h         : [block][nb 4][g 4][lane 64][4 f32]   (residual stream, fp32)
//   weights   : [n-block][ks][lane 64][8 bf16], packed once by pack_weight_frags_kernel
#pragma once
#include "savad_kernels.h"

namespace savad {
namespace bf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define SAVAD_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int FRAG_BYTES = 1024;             // one K-step fragment of 32 rows: 64 lanes x 16 B
constexpr int BLK_BYTES = 8 * FRAG_BYTES;    // 32 rows x 128 features in bf16
constexpr int RING_BYTES = 4 * BLK_BYTES;    // one ring block = 128 output features x 128 k = 32 KiB
constexpr int HBLK_FLOATS = 32 * D;          // fp32 residual block

__device__ __forceinline__ bf16x8 ldfrag(const void* p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ void stfrag(void* p, bf16x8 v) { *reinterpret_cast<u32x4*>(p) = __builtin_bit_cast(u32x4, v); }

// registers 8j..8j+7 of a C-layout 32-feature block -> the fragment of K-step j of that block
__device__ __forceinline__ bf16x8 pack_half(const f32x16& v, int j) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)v[8 * j + e];
    return r;
}
// LayerNorm'ed row (xg[G][s] = feature 8G+4h+s) -> the 8 K-step fragments
__device__ __forceinline__ void pack_row(const f32x4 (&xg)[16], bf16x8 (&xp)[8]) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xp[ks][e] = (__bf16)xg[2 * ks + (e >> 2)][e & 3];
    }
}

// block-space slot -> (valid, flat row, frame index)
__device__ __forceinline__ bool slot_row(int B, int T, int blk, int m, size_t& row, int& t) {
    if (T > 32) {
        const int QB = (T + 31) / 32;
        const int b = blk / QB;
        t = 32 * (blk % QB) + m;
        row = (size_t)b * T + t;
        return b < B && t < T;
    }
    const int G = 32 / T;
    const int seq = blk * G + m / T;
    t = m % T;
    row = (size_t)seq * T + t;
    return m < G * T && seq < B;
}

__device__ __forceinline__ void load_hblock(f32x16 (&x)[4], const float* hb, int lane) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = ld4(hb + ((nb * 4 + g) * 64 + lane) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) x[nb][4 * g + s] += t[s];
        }
}
__device__ __forceinline__ void store_hblock(float* hb, const f32x16 (&x)[4], int lane) {
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

// ---- DMA of one 8 KiB contiguous segment (8 fragments) into LDS; each wave issues 2 of the 8
//      1-KiB instructions (i = 2w, 2w+1).  Source and LDS image are both lane-linear.
__device__ __forceinline__ void dma_seg8(const void* __restrict__ src /* wave-uniform */, void* lds_dst, int w, int lane) {
    if (SAVAD_ABLATE & 1) return;
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_dst;
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr) + 2048u * (unsigned)w;
    const char* base = reinterpret_cast<const char*>(src) + 2048 * w;
    const unsigned off0 = (unsigned)lane * 16u, off1 = off0 + 1024u;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_add_u32 m0, m0, 1024\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(off0), "v"(off1), "s"(base), "s"(m0v)
        : "memory");
}
// 32 KiB contiguous ring block
__device__ __forceinline__ void dma_ring32(const void* __restrict__ src, char* lds_dst, int w, int lane) {
#pragma unroll
    for (int sgm = 0; sgm < 4; ++sgm) dma_seg8(reinterpret_cast<const char*>(src) + sgm * BLK_BYTES, lds_dst + sgm * BLK_BYTES, w, lane);
}
// W2 chunk c: for every output block nb the 8 K-steps 8c..8c+7 (K = 512 -> 32 K-steps per n-block)
__device__ __forceinline__ void dma_ring_w2(const void* __restrict__ w2frag, int c, char* lds_dst, int w, int lane) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
        dma_seg8(reinterpret_cast<const char*>(w2frag) + (size_t)(nb * 32 + 8 * c) * FRAG_BYTES, lds_dst + nb * BLK_BYTES, w, lane);
}
__device__ __forceinline__ void ring_wait() {
    if (SAVAD_ABLATE & 2) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// acc[nbl] += W[ring n-block nbl] . x   (transposed form: lane = data row, registers = output features)
__device__ __forceinline__ void gemm_ring(f32x16 (&acc)[4], const char* ringblk, const bf16x8 (&xp)[8], int lane) {
#pragma unroll
    for (int nbl = 0; nbl < 4; ++nbl)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            acc[nbl] = SAVAD_MFMA_BF16(ldfrag(ringblk + ((nbl * 8 + ks) * 64 + lane) * 16), xp[ks], acc[nbl]);
}
// operands swapped: lane = output feature, registers = data rows (used for V^T)
__device__ __forceinline__ void gemm_ring_swapped(f32x16 (&acc)[4], const char* ringblk, const bf16x8 (&xp)[8], int lane) {
#pragma unroll
    for (int nbl = 0; nbl < 4; ++nbl)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            acc[nbl] = SAVAD_MFMA_BF16(xp[ks], ldfrag(ringblk + ((nbl * 8 + ks) * 64 + lane) * 16), acc[nbl]);
}

// LN -> fragments -> Q, K (transposed form) and V^T (swapped form); ring block 0 (Wq) must be in
// flight into ring buffer `first_buf`.
__device__ __forceinline__ void qkv_tail(const f32x4 (&xg)[16], const char* __restrict__ wqkv_frag, const float* lbq,
                                         char* __restrict__ qf, char* __restrict__ kf, char* __restrict__ vtf, int blk,
                                         char* ring, int first_buf, int w, int lane) {
    const int n = lane & 31, h = lane >> 5;
    bf16x8 xp[8];
    pack_row(xg, xp);
    char* dst[2] = {qf, kf};
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) {
        const int buf = (first_buf + rb) & 1;
        ring_wait();
        if (rb + 1 < 3) dma_ring32(wqkv_frag + (size_t)(rb + 1) * RING_BYTES, ring + (buf ^ 1) * RING_BYTES, w, lane);
        f32x16 acc[4];
        if (rb < 2) {
#pragma unroll
            for (int nbl = 0; nbl < 4; ++nbl) acc[nbl] = bias_block(lbq + D * rb + 32 * nbl, h);
            gemm_ring(acc, ring + buf * RING_BYTES, xp, lane);
#pragma unroll
            for (int nbl = 0; nbl < 4; ++nbl)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    stfrag(dst[rb] + ((size_t)blk * 8 + 2 * nbl + j) * FRAG_BYTES + lane * 16, pack_half(acc[nbl], j));
        } else {
#pragma unroll
            for (int nbl = 0; nbl < 4; ++nbl) {
                const float bv = lbq[2 * D + 32 * nbl + n];  // lane = output feature
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nbl][r] = bv;
            }
            gemm_ring_swapped(acc, ring + buf * RING_BYTES, xp, lane);
#pragma unroll
            for (int nbl = 0; nbl < 4; ++nbl)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    stfrag(vtf + ((size_t)blk * 8 + 2 * nbl + j) * FRAG_BYTES + lane * 16, pack_half(acc[nbl], j));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel 1 (bf16): input Linear + PE -> h (fp32) -> LN -> Q, K, V^T fragments.  4 waves = 4 blocks.
// XT = float or __bf16 input features.
// ---------------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(256, 2) void input_qkv_kernel_bf16(
    const XT* __restrict__ x, int B, int T, int F, int nblk, const char* __restrict__ win_frag,
    const float* __restrict__ bin, const float* __restrict__ pe, const char* __restrict__ wqkv_frag,
    const float* __restrict__ bqkv, float* __restrict__ hbuf, char* __restrict__ qf, char* __restrict__ kf,
    char* __restrict__ vtf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* lbq = reinterpret_cast<float*>(smem + 2 * RING_BYTES);
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * 4 + w;
    dma_ring32(wqkv_frag, ring, w, lane);
    stage_bias(lbq, bqkv, 3 * D);
    size_t row;
    int t;
    const bool valid = (blk < nblk) && slot_row(B, T, blk, m, row, t);
    if (!valid) {
        row = 0;
        t = 0;
    }
    f32x16 h0[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        h0[nb] = zero16();
        add_bias(h0[nb], bin + 32 * nb, h);
        add_block(h0[nb], pe + (size_t)t * D + 32 * nb, h);
    }
    const int KS = F / 16;
    const XT* xr = x + row * (size_t)F;
    for (int ks = 0; ks < KS; ++ks) {
        const int f0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
        bf16x8 xf;
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[e] = valid ? (__bf16)(float)xr[f0 + 8 * (e >> 2) + (e & 3)] : (__bf16)0.0f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
            h0[nb] = SAVAD_MFMA_BF16(ldfrag(win_frag + ((size_t)(nb * KS + ks) * 64 + lane) * 16), xf, h0[nb]);
    }
    store_hblock(hbuf + (size_t)blk * HBLK_FLOATS, h0, lane);
    f32x4 xg[16];
    layernorm_regs(h0, xg);
    qkv_tail(xg, wqkv_frag, lbq, qf, kf, vtf, blk, ring, 0, w, lane);
}

// ---------------------------------------------------------------------------------------------
// Kernel 2 (bf16): flash attention on fragments.  T > 32: workgroup = (sequence, group of <= 4
// query blocks); K and V^T fragments of 2 key blocks (64 keys) per stage are DMA'd into LDS,
// double-buffered.  Output: NORMALISED context as B-operand fragments.
// ---------------------------------------------------------------------------------------------
struct AttnState {
    f32x16 O[4];
    float m_run, l_run;
};
__device__ __forceinline__ void attn_tile(AttnState& st, const bf16x8 (&qp)[8], const char* kblk, const char* vtblk,
                                          bool lds, const bool (&keyok)[16], bool need_mask, float c, int lane) {
    (void)lds;
    f32x16 sc = zero16();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) sc = SAVAD_MFMA_BF16(ldfrag(kblk + (ks * 64 + lane) * 16), qp[ks], sc);
    if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = keyok[r] ? sc[r] : NEG_BIG;
    }
    online_softmax(sc, st.m_run, st.l_run, st.O, c);
    const bf16x8 p0 = pack_half(sc, 0), p1 = pack_half(sc, 1);
#pragma unroll
    for (int nbd = 0; nbd < 4; ++nbd) {
        st.O[nbd] = SAVAD_MFMA_BF16(ldfrag(vtblk + ((nbd * 2 + 0) * 64 + lane) * 16), p0, st.O[nbd]);
        st.O[nbd] = SAVAD_MFMA_BF16(ldfrag(vtblk + ((nbd * 2 + 1) * 64 + lane) * 16), p1, st.O[nbd]);
    }
}
// Invalid query slots (padding of the block space) get an exactly-zero context: a fully masked row
// would otherwise carry inf/NaN into h, K and V^T of that slot and poison the NEXT layer's PV product
// (probability 0 x NaN).
__device__ __forceinline__ void store_ctx(char* ctxf, int blk, AttnState& st, bool qvalid, int lane) {
    const float inv = 1.0f / st.l_run;
#pragma unroll
    for (int nbd = 0; nbd < 4; ++nbd) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st.O[nbd][r] = qvalid ? st.O[nbd][r] * inv : 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) stfrag(ctxf + ((size_t)blk * 8 + 2 * nbd + j) * FRAG_BYTES + lane * 16, pack_half(st.O[nbd], j));
    }
}

__global__ __launch_bounds__(256, 2) void attention_kernel_bf16(const char* __restrict__ qf, const char* __restrict__ kf,
                                                                const char* __restrict__ vtf, char* __restrict__ ctxf,
                                                                int B, int T, int NG, float c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][K 2 blocks | VT 2 blocks] = 64 KiB
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int QB = (T + 31) / 32;
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int b = (i / NG) * 8 + xcd;
    if (b >= B) return;
    const int g = i % NG;
    const int qb0 = (g * QB) / NG, qb1 = ((g + 1) * QB) / NG;
    const int qb = qb0 + w;
    const bool active = qb < qb1;
    const int blk_q = b * QB + (active ? qb : qb0);
    const int NST = (QB + 1) / 2;

    bf16x8 qp[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qp[ks] = ldfrag(qf + ((size_t)blk_q * 8 + ks) * FRAG_BYTES + lane * 16);
    AttnState st;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) st.O[nb] = zero16();
    st.m_run = NEG_BIG;
    st.l_run = 0.0f;

    auto issue = [&](int stage, int buf) {
        const size_t kb0 = (size_t)b * QB + 2 * stage;
        char* dst = smem + buf * 4 * BLK_BYTES;
        dma_seg8(kf + kb0 * BLK_BYTES, dst, w, lane);
        dma_seg8(kf + (kb0 + 1) * BLK_BYTES, dst + BLK_BYTES, w, lane);
        dma_seg8(vtf + kb0 * BLK_BYTES, dst + 2 * BLK_BYTES, w, lane);
        dma_seg8(vtf + (kb0 + 1) * BLK_BYTES, dst + 3 * BLK_BYTES, w, lane);
    };
#ifdef SAVAD_TIMING
    long long tacc[4] = {0, 0, 0, 0}, tp = __builtin_readcyclecounter(), tn;
#define SAVAD_TB(i) do { tn = __builtin_readcyclecounter(); tacc[i] += tn - tp; tp = tn; } while (0)
#else
#define SAVAD_TB(i) do {} while (0)
#endif
    issue(0, 0);
    for (int stg = 0; stg < NST; ++stg) {
        SAVAD_TB(3);
        ring_wait();
        SAVAD_TB(0);
        if (stg + 1 < NST) issue(stg + 1, (stg + 1) & 1);
        SAVAD_TB(1);
        if (!active) continue;
        const char* buf = smem + (stg & 1) * 4 * BLK_BYTES;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int jt = 2 * stg + tt;
            if (jt >= QB) break;
            bool keyok[16];
            const bool need_mask = 32 * jt + 32 > T;
#pragma unroll
            for (int r = 0; r < 16; ++r) keyok[r] = (32 * jt + 8 * (r >> 2) + 4 * h + (r & 3)) < T;
            attn_tile(st, qp, buf + tt * BLK_BYTES, buf + (2 + tt) * BLK_BYTES, true, keyok, need_mask, c, lane);
        }
        SAVAD_TB(2);
    }
#ifdef SAVAD_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i2 = 0; i2 < 4; ++i2) g_savad_dbg[24 + i2] = tacc[i2];
#endif
    if (active) store_ctx(ctxf, blk_q, st, 32 * qb + (lane & 31) < T, lane);
}

// T <= 32: each block (floor(32/T) sequences) attends to itself with a block-diagonal mask.
__global__ __launch_bounds__(256, 2) void attention_packed_kernel_bf16(const char* __restrict__ qf,
                                                                       const char* __restrict__ kf,
                                                                       const char* __restrict__ vtf,
                                                                       char* __restrict__ ctxf, int B, int T, int nblk,
                                                                       float c) {
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * 4 + w;
    if (blk >= nblk) return;
    const int G = 32 / T;
    bf16x8 qp[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qp[ks] = ldfrag(qf + ((size_t)blk * 8 + ks) * FRAG_BYTES + lane * 16);
    AttnState st;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) st.O[nb] = zero16();
    st.m_run = NEG_BIG;
    st.l_run = 0.0f;
    bool keyok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int jk = 8 * (r >> 2) + 4 * h + (r & 3);
        keyok[r] = (jk < G * T) && (jk / T == m / T) && (blk * G + jk / T < B);
    }
    attn_tile(st, qp, kf + (size_t)blk * BLK_BYTES, vtf + (size_t)blk * BLK_BYTES, false, keyok, true, c, lane);
    store_ctx(ctxf, blk, st, (m < G * T) && (blk * G + m / T < B), lane);
}

// ---------------------------------------------------------------------------------------------
// Kernel 3 (bf16, per layer): out-projection + residual -> LN -> FFN (4 chunks of 128 hidden units,
// ReLU output repacked in registers) + residual -> next layer's LN + Q/K/V^T, or the classifier.
// Weight stream: 12 ring blocks of 32 KiB through a 2 x 32 KiB LDS ring, one block ahead.
// ---------------------------------------------------------------------------------------------
template <bool LAST>
__global__ __launch_bounds__(256, 2) void row_kernel_bf16(
    const char* __restrict__ ctxf, int B, int T, int nblk, float* __restrict__ hbuf, const char* __restrict__ wo_frag,
    const float* __restrict__ bo, const char* __restrict__ w1_frag, const float* __restrict__ b1,
    const char* __restrict__ w2_frag, const float* __restrict__ b2, const char* __restrict__ wn_frag /* !LAST: Wqkv' */,
    const float* __restrict__ wc /* LAST: Wc' fp32 [2][D] */, const float* __restrict__ bn, char* __restrict__ qf,
    char* __restrict__ kf, char* __restrict__ vtf, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* lbo = reinterpret_cast<float*>(smem + 2 * RING_BYTES);
    float* lb1 = lbo + D;
    float* lb2 = lb1 + DFF;
    float* lbn = lb2 + D;
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * 4 + w;
    dma_ring32(wo_frag, ring, w, lane);
    stage_bias(lbo, bo, D);
    stage_bias(lb1, b1, DFF);
    stage_bias(lb2, b2, D);
    if (!LAST) stage_bias(lbn, bn, 3 * D);
    float* hb = hbuf + (size_t)blk * HBLK_FLOATS;
    f32x16 h1[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) h1[nb] = zero16();
    load_hblock(h1, hb, lane);
    bf16x8 xp[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        xp[ks] = ldfrag(ctxf + ((size_t)blk * 8 + ks) * FRAG_BYTES + lane * 16);
        if (blk >= nblk) xp[ks] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});  // pad blocks: attention never wrote them
    }
    // ---- h1 = h + bo + ctx Wo^T
    ring_wait();
    dma_ring32(w1_frag, ring + RING_BYTES, w, lane);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) h1[nb] += bias_block(lbo + 32 * nb, h);
    gemm_ring(h1, ring, xp, lane);
    store_hblock(hb, h1, lane);  // park the residual stream (fp32) while the FFN runs
    f32x4 xg[16];
    layernorm_regs(h1, xg);
    pack_row(xg, xp);
    // ---- FFN
    f32x16 o[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) o[nb] = zero16();
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
        ring_wait();  // W1 chunk in ring buffer 1
        dma_ring_w2(w2_frag, ch, ring, w, lane);
        f32x16 a[4];
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) a[nbl] = bias_block(lb1 + 128 * ch + 32 * nbl, h);
        gemm_ring(a, ring + RING_BYTES, xp, lane);
        bf16x8 ap[8];
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[nbl][r] = fmaxf(a[nbl][r], 0.0f);
            ap[2 * nbl] = pack_half(a[nbl], 0);
            ap[2 * nbl + 1] = pack_half(a[nbl], 1);
        }
        ring_wait();  // W2 chunk in ring buffer 0
        if (ch + 1 < 4)
            dma_ring32(w1_frag + (size_t)(ch + 1) * RING_BYTES, ring + RING_BYTES, w, lane);
        else if (!LAST)
            dma_ring32(wn_frag, ring + RING_BYTES, w, lane);
        gemm_ring(o, ring, ap, lane);
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) o[nb] += bias_block(lb2 + 32 * nb, h);
    load_hblock(o, hb, lane);  // residual onto the un-normalised stream
    if (!LAST) store_hblock(hb, o, lane);
    layernorm_regs(o, xg);
    if (!LAST) {
        qkv_tail(xg, wn_frag, lbn, qf, kf, vtf, blk, ring, 1, w, lane);
    } else {
        float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
        for (int G = 0; G < 16; ++G) {
            const f32x4 c0 = ld4(wc + 8 * G + 4 * h), c1 = ld4(wc + D + 8 * G + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z0 = __builtin_fmaf(xg[G][e], c0[e], z0);
                z1 = __builtin_fmaf(xg[G][e], c1[e], z1);
            }
        }
        z0 = half_sum(z0) + bn[0];
        z1 = half_sum(z1) + bn[1];
        const float mx = fmaxf(z0, z1);
        const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
        size_t row;
        int t;
        const bool valid = (blk < nblk) && slot_row(B, T, blk, m, row, t);
        if (h == 0 && valid) *reinterpret_cast<f32x2*>(out + row * 2) = f32x2{z0 - lse, z1 - lse};
    }
}

// ---------------------------------------------------------------------------------------------
// Weight packing: fp32 [N][K] (LayerNorm already folded) -> bf16 fragments [N/32][K/16][64][8].
// ---------------------------------------------------------------------------------------------
__global__ void pack_weight_frags_kernel(const float* __restrict__ W, int N, int K, __bf16* __restrict__ out) {
    const int KS = K / 16;
    const size_t total = (size_t)(N / 32) * KS * 64 * 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const size_t f = i >> 9;
        const int ks = (int)(f % KS);
        const int nblk = (int)(f / KS);
        const int n = lane & 31, h = lane >> 5;
        const int k = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (e >> 2) + 4 * h + (e & 3);
        out[i] = (__bf16)W[(size_t)(32 * nblk + n) * K + k];
    }
}

}  // namespace bf
}  // namespace savad
