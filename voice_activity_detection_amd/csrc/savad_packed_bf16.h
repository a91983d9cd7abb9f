// savad_packed_bf16.h -- T <= 32 with bf16 operands: the WHOLE forward in one launch (round 5).
//
// The reference pipeline only ever runs 7-frame windows (vad/predictor.py:180-224), 99 % of whose FLOPs are the
// projections and the FFN -- the case the bf16 MFMA (16x the fp32 rate) helps most.  A WAVE owns one packed block
// (floor(32/T) whole sequences in the 32 slots of an MFMA tile; savad_kernels_bf16.h) for ALL layers: the block's
// Q, K and V^T are produced by the wave that consumes them, so the attention never leaves its registers (S^T = K Q^T
// 8 MFMAs, softmax lane-local, O^T = V^T P^T 8 MFMAs), the residual stream waits in registers as packed fp16 while the
// attention runs, and nothing but x, the weight stream and the log-probabilities crosses the CU boundary.  The NW
// waves of a workgroup share the weight stream -- 12 ring blocks of 32 KiB per layer (Wq Wk Wv Wo, then W1 / W2 chunks
// alternating), staged by global->LDS DMA exactly as in the per-layer kernels -- and every layer's biases in LDS.
//
// Arithmetic, operand for operand, is that of the per-layer bf16 launches (input_qkv_kernel_bf16 ->
// attention_packed_kernel_bf16 -> row_kernel_bf16): same fragments, same accumulation order, the residual stream
// rounded to fp16 at the same two points per layer.  The results are the same bits (tests/test_gpu_parity.py).
//
// Windowed mode (wo.w == T > 0): x is the predictor's feature MATRIX [N][F] and sequence s is its window
// feature[win_base + s + wo.off[0..T-1]] (vad/predictor.py:180-220): the gather is an address computation.
#pragma once
#include "savad_kernels_bf16.h"

namespace savad {
namespace bf {

constexpr int PACKED_BF16_MAX_LAYERS = 6;  // ring (NW = 8: 128 KiB) + 6 x 4.5 KiB of biases fit the 160 KiB of LDS

struct PackedBf16Layer {
    const char *wqkv, *wo, *w1, *w2;  // fragment-major bf16 (pack_weight_frags_kernel), LayerNorm affine folded in
};
struct PackedBf16Model {
    PackedBf16Layer layer[PACKED_BF16_MAX_LAYERS];
    const char* win;    // input Linear fragments [4][F/16]
    const float* bin;   // input bias
    const float* pe;    // positional encoding / sqrt(D)
    const float* bias;  // [L][LBIAS]: b1' | b2 | bqkv' | bo of every layer
    const float *wc, *bc;
    int L;
};

typedef unsigned u32x8 __attribute__((ext_vector_type(8)));

// the residual stream parked as fp16 (what store_hblock / load_hblock do through HBM between the per-layer launches)
__device__ __forceinline__ void park_h(u32x4 (&hp)[8], const f32x16 (&x)[4], unsigned* __restrict__ satcnt) {
    float amax = 0.0f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            f32x8 f;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                amax = fmaxf(amax, fabsf(x[nb][8 * gp + s]));
                f[s] = fminf(fmaxf(x[nb][8 * gp + s], -65504.0f), 65504.0f);
            }
            hp[nb * 2 + gp] = __builtin_bit_cast(u32x4, __builtin_convertvector(f, f16x8));
        }
    if (__any(!(amax <= 65504.0f))) {
        unsigned c = 0;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) c += !(fabsf(x[nb][r]) <= 65504.0f);
        if (c) atomicAdd(satcnt, c);
    }
}
__device__ __forceinline__ void unpark_h(f32x16 (&x)[4], const u32x4 (&hp)[8]) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            const f32x8 f = __builtin_convertvector(__builtin_bit_cast(f16x8, hp[nb * 2 + gp]), f32x8);
#pragma unroll
            for (int s = 0; s < 8; ++s) x[nb][8 * gp + s] = 0.0f + f[s];  // (load_hblock adds onto a zeroed accumulator)
        }
}

// NR = ring slots (the DMA runs NR - 1 blocks ahead); DW = waves that do nothing but move the weight stream.
//   <4, 2, 0>  two 4-wave workgroups per CU cover for each other's ring steps: the throughput regime
//   <4, 4, 4>  fewer workgroups than CUs (the reference's own 1000-window chunks: 250 blocks): a block's chain is alone on its
//              SIMD, and the 288 DMA instructions per forward it would issue itself (~66 cycles each, 15 % of its time:
//              scripts/ubench/phase_timing_packed_bf16.py) go to four extra waves, three ring slots ahead
//   <8, 4, 0>  8-wave workgroups (tuning knob)
template <int NW, int NR, int DW>
__global__ __launch_bounds__(64 * (NW + DW), (NW + DW == 8) ? 1 : 2) void packed_forward_kernel_bf16(
    const float* __restrict__ x, int B, int T, int F, int nblk, PackedBf16Model M, float qscale, float* __restrict__ out,
    WindowOffsets wo, int win_base, unsigned* __restrict__ satcnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using R = Ring<NW, NR>;
    float* lbias = reinterpret_cast<float*>(smem + R::NRING * RING_BYTES);
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * NW + w;
    const bool live = blk < nblk;  // wave-uniform; a wave without a block still moves its share of the weight stream
    const bool mover = DW > 0 && w >= NW;                 // a wave that only moves the weight stream
    const R ring{smem, DW > 0 ? (w >= NW ? w - NW : w) : w, lane};
    const int L = M.L, NB = 12 * L;
    auto issue = [&](int t) {
        const int l = t / 12, i = t - 12 * l;
        const PackedBf16Layer Lw = M.layer[l];
        ring.issue(t, [&](int sgm) -> const char* {
            if (i < 3) return Lw.wqkv + (size_t)i * RING_BYTES + sgm * BLK_BYTES;
            if (i == 3) return Lw.wo + sgm * BLK_BYTES;
            const int c = (i - 4) >> 1;
            return ((i - 4) & 1) ? Lw.w2 + (size_t)(sgm * 32 + 8 * c) * FRAG_BYTES  // output block sgm, K-steps 8c..8c+7
                                 : Lw.w1 + (size_t)c * RING_BYTES + sgm * BLK_BYTES;
        });
    };
#ifdef SAVAD_TIMING
    long long pacc[4] = {0, 0, 0, 0}, pp = __builtin_readcyclecounter(), pn;
#define SAVAD_PACC(i) do { pn = __builtin_readcyclecounter(); pacc[i] += pn - pp; pp = pn; } while (0)
#else
#define SAVAD_PACC(i) do {} while (0)
#endif
    // acquire block t (wave-uniform), then keep the DMA DEPTH blocks ahead
    auto advance = [&](int t) {
        SAVAD_PACC(0);  // compute since the last ring step
        ring.acquire(NB - 1 - t < R::DEPTH - 1 ? NB - 1 - t : R::DEPTH - 1);
        SAVAD_PACC(1);  // waiting for the block and the other waves
        if ((DW == 0 || mover) && t + R::DEPTH < NB) issue(t + R::DEPTH);
        SAVAD_PACC(2);  // issuing the DMA
    };
    if (DW == 0 || mover) {
#pragma unroll
        for (int t = 0; t < R::DEPTH; ++t) issue(t);
    }
    for (int i = threadIdx.x * 4; i < L * LBIAS; i += 64 * (NW + DW) * 4) st4(lbias + i, ld4(M.bias + i));  // published by the first ring barrier
    float* lwc = lbias + L * LBIAS;   // the classifier's folded weights [2][D] + bias [2] behind the biases
    for (int i = threadIdx.x * 4; i < 2 * D; i += 64 * (NW + DW) * 4) st4(lwc + i, ld4(M.wc + i));
    if (threadIdx.x < 2) lwc[2 * D + threadIdx.x] = M.bc[threadIdx.x];
    if (mover) {  // one barrier per ring block, like the waves that compute
        for (int t = 0; t < NB; ++t) advance(t);
        return;
    }

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
    // ---- input Linear + positional encoding (self_attention.py:12-16,24), as input_qkv_kernel_bf16
    f32x16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        acc[nb] = zero16();
        add_bias(acc[nb], M.bin + 32 * nb, h);
        add_block(acc[nb], M.pe + (size_t)(valid ? t_frame : 0) * D + 32 * nb, h);
    }
    {
        const size_t src_row = wo.w > 0 ? (size_t)win_base + (valid ? seq : 0) + wo.off[valid ? t_frame : 0] : row;
        const float* xr = x + src_row * (size_t)F;
        const int KS = F / 16;
        // Every feature piece of the rows is requested before the first MFMA (up to 128 features), the input weights' fragments one
        // K-step ahead of the MFMAs that use them (round 5; before: load, wait, four MFMAs per K-step -- five dependent round trips at
        // the head of every block, the rows' cache lines among them).  The same MFMAs in the same order.
        constexpr int KSMAX = 8;
        bf16x8 xf[KSMAX], wcur[4], wnxt[4];
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            const int kc = ks < KS ? ks : KS - 1;   // past the end: a valid address, unused
            xf[ks] = load_x_frag(xr + 32 * (kc >> 1) + 16 * (kc & 1) + 4 * h, valid);
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) wcur[nb] = ldfrag(M.win + ((size_t)(nb * KS) * 64 + lane) * 16);
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            if (ks < KS) {   // wave-uniform
                const int kn = ks + 1 < KS ? ks + 1 : KS - 1;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) wnxt[nb] = ldfrag(M.win + ((size_t)(nb * KS + kn) * 64 + lane) * 16);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[nb] = SAVAD_MFMA_BF16(wcur[nb], xf[ks], acc[nb]);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) wcur[nb] = wnxt[nb];
            }
        }
        for (int ks = KSMAX; ks < KS; ++ks) {
            const int f0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
            const bf16x8 xfl = load_x_frag(xr + f0, valid);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                acc[nb] = SAVAD_MFMA_BF16(ldfrag(M.win + ((size_t)(nb * KS + ks) * 64 + lane) * 16), xfl, acc[nb]);
        }
    }
    u32x4 hp[8];
    park_h(hp, acc, satcnt);
    f32x4 xg[16];
    layernorm_regs(acc, xg);
    bf16x8 xp[8];
    pack_row(xg, xp);

#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const int t0 = 12 * l;
        const float* lb = lbias + l * LBIAS;
        const float *lb1 = lb, *lb2 = lb + DFF, *lbn = lb + DFF + D, *lbo = lb + DFF + 4 * D;
        // ---- Q (pre-scaled by log2(e)/sqrt(D)) and K in row layout -> fragments (qkv_block_bf16)
        bf16x8 qp[8], kp[8];
        advance(t0);
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) acc[nbl] = bias_block(lbn + 32 * nbl, h);
        gemm_ring(acc, ring.slot(t0), xp, lane);
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
            acc[nbl] *= qscale;
            qp[2 * nbl] = pack_half(acc[nbl], 0);
            qp[2 * nbl + 1] = pack_half(acc[nbl], 1);
        }
        advance(t0 + 1);
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) acc[nbl] = bias_block(lbn + D + 32 * nbl, h);
        gemm_ring(acc, ring.slot(t0 + 1), xp, lane);
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
            kp[2 * nbl] = pack_half(acc[nbl], 0);
            kp[2 * nbl + 1] = pack_half(acc[nbl], 1);
        }
        // ---- scores and softmax of the single key tile (attn_tile with first = true)
        AttnState st;
        st.negm = zero16();
        st.l_run = 0.0f;
        f32x16 sc = st.negm;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) sc = SAVAD_MFMA_BF16(kp[ks], qp[ks], sc);
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = keyok[r] ? sc[r] : NEG_BIG;
        {   // online_softmax_shifted(sc, st, true) without the accumulators it would rescale (they do not exist yet)
            float mx = sc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
            mx = half_max(mx);
            const bool move = (mx > RESCALE_LOG2) || (mx < -RESCALE_LOG2);
            if (__any(move)) {
                const float d = move ? mx : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] -= d;
            }
            float rs = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = __builtin_amdgcn_exp2f(sc[r]);
                rs += sc[r];
            }
            st.l_run += rs;
        }
        const bf16x8 p0 = pack_half(sc, 0), p1 = pack_half(sc, 1);
        // ---- V^T (operands swapped: lane = feature, registers = keys) and O^T = V^T P^T, normalised -> the context fragments
        advance(t0 + 2);
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
            const float bv = lbn[2 * D + 32 * nbl + m];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nbl][r] = bv;
        }
        gemm_ring_swapped(acc, ring.slot(t0 + 2), xp, lane);
        {
            const float inv = 1.0f / half_sum(st.l_run);
#pragma unroll
            for (int nbd = 0; nbd < 4; ++nbd) {
                f32x16 O = zero16();
                O = SAVAD_MFMA_BF16(pack_half(acc[nbd], 0), p0, O);
                O = SAVAD_MFMA_BF16(pack_half(acc[nbd], 1), p1, O);
#pragma unroll
                for (int r = 0; r < 16; ++r) O[r] = valid ? O[r] * inv : 0.0f;
                xp[2 * nbd] = pack_half(O, 0);
                xp[2 * nbd + 1] = pack_half(O, 1);
            }
        }
        // ---- h1 = h + bo + ctx Wo^T; LN; FFN on top of the residual stream (row_stage_bf16)
        f32x16(&h1)[4] = acc;
        unpark_h(h1, hp);
        advance(t0 + 3);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) h1[nb] += bias_block(lbo + 32 * nb, h);
        gemm_ring(h1, ring.slot(t0 + 3), xp, lane);
        layernorm_regs(h1, xg);
        pack_row(xg, xp);
        f32x16(&o)[4] = h1;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) o[nb] += bias_block(lb2 + 32 * nb, h);
#pragma unroll 1
        for (int ch = 0; ch < 4; ++ch) {
            advance(t0 + 4 + 2 * ch);  // W1 chunk
            f32x16 a[4];
#pragma unroll
            for (int nbl = 0; nbl < 4; ++nbl) a[nbl] = bias_block(lb1 + 128 * ch + 32 * nbl, h);
            gemm_ring(a, ring.slot(t0 + 4 + 2 * ch), xp, lane);
            bf16x8 ap[8];
#pragma unroll
            for (int nbl = 0; nbl < 4; ++nbl) {
                ap[2 * nbl] = relu_frag(pack_half(a[nbl], 0));
                ap[2 * nbl + 1] = relu_frag(pack_half(a[nbl], 1));
            }
            advance(t0 + 5 + 2 * ch);  // W2 chunk
            gemm_ring(o, ring.slot(t0 + 5 + 2 * ch), ap, lane);
        }
        if (l + 1 < L) park_h(hp, o, satcnt);
        layernorm_regs(o, xg);
        if (l + 1 < L) pack_row(xg, xp);
    }
    // ---- final LayerNorm (folded into the classifier) + Linear(D, 2) + log-softmax (self_attention.py:26-28)
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
#ifdef SAVAD_TIMING
    SAVAD_PACC(0);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 4; ++i) g_savad_dbg[32 + i] = pacc[i];
#endif
}


// ---------------------------------------------------------------------------------------------
// The LATENCY variant (round 5): ONE packed block per workgroup, its four waves split every GEMM's OUTPUT features (wave w:
// features [32w, 32w + 32) of Q, K, V^T, the out-projection, FFN2; hidden units [128w, 128w + 128) of FFN1) -- a block's chain is
// 106 MFMAs per wave and layer instead of 400, and 250 blocks (the reference's 1000-window chunk) run on 250 CUs instead of 63.
// Every output element is still accumulated over the same K-steps in the same order from the same operands as in the wave-per-block
// kernel above, so the results are THE SAME BITS; what the waves do not own they get through LDS:
//   full rows for the two LayerNorms (fp32, as packed_forward_kernel), the eight Q / K fragments for the scores (every wave
//   computes the whole 32 x 32 score tile: 8 MFMAs, redundant but identical), the eight context fragments for the out-projection,
//   the 32 ReLU'd hidden fragments for FFN2 -- five barriers per layer.
// Weights: a wave reads only ITS fragments, straight from L2 into AGPRs in blocks of eight (8 KiB), requested by hand one block
// (8 MFMAs... the previous block's) ahead with counted vmcnt waits (wload_frag / wwait of savad_kernels.h, bf16 edition; no LDS
// ring, no DMA: 1.18 MB per workgroup and forward from the L2s).
// ---------------------------------------------------------------------------------------------
struct WB8 {
    u32x4 v[8];
};
__device__ __forceinline__ void wload8(WB8& wb, const char* __restrict__ base /* wave-uniform: 8 consecutive fragments */, unsigned voff /* lane * 16 */) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(wb.v[i]) : "v"(voff), "s"(base + (i >> 2) * 4096), "n"((i & 3) * 1024));
}
template <int PENDING>  // younger requests allowed to stay in flight (8 per block)
__device__ __forceinline__ void wwait8(WB8& wb) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+a"(wb.v[0]), "+a"(wb.v[1]), "+a"(wb.v[2]), "+a"(wb.v[3]), "+a"(wb.v[4]), "+a"(wb.v[5]), "+a"(wb.v[6]), "+a"(wb.v[7])
                 : "n"(PENDING));
}
__device__ __forceinline__ bf16x8 wfrag(const WB8& wb, int i) { return __builtin_bit_cast(bf16x8, wb.v[i]); }

constexpr int NS_XB_FLOATS = TILE * XLD;                 // one fp32 row-exchange buffer
constexpr int NS_FRAG_BYTES = 32 * FRAG_BYTES;           // 32 fragments: the hidden activations; Q / K (16) and the context (8) alias it
inline constexpr int ns_lds_bytes(int L) { return 2 * NS_XB_FLOATS * 4 + NS_FRAG_BYTES + (L * LBIAS + 2 * D + 4) * 4; }

__global__ __launch_bounds__(256, 1) void packed_forward_kernel_bf16_ns(const float* __restrict__ x, int B, int T, int F, int nblk,
                                                                        PackedBf16Model M, float qscale, float* __restrict__ out,
                                                                        WindowOffsets wo, int win_base, unsigned* __restrict__ satcnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xb0 = reinterpret_cast<float*>(smem);
    float* xb1 = xb0 + NS_XB_FLOATS;
    char* fbuf = smem + 2 * NS_XB_FLOATS * 4;
    float* lbias = reinterpret_cast<float*>(fbuf + NS_FRAG_BYTES);
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned voff = (unsigned)lane * 16u;
    const int blk = blockIdx.x, L = M.L;
    SAVAD_STAMP(40);
    float* lwc = lbias + L * LBIAS;  // the classifier's folded weights [2][D] + bias [2], staged with the biases

    const int G = 32 / T, seq = blk * G + m / T, t_frame = m % T;
    const bool valid = blk < nblk && m < G * T && seq < B;
    const size_t row = valid ? (size_t)seq * T + t_frame : 0;
    bool keyok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int jk = 8 * (r >> 2) + 4 * h + (r & 3);
        keyok[r] = (jk < G * T) && (jk / T == m / T) && (blk * G + jk / T < B);
    }
    // ---- input Linear + positional encoding: this wave's 32 features.  Every global load of the prologue is requested before the
    // first use of any (the features, the input weights' fragments, bias and PE rows, then the biases and the classifier for LDS).
    // (scripts/ubench/phase_timing_packed_bf16_ns.py: the prologue is 10 k of the forward's 57-60 k cycles either way -- kernel
    // arguments, the window offsets, 17 integer divisions for the mask -- and the order of its loads does not move it.)
    f32x16 own = zero16();
    {
        const size_t src_row = wo.w > 0 ? (size_t)win_base + (valid ? seq : 0) + wo.off[valid ? t_frame : 0] : row;
        const float* xr = x + src_row * (size_t)F;
        const int KS = F / 16;
        constexpr int KSMAX = 8;  // fragments requested up front (F <= 128; beyond that the loop below loads as it goes)
        bf16x8 xf[KSMAX], wf[KSMAX];
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            const int kc = ks < KS ? ks : KS - 1;
            xf[ks] = load_x_frag(xr + 32 * (kc >> 1) + 16 * (kc & 1) + 4 * h, valid);
            wf[ks] = ldfrag(M.win + ((size_t)(w * KS + kc) * 64 + lane) * 16);
        }
        add_bias(own, M.bin + 32 * w, h);
        add_block(own, M.pe + (size_t)(valid ? t_frame : 0) * D + 32 * w, h);
        constexpr int NBT = (PACKED_BF16_MAX_LAYERS * LBIAS / 4 + 255) / 256 + 1;
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
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks)
            if (ks < KS) own = SAVAD_MFMA_BF16(wf[ks], xf[ks], own);
        for (int ks = KSMAX; ks < KS; ++ks) {
            const int f0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
            own = SAVAD_MFMA_BF16(ldfrag(M.win + ((size_t)(w * KS + ks) * 64 + lane) * 16), load_x_frag(xr + f0, valid), own);
        }
#pragma unroll
        for (int j = 0; j < NBT; ++j) {
            const int i4 = (int)threadIdx.x + 256 * j;
            if (i4 < nb4) st4(lbias + 4 * i4, bt[j]);   // published by the first barrier
        }
    }
    // the residual stream between layers is what fp16 holds (park_h / unpark_h of the wave-per-block kernel, one 32-feature block)
    auto park = [&](u32x4 (&hp)[2], const f32x16& v) {
        float amax = 0.0f;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            f32x8 f;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                amax = fmaxf(amax, fabsf(v[8 * gp + s]));
                f[s] = fminf(fmaxf(v[8 * gp + s], -65504.0f), 65504.0f);
            }
            hp[gp] = __builtin_bit_cast(u32x4, __builtin_convertvector(f, f16x8));
        }
        if (__any(!(amax <= 65504.0f))) {
            unsigned c = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) c += !(fabsf(v[r]) <= 65504.0f);
            if (c) atomicAdd(satcnt, c);
        }
    };
    // full rows through LDS -> the wave-per-block kernel's layernorm_regs on the same register image -> the 8 K-step fragments
    auto rows_ln = [&](float* xb, const f32x16& mine, f32x4 (&xg)[16], bf16x8 (&xp)[8], bool pack) {
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
        if (pack) pack_row(xg, xp);
    };
    u32x4 hp[2];
    park(hp, own);
    f32x4 xg[16];
    bf16x8 xp[8];
    // Weight stream of this wave: 12 blocks of 8 fragments per layer -- 0 Wq, 1 Wk, 2 Wv, 3 Wo, 4..7 W1 (hidden blocks 4w..4w+3),
    // 8..11 W2 (output block w, K-steps 8(i-8)..) -- through FOUR register buffers, block i in buffer i & 3, every block requested
    // THREE blocks before its use: a block is only 8 MFMAs (256 cycles) of work, an L2 round trip several times that.
    WB8 wq[4];
    auto block_addr = [&](int l, int i) -> const char* {
        const PackedBf16Layer Lw = M.layer[l];
        return i < 3 ? Lw.wqkv + (size_t)(i * 4 + w) * BLK_BYTES
             : i == 3 ? Lw.wo + (size_t)w * BLK_BYTES
             : i < 8 ? Lw.w1 + (size_t)(4 * w + i - 4) * BLK_BYTES
                     : Lw.w2 + (size_t)(w * 32 + 8 * (i - 8)) * FRAG_BYTES;
    };
#define SAVAD_NS_REQ(l_, i_) wload8(wq[(i_) & 3], block_addr((i_) < 12 ? (l_) : ((l_) + 1 < L ? (l_) + 1 : (l_)), (i_) < 12 ? (i_) : (i_) - 12), voff)
#define SAVAD_NS_GET(i_) wwait8<24>(wq[(i_) & 3])   /* three younger blocks may stay in flight */
    SAVAD_STAMP(41);
    SAVAD_NS_REQ(0, 0);
    SAVAD_NS_REQ(0, 1);
    SAVAD_NS_REQ(0, 2);
    rows_ln(xb0, own, xg, xp, true);
    SAVAD_STAMP(42);

#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const float* lb = lbias + l * LBIAS;
        const float *lb1 = lb, *lb2 = lb + DFF, *lbn = lb + DFF + D, *lbo = lb + DFF + 4 * D;
        // ---- Q (pre-scaled), K -> fragments 2w, 2w + 1 of the exchange; V^T of this wave's features stays here
        SAVAD_NS_REQ(l, 3);
        f32x16 acc = bias_block(lbn + 32 * w, h);
        SAVAD_NS_GET(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = SAVAD_MFMA_BF16(wfrag(wq[0], ks), xp[ks], acc);
        acc *= qscale;
        stfrag(fbuf + (2 * w + 0) * FRAG_BYTES + lane * 16, pack_half(acc, 0));
        stfrag(fbuf + (2 * w + 1) * FRAG_BYTES + lane * 16, pack_half(acc, 1));
        SAVAD_NS_REQ(l, 4);
        acc = bias_block(lbn + D + 32 * w, h);
        SAVAD_NS_GET(1);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = SAVAD_MFMA_BF16(wfrag(wq[1], ks), xp[ks], acc);
        stfrag(fbuf + (8 + 2 * w + 0) * FRAG_BYTES + lane * 16, pack_half(acc, 0));
        stfrag(fbuf + (8 + 2 * w + 1) * FRAG_BYTES + lane * 16, pack_half(acc, 1));
        SAVAD_NS_REQ(l, 5);
        {
            const float bv = lbn[2 * D + 32 * w + m];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bv;
        }
        SAVAD_NS_GET(2);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = SAVAD_MFMA_BF16(xp[ks], wfrag(wq[2], ks), acc);
        const bf16x8 vt0 = pack_half(acc, 0), vt1 = pack_half(acc, 1);
        SAVAD_NS_REQ(l, 6);
        SAVAD_STAMP(43);
        __syncthreads();  // Q and K fragments of all four waves
        SAVAD_STAMP(44);
        // ---- the whole score tile in every wave (same operands, same order: the same bits), softmax, this wave's O^T block
        f32x16 sc = zero16();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            sc = SAVAD_MFMA_BF16(ldfrag(fbuf + (8 + ks) * FRAG_BYTES + lane * 16), ldfrag(fbuf + ks * FRAG_BYTES + lane * 16), sc);
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = keyok[r] ? sc[r] : NEG_BIG;
        float l_run;
        {
            float mx = sc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
            mx = half_max(mx);
            const bool move = (mx > RESCALE_LOG2) || (mx < -RESCALE_LOG2);
            if (__any(move)) {
                const float d = move ? mx : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] -= d;
            }
            float rs = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = __builtin_amdgcn_exp2f(sc[r]);
                rs += sc[r];
            }
            l_run = 0.0f + rs;
        }
        const bf16x8 p0 = pack_half(sc, 0), p1 = pack_half(sc, 1);
        {
            const float inv = 1.0f / half_sum(l_run);
            f32x16 O = zero16();
            O = SAVAD_MFMA_BF16(vt0, p0, O);
            O = SAVAD_MFMA_BF16(vt1, p1, O);
#pragma unroll
            for (int r = 0; r < 16; ++r) O[r] = valid ? O[r] * inv : 0.0f;
            __syncthreads();  // every wave has read the Q / K fragments: the context fragments may take their place
            stfrag(fbuf + (2 * w + 0) * FRAG_BYTES + lane * 16, pack_half(O, 0));
            stfrag(fbuf + (2 * w + 1) * FRAG_BYTES + lane * 16, pack_half(O, 1));
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xp[ks] = ldfrag(fbuf + ks * FRAG_BYTES + lane * 16);
        SAVAD_STAMP(45);
        // ---- h1 = h + bo + ctx Wo^T (this wave's block), LN2
        f32x16 h1;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            const f32x8 f = __builtin_convertvector(__builtin_bit_cast(f16x8, hp[gp]), f32x8);
#pragma unroll
            for (int s = 0; s < 8; ++s) h1[8 * gp + s] = 0.0f + f[s];
        }
        h1 += bias_block(lbo + 32 * w, h);
        SAVAD_NS_GET(3);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) h1 = SAVAD_MFMA_BF16(wfrag(wq[3], ks), xp[ks], h1);
        SAVAD_NS_REQ(l, 7);
        SAVAD_STAMP(46);
        rows_ln(xb1, h1, xg, xp, true);  // (its barrier also retires the context fragments' readers)
        SAVAD_STAMP(47);
        // ---- FFN1: hidden units [128 w, 128 w + 128) in four blocks, ReLU'd fragments 8 w .. 8 w + 7 of the exchange
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            f32x16 a = bias_block(lb1 + 128 * w + 32 * ch, h);
            SAVAD_NS_GET(4 + ch);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) a = SAVAD_MFMA_BF16(wfrag(wq[ch], ks), xp[ks], a);
            SAVAD_NS_REQ(l, 8 + ch);   // the W2 block that reuses this buffer
            stfrag(fbuf + (8 * w + 2 * ch + 0) * FRAG_BYTES + lane * 16, relu_frag(pack_half(a, 0)));
            stfrag(fbuf + (8 * w + 2 * ch + 1) * FRAG_BYTES + lane * 16, relu_frag(pack_half(a, 1)));
        }
        SAVAD_STAMP(48);
        __syncthreads();
        SAVAD_STAMP(49);
        // ---- FFN2 on top of the residual stream: K-steps 0..31 in order (the wave-per-block kernel's four chunks)
        f32x16 o = h1;
        o += bias_block(lb2 + 32 * w, h);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            SAVAD_NS_GET(8 + c);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) o = SAVAD_MFMA_BF16(wfrag(wq[c], ks), ldfrag(fbuf + (8 * c + ks) * FRAG_BYTES + lane * 16), o);
            if (c < 3) SAVAD_NS_REQ(l, 12 + c);   // the next layer's Q, K, V blocks (behind the last layer: re-read, waited for below)
        }
        own = o;
        SAVAD_STAMP(50);
        if (l + 1 < L) {
            park(hp, own);
            rows_ln(xb0, own, xg, xp, true);   // (xb0's readers passed four barriers since; its barrier retires the hidden fragments' readers)
        }
    }
    SAVAD_STAMP(51);
    asm volatile("s_waitcnt vmcnt(0)" : "+a"(wq[0].v[0]), "+a"(wq[1].v[0]), "+a"(wq[2].v[0]));  // the blocks requested behind the last layer are never used, but must have landed
#undef SAVAD_NS_REQ
#undef SAVAD_NS_GET
    // ---- final LayerNorm (folded into the classifier) + Linear(D, 2) + log-softmax
    rows_ln(xb0, own, xg, xp, false);
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

}  // namespace bf
}  // namespace savad
