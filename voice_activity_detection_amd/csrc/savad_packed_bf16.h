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
        for (int ks = 0; ks < KS; ++ks) {
            const int f0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
            const bf16x8 xf = load_x_frag(xr + f0, valid);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                acc[nb] = SAVAD_MFMA_BF16(ldfrag(M.win + ((size_t)(nb * KS + ks) * 64 + lane) * 16), xf, acc[nb]);
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
        const f32x4 c0 = ld4(M.wc + 8 * Gq + 4 * h), c1 = ld4(M.wc + D + 8 * Gq + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            z0 = __builtin_fmaf(xg[Gq][e], c0[e], z0);
            z1 = __builtin_fmaf(xg[Gq][e], c1[e], z1);
        }
    }
    z0 = half_sum(z0) + M.bc[0];
    z1 = half_sum(z1) + M.bc[1];
    const float mx = fmaxf(z0, z1);
    const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
    if (h == 0 && valid) *reinterpret_cast<f32x2*>(out + row * 2) = f32x2{z0 - lse, z1 - lse};
#ifdef SAVAD_TIMING
    SAVAD_PACC(0);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 4; ++i) g_savad_dbg[32 + i] = pacc[i];
#endif
}

}  // namespace bf
}  // namespace savad
