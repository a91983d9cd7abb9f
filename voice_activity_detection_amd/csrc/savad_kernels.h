// savad_kernels.h -- gfx950 (MI355X / CDNA4) device code of the self-attentive-VAD forward pass.
//
// Everything GEMM-shaped runs on the exact-fp32 matrix core instruction
// v_mfma_f32_32x32x2_f32 (64 cycles, 4096 FLOP; 157.3 TF chip peak) in the TRANSPOSED form
//
//      Out^T[n][m] = sum_k W[n][k] * X[m][k]          A operand = weight rows, B operand = activation rows
//
// so that every activation a wave touches lives in ONE register layout, the "row layout":
//
//      lane = (m, h) = (lane & 31, lane >> 5);  a 32-feature block is 16 registers r = 4g+s,
//      register r of lane (m,h) holds X[row m][feature 8g + 4h + s]                     (g,s in 0..3)
//
// which is simultaneously (i) the MFMA C/D layout of Out^T (row = (r&3)+8(r>>2)+4h = feature,
// column = lane&31 = data row), (ii) a legal B-operand k-ordering for the next GEMM (the k-sum is
// order-free as long as the A operand uses the same permutation: both halves read 4 consecutive
// floats at 8G+4h, one 16-byte load), and (iii) row-per-lane, so LayerNorm, softmax max/sum, the
// online-softmax rescale and log-softmax are lane-local plus ONE exchange between lane and lane^32.
// No transposes, no LDS staging of operands: weights / K / V stream from L2 straight into the
// A operand, activations stay in registers between chained GEMMs.
//
// Reference being restated (paths relative to /root/reference):
//   vad/models/self_attention.py:23-28, vad/modeling/transformer.py:24-61,227-238,258-363,366-382,385-414.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace savad {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int D = 128;        // d_model (= d_head, n_heads = 1: vad/models/self_attention.py:18)
constexpr int DFF = 4 * D;    // vad/models/self_attention.py:10
constexpr int TILE = 32;      // data rows per MFMA tile / per workgroup of the row kernels
constexpr int XLD = D + 4;    // LDS row stride (floats) of the 32x128 exchange buffer: conflict-free b128
constexpr int PLD = 32 + 4;   // LDS row stride of one 32x32 split-K partial block
constexpr float LN_EPS = 1e-5f;
constexpr float NEG_BIG = -1.0e30f;

#define SAVAD_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// A data row lives in lanes m and m+32: one v_permlane32_swap hands every lane both halves' values
// (no LDS round trip as with ds_bpermute), in the same order on both lanes (bit-identical results).
// NB: copy the two results into scalars before __builtin_bit_cast -- bit_cast applied directly to a
// vector-element lvalue (r[1]) reads element 0 (clang quirk; it cost a parity failure to find).
__device__ __forceinline__ void both_halves(float v, float& lo, float& hi) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r[0] = low half's value, r[1] = high half's
    const unsigned a = r[0], b = r[1];
    lo = __builtin_bit_cast(float, a);
    hi = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float half_sum(float v) {
    float lo, hi;
    both_halves(v, lo, hi);
    return lo + hi;
}
__device__ __forceinline__ float half_max(float v) {
    float lo, hi;
    both_halves(v, lo, hi);
    return fmaxf(lo, hi);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}

// bias for the wave's feature block in row layout: b[n0 + 8g + 4h + s]
__device__ __forceinline__ void add_bias(f32x16& acc, const float* __restrict__ b, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = ld4(b + 8 * g + 4 * h);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[4 * g + s] += b4[s];
    }
}

// element offset of flat row r = b T + t of the features: sequences xbs elements apart (xbs = T F: the contiguous [B,T,F]
// tensor; smaller: overlapping windows read in place out of a feature matrix -- the streaming mode, savad_forward_strided)
__device__ __forceinline__ size_t x_row_offset(size_t r, int T, int F, long xbs) {
    return xbs == (long)T * F ? r * (size_t)F : (r / (size_t)T) * (size_t)xbs + (r % (size_t)T) * (size_t)F;
}

// row-layout 32-feature block <-> memory row (global or LDS): 4 x 16-byte pieces at 8g + 4h
__device__ __forceinline__ void store_block(float* rowp /* &X[row][n0] */, const f32x16& v, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 t;
#pragma unroll
        for (int s = 0; s < 4; ++s) t[s] = v[4 * g + s];
        st4(rowp + 8 * g + 4 * h, t);
    }
}
__device__ __forceinline__ void add_block(f32x16& v, const float* rowp, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = ld4(rowp + 8 * g + 4 * h);
#pragma unroll
        for (int s = 0; s < 4; ++s) v[4 * g + s] += t[s];
    }
}

// Read full rows back from the LDS exchange buffer and apply LayerNorm WITHOUT the affine part
// (gamma/beta are folded into the next Linear on the host side of the library, see fold_ln_kernel).
// nn.LayerNorm: biased variance, eps = 1e-5 (vad/modeling/transformer.py:22,231); two-pass statistics.
__device__ __forceinline__ void read_rows_layernorm(const float* xbuf, int m, int h, f32x4 (&xg)[16]) {
    float s = 0.0f;
#pragma unroll
    for (int G = 0; G < 16; ++G) {
        xg[G] = ld4(xbuf + m * XLD + 8 * G + 4 * h);
        s += (xg[G][0] + xg[G][1]) + (xg[G][2] + xg[G][3]);
    }
    s = half_sum(s);
    const float mean = s * (1.0f / D);
    float ss = 0.0f;
#pragma unroll
    for (int G = 0; G < 16; ++G) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = xg[G][e] - mean;
            xg[G][e] = d;
            ss += d * d;
        }
    }
    ss = half_sum(ss);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + LN_EPS);
#pragma unroll
    for (int G = 0; G < 16; ++G) xg[G] *= rstd;
}

// ---------------------------------------------------------------------------------------------
// Kernel 2: single-head scaled-dot-product attention, flash style (vad/modeling/transformer.py:
// 305-346,351-363): S = q k^T / sqrt(d_head), softmax over keys, ctx = A v -- the [T,T] matrix is
// never materialised (the reference discards it: transformer.py:50).
// One WAVE per (sequence, 32-query block, key split): Q stays in registers as the B operand;
// S^T = K Q^T lands in row layout (query row per lane, 16 key scores per lane), so the online
// softmax is lane-local; P feeds the PV MFMA as B operand unchanged; O^T = V^T P^T accumulates in
// row layout.  Output: UNNORMALISED O plus (running max, running sum) per row and split; the row
// kernel combines the splits and divides.
// ---------------------------------------------------------------------------------------------

// Drain this wave's vector-memory operations (global loads AND the LDS-DMA issued through inline asm).
// It must be the BUILTIN, not an asm string: the compiler's wait-count pass cannot see into asm, would
// still believe earlier global loads (e.g. the Q rows read before the key loop) to be pending, and
// would then guard their first use INSIDE the loop with vmcnt(N) waits -- which the hardware counts
// against the DMA just issued for the next tile, exposing its full latency on every iteration.
__device__ __forceinline__ void wait_vmem_all() {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
    asm volatile("" ::: "memory");
}

// One 32-key tile of the flash loop, K and V tiles supplied by functors (LDS or global).
// sc in/out: raw scores S^T (already masked) -> probabilities p.
// Deferred rescale: the running reference point m_run only moves when some row's maximum has grown by
// more than 2^RESCALE_LOG2 since it was set (wave-uniform decision); until then probabilities are taken
// relative to the stale reference, p = 2^((s - m_run) c) <= 2^RESCALE_LOG2.  O, l and p stay mutually
// consistent, so O / l is unchanged (fp32 accumulators: no precision cost), and the 64-register rescale of O
// plus one exp -- needed on ~3/4 of the tiles with a per-tile reference on random data -- all but disappears.
constexpr float RESCALE_LOG2 = 16.0f;
__device__ __forceinline__ void online_softmax(f32x16& sc, float& m_run, float& l_run, f32x16 (&O)[4], float c) {
    float mx = sc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
    mx = half_max(mx);
    if (__any((mx - m_run) * c > RESCALE_LOG2)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);  // base-2 domain: p = 2^((s - m) c)
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) O[nb] *= alpha;
    }
    const float mc = m_run * c;
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        sc[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c, -mc));
        rs += sc[r];
    }
    l_run += half_sum(rs);
}

__device__ __forceinline__ void store_attention_partial(float* __restrict__ Opart, float* __restrict__ ml, size_t prow,
                                                        const f32x16 (&O)[4], float m_run, float l_run, int h) {
    float* op = Opart + prow * D;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) store_block(op + 32 * nb, O[nb], h);
    if (h == 0) *reinterpret_cast<f32x2*>(ml + prow * 2) = f32x2{m_run, l_run};
}

// ---- T <= 32: one item = floor(32/T) whole sequences (block-diagonal mask), one key tile.  ONE WAVE PER
// WORKGROUP (the 7-frame windows of the reference pipeline give only B/4 blocks: 250 for a 10 s clip, so
// the blocks are spread over as many CUs as possible) and every operand of the tile is requested up front
// (Q, K: 16 x 16-byte loads each, V: 64 dword loads) so that the wave pays the L2 latency once, not per MFMA
// group; with one wave per SIMD the 256-VGPR budget of the multi-wave kernels does not apply.
// Attention of one packed tile (floor(32/T) whole sequences, rows k0 .. k0+rowsPB-1) for the calling wave:
// O^T (unnormalised, row layout) and the row sums.  Q and K are requested together, V after the scores (the
// caller may hold other live registers); over-read rows only produce masked scores / zero probabilities.
__device__ __forceinline__ void packed_attention_tile(f32x16 (&O)[4], float& m_run, float& l_run, const float* __restrict__ q,
                                                      const float* __restrict__ k, const float* __restrict__ v, size_t k0,
                                                      int rowsPB, int T, int rows, float c, int n, int h) {
    const int m = n;
    const size_t qrow = k0 + m;
    const int tq = m / T;
    f32x4 qg[16], kg[16];
    const float* kp = k + (k0 + n) * D + 4 * h;
    const float* vp = v + (k0 + 4 * h) * D + n;
#pragma unroll
    for (int G8 = 0; G8 < 16; ++G8) {
        qg[G8] = ld4(q + qrow * D + 8 * G8 + 4 * h);
        kg[G8] = ld4(kp + 8 * G8);
    }
    float vv[4][16];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[nb][r] = vp[(8 * (r >> 2) + (r & 3)) * D + 32 * nb];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) O[nb] = zero16();
    m_run = NEG_BIG;
    l_run = 0.0f;
    f32x16 sc = zero16();
#pragma unroll
    for (int G8 = 0; G8 < 16; ++G8)
#pragma unroll
        for (int e = 0; e < 4; ++e) sc = SAVAD_MFMA(kg[G8][e], qg[G8][e], sc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int jk = 8 * (r >> 2) + 4 * h + (r & 3);
        const bool ok = (jk < rowsPB) && (jk / T == tq) && (k0 + jk < (size_t)rows);
        sc[r] = ok ? sc[r] : NEG_BIG;
    }
    online_softmax(sc, m_run, l_run, O, c);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) O[nb] = SAVAD_MFMA(vv[nb][r], sc[r], O[nb]);
    }
}

__global__ __launch_bounds__(64) void attention_packed_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, float* __restrict__ Opart,
                                                              float* __restrict__ ml, int B, int T, int rows, float c) {
    const int lane = threadIdx.x & 63, n = lane & 31, m = n, h = lane >> 5;
    const int G = 32 / T, rowsPB = G * T;
    const size_t k0 = (size_t)blockIdx.x * rowsPB;
    const size_t qrow = k0 + m;
    const bool qvalid = (m < rowsPB) && (qrow < (size_t)rows);
    // q/k/v carry 32 rows of slack behind rows_pad, so the tile may over-read without clamping:
    // over-read K rows only produce masked scores, over-read V rows are zero (input_qkv_kernel).
    f32x16 O[4];
    float m_run, l_run;
    packed_attention_tile(O, m_run, l_run, q, k, v, k0, rowsPB, T, rows, c, n, h);
    if (qvalid) store_attention_partial(Opart, ml, qrow, O, m_run, l_run, h);
}

// ---- T > 32: a workgroup = (sequence, key split, group of up to 4 query blocks); its 4 waves walk
// the SAME key tiles, which are staged ONCE per workgroup into LDS by asynchronous global->LDS DMA
// (global_load_lds_dwordx4, no VGPR round trip), double-buffered one tile ahead: the DMA of tile
// j+1 is issued right after the barrier that publishes tile j, so its latency hides under the 128
// MFMAs (8192 cycles) of tile j and the barrier's vmcnt(0) is free.
//   K tile [32 keys][128] : A operand of S^T = K Q^T, read with ds_read_b128 (lane = key row); the
//                           16-byte chunk c of row r sits at chunk position c ^ (r & 15), which
//                           makes every 16-lane read group hit 16 distinct 16-byte slots.  The DMA
//                           writes LDS linearly (base + lane*16), so the swizzle is applied to the
//                           per-lane SOURCE address.
//   V tile [32 keys][128] : A operand of O^T = V^T P^T, read with ds_read_b32 (lane = feature d;
//                           32 consecutive floats of one key row: conflict-free), row-major.
constexpr int KV_TILE_FLOATS = 32 * D;

// O^T += V^T P^T for one LDS-staged 32-key tile (`vb`: [32 keys][128] row-major; lane (n, h) of K-step r needs
// V[8 (r >> 2) + 4 h + (r & 3)][n + 32 nb]).  The 32 pairs of values (two consecutive keys, one ds_read2st64_b32 each) are read by
// HAND-issued loads SAVAD_PV_PIPE pairs ahead of the MFMAs that consume them, with counted lgkmcnt waits (round 5).  Left to the
// compiler every pair landed in the SAME register pair -- read, s_waitcnt lgkmcnt(0), two MFMAs, read ... -- so that each 128
// cycles of matrix work waited out one LDS round trip that could only be requested once the pair before had issued: the one
// wave per SIMD of the fused launch has nothing else to cover it with.  dma(nb) issues the wave's nb-th DMA piece of the next
// tile (vector-memory counter: not part of the counting here).  The same MFMAs in the same order: the same bits.
#ifndef SAVAD_PV_PIPE
#define SAVAD_PV_PIPE 4
#endif
template <class Dma>
__device__ __forceinline__ void pv_tile_lds(f32x16 (&O)[4], const f32x16& sc, const float* vb, int n, int h, Dma&& dma) {
#if SAVAD_PV_PIPE > 0
    constexpr int P = SAVAD_PV_PIPE;
    static_assert(P >= 1 && P <= 8, "SAVAD_PV_PIPE");
    // LDS byte address (low half of the flat address); odd feature blocks from a second base 32 floats on: the instruction's two
    // offsets count units of 64 dwords (a key row is two units, two feature blocks are one)
    const unsigned a0 = (unsigned)(size_t)(vb + 4 * h * D + n), a1 = a0 + 128u;
    f32x2 f[P];
    // pair i = 8 nb + p holds K-steps r = 2 p, 2 p + 1: keys R, R + 1 with R = 8 (p >> 1) + 2 (p & 1)
#define SAVAD_PV_OFF0(i) (2 * (8 * (((i) & 7) >> 1) + 2 * ((i) & 1)) + ((i) >> 4))
#define SAVAD_PV_LOAD(i)                                                                               \
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3"                                       \
                 : "=v"(f[(i) % P])                                                                    \
                 : "v"((((i) >> 3) & 1) ? a1 : a0), "n"(SAVAD_PV_OFF0(i)), "n"(SAVAD_PV_OFF0(i) + 2))
#define SAVAD_PV_STEP(i)                                                                               \
    {                                                                                                  \
        constexpr int newer_ = 31 - (i) < P - 1 ? 31 - (i) : P - 1; /* younger reads that may stay in flight */ \
        if constexpr (((i) & 7) == 0) dma((i) >> 3);                                                   \
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f[(i) % P]) : "n"(newer_));                        \
        O[(i) >> 3] = SAVAD_MFMA(f[(i) % P][0], sc[2 * ((i) & 7)], O[(i) >> 3]);                       \
        O[(i) >> 3] = SAVAD_MFMA(f[(i) % P][1], sc[2 * ((i) & 7) + 1], O[(i) >> 3]);                   \
        if constexpr ((i) + P < 32) SAVAD_PV_LOAD((i) + P);                                            \
    }
#define SAVAD_PV_STEP4(i) SAVAD_PV_STEP(i) SAVAD_PV_STEP((i) + 1) SAVAD_PV_STEP((i) + 2) SAVAD_PV_STEP((i) + 3)
    SAVAD_PV_LOAD(0);
    if constexpr (P > 1) SAVAD_PV_LOAD(1);
    if constexpr (P > 2) SAVAD_PV_LOAD(2);
    if constexpr (P > 3) SAVAD_PV_LOAD(3);
    if constexpr (P > 4) SAVAD_PV_LOAD(4);
    if constexpr (P > 5) SAVAD_PV_LOAD(5);
    if constexpr (P > 6) SAVAD_PV_LOAD(6);
    if constexpr (P > 7) SAVAD_PV_LOAD(7);
    SAVAD_PV_STEP4(0) SAVAD_PV_STEP4(4) SAVAD_PV_STEP4(8) SAVAD_PV_STEP4(12) SAVAD_PV_STEP4(16) SAVAD_PV_STEP4(20)
    SAVAD_PV_STEP4(24) SAVAD_PV_STEP4(28)
#undef SAVAD_PV_STEP4
#undef SAVAD_PV_STEP
#undef SAVAD_PV_LOAD
#undef SAVAD_PV_OFF0
#else
    const float* vp = vb + 4 * h * D + n;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        dma(nb);
#pragma unroll
        for (int r = 0; r < 16; ++r) O[nb] = SAVAD_MFMA(vp[(8 * (r >> 2) + (r & 3)) * D + 32 * nb], sc[r], O[nb]);
    }
#endif
}

// Workgroup -> (sequence, index among the sequence's `per_seq` workgroups) for the attention-type kernels.
// Workgroups go to the 8 XCDs round-robin by blockIdx (each XCD has its own L2); the linear order (sequence major)
// is cut into 8 equal chunks, one per XCD, so that a sequence's workgroups -- which all read the same K/V -- share an
// L2 (a sequence straddles at most two XCDs) AND every XCD gets the same number of workgroups for any batch size.
// (Binding whole sequences to XCDs, b = 8 j + xcd, left XCDs 0-1 with twice the work at B = 10: 76 us against 47.)
// Launch 8 * ceil(B * per_seq / 8) workgroups.
__device__ __forceinline__ bool xcd_balanced_map(int B, int per_seq, int& b, int& rr) {
    const long W = (long)B * per_seq;
    const int per_xcd = (int)((W + 7) / 8);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long lin = (long)xcd * per_xcd + slot;
    if (slot >= per_xcd || lin >= W) return false;
    b = (int)(lin / per_seq);
    rr = (int)(lin % per_seq);
    return true;
}

// Phase timing (experiments only, scripts/ablate.sh -DSAVAD_TIMING): wave 0 of workgroup 0 stamps
// s_memtime at phase boundaries into g_savad_dbg.
#ifdef SAVAD_TIMING
__device__ long long g_savad_dbg[64];
#define SAVAD_STAMP(i)                                                                  \
    do {                                                                                \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_savad_dbg[i] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define SAVAD_STAMP(i) \
    do {               \
    } while (0)
#endif
// Fault injection for the NEGATIVE tests of the hazard checkers (tests/test_async_load_hazards.py, tests/test_gpu_cache_pressure.py;
// never defined in the product build): bit 1 = the single-launch fp32 forward re-targets the query block's registers (a request
// for the key block) while the query GEMM still reads them: the load lands inside the GEMM on any box, whatever the caches hold
// (until round 6 the fault was a missing wait behind a request one LayerNorm earlier, which an L2 hit survived), bit 16 = the bf16 row chain hands ring
// block 2 over at a barrier without its wait, the DMA issued right in front of it (every wave then reads block 0's bytes), bit 2 = the bf16
// ring GEMMs wait for one LDS fragment too few, bit 4 = the bf16 weight ring skips its workgroup barrier, bit 8 = the ring waits that
// leave a wave's own stores in flight allow one operation too many (the newest DMA piece may not have landed at the barrier).
#ifndef SAVAD_FAULT_INJECT
#define SAVAD_FAULT_INJECT 0
#endif
#ifndef SAVAD_ABLATE
#define SAVAD_ABLATE 0  // experiment switch (scripts/ablate.sh): 1 = no DMA, 2 = no ring barrier, 4 = no softmax, 8 = no LDS operand reads (bf16 attention)
#endif
// ---- asynchronous global -> LDS DMA (global_load_lds_dwordx4): 64 lanes x 16 B from
// base (SGPR pair, wave-uniform) + per-lane byte offset (VGPR) to LDS at M0 + lane*16.
// A 16 KB block is 16 such instructions; each of the 4 waves issues 4 of them (instruction index
// i = 4*i4 + w, LDS destination block_base + i KiB).  The per-lane offsets depend only on the block
// KIND, so they are computed once per kernel (DmaLanes) and every block costs 4 DMA + 6 SALU.
// Issued as inline asm ON PURPOSE: with the builtin, hipcc (ROCm 7.2) degrades every LDS wait in a
// region where an LDS-DMA may be outstanding to lgkmcnt(0); the asm form is invisible to that
// bookkeeping and completion is tracked by hand (s_waitcnt vmcnt(0), then the workgroup barrier).
struct DmaLanes {
    unsigned off[4];
};
// kind A / K tile: 32 rows x 128 floats, row stride ld floats; chunk c of row r stored at c ^ (r & 15)
__device__ __forceinline__ DmaLanes dma_lanes_rows32(int ld, bool swizzle, int w, int lane) {
    DmaLanes L;
    const int sub = lane >> 5, p = lane & 31;
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
        const int r = 2 * (4 * i4 + w) + sub;  // 1 KiB = rows 2i, 2i+1
        L.off[i4] = (unsigned)(r * ld + 4 * (swizzle ? (p ^ (r & 15)) : p)) * 4u;
    }
    return L;
}
// kind B: 128 rows x 32 floats (column slice, row stride ld); chunk c of row r stored at c ^ ((r >> 1) & 7)
__device__ __forceinline__ DmaLanes dma_lanes_rows128(int ld, int w, int lane) {
    DmaLanes L;
    const int sub = lane >> 3, p = lane & 7;
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
        const int r = 8 * (4 * i4 + w) + sub;  // 1 KiB = rows 8i .. 8i+7 (128 B each)
        L.off[i4] = (unsigned)(r * ld + 4 * (p ^ ((r >> 1) & 7))) * 4u;
    }
    return L;
}
__device__ __forceinline__ void dma_block(const float* __restrict__ base /* wave-uniform */, const DmaLanes& L,
                                          float* lds_block, int w) {
    if (SAVAD_ABLATE & 1) return;
    const unsigned lds_addr = (unsigned)(size_t)lds_block;  // LDS byte address = low half of the flat address (no cast: no null check)
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr) + 1024u * (unsigned)w;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_add_u32 m0, m0, 4096\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %5\n\t"
        "s_add_u32 m0, m0, 4096\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %5\n\t"
        "s_add_u32 m0, m0, 4096\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(L.off[0]), "v"(L.off[1]), "v"(L.off[2]), "v"(L.off[3]), "s"(base), "s"(m0v)
        : "memory", "scc");  // s_add_u32 writes SCC: without the clobber the compiler keeps a compare's result live
                             // across the statement (seen: the null check of a generic->LDS cast, selecting -1)
}

// ONE of the four 1 KiB pieces a wave contributes to a 16 KiB block (piece i4 lands 4 KiB * i4 behind the wave's
// first one).  Issued back to back, DMA instructions stall the wave (~100 cycles each); as separate statements they
// can sit between the MFMAs of the tile being computed.
__device__ __forceinline__ void dma_piece(const float* __restrict__ base /* wave-uniform */, const DmaLanes& L, float* lds_block,
                                          int w, int i4) {
    if (SAVAD_ABLATE & 1) return;
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_block) + 1024u * (unsigned)w + 4096u * (unsigned)i4;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(L.off[i4]), "s"(base), "s"(m0v)
        : "memory");
}

__global__ __launch_bounds__(256, 2) void attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, float* __restrict__ Opart,
                                                           float* __restrict__ ml, int B, int T, int rows_pad, int S,
                                                           int NG /* query-block groups per sequence */, float c) {
    __shared__ __attribute__((aligned(16))) float lds[4 * KV_TILE_FLOATS];  // [buffer 2][K, V]
    const int lane = threadIdx.x & 63, n = lane & 31, m = n, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int QB = (T + 31) / 32, NT = QB;
    const int per_seq = NG * S;
    int b, rr;
    if (!xcd_balanced_map(B, per_seq, b, rr)) return;
    const int s = rr / NG, g = rr % NG;
    const int qb0 = (g * QB) / NG, qb1 = ((g + 1) * QB) / NG;
    const int qb = qb0 + w;
    const bool active = qb < qb1;  // wave-uniform
    const int jt0 = (int)(((long)s * NT) / S), jt1 = (int)(((long)(s + 1) * NT) / S);
    const size_t kbase = (size_t)b * T;
    const size_t qrow = kbase + 32 * (size_t)(active ? qb : qb0) + m;
    const bool qvalid = active && (32 * qb + m) < T;

    f32x4 qg[16];
#pragma unroll
    for (int G8 = 0; G8 < 16; ++G8) qg[G8] = ld4(q + qrow * D + 8 * G8 + 4 * h);
    f32x16 O[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) O[nb] = zero16();
    float m_run = NEG_BIG, l_run = 0.0f;

    const DmaLanes LK = dma_lanes_rows32(D, true, w, lane), LV = dma_lanes_rows32(D, false, w, lane);
    dma_block(k + (kbase + 32 * (size_t)jt0) * D, LK, lds, w);
    dma_block(v + (kbase + 32 * (size_t)jt0) * D, LV, lds + KV_TILE_FLOATS, w);
#ifdef SAVAD_TIMING
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tp = __builtin_readcyclecounter(), tn;
#define SAVAD_TACC(i) do { tn = __builtin_readcyclecounter(); tacc[i] += tn - tp; tp = tn; } while (0)
#else
#define SAVAD_TACC(i) do {} while (0)
#endif
    for (int jt = jt0; jt < jt1; ++jt) {
        SAVAD_TACC(5);
        float* kb = lds + ((jt - jt0) & 1) * 2 * KV_TILE_FLOATS;
        float* vb = kb + KV_TILE_FLOATS;
        // Explicit drain of this wave's DMA before the barrier: hipcc (ROCm 7.2) does NOT emit the
        // vmcnt(0) for LDS-DMA issued in the previous loop iteration (checked in the ISA), so the
        // publish "my part of tile jt is in LDS" must be stated by hand.  It is free: the DMA was
        // issued a whole tile (8192+ MFMA cycles) ago.
        wait_vmem_all();
        __syncthreads();  // tile jt has landed for every wave; everyone is done reading the other buffer
        SAVAD_TACC(0);
        const bool more = jt + 1 < jt1;
        float* kn = lds + ((jt + 1 - jt0) & 1) * 2 * KV_TILE_FLOATS;
        const float* knext = k + (kbase + 32 * (size_t)(jt + 1)) * D;
        const float* vnext = v + (kbase + 32 * (size_t)(jt + 1)) * D;
        if (!active) {  // a wave without a query block still moves its share of the next tile
            if (more) {
                dma_block(knext, LK, kn, w);
                dma_block(vnext, LV, kn + KV_TILE_FLOATS, w);
            }
            continue;
        }
        SAVAD_TACC(1);
        // ---- S^T tile = K Q^T
        f32x16 sc = zero16();
        const float* krow = kb + n * D;
#pragma unroll
        for (int G8 = 0; G8 < 16; ++G8) {
            const f32x4 k4 = ld4(krow + 4 * ((2 * G8 + h) ^ (n & 15)));
            if (more && (G8 & 3) == 1) dma_piece(knext, LK, kn, w, G8 >> 2);  // the next tile's pieces ride between the MFMAs
#pragma unroll
            for (int e = 0; e < 4; ++e) sc = SAVAD_MFMA(k4[e], qg[G8][e], sc);
        }
        SAVAD_TACC(2);
        // lane (m,h), register r: score of query m against key jk = 8(r>>2) + 4h + (r&3)
        if (32 * jt + 32 > T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jk = 8 * (r >> 2) + 4 * h + (r & 3);
                sc[r] = (32 * jt + jk < T) ? sc[r] : NEG_BIG;
            }
        }
        online_softmax(sc, m_run, l_run, O, c);
        SAVAD_TACC(3);
        // ---- O^T += V^T P^T
        pv_tile_lds(O, sc, vb, n, h, [&](int nb) {
            if (more) dma_piece(vnext, LV, kn + KV_TILE_FLOATS, w, nb);
        });
        SAVAD_TACC(4);
    }
#ifdef SAVAD_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) g_savad_dbg[16 + i] = tacc[i];
#endif
    if (qvalid) store_attention_partial(Opart, ml, (size_t)s * rows_pad + qrow, O, m_run, l_run, h);
}

// ---------------------------------------------------------------------------------------------
// Kernel 3 (per layer): everything row-wise between two attention stages, for 32 data rows:
//   combine attention splits -> final_projection + residual (transformer.py:347,237)
//   -> LN + Linear(D,4D) + ReLU + Linear(4D,D) + residual (transformer.py:234-237,370-375)
//   -> !LAST: next layer's LN + QKV (transformer.py:281-284)
//       LAST: encoder LayerNorm + classifier Linear(D,2) + LogSoftmax (transformer.py:33;
//             vad/models/self_attention.py:26-27)
// 4 waves; wave w owns output features [32w,32w+32) of out-proj / QKV (N split) and hidden units
// [128w,128w+128) of the FFN (K split, partial sums reduce-scattered through LDS once).
// ---------------------------------------------------------------------------------------------
// Biases live in LDS for the whole kernel (loaded once, published by the first ring barrier): a
// global load inside the block loop would be drained by the ring's vmcnt(0) at full L2 latency,
// once per block.  For the same reason results are stored one block LATE (after the next block's
// barrier), so that a store's write-ack is never waited for.
__device__ __forceinline__ void stage_bias(float* dst, const float* __restrict__ src, int count) {
    for (int i = threadIdx.x * 4; i < count; i += 256 * 4) st4(dst + i, ld4(src + i));
}
// Several pieces at once (each <= 1024 floats, a multiple of 4; count 0 = none): ALL the loads first, then the LDS stores.
// Piece by piece (stage_bias four times in a row) the compiler emits load -> s_waitcnt vmcnt(0) -> ds_write per piece: four
// dependent round trips to the L2 at the start of every row chain, the first of which also drains whatever else the wave has
// in flight (round 5, read off the disassembly; the loads are unconditional -- a lane past a piece's end re-reads its last
// float4 -- so that no branch separates them).
struct BiasPiece {
    float* dst;
    const float* src;
    int count;
};
template <int N>
struct BiasRegs {
    f32x4 v[N];
};
template <int N>
__device__ __forceinline__ BiasRegs<N> request_bias_pieces(const BiasPiece (&p)[N]) {
    const int i = threadIdx.x * 4;
    BiasRegs<N> r;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int j = i < p[n].count ? i : (p[n].count >= 4 ? p[n].count - 4 : 0);
        r.v[n] = ld4(p[n].src + j);
    }
    return r;
}
template <int N>
__device__ __forceinline__ void commit_bias_pieces(const BiasPiece (&p)[N], const BiasRegs<N>& r) {
    const int i = threadIdx.x * 4;
#pragma unroll
    for (int n = 0; n < N; ++n)
        if (i < p[n].count) st4(p[n].dst + i, r.v[n]);
}
template <int N>
__device__ __forceinline__ void stage_bias_pieces(const BiasPiece (&p)[N]) {
    commit_bias_pieces(p, request_bias_pieces(p));
}
__device__ __forceinline__ f32x16 bias_block(const float* lds_bias /* &bias[n0] */, int h) {
    f32x16 r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = ld4(lds_bias + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[4 * g + e] = b4[e];
    }
    return r;
}

// ---- N-split weight streaming: every wave owns its slices of the weight matrices (nothing to share), so
// they go straight from L2 into the A operand -- but as whole 16-KB BLOCKS (16 x 16-byte loads = 64 VGPRs =
// 64 MFMAs of work), double-buffered in registers: the next block is requested before the current block's
// MFMAs start, so the L2 latency hides under 4096 MFMA cycles instead of being paid per group of 4.
// The N-split kernels run at one workgroup per CU (small batches), hence the 512-VGPR budget.
struct WBlock {
    f32x4 v[16];
};
__device__ __forceinline__ void wmma_k128(f32x16& acc, const WBlock& wb, const f32x4 (&xg)[16]) {
#pragma unroll
    for (int G = 0; G < 16; ++G)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = SAVAD_MFMA(wb.v[G][e], xg[G][e], acc);
}
__device__ __forceinline__ void wmma_w2(f32x16 (&o)[4], const WBlock& wb, const f32x16& a) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[nb] = SAVAD_MFMA(wb.v[4 * nb + g][e], a[4 * g + e], o[nb]);
}

// Weights in FRAGMENT ORDER (pack_frag32_kernel): the 16 KB a wave consumes as one A-operand block are contiguous and
// register i of lane l sits at (i * 64 + l) * 16 bytes, so one load instruction reads 1 KB = 8 whole cache lines.
// (Row-major blocks make every load touch 32 lines for 32 bytes each, four instructions per line, all in flight
// together: the L2 sees the stream four times over.)  48 blocks per layer:
//   0..11 Wqkv' rows 32b..32b+31 | 12..15 Wo rows | 16..31 W1' rows | 32..47 W2 columns 32(b-32).. (all 128 rows)
constexpr int FRAG_BLOCK = 4096, FRAG_LAYER = 48 * FRAG_BLOCK;  // floats
// The request and the wait are written out by hand: left to itself hipcc sinks the 16 loads of a block down to
// their first use once the Q / K / V accumulators join the two weight buffers and the activation rows in the
// register file ("global_load; s_waitcnt vmcnt(0); 4 MFMAs", 16 times per block -- the prefetch is gone).  The
// waits name the buffer as an in/out operand so that no MFMA reading it can be scheduled above them; the layer
// loop issues no other vector-memory instruction, so the counts are exact.
__device__ __forceinline__ void wload_frag(WBlock& wb, const float* __restrict__ frag, int block, int voff /* lane * 16 bytes */) {
    const float* p = frag + (size_t)block * FRAG_BLOCK;  // wave-uniform
#pragma unroll
    for (int i = 0; i < 16; ++i)
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(wb.v[i]) : "v"(voff), "s"(p + (i >> 2) * 1024), "n"((i & 3) * 1024));
}
template <int PENDING>  // younger requests allowed to stay in flight (16 per block)
__device__ __forceinline__ void wwait(WBlock& wb) {
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+a"(wb.v[0]), "+a"(wb.v[1]), "+a"(wb.v[2]), "+a"(wb.v[3]), "+a"(wb.v[4]), "+a"(wb.v[5]), "+a"(wb.v[6]), "+a"(wb.v[7]),
                   "+a"(wb.v[8]), "+a"(wb.v[9]), "+a"(wb.v[10]), "+a"(wb.v[11]), "+a"(wb.v[12]), "+a"(wb.v[13]), "+a"(wb.v[14]),
                   "+a"(wb.v[15])
                 : "n"(PENDING));
}

// ctx = sum_s w_s O_s / sum_s w_s l_s over the S key-split partials of the lane's row (lane-local scalars).
// Pad rows (row >= rows) were never written by the attention stage: they get ctx = 0 so that the whole pipeline
// stays finite (their V rows are multiplied by probability 0 downstream).  S == 1 (no key split: every packed
// T <= 32 launch and every large batch) needs no weights, hence ONE memory round trip instead of two.
__device__ __forceinline__ void combine_splits(f32x4 (&xg)[16], const float* __restrict__ Opart, const float* __restrict__ ml,
                                               int S, size_t row, int rows, int rows_pad, float c, int h) {
    const bool valid = row < (size_t)rows;
    if (S == 1) {
        const float l = ml[row * 2 + 1];
        const float* op = Opart + row * D + 4 * h;
#pragma unroll
        for (int G = 0; G < 16; ++G) xg[G] = ld4(op + 8 * G);
        const float inv = valid ? 1.0f / l : 0.0f;
#pragma unroll
        for (int G = 0; G < 16; ++G) xg[G] = valid ? xg[G] * inv : f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    // every split's partial is requested one split ahead of its use, the first one before the maxima are known: the
    // loads of a split are one L2 / MALL round trip (the attention launch wrote them from another XCD)
    f32x4 nxt[16];
    {
        const float* op0 = Opart + row * D + 4 * h;
#pragma unroll
        for (int G = 0; G < 16; ++G) nxt[G] = ld4(op0 + 8 * G);
    }
    float M = NEG_BIG;
    for (int s = 0; s < S; ++s) {
        const float ms = ml[((size_t)s * rows_pad + row) * 2];
        M = fmaxf(M, valid ? ms : 0.0f);
    }
    float den = 0.0f;
#pragma unroll
    for (int G = 0; G < 16; ++G) xg[G] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        f32x4 cur[16];
#pragma unroll
        for (int G = 0; G < 16; ++G) cur[G] = nxt[G];
        if (s + 1 < S) {
            const float* op = Opart + ((size_t)(s + 1) * rows_pad + row) * D + 4 * h;
#pragma unroll
            for (int G = 0; G < 16; ++G) nxt[G] = ld4(op + 8 * G);
        }
        f32x2 t = *reinterpret_cast<const f32x2*>(ml + ((size_t)s * rows_pad + row) * 2);
        if (!valid) t = f32x2{0.0f, 1.0f};
        const float ws = __builtin_amdgcn_exp2f((t[0] - M) * c);
        den += ws * t[1];
#pragma unroll
        for (int G = 0; G < 16; ++G) xg[G] += ws * (valid ? cur[G] : f32x4{0.f, 0.f, 0.f, 0.f});
    }
    const float inv = 1.0f / den;
#pragma unroll
    for (int G = 0; G < 16; ++G) xg[G] *= inv;
}

// ---------------------------------------------------------------------------------------------
// Kernel 1: input Linear(F, D) + sinusoidal PE / sqrt(D)  (vad/models/self_attention.py:12-16,24;
// vad/modeling/transformer.py:392-401), then layer-0 LN + QKV (transformer.py:234-237,281-284).
// One workgroup (4 waves) per 32 data rows; wave w owns output features [32w, 32w+32).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void input_qkv_kernel(
    const float* __restrict__ x, long xbs, int rows, int T, int F, const float* __restrict__ Win, const float* __restrict__ bin,
    const float* __restrict__ pe /* [T][D], already / sqrt(D) */, const float* __restrict__ frag0 /* layer 0, fragment order */,
    const float* __restrict__ bqkv, float* __restrict__ hbuf, float* __restrict__ q, float* __restrict__ k,
    float* __restrict__ v) {
    __shared__ __attribute__((aligned(16))) float lds[TILE * XLD + 3 * D];
    float* xbuf = lds;
    float* lbn = lds + TILE * XLD;
    const int lane = threadIdx.x & 63, n = lane & 31, m = n, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int voff = lane * 16;
    const size_t row = (size_t)blockIdx.x * TILE + m;
    const bool valid = row < (size_t)rows;
    const float* xp = x + x_row_offset(valid ? row : 0, T, F, xbs) + 4 * h;
    const float* wp = Win + (size_t)(32 * w + n) * F + 4 * h;
    WBlock wa, wb;
    stage_bias(lbn, bqkv, 3 * D);

    // K = F in chunks of 128: all loads of a chunk requested before its first MFMA (F is a multiple of 16)
    f32x16 acc = zero16();
    const int nG = F / 8;
    for (int G0 = 0; G0 < nG; G0 += 16) {
        f32x4 xin[16];
#pragma unroll
        for (int G = 0; G < 16; ++G) {
            const int Gc = G0 + G < nG ? G0 + G : nG - 1;  // past the end: a valid address, the product is dropped below
            xin[G] = ld4(xp + 8 * Gc);
            wb.v[G] = ld4(wp + 8 * Gc);
        }
#pragma unroll
        for (int G = 0; G < 16; G += 2) {
            if (G0 + G < nG) {
#pragma unroll
                for (int g = G; g < G + 2; ++g) {
                    const f32x4 x4 = valid ? xin[g] : f32x4{0.f, 0.f, 0.f, 0.f};  // rows past the batch stay finite (never stored to `out`)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = SAVAD_MFMA(wb.v[g][e], x4[e], acc);
                }
            }
        }
    }
    add_bias(acc, bin + 32 * w, h);
    const int t = (int)((valid ? row : 0) % (size_t)T);
    add_block(acc, pe + (size_t)t * D + 32 * w, h);
    // layer 0's query / key blocks (fragment order, hand-placed waits: see wload_frag); requested behind the
    // compiler-managed loads above, whose waits would otherwise cover these too
    wload_frag(wa, frag0, w, voff);
    wload_frag(wb, frag0, 4 + w, voff);
    store_block(hbuf + row * D + 32 * w, acc, h);  // residual stream h0
    // V's 32 slack rows behind the last tile feed the PV product with probability exactly 0:
    // they must be finite, so the last workgroup zeroes them once per forward.
    if (blockIdx.x == gridDim.x - 1) store_block(v + (row + TILE) * D + 32 * w, zero16(), h);
    store_block(xbuf + m * XLD + 32 * w, acc, h);
    __syncthreads();
    f32x4 xg[16];
    read_rows_layernorm(xbuf, m, h, xg);
    f32x16 qa = bias_block(lbn + 32 * w, h);
    wwait<16>(wa);
    wmma_k128(qa, wa, xg);
    wload_frag(wa, frag0, 8 + w, voff);
    f32x16 ka = bias_block(lbn + D + 32 * w, h);
    wwait<16>(wb);
    wmma_k128(ka, wb, xg);
    f32x16 va = bias_block(lbn + 2 * D + 32 * w, h);
    wwait<0>(wa);
    wmma_k128(va, wa, xg);
    store_block(q + row * D + 32 * w, qa, h);
    store_block(k + row * D + 32 * w, ka, h);
    store_block(v + row * D + 32 * w, va, h);
}

template <bool LAST>
__global__ __launch_bounds__(256, 1) void row_kernel(
    const float* __restrict__ Opart, const float* __restrict__ ml, int S, int rows, int rows_pad, float c,
    float* __restrict__ hbuf, const float* __restrict__ frag /* this layer's weights in fragment order */,
    const float* __restrict__ bo, const float* __restrict__ b1, const float* __restrict__ b2,
    const float* __restrict__ nfrag /* LAST ? unused : the next layer's fragments (Wqkv' blocks 0..11) */,
    const float* __restrict__ Wn /* LAST: Wc'[2][D] */, const float* __restrict__ bn /* LAST ? bc'[2] : bqkv'[3D] */,
    float* __restrict__ q, float* __restrict__ k, float* __restrict__ v, float* __restrict__ out /* [rows][2] */) {
    __shared__ __attribute__((aligned(16))) float lds[TILE * XLD + 12 * TILE * PLD + 8 * D];
    float* xbuf = lds;
    float* pbuf = lds + TILE * XLD;  // [dest block 4][src slot 3][32 rows][PLD]
    float* lb1 = pbuf + 12 * TILE * PLD;  // biases b1[512] b2[128] bqkv[384] in LDS (published by the first barrier):
    float* lb2 = lb1 + DFF;                // a global load per use would cost an exposed L2 round trip each
    float* lbn = lb2 + D;
    const int lane = threadIdx.x & 63, n = lane & 31, m = n, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int voff = lane * 16;
    const size_t row = (size_t)blockIdx.x * TILE + m;

    SAVAD_STAMP(32);
    // Weight stream of this wave (16 KB blocks in fragment order, wload_frag / wwait): Wo, then W1 / W2 slices
    // alternating, then the next layer's Q, K, V blocks; two register buffers, every block requested one block ahead.
    // The waits count LOADS only: result stores issued in between may retire out of order with respect to loads, so they
    // are never part of the allowance (a wait can only come out longer than needed, never shorter).
    WBlock wa, wb;
    wload_frag(wb, frag, 12 + w, voff);      // Wo rows 32w..
    wload_frag(wa, frag, 16 + 4 * w, voff);  // first FFN block
    // b1 | b2 | bqkv' are 256 float4 (160 without the QKV bias): ONE load per thread, requested with everything else of
    // this phase and stored to LDS after it (a load -> wait -> store sequence per array costs a round trip each)
    const int t4 = threadIdx.x;
    const bool bias_lane = LAST ? t4 < 160 : true;
    const float* bsrc = t4 < 128 ? b1 + 4 * t4 : (t4 < 160 ? b2 + 4 * (t4 - 128) : bn + 4 * (t4 - 160));
    const f32x4 bval = bias_lane ? ld4(bsrc) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 h1 = zero16();  // the out-projection accumulator starts at the residual stream + bias
    add_bias(h1, bo + 32 * w, h);
    add_block(h1, hbuf + row * D + 32 * w, h);
    // ---- phase 0: ctx = combination of the key-split partials (rows are lane-local: all scalars per lane)
    f32x4 xg[16];
    combine_splits(xg, Opart, ml, S, row, rows, rows_pad, c, h);
    if (bias_lane) st4(lb1 + 4 * t4, bval);  // published by the first barrier below
    SAVAD_STAMP(33);
    // ---- phase 1: h1 = ctx Wo^T + bo + h   (wave's 32 features)
    wwait<16>(wb);
    wmma_k128(h1, wb, xg);
    store_block(xbuf + m * XLD + 32 * w, h1, h);
    __syncthreads();
    read_rows_layernorm(xbuf, m, h, xg);
    SAVAD_STAMP(34);
    // ---- phase 2: FFN, hidden units [128w, 128w+128) in 4 chunks of 32; ReLU output feeds the
    //      second GEMM as B operand straight from the accumulator registers
    f32x16 o[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) o[nb] = zero16();
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
        wload_frag(wb, frag, 32 + 4 * w + ch, voff);
        f32x16 a = bias_block(lb1 + 128 * w + 32 * ch, h);
        wwait<16>(wa);
        wmma_k128(a, wa, xg);
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.0f);
        // the next W1 slice, or the next layer's query block; the LAST launch re-reads a block it will not use (keeps the
        // wait count uniform and the loop free of branches: a request in one arm of a branch makes the compiler copy blocks
        // that are still in flight where the arms meet)
        wload_frag(wa, (ch + 1 < 4 || LAST) ? frag : nfrag, ch + 1 < 4 ? 16 + 4 * w + ch + 1 : w, voff);
        wwait<16>(wb);
        wmma_w2(o, wb, a);
    }
    // ... but it WAITS for that block: a request left in flight lands on whatever the compiler keeps in those registers by
    // then (round 4: one wrong 32-row tile in a few percent of the runs with other streams' kernels pressing on the L2)
    if (LAST) wwait<0>(wa);
    SAVAD_STAMP(35);
    // reduce-scatter the 4 K-split partials: wave w ends up with feature block w
    f32x16 own = o[0];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        if (nb == w) {
            own = o[nb];
        } else {
            const int slot = (w - nb - 1) & 3;
            store_block(pbuf + ((nb * 3 + slot) * TILE + m) * PLD, o[nb], h);
        }
    }
    __syncthreads();
#pragma unroll
    for (int slot = 0; slot < 3; ++slot) add_block(own, pbuf + ((w * 3 + slot) * TILE + m) * PLD, h);
    own += bias_block(lb2 + 32 * w, h);
    own += h1;  // residual onto the un-normalised stream (transformer.py:235-237)
    if (!LAST) store_block(hbuf + row * D + 32 * w, own, h);
    SAVAD_STAMP(36);
    // ---- phase 3
    store_block(xbuf + m * XLD + 32 * w, own, h);  // xbuf's last readers all passed the barrier above
    __syncthreads();
    read_rows_layernorm(xbuf, m, h, xg);
    SAVAD_STAMP(37);
    if (!LAST) {
        // Q (in flight in wa), K, V blocks of this wave's 32 features; results are stored after the last wait
        wload_frag(wb, nfrag, 4 + w, voff);
        f32x16 qa = bias_block(lbn + 32 * w, h);
        wwait<16>(wa);
        wmma_k128(qa, wa, xg);
        wload_frag(wa, nfrag, 8 + w, voff);
        f32x16 ka = bias_block(lbn + D + 32 * w, h);
        wwait<16>(wb);
        wmma_k128(ka, wb, xg);
        f32x16 va = bias_block(lbn + 2 * D + 32 * w, h);
        wwait<0>(wa);
        wmma_k128(va, wa, xg);
        store_block(q + row * D + 32 * w, qa, h);
        store_block(k + row * D + 32 * w, ka, h);
        store_block(v + row * D + 32 * w, va, h);
    } else if (w == 0) {
        float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
        for (int G = 0; G < 16; ++G) {
            const f32x4 c0 = ld4(Wn + 8 * G + 4 * h), c1 = ld4(Wn + D + 8 * G + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z0 = __builtin_fmaf(xg[G][e], c0[e], z0);
                z1 = __builtin_fmaf(xg[G][e], c1[e], z1);
            }
        }
        z0 = half_sum(z0);
        z1 = half_sum(z1);
        z0 += bn[0];
        z1 += bn[1];
        const float mx = fmaxf(z0, z1);
        const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
        if (h == 0 && row < (size_t)rows) *reinterpret_cast<f32x2*>(out + row * 2) = f32x2{z0 - lse, z1 - lse};
    }
    SAVAD_STAMP(38);
}

// ---------------------------------------------------------------------------------------------
// T <= 32, whole forward in ONE launch (the reference pipeline's own shape: 7-frame windows,
// vad/predictors/vad_predictor.py:77-112).  A workgroup owns one packed tile (floor(32/T) whole sequences) for
// ALL layers; nothing but x, the weights and the log-probabilities crosses the CU boundary: the residual
// stream lives in registers (wave w: features [32w, 32w+32) in row layout), LayerNorm rows are exchanged
// through LDS as in row_kernel, and the attention never leaves the registers either:
//   * wave w projects ITS 32 features of Q and K in row layout and ITS 32 features of V TRANSPOSED (MFMA
//     operands swapped: A = activation rows, B = weight rows -> lane = feature, registers = rows);
//   * S^T = K Q^T is a sum over features: every wave contracts its own 32 (16 MFMAs, operands straight from
//     the Q / K accumulators), the four partial score tiles are summed through LDS in a fixed order (all waves
//     hold identical scores), the softmax is lane-local and redundant per wave (16 exps);
//   * O^T = V^T P^T for the wave's own features: A = the transposed V accumulator, B = the probabilities, again
//     16 MFMAs; the context block joins the other three through the LDS exchange buffer for the out-projection.
// 32 attention MFMAs per wave and layer instead of the 128 redundant ones of row_kernel's packed mode, no
// q / k / v / h round trip through L2 and one launch instead of four.
// ---------------------------------------------------------------------------------------------
// window geometry of the predictor (vad/predictor.py:186-212): W relative frame offsets
struct WindowOffsets {
    int w;
    int off[64];
};
constexpr int PACKED_MAX_LAYERS = 8;
struct PackedLayer {
    const float* frag;  // Wqkv / W1: LayerNorm affine folded in
};
struct PackedModel {
    PackedLayer layer[PACKED_MAX_LAYERS];
    const float *win, *bin, *pe, *wc, *bc;
    const float* bias;  // [L][LBIAS]: b1' | b2 | bqkv' | bo of every layer, contiguous
    int L;
};
constexpr int LBIAS = DFF + D + 3 * D + D;  // b1 | b2 | bqkv | bo

// Windowed mode (wo.w == T > 0): x is the predictor's feature MATRIX [N][F] and sequence s is its window
// feature[win_base + s + wo.off[0..T-1]] (a13, vad/predictor.py:180-220) -- the gather is an address computation here
// instead of a launch that writes 7 copies of every frame and a forward that reads them back.
__global__ __launch_bounds__(256, 1) void packed_forward_kernel(const float* __restrict__ x, int rows, int T, int F, PackedModel M,
                                                                float c, float* __restrict__ out, int tile_rows, WindowOffsets wo,
                                                                int win_base) {
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE * XLD + 12 * TILE * PLD + PACKED_MAX_LAYERS * LBIAS];
    float* xb0 = lds;
    float* xb1 = lds + TILE * XLD;
    float* pbuf = lds + 2 * TILE * XLD;  // reduce-scatter blocks [dest 4][slot 3][32][PLD]; also the 4 partial score tiles
    float* lbias = pbuf + 12 * TILE * PLD;  // every layer's biases, staged once
    const int lane = threadIdx.x & 63, n = lane & 31, m = n, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t row = (size_t)blockIdx.x * tile_rows + m;
    const bool in_batch = row < (size_t)rows;
    const bool lane_ok = m < tile_rows && in_batch;  // lanes past the tile alias the next tile's rows: computed, never stored
    const int tq = m / T;
    const int voff = lane * 16;

    SAVAD_STAMP(48);
    // Weight stream: 12 blocks of 16 KB per wave and layer (Q K V Wo, then W1 / W2 slices alternating), two register
    // buffers, every block requested one block (64 MFMAs = 4096 cycles) before its first use.
    WBlock wa, wb;
    // all biases -> LDS (requested now, stored after the input GEMM, published by the first barrier)
    constexpr int NBT = (PACKED_MAX_LAYERS * LBIAS / 4 + 255) / 256;
    const int nb4 = M.L * (LBIAS / 4);
    f32x4 bt[NBT];
#pragma unroll
    for (int j = 0; j < NBT; ++j)
        if ((int)threadIdx.x + 256 * j < nb4) bt[j] = ld4(M.bias + 4 * (threadIdx.x + 256 * j));

    // ---- input Linear + positional encoding (self_attention.py:12-16,24): K = F in chunks of 128, all loads of a
    // chunk requested before its first MFMA
    f32x16 own = zero16();
    {
        const size_t rr = in_batch ? row : 0;
        const size_t src_row = wo.w > 0 ? (size_t)win_base + rr / (size_t)T + wo.off[rr % (size_t)T] : rr;
        const float* xp = x + src_row * (size_t)F + 4 * h;
        const float* wp = M.win + (size_t)(32 * w + n) * F + 4 * h;
        const int nG = F / 8;
        for (int G0 = 0; G0 < nG; G0 += 16) {
            f32x4 xin[16];
#pragma unroll
            for (int G = 0; G < 16; ++G) {
                const int Gc = G0 + G < nG ? G0 + G : nG - 1;  // past the end: a valid address, the product is dropped below
                xin[G] = ld4(xp + 8 * Gc);
                wb.v[G] = ld4(wp + 8 * Gc);
            }
#pragma unroll
            for (int G = 0; G < 16; G += 2) {  // F is a multiple of 16: chunks come in pairs
                if (G0 + G < nG) {
#pragma unroll
                    for (int g = G; g < G + 2; ++g) {
                        const f32x4 x4 = in_batch ? xin[g] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < 4; ++e) own = SAVAD_MFMA(wb.v[g][e], x4[e], own);
                    }
                }
            }
        }
        add_bias(own, M.bin + 32 * w, h);
        const int t = (int)((in_batch ? row : 0) % (size_t)T);
        add_block(own, M.pe + (size_t)t * D + 32 * w, h);
    }
#pragma unroll
    for (int j = 0; j < NBT; ++j)
        if ((int)threadIdx.x + 256 * j < nb4) st4(lbias + 4 * (threadIdx.x + 256 * j), bt[j]);

    wload_frag(wa, M.layer[0].frag, w, voff);  // layer 0's query block (after the prologue's compiler-managed loads:
                                               // their waits would otherwise cover this request too)
    SAVAD_STAMP(49);
    f32x4 xg[16];
#pragma unroll 1
    for (int l = 0; l < M.L; ++l) {
        const float* frag = M.layer[l].frag;
        const float* lb = lbias + l * LBIAS;
        const float *lb1 = lb, *lb2 = lb + DFF, *lbn = lb + DFF + D, *lbo = lb + DFF + 4 * D;
        // ---- LN1 + Q, K (row layout) and V^T blocks of this wave's 32 features; wa = query block (requested a block ago)
        store_block(xb0 + m * XLD + 32 * w, own, h);
        __syncthreads();
        read_rows_layernorm(xb0, m, h, xg);
        SAVAD_STAMP(50);
        wload_frag(wb, frag, 4 + w, voff);
        f32x16 qb = bias_block(lbn + 32 * w, h);
        wwait<16>(wa);
        // the planted fault (bit 1): the buffer is re-targeted (the key block) while the query GEMM still reads it -- the load lands
        // somewhere inside the GEMM's 64 MFMAs (4 096 cycles) whatever the caches hold, and the MFMAs behind it read key weights
        if (SAVAD_FAULT_INJECT & 1) wload_frag(wa, frag, 4 + w, voff);
        wmma_k128(qb, wa, xg);
        wload_frag(wa, frag, 8 + w, voff);
        f32x16 kb = bias_block(lbn + D + 32 * w, h);
        wwait<16>(wb);
        wmma_k128(kb, wb, xg);
        wload_frag(wb, frag, 12 + w, voff);
        f32x16 vT;  // vT[4g+s] = v[row 8g+4h+s][feature 32w + n]
        {
            const float bv = lbn[2 * D + 32 * w + n];
#pragma unroll
            for (int r = 0; r < 16; ++r) vT[r] = bv;
            wwait<16>(wa);
#pragma unroll
            for (int G = 0; G < 16; ++G)
#pragma unroll
                for (int e = 0; e < 4; ++e) vT = SAVAD_MFMA(xg[G][e], wa.v[G][e], vT);
        }
        wload_frag(wa, frag, 16 + 4 * w, voff);  // first FFN block
        SAVAD_STAMP(51);
        // ---- scores: this wave's 32-feature share, summed over the waves through LDS
        f32x16 sc = zero16();
#pragma unroll
        for (int r = 0; r < 16; ++r) sc = SAVAD_MFMA(kb[r], qb[r], sc);
#pragma unroll
        for (int g = 0; g < 4; ++g) st4(pbuf + ((w * 4 + g) * 64 + lane) * 4, f32x4{sc[4 * g], sc[4 * g + 1], sc[4 * g + 2], sc[4 * g + 3]});
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 s4 = ld4(pbuf + ((0 * 4 + g) * 64 + lane) * 4);
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) s4 += ld4(pbuf + ((ww * 4 + g) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) sc[4 * g + e] = s4[e];
        }
        SAVAD_STAMP(52);
        // block-diagonal mask (a key belongs to the query's own sequence), softmax over the single key tile
        float mx = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jk = 8 * (r >> 2) + 4 * h + (r & 3);
            const bool ok = (jk < tile_rows) && (jk / T == tq) && ((size_t)blockIdx.x * tile_rows + jk < (size_t)rows);
            sc[r] = ok ? sc[r] : NEG_BIG;
            mx = fmaxf(mx, sc[r]);
        }
        mx = half_max(mx);
        const float mc = mx * c;
        float rs = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c, -mc));
            rs += sc[r];
        }
        const float inv = lane_ok ? 1.0f / half_sum(rs) : 0.0f;
        f32x16 ctx = zero16();
#pragma unroll
        for (int r = 0; r < 16; ++r) ctx = SAVAD_MFMA(vT[r], sc[r], ctx);
        ctx *= inv;
        SAVAD_STAMP(53);
        store_block(xb1 + m * XLD + 32 * w, ctx, h);
        __syncthreads();
#pragma unroll
        for (int G = 0; G < 16; ++G) xg[G] = ld4(xb1 + m * XLD + 8 * G + 4 * h);
        SAVAD_STAMP(54);
        // ---- out-projection + residual (accumulator starts at the residual stream), LN2
        f32x16 h1 = own + bias_block(lbo + 32 * w, h);
        wwait<16>(wb);
        wmma_k128(h1, wb, xg);
        store_block(xb0 + m * XLD + 32 * w, h1, h);  // xb0's LN1 readers passed two barriers since
        __syncthreads();
        read_rows_layernorm(xb0, m, h, xg);
        SAVAD_STAMP(55);
        // ---- FFN: hidden units [128w, 128w+128) in 4 chunks of 32 (wa = W1 slice, wb = W2 slice), as in row_kernel
        f32x16 o[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) o[nb] = zero16();
        const float* nfrag = M.layer[l + 1 < M.L ? l + 1 : l].frag;
#pragma unroll 1
        for (int ch = 0; ch < 4; ++ch) {
            wload_frag(wb, frag, 32 + 4 * w + ch, voff);
            f32x16 a = bias_block(lb1 + 128 * w + 32 * ch, h);
            SAVAD_STAMP(40);
            wwait<16>(wa);
            SAVAD_STAMP(41);
            wmma_k128(a, wa, xg);
            SAVAD_STAMP(42);
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.0f);
            // the next W1 slice, or the next layer's query block; behind the last layer the stream simply re-reads a
            // block it will not use (keeps the wait count uniform; waited for behind the layer loop)
            wload_frag(wa, ch + 1 < 4 ? frag : nfrag, ch + 1 < 4 ? 16 + 4 * w + ch + 1 : w, voff);
            SAVAD_STAMP(43);
            wwait<16>(wb);
            SAVAD_STAMP(44);
            wmma_w2(o, wb, a);
            SAVAD_STAMP(45);
        }
        SAVAD_STAMP(56);
        // reduce-scatter the 4 K-split partials: wave w ends up with feature block w (the score tiles' readers
        // passed two barriers since)
        own = o[0];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb == w) {
                own = o[nb];
            } else {
                const int slot = (w - nb - 1) & 3;
                store_block(pbuf + ((nb * 3 + slot) * TILE + m) * PLD, o[nb], h);
            }
        }
        __syncthreads();
#pragma unroll
        for (int slot = 0; slot < 3; ++slot) add_block(own, pbuf + ((w * 3 + slot) * TILE + m) * PLD, h);
        own += bias_block(lb2 + 32 * w, h);
        own += h1;  // residual onto the un-normalised stream (transformer.py:235-237)
        SAVAD_STAMP(57);
    }
    wwait<0>(wa);  // the block requested behind the last layer is never used, but must not land on reused registers (see row_kernel)
    // ---- final LayerNorm (folded into the classifier) + Linear(D, 2) + log-softmax (self_attention.py:26-28)
    store_block(xb0 + m * XLD + 32 * w, own, h);
    __syncthreads();
    if (w == 0) {
        read_rows_layernorm(xb0, m, h, xg);
        float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
        for (int G = 0; G < 16; ++G) {
            const f32x4 c0 = ld4(M.wc + 8 * G + 4 * h), c1 = ld4(M.wc + D + 8 * G + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z0 = __builtin_fmaf(xg[G][e], c0[e], z0);
                z1 = __builtin_fmaf(xg[G][e], c1[e], z1);
            }
        }
        z0 = half_sum(z0) + M.bc[0];
        z1 = half_sum(z1) + M.bc[1];
        const float mx = fmaxf(z0, z1);
        const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
        if (h == 0 && lane_ok) *reinterpret_cast<f32x2*>(out + row * 2) = f32x2{z0 - lse, z1 - lse};
    }
    SAVAD_STAMP(58);
}

// =============================================================================================
// M-split row kernels (used when the batch is large enough to fill the chip with 128-row tiles).
// A workgroup = 128 data rows = 4 waves x 32 rows; every wave runs the WHOLE row-wise chain for
// its own rows register-to-register (row layout in, row layout out: no activation ever touches
// LDS, LayerNorm is lane-local, the FFN's ReLU output feeds the second GEMM straight from the
// accumulators).  The only shared operand is the weight stream (768 KB per layer): it is staged
// ONCE per workgroup into a 2 x 16 KB LDS ring by asynchronous global->LDS DMA, one 16 KB block
// (= 64 MFMAs per wave = 4096 cycles) ahead of its use, and read by all four waves with
// conflict-free ds_read_b128.  Block kinds:
//   A: 32 output features x 128 k   (Wo / W1 chunk / Wqkv block); chunk c of row r at c ^ (r & 15)
//   B: 128 output features x 32 k   (column slice of W2 [128][512]); chunk c of row r at c ^ ((r >> 1) & 7)
// =============================================================================================
constexpr int WBLK = 4096;  // floats per ring buffer (16 KB)

// publish / acquire one ring block: my DMA has landed, everybody's has, and everybody is done with
// the previous block (so its buffer may be refilled right after this returns)
__device__ __forceinline__ void ring_acquire() {
    if (SAVAD_ABLATE & 2) return;
    wait_vmem_all();  // hipcc does not count LDS-DMA across the loop back-edge
    __syncthreads();
}
__device__ __forceinline__ void gemm_lds_a(f32x16& acc, const float* buf, int n, int h, const f32x4 (&xg)[16]) {
    const float* row = buf + n * D;
#pragma unroll
    for (int G = 0; G < 16; ++G) {
        const f32x4 w4 = ld4(row + 4 * ((2 * G + h) ^ (n & 15)));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = SAVAD_MFMA(w4[e], xg[G][e], acc);
    }
}
__device__ __forceinline__ void gemm_lds_b(f32x16 (&o)[4], const float* buf, int n, int h, const f32x16& a) {
    const int sw = (n >> 1) & 7;  // ((32 nb + n) >> 1) & 7
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const float* row = buf + (32 * nb + n) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w4 = ld4(row + 4 * ((2 * g + h) ^ sw));
#pragma unroll
            for (int e = 0; e < 4; ++e) o[nb] = SAVAD_MFMA(w4[e], a[4 * g + e], o[nb]);
        }
    }
}
// LayerNorm (no affine) of full rows held as 4 row-layout blocks -> B-operand registers
__device__ __forceinline__ void layernorm_regs(const f32x16 (&x)[4], f32x4 (&xg)[16]) {
    float s = 0.0f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += x[nb][r];
    s = half_sum(s);
    const float mean = s * (1.0f / D);
    float ss = 0.0f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = x[nb][r] - mean;
            xg[4 * nb + (r >> 2)][r & 3] = d;
            ss += d * d;
        }
    ss = half_sum(ss);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + LN_EPS);
#pragma unroll
    for (int G = 0; G < 16; ++G) xg[G] *= rstd;
}

// QKV tail shared by both M-split kernels: 12 ring blocks (Wqkv rows 32j .. 32j+31); block 0 must
// already be in flight into ring buffer 0; bq = LDS copy of the packed QKV bias [384].
__device__ __forceinline__ void qkv_tail_m(const f32x4 (&xg)[16], const float* __restrict__ Wqkv, const float* bq,
                                           float* __restrict__ q, float* __restrict__ k, float* __restrict__ v,
                                           size_t row, float* ring, const DmaLanes& LA, int w, int n, int h,
                                           bool store_ok = true /* per lane: this lane's row exists */,
                                           bool act = true /* wave-uniform: the wave owns rows (else: DMA + barriers only) */) {
    float* dst[3] = {q, k, v};
    f32x16 prev = zero16();
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        ring_acquire();
        if (j + 1 < 12) dma_block(Wqkv + (size_t)(32 * (j + 1)) * D, LA, ring + ((j + 1) & 1) * WBLK, w);
        if (j > 0 && store_ok) store_block(dst[(j - 1) >> 2] + row * D + 32 * ((j - 1) & 3), prev, h);
        f32x16 acc = bias_block(bq + 32 * j, h);
        if (act) gemm_lds_a(acc, ring + (j & 1) * WBLK, n, h, xg);
        prev = acc;
    }
    if (store_ok) store_block(dst[2] + row * D + 96, prev, h);
}

__global__ __launch_bounds__(256, 2) void input_qkv_kernel_m(
    const float* __restrict__ x, long xbs, int rows, int T, int F, const float* __restrict__ Win, const float* __restrict__ bin,
    const float* __restrict__ pe, const float* __restrict__ Wqkv, const float* __restrict__ bqkv,
    float* __restrict__ hbuf, float* __restrict__ q, float* __restrict__ k, float* __restrict__ v) {
    __shared__ __attribute__((aligned(16))) float lds[2 * WBLK + 3 * D];
    float* ring = lds;
    float* bq = lds + 2 * WBLK;
    const int lane = threadIdx.x & 63, n = lane & 31, m = n, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t row = (size_t)blockIdx.x * 128 + 32 * w + m;
    const bool valid = row < (size_t)rows;
    SAVAD_STAMP(40);
    const DmaLanes LA = dma_lanes_rows32(D, true, w, lane);
    dma_block(Wqkv, LA, ring, w);  // first QKV block flies while the input projection runs
    stage_bias(bq, bqkv, 3 * D);
    const float* xp = x + x_row_offset(valid ? row : 0, T, F, xbs) + 4 * h;
    const float* wp = Win + (size_t)n * F + 4 * h;
    const int t = (int)((valid ? row : 0) % (size_t)T);
    f32x16 h0[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {  // accumulators start at bias + PE (issued first, consumed last)
        h0[nb] = zero16();
        add_bias(h0[nb], bin + 32 * nb, h);
        add_block(h0[nb], pe + (size_t)t * D + 32 * nb, h);
    }
    SAVAD_STAMP(41);
    for (int G = 0; G < F / 8; ++G) {  // the input weights (40 KB) come straight from L2: 2.6 % of the MFMAs
        f32x4 x4 = ld4(xp + 8 * G);
        if (!valid) x4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const f32x4 w4 = ld4(wp + (size_t)(32 * nb) * F + 8 * G);
#pragma unroll
            for (int e = 0; e < 4; ++e) h0[nb] = SAVAD_MFMA(w4[e], x4[e], h0[nb]);
        }
    }
    SAVAD_STAMP(42);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        store_block(hbuf + row * D + 32 * nb, h0[nb], h);
        if (blockIdx.x == gridDim.x - 1 && w == 3) store_block(v + (row + TILE) * D + 32 * nb, zero16(), h);  // V slack
    }
    SAVAD_STAMP(43);
    f32x4 xg[16];
    layernorm_regs(h0, xg);
    SAVAD_STAMP(44);
    qkv_tail_m(xg, Wqkv, bq, q, k, v, row, ring, LA, w, n, h);
    SAVAD_STAMP(45);
}

// The row-wise chain of one layer for the 32 rows of a wave, weights through the workgroup's LDS ring:
//   h1 = h + bo + ctx Wo^T -> LN -> FFN (+ residual) -> next layer's LN + QKV, or final LN + classifier.
// In : xg = attention context of the wave's rows (row layout), h1 = their residual-stream rows.
// Pre: block 0 of Wo is in flight into ring buffer 0; lbo / lb1 / lb2 / lbn are being staged (the first ring
//      barrier publishes both).  store_ok / out_ok (per lane): the lane's row exists and may be written.
// act (wave-uniform): the wave owns rows.  A wave without a query block (ragged group) only keeps the weight stream and
// the barriers going: it issues its share of the DMA and no MFMA (round 1 let it run the chain on zeros: 6 % of the
// launch's MFMA instructions at T = 800, 3 idle slots in 28).
template <bool LAST>
__device__ __forceinline__ void row_chain_m(f32x4 (&xg)[16], f32x16 (&h1)[4], size_t row, bool store_ok, bool out_ok, bool act,
                                            float* ring, const float* lbo, const float* lb1, const float* lb2, const float* lbn,
                                            const DmaLanes& LA, const DmaLanes& LB, const float* __restrict__ Wo,
                                            const float* __restrict__ W1, const float* __restrict__ W2,
                                            const float* __restrict__ Wn, const float* __restrict__ bn,
                                            float* __restrict__ hbuf, float* __restrict__ q, float* __restrict__ k,
                                            float* __restrict__ v, float* __restrict__ out, int w, int n, int h) {
    SAVAD_STAMP(1);
    // ---- h1 = h + bo + ctx Wo^T  (ring blocks 0..3)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        ring_acquire();
        if (nb < 3)
            dma_block(Wo + (size_t)(32 * (nb + 1)) * D, LA, ring + ((nb + 1) & 1) * WBLK, w);
        else
            dma_block(W1, LA, ring, w);
        if (act) {
            h1[nb] += bias_block(lbo + 32 * nb, h);
            gemm_lds_a(h1[nb], ring + (nb & 1) * WBLK, n, h, xg);
        }
    }
    SAVAD_STAMP(2);
    if (act) layernorm_regs(h1, xg);
    SAVAD_STAMP(3);
    // ---- FFN: 16 hidden chunks of 32; W1 chunk in ring buffer 0, W2 column slice in buffer 1.  The
    // accumulators START from the residual stream (h1 + b2: transformer.py:235-237), so h1 needs no
    // registers of its own across the FFN and the kernel stays spill-free at 2 waves per SIMD.
    f32x16(&o)[4] = h1;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) o[nb] += bias_block(lb2 + 32 * nb, h);
    SAVAD_STAMP(4);
#pragma unroll 1
    for (int ch = 0; ch < 16; ++ch) {
        ring_acquire();
        dma_block(W2 + 32 * ch, LB, ring + WBLK, w);
        f32x16 a = bias_block(lb1 + 32 * ch, h);
        if (act) {
            gemm_lds_a(a, ring, n, h, xg);
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.0f);
        }
        ring_acquire();
        if (ch + 1 < 16)
            dma_block(W1 + (size_t)(32 * (ch + 1)) * D, LA, ring, w);
        else if (!LAST)
            dma_block(Wn, LA, ring, w);
        if (act) gemm_lds_b(o, ring + WBLK, n, h, a);
    }
    SAVAD_STAMP(5);
    if (!LAST) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) if (store_ok) store_block(hbuf + row * D + 32 * nb, o[nb], h);
    }
    SAVAD_STAMP(6);
    if (act) layernorm_regs(o, xg);
    SAVAD_STAMP(7);
    if (!LAST) {
        qkv_tail_m(xg, Wn, lbn, q, k, v, row, ring, LA, w, n, h, store_ok, act);
        SAVAD_STAMP(8);
    } else {
        float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
        for (int G = 0; G < 16; ++G) {
            const f32x4 c0 = ld4(Wn + 8 * G + 4 * h), c1 = ld4(Wn + D + 8 * G + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z0 = __builtin_fmaf(xg[G][e], c0[e], z0);
                z1 = __builtin_fmaf(xg[G][e], c1[e], z1);
            }
        }
        z0 = half_sum(z0);
        z1 = half_sum(z1);
        z0 += bn[0];
        z1 += bn[1];
        const float mx = fmaxf(z0, z1);
        const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
        if (h == 0 && out_ok) *reinterpret_cast<f32x2*>(out + row * 2) = f32x2{z0 - lse, z1 - lse};
    }
}

template <bool LAST>
__global__ __launch_bounds__(256, 2) void row_kernel_m(
    const float* __restrict__ Opart, const float* __restrict__ ml, int S, int rows, int rows_pad, float c,
    float* __restrict__ hbuf, const float* __restrict__ Wo, const float* __restrict__ bo,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
    const float* __restrict__ b2, const float* __restrict__ Wn, const float* __restrict__ bn, float* __restrict__ q,
    float* __restrict__ k, float* __restrict__ v, float* __restrict__ out) {
    // LDS: weight ring 2 x 16 KB, then biases bo[128] b1[512] b2[128] bqkv[384]
    __shared__ __attribute__((aligned(16))) float lds[2 * WBLK + 9 * D];
    float* ring = lds;
    float* lbo = lds + 2 * WBLK;
    float* lb1 = lbo + D;
    float* lb2 = lb1 + DFF;
    float* lbn = lb2 + D;
    const int lane = threadIdx.x & 63, n = lane & 31, m = n, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t row = (size_t)blockIdx.x * 128 + 32 * w + m;
    SAVAD_STAMP(0);
    const DmaLanes LA = dma_lanes_rows32(D, true, w, lane), LB = dma_lanes_rows128(DFF, w, lane);
    dma_block(Wo, LA, ring, w);
    stage_bias_pieces<4>({{lbo, bo, D}, {lb1, b1, DFF}, {lb2, b2, D}, {lbn, LAST ? bo : bn, LAST ? 0 : 3 * D}});
    // the out-projection accumulators start at the residual stream.  Without key splits the loads are issued now and
    // consumed after phase 0; with splits they wait until the partials are combined: the combine keeps two partials
    // (128 registers) in flight next to the 64 accumulators, and 64 more live registers made the kernel spill
    // (24 VGPRs / 100 bytes of scratch in round 2; tests/test_abi_and_host.py::test_no_kernel_spills).
    f32x16 h1[4];
    f32x4 xg[16];
    auto load_residual = [&]() {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            h1[nb] = zero16();
            add_block(h1[nb], hbuf + row * D + 32 * nb, h);
        }
    };
    // ---- combine the attention splits: ctx = sum_s w_s O_s / sum_s w_s l_s (lane-local)
    if (S == 1) {
        load_residual();
        combine_splits(xg, Opart, ml, 1, row, rows, rows_pad, c, h);
    } else {
        combine_splits(xg, Opart, ml, S, row, rows, rows_pad, c, h);
        load_residual();
    }
    row_chain_m<LAST>(xg, h1, row, true, row < (size_t)rows, true, ring, lbo, lb1, lb2, lbn, LA, LB, Wo, W1, W2, Wn, bn, hbuf, q, k, v, out,
                      w, n, h);
}

// ---------------------------------------------------------------------------------------------
// Fused stage (T > 32, no key split, M-split regime): attention of one (sequence, group of <= 4 query
// blocks) IMMEDIATELY followed by the row chain of those same rows, in the same workgroup.  The
// unnormalised context O^T a wave holds after its last key tile IS the row chain's B operand (row
// layout), so the context never leaves registers: no partial buffer, one launch per layer instead of
// two, and the chain's weight ring reuses the K/V staging LDS.  Because a workgroup now writes the NEXT
// layer's K/V rows while other workgroups of the sequence may still be reading this layer's, q/k/v are
// double-buffered between layers (read q,k,v -- write qn,kn,vn).
// Rows are addressed per sequence (flat row b*T + 32*qb + m); lanes past T exist only to fill the MFMA
// tile: they compute on zeros and never store.
// ---------------------------------------------------------------------------------------------
template <bool LAST>
__global__ __launch_bounds__(256, 2) void attention_row_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int B, int T, int NG, float c,
    float* __restrict__ hbuf, const float* __restrict__ Wo, const float* __restrict__ bo, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ Wn,
    const float* __restrict__ bn, float* __restrict__ qn, float* __restrict__ kn, float* __restrict__ vn,
    float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[4 * KV_TILE_FLOATS];  // attention: [buffer 2][K, V]; then ring + biases
    static_assert(4 * KV_TILE_FLOATS >= 2 * WBLK + 9 * D, "the row chain's ring and biases must fit the K/V staging area");
    const int lane = threadIdx.x & 63, n = lane & 31, m = n, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int QB = (T + 31) / 32, NT = QB;
    int b, g;
    if (!xcd_balanced_map(B, NG, b, g)) return;
    const int qb0 = (g * QB) / NG, qb1 = ((g + 1) * QB) / NG;
    const int qb = qb0 + w;
    const bool active = qb < qb1;  // wave-uniform
    const size_t kbase = (size_t)b * T;
    const size_t row = kbase + 32 * (size_t)(active ? qb : qb0) + m;
    const bool qvalid = active && (32 * qb + m) < T;

    f32x4 xg[16];  // Q rows first, the normalised context afterwards
#pragma unroll
    for (int G8 = 0; G8 < 16; ++G8) xg[G8] = ld4(q + row * D + 8 * G8 + 4 * h);
    f32x16 O[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) O[nb] = zero16();
    float m_run = NEG_BIG, l_run = 0.0f;

    const DmaLanes LK = dma_lanes_rows32(D, true, w, lane), LV = dma_lanes_rows32(D, false, w, lane);
    dma_block(k + kbase * D, LK, lds, w);
    dma_block(v + kbase * D, LV, lds + KV_TILE_FLOATS, w);
    for (int jt = 0; jt < NT; ++jt) {
        float* kb = lds + (jt & 1) * 2 * KV_TILE_FLOATS;
        float* vb = kb + KV_TILE_FLOATS;
        wait_vmem_all();
        __syncthreads();  // tile jt has landed for every wave; everyone is done reading the other buffer
        const bool more = jt + 1 < NT;
        float* kn2 = lds + ((jt + 1) & 1) * 2 * KV_TILE_FLOATS;
        const float* knext = k + (kbase + 32 * (size_t)(jt + 1)) * D;
        const float* vnext = v + (kbase + 32 * (size_t)(jt + 1)) * D;
        if (!active) {  // a wave without a query block still moves its share of the next tile
            if (more) {
                dma_block(knext, LK, kn2, w);
                dma_block(vnext, LV, kn2 + KV_TILE_FLOATS, w);
            }
            continue;
        }
        // the wave's 8 DMA pieces of the next tile are spread over the 128 MFMAs of this one
        f32x16 sc = zero16();
        const float* krow = kb + n * D;
#pragma unroll
        for (int G8 = 0; G8 < 16; ++G8) {
            const f32x4 k4 = ld4(krow + 4 * ((2 * G8 + h) ^ (n & 15)));
            if (more && (G8 & 3) == 1) dma_piece(knext, LK, kn2, w, G8 >> 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) sc = SAVAD_MFMA(k4[e], xg[G8][e], sc);
        }
        if (32 * jt + 32 > T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jk = 8 * (r >> 2) + 4 * h + (r & 3);
                sc[r] = (32 * jt + jk < T) ? sc[r] : NEG_BIG;
            }
        }
        online_softmax(sc, m_run, l_run, O, c);
        pv_tile_lds(O, sc, vb, n, h, [&](int nb) {
            if (more) dma_piece(vnext, LV, kn2 + KV_TILE_FLOATS, w, nb);
        });
    }
    // ---- hand-over: everyone is done with the K/V tiles; the staging area becomes weight ring + biases
    float* ring = lds;
    float* lbo = lds + 2 * WBLK;
    float* lb1 = lbo + D;
    float* lb2 = lb1 + DFF;
    float* lbn = lb2 + D;
    f32x16 h1[4];  // residual rows: requested before the barrier, consumed after the first ring block
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        h1[nb] = zero16();
        add_block(h1[nb], hbuf + row * D + 32 * nb, h);
    }
    // The next layer's last key tile over-reads up to 31 V rows behind the batch; rows past B*T are never
    // written by this kernel, and probability 0 times a non-finite value would poison the context.
    if (!LAST && b == B - 1 && g == NG - 1) store_block(vn + ((size_t)B * T + m) * D + 32 * w, zero16(), h);
    __syncthreads();
    const DmaLanes LA = dma_lanes_rows32(D, true, w, lane), LB = dma_lanes_rows128(DFF, w, lane);
    dma_block(Wo, LA, ring, w);
    stage_bias_pieces<4>({{lbo, bo, D}, {lb1, b1, DFF}, {lb2, b2, D}, {lbn, LAST ? bo : bn, LAST ? 0 : 3 * D}});
    {
        const float inv = qvalid ? 1.0f / l_run : 0.0f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) xg[4 * nb + (r >> 2)][r & 3] = qvalid ? O[nb][r] * inv : 0.0f;
    }
    row_chain_m<LAST>(xg, h1, row, qvalid, qvalid, active, ring, lbo, lb1, lb2, lbn, LA, LB, Wo, W1, W2, Wn, bn, hbuf, qn, kn, vn, out, w, n, h);
}

// ---------------------------------------------------------------------------------------------
// Weight preparation: fold a LayerNorm's affine parameters into the Linear that consumes it:
//   LN(x) W^T + b = xhat (W diag(gamma))^T + (b + W beta)
// One block per output row; beta term accumulated in fp64.
// ---------------------------------------------------------------------------------------------
// fp32 weights in fragment order (PackedLayer::frag).  mode 0: `count` row blocks of W [32 count][128] (register i =
// k-chunk G: W[32 b + n][8 G + 4 h + e]); mode 1: `count` column blocks of W2 [128][512] (register i = 4 nb + g:
// W2[32 nb + n][32 b + 8 g + 4 h + e]).
__global__ void pack_frag32_kernel(const float* __restrict__ W, int mode, int count, float* __restrict__ out) {
    const size_t total = (size_t)count * FRAG_BLOCK;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / FRAG_BLOCK), r = (int)(idx % FRAG_BLOCK);
        const int i = r >> 8, lane = (r >> 2) & 63, e = r & 3, n = lane & 31, h = lane >> 5;
        out[idx] = mode == 0 ? W[(size_t)(32 * b + n) * D + 8 * i + 4 * h + e]
                             : W[(size_t)(32 * (i >> 2) + n) * DFF + 32 * b + 8 * (i & 3) + 4 * h + e];
    }
}

__global__ void fold_ln_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ gamma,
                               const float* __restrict__ beta, float* __restrict__ Wout, float* __restrict__ bout,
                               int K) {
    __shared__ double red[256];
    const int nrow = blockIdx.x;
    double acc = 0.0;
    for (int kk = threadIdx.x; kk < K; kk += blockDim.x) {
        const float wv = W[(size_t)nrow * K + kk];
        Wout[(size_t)nrow * K + kk] = wv * gamma[kk];
        acc += (double)wv * (double)beta[kk];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) bout[nrow] = (float)((double)b[nrow] + red[0]);
}

// ---------------------------------------------------------------------------------------------
// a13: window gather (vad/predictor.py:180-220).  One thread per output float4.
// ---------------------------------------------------------------------------------------------
__global__ void gather_windows_kernel(const float* __restrict__ feature, int F, int half, int first, int count,
                                      WindowOffsets wo, float* __restrict__ windows, int64_t* __restrict__ positions) {
    const int f4 = F / 4;
    const size_t total = (size_t)count * wo.w * f4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % f4);
        const size_t iw = i / f4;
        const int wi = (int)(iw % wo.w);
        const size_t item = iw / wo.w;
        const size_t pos = (size_t)half + first + item + wo.off[wi];
        st4(windows + iw * F + 4 * c, ld4(feature + pos * F + 4 * c));
        if (c == 0 && positions) positions[iw] = (int64_t)pos;
    }
}

// ---------------------------------------------------------------------------------------------
// a14: boosted prediction (vad/predictor.py:238-258, :95): scatter, then softmax[...,1] and mean.
// ---------------------------------------------------------------------------------------------
__global__ void boost_scatter_kernel(const float* __restrict__ logp, const int64_t* __restrict__ positions,
                                     size_t count_w, int W, float* __restrict__ boosted) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count_w; i += (size_t)gridDim.x * blockDim.x) {
        const int wi = (int)(i % W);
        const size_t p = (size_t)positions[i];
        *reinterpret_cast<f32x2*>(boosted + (p * W + wi) * 2) = *reinterpret_cast<const f32x2*>(logp + i * 2);
    }
}
__global__ void boost_softmax_kernel(const float* __restrict__ boosted, int N, int W, float* __restrict__ probs,
                                     float* __restrict__ mean) {
    for (int nrow = blockIdx.x * blockDim.x + threadIdx.x; nrow < N; nrow += gridDim.x * blockDim.x) {
        float acc = 0.0f;
        for (int wi = 0; wi < W; ++wi) {
            const f32x2 z = *reinterpret_cast<const f32x2*>(boosted + ((size_t)nrow * W + wi) * 2);
            const float mx = fmaxf(z[0], z[1]);
            const float e0 = expf(z[0] - mx), e1 = expf(z[1] - mx);
            const float p = e1 / (e0 + e1);
            probs[(size_t)nrow * W + wi] = p;
            acc += p;
        }
        if (mean) mean[nrow] = acc / (float)W;
    }
}

// The same result as scatter + softmax, read the other way round: slot (n, w) of the boosted array was written by
// window b = n - half - off[w] if that window exists, else it still holds (0, 0) -> 0.5.  Same arithmetic on the same
// values, no memset, no scatter launch, no positions array.
__global__ void boost_gather_kernel(const float* __restrict__ logp, int n_items, int N, int half, WindowOffsets wo,
                                    float* __restrict__ probs, float* __restrict__ mean) {
    const int W = wo.w;
    for (int nrow = blockIdx.x * blockDim.x + threadIdx.x; nrow < N; nrow += gridDim.x * blockDim.x) {
        float acc = 0.0f;
        for (int wi = 0; wi < W; ++wi) {
            const int b = nrow - half - wo.off[wi];
            f32x2 z = f32x2{0.0f, 0.0f};
            if (b >= 0 && b < n_items) z = *reinterpret_cast<const f32x2*>(logp + ((size_t)b * W + wi) * 2);
            const float mx = fmaxf(z[0], z[1]);
            const float e0 = expf(z[0] - mx), e1 = expf(z[1] - mx);
            const float p = e1 / (e0 + e1);
            probs[(size_t)nrow * W + wi] = p;
            acc += p;
        }
        if (mean) mean[nrow] = acc / (float)W;
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming long-form mode (BASELINE configs[4]; not a reference mode -- the reference would cut
// 7-frame windows): window w covers frames [hop*w, hop*w + T) of feature[N][F], zero-padded past N.
// ---------------------------------------------------------------------------------------------
__global__ void gather_strided_kernel(const float* __restrict__ feature, int N, int F, int T, int hop, int first,
                                      int count, float* __restrict__ windows) {
    const int f4 = F / 4;
    const size_t total = (size_t)count * T * f4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % f4);
        const size_t wt = i / f4;
        const int t = (int)(wt % T);
        const size_t wi = wt / T;
        const long frame = (long)hop * (long)(first + wi) + t;
        f32x4 val = f32x4{0.f, 0.f, 0.f, 0.f};
        if (frame < N) val = ld4(feature + (size_t)frame * F + 4 * c);
        st4(windows + wt * F + 4 * c, val);
    }
}
// probs[n] = mean over the windows w covering frame n (hop*w <= n < hop*w + T, 0 <= w < W) of
// softmax(logp[w][n - hop*w])[1]
__global__ void overlap_merge_kernel(const float* __restrict__ logp, int W, int N, int T, int hop,
                                     float* __restrict__ probs) {
    for (int nrow = blockIdx.x * blockDim.x + threadIdx.x; nrow < N; nrow += gridDim.x * blockDim.x) {
        int w_hi = nrow / hop;
        if (w_hi > W - 1) w_hi = W - 1;
        float acc = 0.0f;
        int cnt = 0;
        for (int wi = w_hi; wi >= 0 && nrow - hop * wi < T; --wi) {
            const f32x2 z = *reinterpret_cast<const f32x2*>(logp + ((size_t)wi * T + (nrow - hop * wi)) * 2);
            const float mx = fmaxf(z[0], z[1]);
            const float e0 = expf(z[0] - mx), e1 = expf(z[1] - mx);
            acc += e1 / (e0 + e1);
            ++cnt;
        }
        probs[nrow] = cnt ? acc / (float)cnt : 0.5f;
    }
}

// ---------------------------------------------------------------------------------------------
// Arbitrary feature sizes (the reference's transforms give 80 mels, 257 spectrogram bins, n_mfcc ...):
// the MFMA kernels read K in groups of 8 (fp32) / 16 (bf16) with 16-byte loads, so rows are zero-padded
// to FP = round_up(F, 16) when F is not a multiple of 16: the input weight once (prepare_weights), the
// features once per forward.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pad_rows_kernel(const T* __restrict__ src, size_t rows, int F, int FP, float* __restrict__ dst) {
    const size_t total = rows * (size_t)FP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % FP);
        const size_t r = i / FP;
        dst[i] = c < F ? (float)src[r * F + c] : 0.0f;
    }
}

}  // namespace savad
