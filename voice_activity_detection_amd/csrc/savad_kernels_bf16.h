// savad_kernels_bf16.h -- bf16-operand variant of the forward pass (BASELINE.json configs[2..3]:
// bf16 weights / activations, fp32 accumulation, fp32 softmax and LayerNorm statistics, residual
// stream fp32 in registers and fp16 in HBM).  Same row-layout idea as savad_kernels.h, on v_mfma_f32_32x32x16_bf16
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
//   h         : [block][nb 4][gp 2][lane 64][8 f16]  (residual stream as stored between kernels)
//   weights   : [n-block][ks][lane 64][8 bf16], packed once by pack_weight_frags_kernel
#pragma once
#include "savad_kernels.h"
#include <type_traits>

namespace savad {
namespace bf {

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for_c(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for_c<B + 1, E>(f);
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define SAVAD_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int FRAG_BYTES = 1024;             // one K-step fragment of 32 rows: 64 lanes x 16 B
constexpr int BLK_BYTES = 8 * FRAG_BYTES;    // 32 rows x 128 features in bf16
constexpr int RING_BYTES = 4 * BLK_BYTES;    // one ring block = 128 output features x 128 k = 32 KiB
constexpr int HBLK_FLOATS = 32 * D;          // elements of one residual block

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
// ReLU on a packed fragment: max(x, 0) of a bf16 is max of its bit pattern as a signed 16-bit integer with 0 (negative values have
// the sign bit set; -0.0 becomes +0.0) -- 4 v_pk_max_i16 per fragment instead of 8 v_max_f32 before the pack; the same bits for
// every non-NaN input (cvt is monotonic and keeps the sign).
typedef short i16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 relu_frag(bf16x8 v) {
    const i16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(i16x8, v), z));
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

// The residual stream h lives in HBM between kernels as FP16 (all arithmetic on it is fp32 in registers):
// at B=256, T=800 the fp32 image was a third of the row stage's 430 MB of HBM traffic per launch, and that
// stage runs at the HBM limit.  fp16 keeps 11 significant bits -- 8x finer than the bf16 rounding every
// consumer of h applies right after its LayerNorm -- and saturates at +-65504 instead of overflowing.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hres_t;
__device__ __forceinline__ void load_hblock(f32x16 (&x)[4], const hres_t* hb, int lane) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            const f16x8 t = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(hb + ((nb * 2 + gp) * 64 + lane) * 8));
            const f32x8 f = __builtin_convertvector(t, f32x8);
#pragma unroll
            for (int s = 0; s < 8; ++s) x[nb][8 * gp + s] += f[s];
        }
}
// Saturation is not silent: a wave that clamps anything adds the number of clamped elements to *satcnt (one atomic
// per wave and stored block, only when it happens; savad_residual_saturations reads and clears the counter).  The reference's
// fp32 residual cannot saturate; a model whose residual stream leaves +-65504 needs the fp32 path (precision "fp32").
__device__ __forceinline__ void store_hblock(hres_t* hb, const f32x16 (&x)[4], int lane, unsigned* __restrict__ satcnt) {
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
            const f16x8 t = __builtin_convertvector(f, f16x8);
            *reinterpret_cast<u32x4*>(hb + ((nb * 2 + gp) * 64 + lane) * 8) = __builtin_bit_cast(u32x4, t);
        }
    if (__any(!(amax <= 65504.0f))) {  // rare: count exactly (NaN counts too)
        unsigned c = 0;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) c += !(fabsf(x[nb][r]) <= 65504.0f);
        if (c) atomicAdd(satcnt, c);
    }
}

// ---- LDS ring of 32 KiB blocks fed by asynchronous global->LDS DMA, shared by the NW waves of a
// workgroup.  A block = 4 segments of 8 KiB (8 fragments each), every segment contiguous in global
// memory; source and LDS image are both lane-linear, so one DMA instruction moves 1 KiB (64 lanes x
// 16 B) and wave w issues instructions i = w, w+NW, ... of the block's 32.
//   NW = 4: 2 slots, the DMA runs one block ahead (two workgroups per CU hide its latency);
//   NW = 8: 4 slots, the DMA runs THREE blocks ahead -- a bf16 block is only ~2000 cycles of work,
//           less than the DMA latency under load -- and the 8 waves halve the stream per data row.
// Completion is tracked with a COUNTED s_waitcnt: loads return in order, so "at most PER * k
// outstanding" (k = DMA blocks issued after block t) implies block t has landed, whatever other
// vector-memory operations are in flight (they can only make the wait longer, never shorter).
// Ring waits of the 2-slot ring leave a wave's own result stores in flight (round 5): the residual block's and the Q / K blocks'
// eight stores are YOUNGER than the DMA block the next ring step waits for, vector-memory operations retire in order, so
// vmcnt(8) instead of vmcnt(0) says "the DMA has landed" without draining the stores' write acknowledgements three times per
// block (bf16 row launch at [256,800,80]: 105.3 -> 104.0 us, the forward -0.5 %; same-box A/B, scripts/ubench/ab_head.sh).
// Round 2 had tried this on the fp32-residual kernels and seen no change.  0 = the drained form (A/B builds).
#ifndef SAVAD_STORES_IN_FLIGHT
#define SAVAD_STORES_IN_FLIGHT 1
#endif
template <int NW, int NR = (NW == 8 ? 4 : 2), int DP = NR - 1>
struct Ring {
    static constexpr int NRING = NR;
    static constexpr int DEPTH = DP;  // DMA blocks in flight beyond the one being consumed (NRING - DEPTH blocks stay readable)
    static constexpr int PER = 32 / NW;
    // gfx9 s_waitcnt immediate: vmcnt = bits 3:0 and 15:14, expcnt / lgkmcnt "no wait"
    static constexpr int vmcnt_imm(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }
    char* base;
    int w, lane;

    __device__ __forceinline__ char* slot(int t) const { return base + (t & (NRING - 1)) * RING_BYTES; }

    // One DMA instruction moves 1 KiB (64 lanes x 16 B); piece i of the block's 32 reads KiB (i & 7) of segment i >> 3 and
    // lands at slot + i KiB.  A wave's pieces are CONSECUTIVE KiB of ONE segment, four per M0 write: the instruction's
    // immediate offset moves the global source and the LDS destination together (measured: scripts/ubench/dma_issue.hip),
    // so a group of four costs s_add_u32 m0 + the hazard nop + 4 loads (round 2 paid the M0 write and the nop per piece).
    // M0 is not saved / restored: nothing else in these kernels uses it (gfx9 DS instructions do not need it;
    // tests/test_abi_and_host.py checks the disassembly).
    template <int IMM>
    static __device__ __forceinline__ void dma4k(const char* src, unsigned ldsb, unsigned voff) {
        asm volatile(
            "s_add_u32 m0, %2, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %1\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:3072"
            :
            : "v"(voff), "s"(src), "s"(ldsb), "n"(IMM)
            : "memory", "scc");
    }
    template <class SegSrc>
    __device__ __forceinline__ void issue(int t, SegSrc seg_src) const {
        if (SAVAD_ABLATE & 1) return;
        // LDS byte address = low 32 bits of the flat address (the LDS aperture is 4 GiB aligned).  An explicit
        // generic -> address_space(3) cast adds a null check that hipcc 7.2 mis-selects in one instantiation
        // ("V_CMP_NE_U32 0, $src_shared_base: operand has incorrect register class").
        const unsigned slot0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)slot(t));
        const unsigned off = (unsigned)lane * 16u, off4 = off + 4u * FRAG_BYTES;
        if (NW == 4) {  // wave w moves segment w: KiB 8w .. 8w+7 of the block
            const char* s0 = seg_src(w);
            const unsigned ldsb = slot0 + (unsigned)w * BLK_BYTES;
            dma4k<0>(s0, ldsb, off);
            dma4k<4096>(s0, ldsb, off4);
        } else {  // NW == 8: wave w moves half a segment: KiB 4w .. 4w+3
            const char* s0 = seg_src(w >> 1) + (size_t)(w & 1) * 4 * FRAG_BYTES;
            dma4k<0>(s0, slot0 + (unsigned)w * 4 * FRAG_BYTES, off);
        }
    }
    // wait until block t has landed for every wave; `newer` = DMA blocks issued after block t (wave-uniform)
    // stores_after (wave-uniform, 0 or 8): result stores this wave issued AFTER its newest DMA block -- they may stay in flight
    __device__ __forceinline__ void acquire(int newer, int stores_after = 0) const {
        if (SAVAD_ABLATE & 2) return;
        // builtin, not asm: see wait_vmem_all().  gfx9 encoding: vmcnt in bits 3:0 (PER <= 8), others "no wait"
        if (DEPTH >= 3 && newer >= 2)
            __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 * PER));
        else if (DEPTH >= 2 && newer == 1)
            __builtin_amdgcn_s_waitcnt(vmcnt_imm(PER));
        else if (SAVAD_STORES_IN_FLIGHT && DEPTH == 1 && stores_after == 8)
            __builtin_amdgcn_s_waitcnt(vmcnt_imm((SAVAD_FAULT_INJECT & 8) ? 9 : 8));
        else
            __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
        asm volatile("" ::: "memory");
        if (!(SAVAD_FAULT_INJECT & 4)) __syncthreads();
    }
};

// acc[nbl] += W[ring n-block nbl] . x   (transposed form: lane = data row, registers = output features); SWAP: operands
// swapped (lane = output feature, registers = data rows: V^T).
// The 32 weight fragments of a ring block are read from LDS by HAND-issued ds_read_b128 with counted lgkmcnt waits, RING_PIPE
// fragments ahead of the MFMA that consumes them (round 5).  Left to the compiler, the reads run two fragments ahead whatever
// the register budget (ds_read, ds_read, wait, MFMA, wait, MFMA, ...): every pair of MFMAs (64 cycles) then waits out an LDS
// round trip, and a wave alone on its SIMD kept the matrix pipe ~40 % busy (scripts/ubench/phase_timing_packed_bf16.py).
// Counting is safe against LDS / scalar-memory operations the compiler issues in between: LDS data returns in order, so
// "at most k operations outstanding" can only be reached once everything older than the last k has landed -- extra
// operations make the wait longer, never shorter.  The same order of accumulation as before: the same bits.
#ifndef SAVAD_RING_PIPE
#define SAVAD_RING_PIPE 6
#endif
#ifndef SAVAD_RING_INTERLEAVE
#define SAVAD_RING_INTERLEAVE 1   // 0: block after block (A/B builds; [256,800,80] last row launch 77.6 -> 76.4 us, [65536,7,80] 0.648 -> 0.642 ms with 1)
#endif
template <bool SWAP>
__device__ __forceinline__ void gemm_ring_t(f32x16 (&acc)[4], const char* ringblk, const bf16x8 (&xp)[8], int lane) {
#if SAVAD_RING_PIPE > 0
    constexpr int P = SAVAD_RING_PIPE;
    const unsigned a = (unsigned)(size_t)ringblk + (unsigned)lane * 16u;  // LDS byte address (low half of the flat address)
    u32x4 f[P];
    // (macros, not a compile-time loop over a generic lambda: clang rejects asm operands that name captured variables there)
    // step i works on output block NB(i) and K-step KS(i).  SAVAD_RING_INTERLEAVE: the four accumulators take turns (every
    // accumulator still sees its K-steps in order: the same bits), so that no MFMA waits for the one issued just before it
#if SAVAD_RING_INTERLEAVE
#define SAVAD_RING_NB(i) ((i) % 4)
#define SAVAD_RING_KS(i) ((i) / 4)
#else
#define SAVAD_RING_NB(i) ((i) / 8)
#define SAVAD_RING_KS(i) ((i) % 8)
#endif
#define SAVAD_RING_LOAD(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[(i) % P]) : "v"(a), "n"((SAVAD_RING_NB(i) * 8 + SAVAD_RING_KS(i)) * FRAG_BYTES))
#define SAVAD_RING_STEP(i)                                                                                              \
    {                                                                                                                   \
        constexpr int newer_ = (31 - (i) < P - 1 ? 31 - (i) : P - 1) + ((SAVAD_FAULT_INJECT & 2) ? 1 : 0); /* of this statement's reads */ \
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f[(i) % P]) : "n"(newer_));                                         \
        const bf16x8 w_ = __builtin_bit_cast(bf16x8, f[(i) % P]);                                                       \
        acc[SAVAD_RING_NB(i)] = SWAP ? SAVAD_MFMA_BF16(xp[SAVAD_RING_KS(i)], w_, acc[SAVAD_RING_NB(i)])                 \
                                     : SAVAD_MFMA_BF16(w_, xp[SAVAD_RING_KS(i)], acc[SAVAD_RING_NB(i)]);                \
        if constexpr ((i) + P < 32) SAVAD_RING_LOAD((i) + P);                                                           \
    }
#define SAVAD_RING_STEP4(i) SAVAD_RING_STEP(i) SAVAD_RING_STEP((i) + 1) SAVAD_RING_STEP((i) + 2) SAVAD_RING_STEP((i) + 3)
    SAVAD_RING_LOAD(0);
    SAVAD_RING_LOAD(1);
    if constexpr (P > 2) SAVAD_RING_LOAD(2);
    if constexpr (P > 3) SAVAD_RING_LOAD(3);
    if constexpr (P > 4) SAVAD_RING_LOAD(4);
    if constexpr (P > 5) SAVAD_RING_LOAD(5);
    if constexpr (P > 6) SAVAD_RING_LOAD(6);
    if constexpr (P > 7) SAVAD_RING_LOAD(7);
    static_assert(P >= 2 && P <= 8, "SAVAD_RING_PIPE");
    SAVAD_RING_STEP4(0) SAVAD_RING_STEP4(4) SAVAD_RING_STEP4(8) SAVAD_RING_STEP4(12) SAVAD_RING_STEP4(16) SAVAD_RING_STEP4(20)
    SAVAD_RING_STEP4(24) SAVAD_RING_STEP4(28)
#undef SAVAD_RING_STEP4
#undef SAVAD_RING_STEP
#undef SAVAD_RING_LOAD
#undef SAVAD_RING_NB
#undef SAVAD_RING_KS
#else
#pragma unroll
    for (int nbl = 0; nbl < 4; ++nbl)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 w = ldfrag(ringblk + ((nbl * 8 + ks) * 64 + lane) * 16);
            acc[nbl] = SWAP ? SAVAD_MFMA_BF16(xp[ks], w, acc[nbl]) : SAVAD_MFMA_BF16(w, xp[ks], acc[nbl]);
        }
#endif
}
__device__ __forceinline__ void gemm_ring(f32x16 (&acc)[4], const char* ringblk, const bf16x8 (&xp)[8], int lane) {
    gemm_ring_t<false>(acc, ringblk, xp, lane);
}
__device__ __forceinline__ void gemm_ring_swapped(f32x16 (&acc)[4], const char* ringblk, const bf16x8 (&xp)[8], int lane) {
    gemm_ring_t<true>(acc, ringblk, xp, lane);
}

// One of the three QKV ring blocks (rb = 0 query, 1 key: transposed form; 2 value: swapped form -> V^T)
// Q is stored PRE-SCALED by qscale = log2(e)/sqrt(D): the attention stage then gets its scores directly in
// the base-2 exponent domain and spends no VALU instruction on scaling them.
__device__ __forceinline__ void qkv_block_bf16(int rb, const char* ringblk, const bf16x8 (&xp)[8], const float* lbq,
                                               char* __restrict__ qf, char* __restrict__ kf, char* __restrict__ vtf,
                                               int blk, int lane, float qscale, bool live = true /* wave-uniform: store */) {
    const int n = lane & 31, h = lane >> 5;
    f32x16 acc[4];
    if (rb < 2) {
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) acc[nbl] = bias_block(lbq + D * rb + 32 * nbl, h);
        gemm_ring(acc, ringblk, xp, lane);
    } else {
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
            const float bv = lbq[2 * D + 32 * nbl + n];  // lane = output feature
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nbl][r] = bv;
        }
        gemm_ring_swapped(acc, ringblk, xp, lane);
    }
    char* dst = rb == 0 ? qf : (rb == 1 ? kf : vtf);
    if (rb == 0) {
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) acc[nbl] *= qscale;
    }
    if (!live) return;
#pragma unroll
    for (int nbl = 0; nbl < 4; ++nbl)
#pragma unroll
        for (int j = 0; j < 2; ++j) stfrag(dst + ((size_t)blk * 8 + 2 * nbl + j) * FRAG_BYTES + lane * 16, pack_half(acc[nbl], j));
}

// features f0..f0+3 and f0+8..f0+11 of one input row -> the 8 bf16 of an input K-step fragment
__device__ __forceinline__ bf16x8 load_x_frag(const float* p, bool valid) {
    f32x4 a = ld4(p), b = ld4(p + 8);
    if (!valid) a = b = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = (__bf16)a[e];
        r[4 + e] = (__bf16)b[e];
    }
    return r;
}
__device__ __forceinline__ bf16x8 load_x_frag(const __bf16* p, bool valid) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 a = *reinterpret_cast<const u32x2*>(p), b = *reinterpret_cast<const u32x2*>(p + 8);
    if (!valid) a = b = u32x2{0u, 0u};
    return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
}

// ---------------------------------------------------------------------------------------------
// Kernel 1 (bf16): input Linear + PE -> h (fp32) -> LN -> Q, K, V^T fragments.  NW waves = NW blocks.
// XT = float or __bf16 input features.
// ---------------------------------------------------------------------------------------------
template <typename XT, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void input_qkv_kernel_bf16(
    const XT* __restrict__ x, long xbs, int B, int T, int F, int nblk, const char* __restrict__ win_frag,
    const float* __restrict__ bin, const float* __restrict__ pe, const char* __restrict__ wqkv_frag,
    const float* __restrict__ bqkv, hres_t* __restrict__ hbuf, char* __restrict__ qf, char* __restrict__ kf,
    char* __restrict__ vtf, float qscale, unsigned* __restrict__ satcnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using R = Ring<NW>;
    float* lbq = reinterpret_cast<float*>(smem + R::NRING * RING_BYTES);
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * NW + w;
    const R ring{smem, w, lane};
    constexpr int NBLK = 3;
    auto issue = [&](int t) { ring.issue(t, [&](int sgm) { return wqkv_frag + (size_t)t * RING_BYTES + sgm * BLK_BYTES; }); };
#pragma unroll
    for (int t = 0; t < R::DEPTH && t < NBLK; ++t) issue(t);
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
    const XT* xr = x + x_row_offset(row, T, F, xbs);
    for (int ks = 0; ks < KS; ++ks) {
        const int f0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
        const bf16x8 xf = load_x_frag(xr + f0, valid);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
            h0[nb] = SAVAD_MFMA_BF16(ldfrag(win_frag + ((size_t)(nb * KS + ks) * 64 + lane) * 16), xf, h0[nb]);
    }
    store_hblock(hbuf + (size_t)blk * HBLK_FLOATS, h0, lane, satcnt);
    f32x4 xg[16];
    layernorm_regs(h0, xg);
    bf16x8 xp[8];
    pack_row(xg, xp);
#pragma unroll
    for (int t = 0; t < NBLK; ++t) {
        ring.acquire(NBLK - 1 - t < R::DEPTH - 1 ? NBLK - 1 - t : R::DEPTH - 1);
        if (t + R::DEPTH < NBLK) issue(t + R::DEPTH);
        qkv_block_bf16(t, ring.slot(t), xp, lbq, qf, kf, vtf, blk, lane, qscale);
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel 1, persistent form (round 5): the stage's weights -- layer 0's Q / K / V fragments (96 KiB) and the input Linear's
// (4 KS KiB, KS = F / 16) -- are RESIDENT in LDS, one workgroup of NW waves per CU, every wave walking blocks on its own:
// no weight ring, no barrier after the prologue, and the 96 KiB a 4-block workgroup of the kernel above pulls through the
// L2 for every 128 rows (157 MB per launch at [256,800,80], as much again as the stage writes) are pulled once per CU.
// Block order: round r of the launch covers blocks r G NW .. (r+1) G NW - 1 (G workgroups), wave w of workgroup g taking
// block r G NW + w G + g -- neighbouring CUs write neighbouring blocks, and a thin last round spreads over all CUs.
// KSC = KS at compile time (5: the reference's 80 mel bands) adds a software pipeline: the NEXT block's feature pieces and
// positional-encoding rows are requested by hand-issued loads while this block's LayerNorm and Q / K / V products run, and
// waited for with ONE counted s_waitcnt at the bottom of this one -- vmcnt(24): the 24 fragment stores of this block are
// younger than every one of those loads, and vector-memory operations retire in order, so "at most 24 outstanding" means
// all the loads have landed while the stores may still be in flight (the ablation of the un-pipelined form, scripts/ubench/
// input_p_ablate.sh: 72 us with, 47 us without the feature loads -- five dependent round trips to memory per block).
// KSC = 0: any KS, features loaded where they are used.
// Arithmetic, operand for operand, is the kernel above's: the same bits.
// ---------------------------------------------------------------------------------------------
constexpr int input_p_lds_bytes(int KS) { return 3 * RING_BYTES + KS * 4 * FRAG_BYTES + 4 * D * 4; }
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <typename XT>
struct XRaw;
template <>
struct XRaw<float> {
    f32x4 a, b;
};
template <>
struct XRaw<__bf16> {
    u32x2 a, b;
};
// the two pieces of K-step ks of one input row (load_x_frag's addresses), requested without waiting
template <int KS>
__device__ __forceinline__ void request_x(XRaw<float> (&r)[KS], const float* lanep /* &x[row][4 h] */) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r[ks].a) : "v"(lanep), "n"(64 * ks) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r[ks].b) : "v"(lanep), "n"(64 * ks + 32) : "memory");
    }
}
template <int KS>
__device__ __forceinline__ void request_x(XRaw<__bf16> (&r)[KS], const __bf16* lanep) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(r[ks].a) : "v"(lanep), "n"(32 * ks) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(r[ks].b) : "v"(lanep), "n"(32 * ks + 16) : "memory");
    }
}
__device__ __forceinline__ bf16x8 x_frag(const XRaw<float>& r, bool valid) {
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = (__bf16)(valid ? r.a[e] : 0.0f);
        f[4 + e] = (__bf16)(valid ? r.b[e] : 0.0f);
    }
    return f;
}
__device__ __forceinline__ bf16x8 x_frag(const XRaw<__bf16>& r, bool valid) {
    return __builtin_bit_cast(bf16x8, valid ? u32x4{r.a[0], r.a[1], r.b[0], r.b[1]} : u32x4{0u, 0u, 0u, 0u});
}
// a positional-encoding row in add_block's pieces: piece 4 nb + g = features 32 nb + 8 g + 4 h ..
__device__ __forceinline__ void request_pe(f32x4 (&r)[16], const float* lanep /* &pe[t][4 h] */) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r[i]) : "v"(lanep), "n"(128 * (i >> 2) + 32 * (i & 3)) : "memory");
}
template <int PENDING>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PENDING) : "memory");
}
template <class V>
__device__ __forceinline__ void landed(V& v) {   // orders every use of v behind the vm_wait in front of it
    asm volatile("" : "+v"(v));
}

template <typename XT, int NW, int KSC>
__global__ __launch_bounds__(64 * NW, 1) void input_qkv_kernel_bf16_p(
    const XT* __restrict__ x, long xbs, int B, int T, int F, int nblk, int nblk_pad, const char* __restrict__ win_frag,
    const float* __restrict__ bin, const float* __restrict__ pe, const char* __restrict__ wqkv_frag,
    const float* __restrict__ bqkv, hres_t* __restrict__ hbuf, char* __restrict__ qf, char* __restrict__ kf,
    char* __restrict__ vtf, float qscale, unsigned* __restrict__ satcnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KS = KSC ? KSC : F / 16;
    char* lwin = smem + 3 * RING_BYTES;
    float* lbq = reinterpret_cast<float*>(lwin + KS * 4 * FRAG_BYTES);
    float* lbin = lbq + 3 * D;
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {   // prologue: 4-KiB pieces of the two weight images, wave w moving pieces w, w + NW, ... by LDS-DMA
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
        const int npieces = 24 + KS;
        for (int p = w; p < npieces; p += NW) {
            const char* src = p < 24 ? wqkv_frag + (size_t)p * 4 * FRAG_BYTES : win_frag + (size_t)(p - 24) * 4 * FRAG_BYTES;
            Ring<4>::dma4k<0>(src, lds0 + (unsigned)p * 4 * FRAG_BYTES, (unsigned)lane * 16u);
        }
        wait_vmem_all();
        __syncthreads();
        for (int i = threadIdx.x * 4; i < 3 * D; i += 64 * NW * 4) st4(lbq + i, ld4(bqkv + i));
        for (int i = threadIdx.x * 4; i < D; i += 64 * NW * 4) st4(lbin + i, ld4(bin + i));
        __syncthreads();
    }
    const int per_round = gridDim.x * NW;
    int blk = w * gridDim.x + blockIdx.x;
    if (blk >= nblk_pad) return;   // (the padding blocks are written like the kernel above writes them)
    size_t row;
    int t_frame;
    bool valid = (blk < nblk) && slot_row(B, T, blk, m, row, t_frame);
    if (!valid) {
        row = 0;
        t_frame = 0;
    }
    XRaw<XT> xr[KSC ? KSC : 1];
    f32x4 per[16];
    if constexpr (KSC > 0) {
        request_x(xr, x + x_row_offset(row, T, F, xbs) + 4 * h);
        request_pe(per, pe + (size_t)t_frame * D + 4 * h);
        vm_wait<0>();
    }
    for (;;) {
        f32x16 h0[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            h0[nb] = zero16();
            add_bias(h0[nb], lbin + 32 * nb, h);
        }
        const int nxt = blk + per_round;
        const bool has_next = nxt < nblk_pad;   // wave-uniform
        size_t nrow = 0;
        int nt = 0;
        bool nvalid = false;
        if (has_next) {
            nvalid = (nxt < nblk) && slot_row(B, T, nxt, m, nrow, nt);
            if (!nvalid) {
                nrow = 0;
                nt = 0;
            }
        }
        if constexpr (KSC > 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) landed(per[i]);   // (behind the vm_wait in front of the loop / at its bottom)
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks) {
                landed(xr[ks].a);
                landed(xr[ks].b);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) h0[nb][4 * g + s_] += per[4 * nb + g][s_];
            bf16x8 xf[KSC];
            if (__all(valid)) {   // (wave-uniform: all but a sequence's last block)
#pragma unroll
                for (int ks = 0; ks < KSC; ++ks) xf[ks] = x_frag(xr[ks], true);
            } else {
#pragma unroll
                for (int ks = 0; ks < KSC; ++ks) xf[ks] = x_frag(xr[ks], valid);
            }
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    h0[nb] = SAVAD_MFMA_BF16(ldfrag(lwin + ((nb * KSC + ks) * 64 + lane) * 16), xf[ks], h0[nb]);
            if (has_next) request_x(xr, x + x_row_offset(nrow, T, F, xbs) + 4 * h);
        } else {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) add_block(h0[nb], pe + (size_t)t_frame * D + 32 * nb, h);
            const XT* xrow = x + x_row_offset(row, T, F, xbs);
            for (int ks = 0; ks < KS; ++ks) {
                const int f0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
                const bf16x8 xf = (SAVAD_ABLATE & 64) ? ldfrag(lwin + lane * 16 + ks * 64) : load_x_frag(xrow + f0, valid);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) h0[nb] = SAVAD_MFMA_BF16(ldfrag(lwin + ((nb * KS + ks) * 64 + lane) * 16), xf, h0[nb]);
            }
        }
        if (!(SAVAD_ABLATE & 32) || qscale < 0.0f) store_hblock(hbuf + (size_t)blk * HBLK_FLOATS, h0, lane, satcnt);
        f32x4 xg[16];
        layernorm_regs(h0, xg);
        bf16x8 xp[8];
        pack_row(xg, xp);
        if constexpr (KSC > 0) {
            if (has_next) request_pe(per, pe + (size_t)nt * D + 4 * h);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
            qkv_block_bf16(t, smem + t * RING_BYTES, xp, lbq, qf, kf, vtf, blk, lane, qscale, !(SAVAD_ABLATE & 16) || qscale < 0.0f);
        // the next block's requests are older than this block's 24 fragment stores (in front of the exit test: every path from
        // a request to the loop's top passes the wait, which is what scripts/check_async_loads.py can verify)
        if constexpr (KSC > 0) vm_wait<24>();
        if (!has_next) break;
        blk = nxt;
        row = nrow;
        t_frame = nt;
        valid = nvalid;
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel 2 (bf16): flash attention on fragments.  T > 32: workgroup = (sequence, group of <= NW
// query blocks); K and V^T fragments of 2 key blocks (64 keys, 32 KiB) per ring block.
// Output: NORMALISED context as B-operand fragments.
// ---------------------------------------------------------------------------------------------
// Online softmax state of one query block.  Scores arrive in the base-2 exponent domain (Q is pre-scaled)
// and RELATIVE to a per-row reference: negm = -reference rides in as the C operand of the first S^T MFMA,
// so the common tile spends no instruction on "s*c - m*c".  The reference only moves when a row maximum
// drifts more than 2^RESCALE_LOG2 above it (or, on the first tile, away from the initial reference 0 in
// either direction): p = 2^(s - ref) <= 2^16 and sums of 800+ of them are far inside fp32 range.
struct AttnState {
    f32x16 O[4];
    f32x16 negm;  // all 16 registers of a lane hold -reference of the lane's query row
    float l_run;  // this lane's half of the running row sum
};
__device__ __forceinline__ void attn_state_init(AttnState& st) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) st.O[nb] = zero16();
    st.negm = zero16();
    st.l_run = 0.0f;
}
__device__ __forceinline__ void online_softmax_shifted(f32x16& sc, AttnState& st, bool first /* wave-uniform */) {
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
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        sc[r] = __builtin_amdgcn_exp2f(sc[r]);
        rs += sc[r];
    }
    st.l_run += rs;  // this lane's half of the row sum: the two halves meet once, when the context is normalised
}
// (Tried in round 2: taking the exponentials against the standing reference WITHOUT the row maxima and redoing a tile only
// when its row sum leaves a 2^60 window -- 19 fewer VALU instructions per tile, bit-compatible cold path -- made the
// stage 22 % SLOWER (97 -> 119 us at [256,800,80]): the rarely taken branch then sits between the exponentials and
// the PV MFMAs, which the scheduler no longer interleaves; here it sits before the exponentials.)
// `mask(sc)` sets the scores of keys that do not exist to NEG_BIG (lane (m,h), register r <-> key 8(r>>2)+4h+(r&3)).
template <class Mask>
__device__ __forceinline__ void attn_tile(AttnState& st, const bf16x8 (&qp)[8], const char* kblk, const char* vtblk,
                                          Mask mask, bool first, int lane) {
    f32x16 sc = st.negm;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
        sc = SAVAD_MFMA_BF16((SAVAD_ABLATE & 8) ? qp[7 - ks] : ldfrag(kblk + (ks * 64 + lane) * 16), qp[ks], sc);
    mask(sc);
    if (!(SAVAD_ABLATE & 4)) online_softmax_shifted(sc, st, first);
    const bf16x8 p0 = pack_half(sc, 0), p1 = pack_half(sc, 1);
#pragma unroll
    for (int nbd = 0; nbd < 4; ++nbd) {
        st.O[nbd] = SAVAD_MFMA_BF16((SAVAD_ABLATE & 8) ? qp[nbd] : ldfrag(vtblk + ((nbd * 2 + 0) * 64 + lane) * 16), p0, st.O[nbd]);
        st.O[nbd] = SAVAD_MFMA_BF16((SAVAD_ABLATE & 8) ? qp[4 + nbd] : ldfrag(vtblk + ((nbd * 2 + 1) * 64 + lane) * 16), p1, st.O[nbd]);
    }
}
// Invalid query slots (padding of the block space) get an exactly-zero context: a fully masked row
// would otherwise carry inf/NaN into h, K and V^T of that slot and poison the NEXT layer's PV product
// (probability 0 x NaN).
__device__ __forceinline__ void store_ctx(char* ctxf, int blk, AttnState& st, bool qvalid, int lane) {
    const float inv = 1.0f / half_sum(st.l_run);
#pragma unroll
    for (int nbd = 0; nbd < 4; ++nbd) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st.O[nbd][r] = qvalid ? st.O[nbd][r] * inv : 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) stfrag(ctxf + ((size_t)blk * 8 + 2 * nbd + j) * FRAG_BYTES + lane * 16, pack_half(st.O[nbd], j));
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void attention_kernel_bf16(const char* __restrict__ qf, const char* __restrict__ kf,
                                                                    const char* __restrict__ vtf, char* __restrict__ ctxf,
                                                                    int B, int T, int NG) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // ring of [K 2 blocks | V^T 2 blocks] stages
    using R = Ring<NW>;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int QB = (T + 31) / 32;
    int b, g;
    if (!xcd_balanced_map(B, NG, b, g)) return;
    const int qb0 = (g * QB) / NG, qb1 = ((g + 1) * QB) / NG;
    const int qb = qb0 + w;
    const bool active = qb < qb1;
    const int blk_q = b * QB + (active ? qb : qb0);
    const int NST = (QB + 1) / 2;
    const R ring{smem, w, lane};

    bf16x8 qp[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qp[ks] = ldfrag(qf + ((size_t)blk_q * 8 + ks) * FRAG_BYTES + lane * 16);
    AttnState st;
    attn_state_init(st);

    auto issue = [&](int stage) {
        const size_t kb0 = (size_t)b * QB + 2 * stage;
        ring.issue(stage, [&](int sgm) { return (sgm < 2 ? kf : vtf) + (kb0 + (sgm & 1)) * BLK_BYTES; });
    };
    for (int s0 = 0; s0 < R::DEPTH && s0 < NST; ++s0) issue(s0);
#ifdef SAVAD_TIMING
    long long tacc[4] = {0, 0, 0, 0}, tp = __builtin_readcyclecounter(), tn;
#define SAVAD_TB(i) do { tn = __builtin_readcyclecounter(); tacc[i] += tn - tp; tp = tn; } while (0)
#else
#define SAVAD_TB(i) do {} while (0)
#endif
    for (int stg = 0; stg < NST; ++stg) {
        SAVAD_TB(3);
        const int newer = NST - 1 - stg < R::DEPTH - 1 ? NST - 1 - stg : R::DEPTH - 1;
        ring.acquire(newer);
        SAVAD_TB(0);
        if (stg + R::DEPTH < NST) issue(stg + R::DEPTH);
        SAVAD_TB(1);
        if (!active) continue;
        const char* buf = ring.slot(stg);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int jt = 2 * stg + tt;
            if (jt >= QB) break;
            // Only the last tile of a ragged sequence has missing keys.  The empty asm keeps this a REAL
            // (wave-uniform) branch: if-converted, its 16 compares + selects + index arithmetic would run on
            // every tile, and VALU instructions do not hide behind this wave's MFMAs (~1 ns each, measured).
            auto mask = [&](f32x16& sc) {
                if (32 * jt + 32 > T) {
                    asm volatile("" ::: "memory");
                    const int lim = T - 32 * jt - 4 * h;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] = (8 * (r >> 2) + (r & 3) < lim) ? sc[r] : NEG_BIG;
                }
            };
            attn_tile(st, qp, buf + tt * BLK_BYTES, buf + (2 + tt) * BLK_BYTES, mask, jt == 0, lane);
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
                                                                       char* __restrict__ ctxf, int B, int T, int nblk) {
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * 4 + w;
    if (blk >= nblk) return;
    const int G = 32 / T;
    bf16x8 qp[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qp[ks] = ldfrag(qf + ((size_t)blk * 8 + ks) * FRAG_BYTES + lane * 16);
    AttnState st;
    attn_state_init(st);
    bool keyok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int jk = 8 * (r >> 2) + 4 * h + (r & 3);
        keyok[r] = (jk < G * T) && (jk / T == m / T) && (blk * G + jk / T < B);
    }
    auto mask = [&](f32x16& sc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = keyok[r] ? sc[r] : NEG_BIG;
    };
    attn_tile(st, qp, kf + (size_t)blk * BLK_BYTES, vtf + (size_t)blk * BLK_BYTES, mask, true, lane);
    store_ctx(ctxf, blk, st, (m < G * T) && (blk * G + m / T < B), lane);
}

// ---------------------------------------------------------------------------------------------
// Kernel 3 (bf16, per layer): out-projection + residual -> LN -> FFN (4 chunks of 128 hidden units,
// ReLU output repacked in registers) + residual -> next layer's LN + Q/K/V^T, or the classifier.
// Weight stream = ring blocks  0: Wo | 1+2c: W1 chunk c | 2+2c: W2 chunk c (c = 0..3) | 9,10,11: Wq, Wk, Wv.
// ---------------------------------------------------------------------------------------------
struct RowArgsBf16 {
    int B, T, nblk;
    hres_t* hbuf;
    const char* wo_frag;
    const float* bo;
    const char* w1_frag;
    const float* b1;
    const char* w2_frag;
    const float* b2;
    const char* wn_frag;  // !LAST: next layer's Wqkv' fragments
    const float* wc;      // LAST: Wc' fp32 [2][D]
    const float* bn;      // !LAST: bqkv' [384]; LAST: bc' [2]
    char *qf, *kf, *vtf;  // !LAST: written (the NEXT layer's buffers)
    float* out;           // LAST
    float qscale;
    unsigned* satcnt;     // residual-stream saturation counter (store_hblock)
};

// The whole row chain of block `blk` for one wave.  In: xp = the block's attention context as B-operand fragments.
// `smem` = the workgroup's ring (NRING x 32 KiB) followed by 9*D floats of biases; nothing of it may still be in use
// by other waves when this is entered (the first ring barrier inside publishes ring block 0 and the biases).
// live (wave-uniform) = this wave's block exists: a wave without a block still issues its share of the DMA and takes
// part in every barrier, but stores nothing.
// ALWAYS_LIVE: `live` is the constant true (the row launch: every wave has a block) -- only then do the ring waits of the Q / K / V
// steps leave the wave's stores in flight: with a run-time `live` the stores and the wait's count sit behind two branches on the
// same flag, which no static check can tie together (scripts/check_async_loads.py, the publication rule).
template <bool LAST, int NW, class R = Ring<NW>, bool ALWAYS_LIVE = false>
__device__ __forceinline__ void row_stage_bf16(const RowArgsBf16& A, char* smem, bf16x8 (&xp)[8], int blk, bool live, int lane, int w) {
    float* lbo = reinterpret_cast<float*>(smem + R::NRING * RING_BYTES);
    float* lb1 = lbo + D;
    float* lb2 = lb1 + DFF;
    float* lbn = lb2 + D;
    const int m = lane & 31, h = lane >> 5;
    const R ring{smem, w, lane};
    constexpr int NBLK = LAST ? 9 : 12;
    auto issue = [&](int t) {
        ring.issue(t, [&](int sgm) -> const char* {
            if (t == 0) return A.wo_frag + sgm * BLK_BYTES;
            if (t < 9) {
                const int c = (t - 1) >> 1;
                return ((t - 1) & 1) ? A.w2_frag + (size_t)(sgm * 32 + 8 * c) * FRAG_BYTES  // output block sgm, K-steps 8c..8c+7
                                     : A.w1_frag + (size_t)c * RING_BYTES + sgm * BLK_BYTES;
            }
            return A.wn_frag + (size_t)(t - 9) * RING_BYTES + sgm * BLK_BYTES;
        });
    };
    // acquire block t (wave-uniform t), then keep the DMA DEPTH blocks ahead
    auto advance = [&](int t, int stores_after = 0) {
        if ((SAVAD_FAULT_INJECT & 16) && R::DEPTH == 1 && t == 2) {   // the planted publication fault: block 2 requested HERE, handed over unwaited
            issue(2);
            __syncthreads();
        } else {
            ring.acquire(NBLK - 1 - t < R::DEPTH - 1 ? NBLK - 1 - t : R::DEPTH - 1, stores_after);
        }
        if (t + R::DEPTH < NBLK && !((SAVAD_FAULT_INJECT & 16) && R::DEPTH == 1 && t + R::DEPTH == 2)) issue(t + R::DEPTH);
    };
#pragma unroll
    for (int t = 0; t < R::DEPTH; ++t) issue(t);
    // Every request of the prologue goes out before anything waits (round 5; stage_bias_pieces): the residual block first, then
    // the bias pieces -- LAST: the classifier's [2][D] weights and its bias take the unused Q/K/V bias slot, its tail reads them
    // from LDS, not from global memory -- and one round trip later the ring's first barrier publishes them.
    const BiasPiece pieces[4] = {{lbo, A.bo, D}, {lb1, A.b1, DFF}, {lb2, A.b2, D}, {lbn, LAST ? A.wc : A.bn, LAST ? 2 * D : 3 * D}};
    const BiasRegs<4> breg = request_bias_pieces(pieces);
    const float bc = LAST ? A.bn[threadIdx.x & 1] : 0.0f;
    hres_t* hb = A.hbuf + (size_t)blk * HBLK_FLOATS;
    f32x16 h1[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) h1[nb] = zero16();
    load_hblock(h1, hb, lane);
    commit_bias_pieces(pieces, breg);
    if (LAST && threadIdx.x < 2) lbn[2 * D + threadIdx.x] = bc;
    // ---- h1 = h + bo + ctx Wo^T
    advance(0);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) h1[nb] += bias_block(lbo + 32 * nb, h);
    gemm_ring(h1, ring.slot(0), xp, lane);
    f32x4 xg[16];
    layernorm_regs(h1, xg);
    pack_row(xg, xp);
    // ---- FFN; its accumulators START from the residual stream (h1 + b2), so nothing is parked in HBM
    f32x16(&o)[4] = h1;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) o[nb] += bias_block(lb2 + 32 * nb, h);
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
        advance(1 + 2 * ch);  // W1 chunk
        f32x16 a[4];
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) a[nbl] = bias_block(lb1 + 128 * ch + 32 * nbl, h);
        gemm_ring(a, ring.slot(1 + 2 * ch), xp, lane);
        bf16x8 ap[8];
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
            ap[2 * nbl] = relu_frag(pack_half(a[nbl], 0));
            ap[2 * nbl + 1] = relu_frag(pack_half(a[nbl], 1));
        }
        advance(2 + 2 * ch);  // W2 chunk
        gemm_ring(o, ring.slot(2 + 2 * ch), ap, lane);
    }
    if (!LAST && live) store_hblock(hb, o, lane, A.satcnt);
    layernorm_regs(o, xg);
    if (!LAST) {
        pack_row(xg, xp);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            advance(9 + rb, ALWAYS_LIVE ? 8 : 0);   // the residual block's / the previous Q / K block's eight stores are younger than the DMA waited for
            qkv_block_bf16(rb, ring.slot(9 + rb), xp, lbn, A.qf, A.kf, A.vtf, blk, lane, A.qscale, live);
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

template <bool LAST, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void row_kernel_bf16(const char* __restrict__ ctxf, RowArgsBf16 A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * NW + w;
    bf16x8 xp[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        xp[ks] = ldfrag(ctxf + ((size_t)blk * 8 + ks) * FRAG_BYTES + lane * 16);
        if (blk >= A.nblk) xp[ks] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});  // pad blocks: attention never wrote them
    }
    row_stage_bf16<LAST, NW, Ring<NW>, true>(A, smem, xp, blk, true, lane, w);
}

// ---------------------------------------------------------------------------------------------
// Fused stage (bf16; T > 32): attention of one (sequence, group of <= NW query blocks) immediately followed by the
// row chain of those blocks in the same workgroup.  The normalised context is repacked into B-operand fragments in
// registers (what store_ctx used to write and the row kernel to read back: 2 x 52 MB per layer at B=256), the K/V
// ring becomes the weight ring, and the HBM-heavy row phase of one workgroup overlaps the LDS/MFMA-heavy attention
// phase of its co-resident partner.  q/k/v^T are double-buffered between layers (read qf,kf,vtf -- write A.qf,A.kf,
// A.vtf): a workgroup writes the next layer's K/V blocks while others still read this layer's.
// ---------------------------------------------------------------------------------------------
template <bool LAST, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void attention_row_kernel_bf16(const char* __restrict__ qf, const char* __restrict__ kf,
                                                                                      const char* __restrict__ vtf, int NG, RowArgsBf16 A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using R = Ring<NW>;
    const int B = A.B, T = A.T;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int QB = (T + 31) / 32;
    int b, g;
    if (!xcd_balanced_map(B, NG, b, g)) return;
    const int qb0 = (g * QB) / NG, qb1 = ((g + 1) * QB) / NG;
    const int qb = qb0 + w;
    const bool active = qb < qb1;
    const int blk_q = b * QB + (active ? qb : qb0);
    const int NST = (QB + 1) / 2;
    const R ring{smem, w, lane};

    bf16x8 qp[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qp[ks] = ldfrag(qf + ((size_t)blk_q * 8 + ks) * FRAG_BYTES + lane * 16);
    AttnState st;
    attn_state_init(st);
    auto issue = [&](int stage) {
        const size_t kb0 = (size_t)b * QB + 2 * stage;
        ring.issue(stage, [&](int sgm) { return (sgm < 2 ? kf : vtf) + (kb0 + (sgm & 1)) * BLK_BYTES; });
    };
    for (int s0 = 0; s0 < R::DEPTH && s0 < NST; ++s0) issue(s0);
    for (int stg = 0; stg < NST; ++stg) {
        const int newer = NST - 1 - stg < R::DEPTH - 1 ? NST - 1 - stg : R::DEPTH - 1;
        ring.acquire(newer);
        if (stg + R::DEPTH < NST) issue(stg + R::DEPTH);
        if (!active) continue;
        const char* buf = ring.slot(stg);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int jt = 2 * stg + tt;
            if (jt >= QB) break;
            auto mask = [&](f32x16& sc) {  // a REAL branch: see attention_kernel_bf16
                if (32 * jt + 32 > T) {
                    asm volatile("" ::: "memory");
                    const int lim = T - 32 * jt - 4 * h;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] = (8 * (r >> 2) + (r & 3) < lim) ? sc[r] : NEG_BIG;
                }
            };
            attn_tile(st, qp, buf + tt * BLK_BYTES, buf + (2 + tt) * BLK_BYTES, mask, jt == 0, lane);
        }
    }
    // context -> B-operand fragments, in registers (invalid slots and waves without a block: exact zeros)
    bf16x8 xp[8];
    {
        const bool qvalid = active && 32 * qb + (lane & 31) < T;
        const float inv = qvalid ? 1.0f / half_sum(st.l_run) : 0.0f;
#pragma unroll
        for (int nbd = 0; nbd < 4; ++nbd) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st.O[nbd][r] = qvalid ? st.O[nbd][r] * inv : 0.0f;
            xp[2 * nbd] = pack_half(st.O[nbd], 0);
            xp[2 * nbd + 1] = pack_half(st.O[nbd], 1);
        }
    }
    // A key stage of two blocks may over-read ONE V^T block behind the batch in the next layer: never written by
    // this kernel, and probability 0 times a non-finite value would poison the context.
    if (!LAST && b == B - 1 && g == NG - 1 && w == 0) {
#pragma unroll
        for (int f = 0; f < 8; ++f) stfrag(A.vtf + ((size_t)B * QB * 8 + f) * FRAG_BYTES + lane * 16, __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u}));
    }
    __syncthreads();  // everyone is done with the K/V ring: it becomes the weight ring
    row_stage_bf16<LAST, NW>(A, smem, xp, blk_q, active, lane, w);
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
