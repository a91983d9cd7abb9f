// savad_logmel.h -- log-mel front-end on the GPU ("next" row 1).
// Restates, for the reference's only transform configuration (tests/configs/vad/train_config.yaml:
// 18-26: n_fft 512, hop 10 ms, window 25 ms, 80 mels @16 kHz), what
// vad/acoustics/transforms/log_mel_spectrogram.py:19-32 obtains from librosa 0.8.0:
//   frames (center=True, reflect padding, periodic Hann(400) zero-padded to 512) -> |rFFT|^2 ->
//   Slaney mel filterbank (norm="slaney", fmin 0, fmax 8 kHz) -> log(x + 1e-6) -> [N, 80], N = 1 + len/160.
//
// The STFT is a DFT-as-GEMM on the exact-fp32 MFMA, in the same transposed / row-layout form as the
// model kernels: Out^T[dft row][frame] = sum_k Wdft[row][k] * y[160*frame + 56 + k], k = 0..399 (only
// the 400 samples under the window), A operand = window-folded DFT rows (re/im of a bin interleaved
// in adjacent rows, so that |X|^2 is lane-local), B operand = the frame's samples straight from the
// reflect-padded signal (16-byte aligned because 160, 56 and 8G+4h are multiples of 4).  The power
// values feed the mel GEMM from the accumulator registers; log and the [N,80] store finish the
// tile.  One workgroup = 32 frames, its 4 waves take one pass of 4 DFT row blocks (64 bins) each.
#pragma once
#include "savad_kernels.h"
#include <type_traits>

namespace savad {
namespace mel {

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

constexpr int N_FFT = 512, HOP = 160, WIN = 400, N_MELS = 80, LPAD = (N_FFT - WIN) / 2;  // LPAD = 56
constexpr int KG = WIN / 8;                    // 50 k-groups of 8 samples
constexpr int DFT_FRAG_FLOATS = 16 * KG * 256;  // [row block 16][G 50][lane 64][4]
constexpr int MEL_FRAG_FLOATS = 4 * 3 * 4 * 2 * 256;  // [pass 4][mel block 3][row block 4][g pair 2][lane 64][4]

// y_pad[j] = y[reflect(j - 256)], j in [0, n + 512)   (numpy.pad(mode="reflect"))
__global__ void reflect_pad_kernel(const float* __restrict__ y, int n, float* __restrict__ ypad) {
    const long total = (long)n + N_FFT;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (long)gridDim.x * blockDim.x) {
        long i = j - N_FFT / 2;
        if (i < 0) i = -i;
        if (i >= n) i = 2L * (n - 1) - i;
        if (i < 0) i = 0;  // signals shorter than the padding
        ypad[j] = y[i];
    }
}

// dst[i] = y[reflect(j0 + i - 256)], i in [0, count), for up to two stretches of the padded signal (the edge frames of the
// factored kernel); y0[i] = sample i of the whole signal, n = its length
struct PadSeg {
    float* dst;
    long j0;
    int count;
};
__global__ void reflect_pad_segments_kernel(const float* __restrict__ y0, long n, PadSeg a, PadSeg b) {
    const long total = (long)a.count + b.count;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const bool second = t >= a.count;
        const long k = second ? t - a.count : t;
        long i = (second ? b.j0 : a.j0) + k - N_FFT / 2;
        if (i < 0) i = -i;
        if (i >= n) i = 2L * (n - 1) - i;
        if (i < 0) i = 0;
        if (i >= n) i = n - 1;
        (second ? b.dst : a.dst)[k] = y0[i];
    }
}

// One WORKGROUP = 32 frames; wave w runs pass w (64 of the 256 bins: 800 DFT MFMAs + its 96 mel MFMAs) and the four
// partial mel accumulators are summed through LDS in a fixed order.  (First version: one wave ran all four passes of
// its tile -- 3584 dependent-issue MFMAs = 100 us of latency for a 10 s clip, whose 32 tiles occupied 8 CUs.)
// (NWV = 8, two waves per pass with two row blocks each, is no faster for short inputs -- the same 896 MFMAs land on each
// SIMD of the tile's CU -- and measured slower: 37.6 against 31.1 us for a 10 s clip.)
template <int NWV = 4>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void logmel_kernel(const float* __restrict__ ypad, int n_frames,
                                                                          const float* __restrict__ dft_frag,
                                                                          const float* __restrict__ mel_frag, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float part[NWV * 3 * 16 * 64];  // [wave][mel block][register][lane]
    constexpr int RB = 16 / NWV;  // DFT row blocks per wave
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pass = wv / (NWV / 4), rb0 = (wv % (NWV / 4)) * RB;
    const int tile = blockIdx.x;
    int f = tile * 32 + m;
    const bool valid = f < n_frames;
    if (!valid) f = n_frames - 1;
    const float* xp = ypad + (size_t)HOP * f + LPAD + 4 * h;
    f32x16 macc[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) macc[mb] = zero16();
    {
        f32x16 acc[RB];
#pragma unroll
        for (int rbl = 0; rbl < RB; ++rbl) acc[rbl] = zero16();
        // DFT rows and samples of k-group G are requested DEPTH groups ahead, by hand: hipcc sinks such loads down
        // to their first use (compiler finding (vi)) and each of the 50 groups then pays an L2 round trip.
        const float* apu = dft_frag + (size_t)(pass * 4 + rb0) * KG * 256;  // wave-uniform
        const int voff = lane * 16;
        // request depth in k-groups (6 instead of 2 for the two-waves-per-pass variant measured slower: 38 vs 32 us)
        constexpr int DEPTH = 2, NSLOT = DEPTH + 1;
        f32x4 ab[NSLOT][RB], xb[NSLOT];
        auto issue = [&](int G, int slot) {
#pragma unroll
            for (int rbl = 0; rbl < RB; ++rbl)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ab[slot][rbl]) : "v"(voff), "s"(apu + (size_t)(rbl * KG + G) * 256));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xb[slot]) : "v"(xp + 8 * G));
        };
        // mel filter fragments of this wave's row blocks: ALL requested at once (hipcc would again pair every load with its 4
        // MFMAs: 12-24 dependent round trips to the Infinity Cache), and for the two-waves-per-pass variant BEFORE the DFT
        // loop -- older than its loads, so the loop's counted waits are unaffected
        const float* mpu = mel_frag + (size_t)(pass * 3) * 4 * 2 * 256;  // wave-uniform
        f32x4 mf[RB][3][2];
        auto issue_mel = [&]() {
#pragma unroll
            for (int rbl = 0; rbl < RB; ++rbl)
#pragma unroll
                for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp)
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(mf[rbl][mb][gp]) : "v"(voff), "s"(mpu + (size_t)((mb * 4 + rb0 + rbl) * 2 + gp) * 256));
        };
        if (RB != 4) issue_mel();
#pragma unroll
        for (int G = 0; G < DEPTH; ++G) issue(G, G);
        static_for<0, KG>([&](auto Gc) {
            constexpr int G = decltype(Gc)::value, slot = G % NSLOT;
            if (G + DEPTH < KG) issue(G + DEPTH, (G + DEPTH) % NSLOT);
            // loads retire in order: at most the (RB + 1) loads of each group requested after G may still be in flight
            constexpr int newer = (KG - 1 - G < DEPTH ? KG - 1 - G : DEPTH) * (RB + 1);
            if (RB == 4)
                asm volatile("s_waitcnt vmcnt(%5)" : "+v"(ab[slot][0]), "+v"(ab[slot][1]), "+v"(ab[slot][RB - 2]), "+v"(ab[slot][RB - 1]), "+v"(xb[slot]) : "n"(newer));
            else
                asm volatile("s_waitcnt vmcnt(%3)" : "+v"(ab[slot][0]), "+v"(ab[slot][1]), "+v"(xb[slot]) : "n"(newer));
#pragma unroll
            for (int rbl = 0; rbl < RB; ++rbl)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[rbl] = SAVAD_MFMA(ab[slot][rbl][e], xb[slot][e], acc[rbl]);
        });
        if (RB == 4) issue_mel();  // (a wave per pass has no registers to spare across the DFT loop)
        // power spectrum, lane-local: rows 2b (re) and 2b+1 (im) are registers 2i and 2i+1
        float pw[RB][8];
#pragma unroll
        for (int rbl = 0; rbl < RB; ++rbl)
#pragma unroll
            for (int i = 0; i < 8; ++i) pw[rbl][i] = acc[rbl][2 * i] * acc[rbl][2 * i] + acc[rbl][2 * i + 1] * acc[rbl][2 * i + 1];
#pragma unroll
        for (int rbl = 0; rbl < RB; ++rbl)
#pragma unroll
            for (int mb = 0; mb < 3; ++mb) {
                if (rbl == 0 && mb == 0) {
                    if (RB == 4)
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(mf[0][0][0]), "+v"(mf[0][0][1]), "+v"(mf[0][1][0]), "+v"(mf[0][1][1]), "+v"(mf[0][2][0]), "+v"(mf[0][2][1]),
                                     "+v"(mf[1][0][0]), "+v"(mf[1][0][1]), "+v"(mf[1][1][0]), "+v"(mf[1][1][1]), "+v"(mf[1][2][0]), "+v"(mf[1][2][1]),
                                     "+v"(mf[RB - 2][0][0]), "+v"(mf[RB - 2][0][1]), "+v"(mf[RB - 2][1][0]), "+v"(mf[RB - 2][1][1]), "+v"(mf[RB - 2][2][0]), "+v"(mf[RB - 2][2][1]),
                                     "+v"(mf[RB - 1][0][0]), "+v"(mf[RB - 1][0][1]), "+v"(mf[RB - 1][1][0]), "+v"(mf[RB - 1][1][1]), "+v"(mf[RB - 1][2][0]), "+v"(mf[RB - 1][2][1]));
                    else
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(mf[0][0][0]), "+v"(mf[0][0][1]), "+v"(mf[0][1][0]), "+v"(mf[0][1][1]), "+v"(mf[0][2][0]), "+v"(mf[0][2][1]),
                                     "+v"(mf[1][0][0]), "+v"(mf[1][0][1]), "+v"(mf[1][1][0]), "+v"(mf[1][1][1]), "+v"(mf[1][2][0]), "+v"(mf[1][2][1]));
                }
#pragma unroll
                for (int gp = 0; gp < 2; ++gp)
#pragma unroll
                    for (int e = 0; e < 4; ++e) macc[mb] = SAVAD_MFMA(mf[rbl][mb][gp][e], pw[rbl][4 * gp + e], macc[mb]);
            }
    }
#pragma unroll
    for (int mb = 0; mb < 3; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            st4(part + (((wv * 3 + mb) * 4 + g) * 64 + lane) * 4, f32x4{macc[mb][4 * g], macc[mb][4 * g + 1], macc[mb][4 * g + 2], macc[mb][4 * g + 3]});
    __syncthreads();
    const int mb = wv;  // wave mb finishes mel block mb
    if (mb >= 3 || !valid) return;
    float* op = out + (size_t)f * N_MELS;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int mel0 = 32 * mb + 8 * g + 4 * h;
        if (mel0 < N_MELS) {
            f32x4 t = ld4(part + (((0 * 3 + mb) * 4 + g) * 64 + lane) * 4);
#pragma unroll
            for (int w2 = 1; w2 < NWV; ++w2) t += ld4(part + (((w2 * 3 + mb) * 4 + g) * 64 + lane) * 4);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) t[s2] = logf(t[s2] + 1e-6f);
            st4(op + mel0, t);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Round 5: the STFT as a FACTORED DFT (512 = 32 x 16) on the same exact-fp32 MFMA -- 624 MFMAs per 32 frames
// instead of 3584 (the DFT-as-GEMM above is kept for A/B and as an on-device cross-check).
//
//   n = 16 n1 + n2,  k = k1 + 32 k2:   X[k] = sum_n2 W512^(n2 k) * Y[k1][n2],   Y[k1][n2] = sum_n1 w[n] y[n] W32^(n1 k1)
//
//   step 1  (52 MFMAs / wave)  Y^T[k1 re|im][frame] for one n2: A = window-folded W32 rows (only n1 = 3..28 lie under
//           the 400-sample window: 13 k-steps), B = the frames' samples, a float4 per lane and k-step = four n2 at once.
//           The input is real, so k1 = 0..16 suffices; re(k1) sits in the low lane half, im(k1) in the high one
//           (im(0) = 0: its slot carries re(16)), which is exactly the k-pair a step-3 MFMA consumes.
//   exchange through LDS: [k1 >> 2][frame][re|im][n2][k1 & 3]; wave w writes n2 = 4w..4w+3 (a float4 = four consecutive
//           accumulator registers, no moves) and reads k1 = 4w..4w+3 (a float4 = one k-step's B operands of its four groups).
//   step 3  (64 MFMAs / wave)  per k1: X[k1 + 32 k2], k2 = 0..15, A = W512^(n2 (k1 + 32 k2)) (twiddle folded in: 16
//           different 32x32 matrices, no VALU work), B = Y from LDS.  Bins above 256 are the mirror images of the bins
//           32 - k1 + 32 k2' that no wave computes (|X[512-k]| = |X[k]|).  k1 = 0 and 16 (both real) share one pass:
//           bins 32..224 step 32 and 16..240 step 32; bins 0 and 256 have zero weight in every filter.
//   power   re / im of a bin are adjacent registers: lane-local.
//   mel     (40 MFMAs / wave)  the rows of step 3 are ordered so that register pair p holds the 2p-th and (2p+1)-th
//           lowest bins of the group: pair 0 only meets mel block 0 (mels 0..31), pairs 2, 3 block 1, pairs 5..7
//           block 2, pairs 1 and 4 two blocks -- 10 MFMAs per group instead of 24 (checked when the tables are built).
//   the four waves' partial mel tiles are summed through LDS in a fixed order; log; store (one tile late, so that the
//   loop's vmcnt(0) for the NEXT tile's samples never waits for a store acknowledgement).
//
// One persistent workgroup per CU (all A operands of a wave -- 164 registers -- are loaded once), samples are read
// straight from the caller's audio (reflect padding is materialised for the edge frames only).
constexpr int FFT_T1_FLOATS = 4 * 13 * 256;   // [wave 4][k-step 13][lane 64][n2 & 3]
constexpr int FFT_T3_FLOATS = 16 * 4 * 256;   // [group 16][n2 >> 2][lane 64][n2 & 3]
constexpr int FFT_TM_FLOATS = 16 * 3 * 256;   // [group 16][t >> 2][lane 64][t & 3], t = 0..9 (10, 11 unused)
constexpr int FFT_YLD = 132;                  // LDS floats per (reader wave, frame): [re|im][n2 16][k1 & 3] + 4 (conflict-free b128)
constexpr int FFT_Y_FLOATS = 4 * 32 * FFT_YLD;
constexpr int FFT_PART_FLOATS = 4 * 3 * 4 * 256;
// Sample stage: the tile's stretch of the padded signal, padded indices [160 f0 + 48, 160 f0 + 48 + 5376), as 34 blocks of
// 160 samples, each followed by 4 unused floats: frame m's float4 for k-step s then sits at 164 m + 16 h + 4 w +
// (164 (s / 5) + 32 (s % 5)) -- lane stride 164 = 36 mod 64 banks (conflict-free ds_read_b128), the rest an immediate.
// Filled by global -> LDS DMA: 24 instructions of 64 x 16 bytes, 6 per wave (chunk q = 64 i + lane of the stage is
// sample chunk 40 (q / 41) + min(q % 41, 39) of the stretch); every 128-byte line of the audio is requested once per tile.
constexpr int FFT_SBLK = 164, FFT_STAGE_CHUNKS = 24 * 64, FFT_STAGE_FLOATS = FFT_STAGE_CHUNKS * 4;
constexpr int FFT_LDS_FLOATS = FFT_Y_FLOATS + FFT_PART_FLOATS + FFT_STAGE_FLOATS;
constexpr int FFT_LDS_BYTES = FFT_LDS_FLOATS * 4;  // 147 456

// Where the padded signal ypad[j] = y[reflect(j - 256)] is read from: padded indices below jA_end from a padded copy
// (padA[j - jA0]: the stretch that mirrors the signal's head, or the whole span when the audio pointer is unaligned), from
// jB0 on from padB[j - jB0] (the stretch that runs past the last sample), everything between straight from the audio
// (y0[j - 256]).  jA_end, jB0, jA0 are multiples of 4 and every pointer is 16-byte aligned.
struct FftSrc {
    const float* y0;
    const float* padA;
    const float* padB;
    long jA0, jA_end, jB0;
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(256, 1) void logmel_fft_kernel(FftSrc src, int frame_first, int frame_count, int n_tiles,
                                                            const float* __restrict__ t1, const float* __restrict__ t3,
                                                            const float* __restrict__ tm, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float fft_lds[];
    float* Yl = fft_lds;                                     // [k1 >> 2][frame 32][FFT_YLD]
    float* part = fft_lds + FFT_Y_FLOATS;                    // [wave 4][mel block 3][g 4][lane 64][4]
    float* stage = fft_lds + FFT_Y_FLOATS + FFT_PART_FLOATS;  // see FFT_SBLK
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this wave's A operands, for the whole launch
    f32x4 a1[13], a3[4][4], am[4][3];
#pragma unroll
    for (int s = 0; s < 13; ++s) a1[s] = ld4(t1 + ((size_t)(wv * 13 + s) * 64 + lane) * 4);
#pragma unroll
    for (int gl = 0; gl < 4; ++gl) {
#pragma unroll
        for (int c = 0; c < 4; ++c) a3[gl][c] = ld4(t3 + ((size_t)((wv * 4 + gl) * 4 + c) * 64 + lane) * 4);
#pragma unroll
        for (int q = 0; q < 3; ++q) am[gl][q] = ld4(tm + ((size_t)((wv * 4 + gl) * 3 + q) * 64 + lane) * 4);
    }
    // DMA instruction i = wv + 4 k of this wave fills stage chunks 64 i + lane: offsets (floats) inside the tile's stretch
    int rel[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int q = 64 * (wv + 4 * k) + lane, b = q / 41, c = q % 41;
        rel[k] = 160 * b + 4 * (c < 39 ? c : 39);
    }
    const long j_last = (long)HOP * (frame_first + frame_count - 1) + 460;  // the last chunk any frame of the launch reads
    const unsigned stage_m0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)stage) + 1024u * (unsigned)wv;
    // sources as integers: address of padded index j = base + 4 j
    const uintptr_t srcA = (uintptr_t)src.padA - 4 * (uintptr_t)src.jA0, srcB = (uintptr_t)src.padB - 4 * (uintptr_t)src.jB0,
                    srcY = (uintptr_t)src.y0 - 4 * (uintptr_t)(N_FFT / 2);
    // the six source addresses of this wave's part of a tile's stage
    const float* sptr[6];
    auto stage_ptrs = [&](int tile) {
        const long J0 = (long)HOP * (frame_first + tile * 32) + 48;
        const long J1 = J0 + 160 * 37 + 160;  // one past the last chunk the 24 instructions touch
        if (J0 >= src.jA_end && J1 <= src.jB0 && J1 <= j_last + 4) {  // wave-uniform: every chunk straight from the audio
#pragma unroll
            for (int k = 0; k < 6; ++k) sptr[k] = (const float*)(srcY + 4 * (uintptr_t)(J0 + rel[k]));
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                long j = J0 + rel[k];
                if (j > j_last) j = j_last;
                const uintptr_t b = (j < src.jA_end) ? srcA : (j >= src.jB0) ? srcB : srcY;
                sptr[k] = (const float*)(b + 4 * (uintptr_t)j);
            }
        }
    };
    // one DMA instruction (64 x 16 bytes -> stage + 1 KiB * (wv + 4 k)); separate statements, so that they can sit between
    // MFMAs (issued back to back they hold the wave ~100 cycles each); the M0 offset is an immediate (as SGPR operands the
    // six values ran the loop out of SGPRs, and hipcc then hands an "s" operand a VGPR)
    unsigned keep_m0;
#define SAVAD_FFT_DMA(k)                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %2, " #k "*4096\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_m0) : "v"(sptr[k]), "s"(stage_m0) : "memory", "scc")
    f32x16 acc[4];
    // lanes past the launch's last frame redo that frame: they store the same values to the same place, unpredicated
    auto step1 = [&](int tile) {
        const int room = frame_count - 1 - 32 * tile, mm = m < room ? m : (room > 0 ? room : 0);
        const float* sp = stage + FFT_SBLK * mm + 16 * h + 4 * wv;
        f32x4 x[13];
#pragma unroll
        for (int s = 0; s < 13; ++s) x[s] = ld4(sp + FFT_SBLK * (s / 5) + 32 * (s % 5));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = zero16();
#pragma unroll
        for (int s = 0; s < 13; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = SAVAD_MFMA(a1[s][e], x[s][e], acc[e]);
    };
#ifdef SAVAD_TIMING
    long long facc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fp = __builtin_readcyclecounter(), fn;
#define SAVAD_FACC(i) do { fn = __builtin_readcyclecounter(); facc[i] += fn - fp; fp = fn; } while (0)
#else
#define SAVAD_FACC(i) do {} while (0)
#endif
    int tile = blockIdx.x;
    if (tile < n_tiles) {
        stage_ptrs(tile);
        SAVAD_FFT_DMA(0); SAVAD_FFT_DMA(1); SAVAD_FFT_DMA(2); SAVAD_FFT_DMA(3); SAVAD_FFT_DMA(4); SAVAD_FFT_DMA(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        step1(tile);
    }
    SAVAD_FACC(0);
    for (; tile < n_tiles; tile += gridDim.x) {
        // ---- exchange: acc[e][r] = Y[k1 = r][re|im = h][n2 = 4 wv + e][frame m]
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                st4(Yl + (g * 32 + m) * FFT_YLD + 64 * h + 4 * (4 * wv + e), f32x4{acc[e][4 * g], acc[e][4 * g + 1], acc[e][4 * g + 2], acc[e][4 * g + 3]});
        SAVAD_FACC(1);
        lds_barrier();  // Y is complete; every wave has finished reading the stage and last tile's partial sums
        SAVAD_FACC(2);
        const int next = tile + gridDim.x;
        const bool have_next = next < n_tiles;  // wave-uniform
        if (have_next) stage_ptrs(next);
        // ---- step 3, power, mel: group gl + 1's DFT runs in front of group gl's power and mel work
        f32x16 macc[3];
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) macc[mb] = zero16();
        f32x4 yb[16];  // yb[n2][gl]
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) yb[n2] = ld4(Yl + (wv * 32 + m) * FFT_YLD + 64 * h + 4 * n2);
        auto dft3 = [&](int gl, auto&& between) {
            f32x16 d = zero16();
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int e = 0; e < 4; ++e) d = SAVAD_MFMA(a3[gl][c][e], yb[4 * c + e][gl], d);
                between(c);
            }
            return d;
        };
        // the next tile's stage is requested between the MFMAs of the first two groups
        f32x16 d = dft3(0, [&](int c) {
            if (have_next) {
                if (c == 0) SAVAD_FFT_DMA(0);
                if (c == 1) SAVAD_FFT_DMA(1);
                if (c == 2) SAVAD_FFT_DMA(2);
                if (c == 3) SAVAD_FFT_DMA(3);
            }
        });
#pragma unroll
        for (int gl = 0; gl < 4; ++gl) {
            f32x16 dn;
            if (gl < 3)
                dn = dft3(gl + 1, [&](int c) {
                    if (have_next && gl == 0) {
                        if (c == 0) SAVAD_FFT_DMA(4);
                        if (c == 1) SAVAD_FFT_DMA(5);
                    }
                });
            float pw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pw[i] = d[2 * i] * d[2 * i] + d[2 * i + 1] * d[2 * i + 1];
            // pair -> mel blocks: 0:{0} 1:{0,1} 2:{1} 3:{1} 4:{1,2} 5:{2} 6:{2} 7:{2}
            macc[0] = SAVAD_MFMA(am[gl][0][0], pw[0], macc[0]);
            macc[0] = SAVAD_MFMA(am[gl][0][1], pw[1], macc[0]);
            macc[1] = SAVAD_MFMA(am[gl][0][2], pw[1], macc[1]);
            macc[1] = SAVAD_MFMA(am[gl][0][3], pw[2], macc[1]);
            macc[1] = SAVAD_MFMA(am[gl][1][0], pw[3], macc[1]);
            macc[1] = SAVAD_MFMA(am[gl][1][1], pw[4], macc[1]);
            macc[2] = SAVAD_MFMA(am[gl][1][2], pw[4], macc[2]);
            macc[2] = SAVAD_MFMA(am[gl][1][3], pw[5], macc[2]);
            macc[2] = SAVAD_MFMA(am[gl][2][0], pw[6], macc[2]);
            macc[2] = SAVAD_MFMA(am[gl][2][1], pw[7], macc[2]);
            if (gl < 3) d = dn;
        }
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                st4(part + (((wv * 3 + mb) * 4 + g) * 64 + lane) * 4, f32x4{macc[mb][4 * g], macc[mb][4 * g + 1], macc[mb][4 * g + 2], macc[mb][4 * g + 3]});
        SAVAD_FACC(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the next stage has landed (and the last tile's stores are acknowledged)
        SAVAD_FACC(4);
        lds_barrier();
        SAVAD_FACC(5);
        // ---- the next tile's step 1 (MFMA) in one basic block with this tile's sums, log and stores (LDS, VALU): no branch in
        // here (after the last tile step 1 reruns on the stale stage; nobody reads its result)
        {
            const int room = frame_count - 1 - 32 * tile, fl = tile * 32 + (m < room ? m : room);  // frame inside the launch
            float* op = out + (size_t)fl * N_MELS + 4 * h;
            step1(next);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                // piece = 4 (mel block) + g: mels 8 piece + 4 h .. + 3; 10 pieces, wave w takes w, w + 4 and w + 8 (waves 2, 3: piece 9 again)
                const int piece = wv + 4 * i < 10 ? wv + 4 * i : 9;
                f32x4 t = ld4(part + ((0 * 12 + piece) * 64 + lane) * 4);
#pragma unroll
                for (int w2 = 1; w2 < 4; ++w2) t += ld4(part + ((w2 * 12 + piece) * 64 + lane) * 4);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) t[s2] = logf(t[s2] + 1e-6f);
                st4(op + 8 * piece, t);
            }
        }
        SAVAD_FACC(7);
    }
#ifdef SAVAD_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_savad_dbg[24 + i] = facc[i];
#endif
#undef SAVAD_FFT_DMA
}

}  // namespace mel
}  // namespace savad
