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

}  // namespace mel
}  // namespace savad
