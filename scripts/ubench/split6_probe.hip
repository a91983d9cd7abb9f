// split6_probe.hip -- does a GEMM on v_mfma_f32_32x32x16_bf16 with every fp32 operand split into three bf16 pieces
// (hi + mid + lo, six cross products, one fp32 accumulator) reproduce fp32 on THIS hardware?  The matrix core's internal
// summation order / rounding is not documented; this measures it: C = A B^T for 32 x K operands, K = 128 and 800,
// three value distributions, against fp64 on the host, beside the exact-fp32 MFMA and the three-product variant.
// Second half: issue rate of the six-MFMA group fed by three ds_read_b128 (the shape of every fp32s inner loop).
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/split6_probe.bin scripts/ubench/split6_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)a;
    const float r1 = a - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// mode 0: fp32 MFMA; 1: bf16 x 6 (small terms first); 2: bf16 x 3 (hh, hm, mh); 3: bf16 x 6, big term first; 4: bf16 x 1
__global__ void gemm_probe(const float* __restrict__ A, const float* __restrict__ B, int K, int mode, float* __restrict__ C) {
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[n * K + k + h], B[n * K + k + h], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int e = 0; e < 8; ++e) {
                __bf16 x, y, z;
                split3(A[n * K + k0 + 8 * h + e], x, y, z);
                ah[e] = x, am[e] = y, al[e] = z;
                split3(B[n * K + k0 + 8 * h + e], x, y, z);
                bh[e] = x, bm[e] = y, bl[e] = z;
            }
#define MF(a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), acc, 0, 0, 0)
            if (mode == 1) {
                MF(ah, bl); MF(al, bh); MF(am, bm); MF(ah, bm); MF(am, bh); MF(ah, bh);
            } else if (mode == 2) {
                MF(ah, bm); MF(am, bh); MF(ah, bh);
            } else if (mode == 3) {
                MF(ah, bh); MF(ah, bm); MF(am, bh); MF(am, bm); MF(ah, bl); MF(al, bh);
            } else {
                MF(ah, bh);
            }
        }
    }
    // C[row = (r&3) + 8 (r>>2) + 4h][col = n]  (A rows x B rows)
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + n] = acc[r];
}

// ---- issue rate: 4 waves per workgroup, each looping over `iters` groups of {3 ds_read_b128, 6 MFMA} on 4 accumulators
template <int SPLITS>
__global__ __launch_bounds__(256) void rate_probe(int iters, float* out) {
    __shared__ __attribute__((aligned(16))) char lds[48 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 48 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    bf16x8 xh, xm, xl;
    for (int e = 0; e < 8; ++e) xh[e] = (__bf16)(1.0f + lane), xm[e] = (__bf16)0.001f, xl[e] = (__bf16)0.00001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const char* p = lds + ((s * 3) % 45) * 1024 + lane * 16;
            const bf16x8 wh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
            const bf16x8 wm = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + 1024));
            const bf16x8 wl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + 2048));
            f32x16& a = acc[s & 3];
            if (SPLITS == 6) {
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xm, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xm, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xh, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, a, 0, 0, 0);
            } else {
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, a, 0, 0, 0);
            }
        }
    }
    float s = 0.0f;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 12345.678f) out[0] = s;
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
    const char* names[3] = {"normal(0,1)", "uniform(-13.8,4.2)", "heavy (normal^3 * 30)"};
    const char* modes[5] = {"fp32 mfma", "bf16x6 small-first", "bf16x3", "bf16x6 big-first", "bf16x1"};
    float *dA, *dB, *dC;
    const int KMAX = 800;
    hipMalloc(&dA, 32 * KMAX * 4);
    hipMalloc(&dB, 32 * KMAX * 4);
    hipMalloc(&dC, 32 * 32 * 4);
    srand(7);
    for (int K : {128, 512, 800}) {
        for (int dist = 0; dist < 3; ++dist) {
            std::vector<float> A(32 * K), B(32 * K), C(1024);
            for (auto* v : {&A, &B})
                for (auto& x : *v) {
                    if (dist == 0) x = (float)nrand();
                    else if (dist == 1) x = (float)(-13.8 + 18.0 * urand());
                    else { const double t = nrand(); x = (float)(30.0 * t * t * t); }
                }
            std::vector<double> R(1024);
            double scale = 0.0;
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double s = 0.0, sa = 0.0;
                    for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[j * K + k], sa += fabs((double)A[i * K + k] * B[j * K + k]);
                    R[i * 32 + j] = s;
                    scale = fmax(scale, sa);
                }
            hipMemcpy(dA, A.data(), 32 * K * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), 32 * K * 4, hipMemcpyHostToDevice);
            printf("K=%d %-24s (sum|ab| max %.3g):", K, names[dist], scale);
            for (int mode = 0; mode < 5; ++mode) {
                hipLaunchKernelGGL(gemm_probe, dim3(1), dim3(64), 0, 0, dA, dB, K, mode, dC);
                hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
                double e = 0.0;
                for (int i = 0; i < 1024; ++i) e = fmax(e, fabs((double)C[i] - R[i]));
                printf("  %s %.3g", modes[mode], e / scale);
            }
            printf("   (errors relative to max sum|ab|; 2^-24 = 5.96e-8)\n");
        }
    }
    // issue rate
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int splits : {6, 1}) {
        for (int wgs : {256, 512}) {
            const int iters = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (splits == 6) hipLaunchKernelGGL(rate_probe<6>, dim3(wgs), dim3(256), 0, 0, iters, dC);
                else hipLaunchKernelGGL(rate_probe<1>, dim3(wgs), dim3(256), 0, 0, iters, dC);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double mfma = (double)wgs * 4 * iters * 16 * splits;
                if (rep) printf("rate: %d MFMA per 3 ds_read_b128 group, %d workgroups: %.3f ms, %.1f TFLOP/s issued (%.2f of 2500), %.1f cycles/MFMA/SIMD at 2.4 GHz\n",
                                splits, wgs, ms, mfma * 32768 / ms / 1e9, mfma * 32768 / ms / 1e9 / 2500.0,
                                ms * 1e-3 * 2.4e9 / (mfma / (wgs > 256 ? 1024 : 1024)) );
            }
        }
    }
    return 0;
}
