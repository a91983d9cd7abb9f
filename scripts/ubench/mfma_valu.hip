// Microbenchmark: does VALU work interleaved into ONE dependent v_mfma_f32_32x32x2_f32 chain issue for free?
// Per MFMA: NV independent VALU ops (KIND 0: v_fma_f32, 1: v_exp_f32) between two sched_barriers.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NV, int KIND, int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a0 * i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u % CH] = MFMA(a, b, acc[u % CH]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (KIND == 0)
                    x[v & 7] = __builtin_fmaf(x[v & 7], b0, a0);
                else
                    x[v & 7] = __builtin_amdgcn_exp2f(x[v & 7]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int c = 0; c < CH; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int KIND, int CH = 1>
void run(int blocks) {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * blocks);
    int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NV, KIND, CH><<<blocks, 256>>>(out, 10, 1.f, 0.5f);
    hipEventRecord(e0);
    k<NV, KIND, CH><<<blocks, 256>>>(out, iters, 1.f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 16;
    printf("chains=%d %s x%d per MFMA, %d waves/SIMD: %.2f ns per MFMA (%.1f TFLOP/s)\n", CH, KIND ? "v_exp_f32" : "v_fma_f32", NV, blocks / 256,
           ms * 1e6 / n, n * 4096.0 * 4 * blocks / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<0, 0, 1>(256);
    run<1, 0, 1>(256);
    run<4, 0, 1>(256);
    run<16, 0, 1>(256);
    run<0, 0, 2>(256);
    run<1, 0, 2>(256);
    run<4, 0, 2>(256);
    run<8, 0, 2>(256);
    run<16, 0, 2>(256);
    run<4, 1, 2>(256);
    run<1, 0, 4>(256);
    run<4, 0, 4>(256);
    run<16, 0, 4>(256);
    run<4, 1, 4>(256);
    return 0;
}
