// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 for 1 / 2 / 4 independent accumulator chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int CH>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float a0, float b0) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / CH; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = MFMA(a, b, acc[c]);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < CH; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CH>
void run(const char* name, int blocks) {
    float* out;
    long long* cyc;
    hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipMalloc(&cyc, 8);
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<CH><<<blocks, 256>>>(out, cyc, 10, 1.f, 2.f);
    hipEventRecord(e0);
    k<CH><<<blocks, 256>>>(out, cyc, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 16;
    double tf = n * 4096.0 * 4 * blocks / (ms * 1e-3) / 1e12;
    printf("%s chains=%d blocks=%d: %.2f shader-clock ticks per MFMA (s_memtime-like counter), %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA\n", name, CH,
           blocks, (double)c / n, ms, tf, ms * 1e6 / n);
}
int main() {
    run<1>("1wave/SIMD", 256);
    run<2>("1wave/SIMD", 256);
    run<4>("1wave/SIMD", 256);
    run<1>("2wave/SIMD", 512);
    run<4>("2wave/SIMD", 512);
    run<1>("single-CU", 1);
    return 0;
}
