// Microbenchmark: the inner loop of gemm_lds_a (16 x ds_read_b128 + 64 MFMA per block), no DMA, no barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, long long* cyc, int blocks, const float* src) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[i];
    __syncthreads();
    f32x4 xg[16];
    for (int G = 0; G < 16; ++G) xg[G] = ld4(src + 4 * G + lane);
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int b = 0; b < blocks; ++b) {
        const float* buf = lds + (b & 1) * 4096;
        const float* row = buf + n * 128;
        if (MODE == 0) {  // as in gemm_lds_a: compiler-scheduled reads
#pragma unroll
            for (int G = 0; G < 16; ++G) {
                const f32x4 w4 = ld4(row + 4 * ((2 * G + h) ^ (n & 15)));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = MFMA(w4[e], xg[G][e], acc);
            }
        } else if (MODE == 1) {  // all 16 reads up front
            f32x4 wv[16];
#pragma unroll
            for (int G = 0; G < 16; ++G) wv[G] = ld4(row + 4 * ((2 * G + h) ^ (n & 15)));
#pragma unroll
            for (int G = 0; G < 16; ++G)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = MFMA(wv[G][e], xg[G][e], acc);
        } else if (MODE == 3) {  // one ds_read_b32 per MFMA (the PV pattern: A = V^T column from a row-major tile)
            const float* vp = buf + 4 * h * 128 + n;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc = MFMA(vp[(8 * (r >> 2) + (r & 3)) * 128 + 32 * nb], xg[r][nb], acc);
        } else if (MODE == 4) {  // one ds_read_b64 per 2 MFMA
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int G = 0; G < 32; ++G) {
                const f32x2 w2 = *reinterpret_cast<const f32x2*>(buf + 2 * (G * 64 + lane));
                acc = MFMA(w2[0], xg[G & 15][0], acc);
                acc = MFMA(w2[1], xg[G & 15][1], acc);
            }
        } else {  // no LDS: A operand from registers
#pragma unroll
            for (int G = 0; G < 16; ++G)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = MFMA(xg[(G + 1) & 15][e], xg[G][e], acc);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(int wgs) {
    float *out, *src;
    long long* cyc;
    hipMalloc(&out, sizeof(float) * 256 * wgs);
    hipMalloc(&src, sizeof(float) * 8192);
    hipMemset(src, 0, sizeof(float) * 8192);
    hipMalloc(&cyc, 8);
    int blocks = 480;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<wgs, 256>>>(out, cyc, 8, src);
    hipEventRecord(e0);
    k<MODE><<<wgs, 256>>>(out, cyc, blocks, src);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)blocks * 64;
    printf("mode %d wgs=%d: %.2f ticks/MFMA (wave 0), kernel %.3f ms -> %.1f ns/MFMA\n", MODE, wgs, (double)c / n, ms, ms * 1e6 / n);
}
int main() {
    run<2>(25); run<0>(25); run<1>(25); run<3>(25); run<4>(25); run<3>(512); run<4>(512);
    run<2>(256); run<0>(256); run<1>(256);
    run<0>(512); run<1>(512);
    return 0;
}
