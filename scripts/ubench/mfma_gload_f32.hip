// Microbenchmark for a global-fed fp32 attention: each wave walks the K / V^T fragments (1 KiB = 64 lanes x
// 16 B, feeding 4 x v_mfma_f32_32x32x2_f32) of ITS sequence straight from L2 with a rolling prefetch of P
// fragments; no LDS, no barriers.  32 sequences x 800 KiB, sequence -> XCD by blockIdx & 7 as in the real kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
constexpr int NFRAG = 25 * 32;  // fragments per sequence: 25 key tiles x (16 K + 16 V^T)

template <int P>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ kv, float* out, int nseq, int tiles, float b0) {
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int seq = ((i * 4 + (threadIdx.x >> 6)) % (nseq / 8)) * 8 + xcd;
    const float* base = kv + (size_t)seq * NFRAG * 256 + lane * 4;
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f32x4 buf[P];
#pragma unroll
    for (int j = 0; j < P; ++j) buf[j] = *reinterpret_cast<const f32x4*>(base + j * 256);
    const int nf = tiles * 32;
    for (int f = 0; f < nf; f += P) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j & 3] = MFMA(buf[j][e], b0 + e, acc[j & 3]);
            int nx = f + P + j;
            nx = nx >= nf ? nx - nf : nx;
            buf[j] = *reinterpret_cast<const f32x4*>(base + (size_t)nx * 256);
        }
    }
    float s = 0;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int P>
void run(int blocks, const float* kv) {
    float* out;
    (void)hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int tiles = 25;
    k<P><<<blocks, 256>>>(kv, out, 32, tiles, 0.5f);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k<P><<<blocks, 256>>>(kv, out, 32, tiles, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("global-fed fp32 MFMA, prefetch %2d fragments, %d workgroups (%.2f waves/SIMD): %.1f us per launch, %.3f us per key tile (128 MFMAs; "
           "3.64 us = MFMA-bound at 2.25 GHz)\n", P, blocks, blocks * 4 / 1024.0, ms * 1e3, ms * 1e3 / tiles);
    (void)hipFree(out);
}
int main() {
    float* kv;
    (void)hipMalloc(&kv, (size_t)32 * NFRAG * 1024);
    (void)hipMemset(kv, 0, (size_t)32 * NFRAG * 1024);
    run<4>(256, kv);
    run<8>(256, kv);
    run<16>(256, kv);
    run<8>(200, kv);
    run<8>(512, kv);
    run<16>(512, kv);
    return 0;
}
