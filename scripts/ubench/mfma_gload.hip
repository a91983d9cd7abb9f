// Microbenchmark: bf16 MFMA whose A operand (weight fragment, 1 KiB per wave) is loaded STRAIGHT from global
// memory (L1/L2 hits: every wave of the chip reads the same 384 KiB), P fragments ahead, no LDS, no barriers.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
constexpr int NF = 384;  // fragments per pass (one layer's weights)

template <int P>
__global__ __launch_bounds__(256, 2) void k(const char* __restrict__ w, float* out, int iters, float b0) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    bf16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (__bf16)b0;
    const char* wl = w + lane * 16;
    u32x4 buf[P];
#pragma unroll
    for (int j = 0; j < P; ++j) buf[j] = *reinterpret_cast<const u32x4*>(wl + j * 1024);
    for (int it = 0; it < iters; ++it) {
        for (int f = 0; f < NF; f += P) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                acc[j & 3] = MFMA(__builtin_bit_cast(bf16x8, buf[j]), b, acc[j & 3]);
                int nf = f + P + j;
                nf = nf >= NF ? nf - NF : nf;
                buf[j] = *reinterpret_cast<const u32x4*>(wl + nf * 1024);
            }
        }
    }
    float s = 0;
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int P>
void run(int blocks, const char* w) {
    float* out;
    (void)hipMalloc(&out, sizeof(float) * 256 * blocks);
    int iters = 20;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<P><<<blocks, 256>>>(w, out, 2, 0.5f);
    (void)hipEventRecord(e0);
    k<P><<<blocks, 256>>>(w, out, iters, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * NF;
    printf("global-fed bf16 MFMA, prefetch %2d fragments, %d waves/SIMD: %.1f ns per MFMA per wave -> %.0f TFLOP/s\n", P, blocks / 256,
           ms * 1e6 / n, n * 32768.0 * 4 * blocks / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
int main() {
    char* w;
    (void)hipMalloc(&w, NF * 1024);
    (void)hipMemset(w, 0x3c, NF * 1024);
    run<4>(256, w);
    run<8>(256, w);
    run<16>(256, w);
    run<4>(512, w);
    run<8>(512, w);
    run<16>(512, w);
    run<24>(512, w);
    return 0;
}
