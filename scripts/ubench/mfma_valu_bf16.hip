// Microbenchmark: VALU / LDS-read work interleaved with v_mfma_f32_32x32x16_bf16 (CH independent chains).
// KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_pk_fma_f32, 3: ds_read_b128 feeding the A operand.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

template <int NV, int KIND, int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(a0 + threadIdx.x * 1e-3f);
        b[i] = (__bf16)b0;
    }
    for (int i = threadIdx.x; i < 8192; i += 256) ((float*)lds)[i] = a0 * i;
    __syncthreads();
    float x[8];
    f32x2 xp[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = a0 * i + threadIdx.x;
        xp[i] = f32x2{x[i], x[i] + 1};
    }
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 3) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(lds + ((u * 64 + lane) * 16 + (it & 1) * 16384));
                acc[u % CH] = MFMA(av, b, acc[u % CH]);
            } else {
                acc[u % CH] = MFMA(a, b, acc[u % CH]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (KIND == 0)
                        x[v & 7] = __builtin_fmaf(x[v & 7], b0, a0);
                    else if (KIND == 1)
                        x[v & 7] = __builtin_amdgcn_exp2f(x[v & 7]);
                    else
                        xp[v & 7] = xp[v & 7] * f32x2{b0, b0} + f32x2{a0, a0};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int c = 0; c < CH; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < 8; ++i) s += x[i] + xp[i][0] + xp[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int KIND, int CH>
void run(int blocks) {
    float* out;
    (void)hipMalloc(&out, sizeof(float) * 256 * blocks);
    int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<NV, KIND, CH><<<blocks, 256>>>(out, 10, 1.f, 0.5f);
    (void)hipEventRecord(e0);
    k<NV, KIND, CH><<<blocks, 256>>>(out, iters, 1.f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 16;
    const char* names[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "ds_read_b128-fed"};
    printf("bf16 chains=%d %s x%d per MFMA, %d waves/SIMD: %.2f ns per MFMA (%.0f TFLOP/s)\n", CH, names[KIND], NV, blocks / 256,
           ms * 1e6 / n, n * 32768.0 * 4 * blocks / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
int main() {
    run<0, 0, 1>(256);
    run<0, 0, 4>(256);
    run<1, 0, 4>(256);
    run<2, 0, 4>(256);
    run<4, 0, 4>(256);
    run<8, 0, 4>(256);
    run<2, 1, 4>(256);
    run<4, 2, 4>(256);
    run<0, 3, 4>(256);
    run<0, 0, 4>(512);
    run<4, 0, 4>(512);
    run<8, 0, 4>(512);
    run<2, 1, 4>(512);
    run<0, 3, 4>(512);
    run<0, 3, 1>(512);
    return 0;
}
