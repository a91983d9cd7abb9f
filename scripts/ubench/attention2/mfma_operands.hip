// MFMA issue rate of v_mfma_f32_32x32x16_bf16 by operand register file (A, B from VGPR or AGPR; C/D in VGPR or AGPR).
// One wave per SIMD (256 threads per block, 256 blocks), 8 independent accumulators, back to back.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(long long* out, int iters) {
    asm volatile("" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31",
                 "a128","a129","a130","a131","a132","a133","a134","a135","a192","a193","a194","a195","a196","a197","a198","a199", "a255");
    for (int i = 0; i < 256; ++i) asm volatile("v_accvgpr_write_b32 a[%0], 0" :: "n"(0));
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // A vgpr, B vgpr, CD agpr
#define X(i) asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], v[8:11], v[12:15], a[%0:%1]" :: "n"(16 * i), "n"(16 * i + 15));
            REP8(X) REP8(X)
#undef X
        } else if (MODE == 1) {  // A agpr, B vgpr, CD agpr
#define X(i) asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], a[192:195], v[12:15], a[%0:%1]" :: "n"(16 * i), "n"(16 * i + 15));
            REP8(X) REP8(X)
#undef X
        } else if (MODE == 2) {  // A agpr, B agpr, CD vgpr
#define X(i) asm volatile("v_mfma_f32_32x32x16_bf16 v[%0:%1], a[192:195], a[128:131], v[%0:%1]" :: "n"(16 + 16 * i), "n"(16 * i + 31));
            REP8(X) REP8(X)
#undef X
        } else if (MODE == 3) {  // A vgpr, B agpr, CD vgpr
#define X(i) asm volatile("v_mfma_f32_32x32x16_bf16 v[%0:%1], v[8:11], a[128:131], v[%0:%1]" :: "n"(16 + 16 * i), "n"(16 * i + 31));
            REP8(X) REP8(X)
#undef X
        } else if (MODE == 4) {  // A vgpr, B vgpr, CD vgpr
#define X(i) asm volatile("v_mfma_f32_32x32x16_bf16 v[%0:%1], v[8:11], v[12:15], v[%0:%1]" :: "n"(16 + 16 * i), "n"(16 * i + 31));
            REP8(X) REP8(X)
#undef X
        } else if (MODE == 5) {  // A agpr, B agpr, CD agpr
#define X(i) asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], a[192:195], a[196:199], a[%0:%1]" :: "n"(16 * i), "n"(16 * i + 15));
            REP8(X) REP8(X)
#undef X
        }
    }
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" ::: "v8","v9","v10","v11","v12","v13","v14","v15","v16","v31","v47","v63","v79","v95","v111","v127","v143");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[MODE] = t1 - t0;
}
int main() {
    long long* d; hipMalloc(&d, 64);
    const int iters = 2000;
    const char* names[6] = {"A v, B v, CD a", "A a, B v, CD a", "A a, B a, CD v", "A v, B a, CD v", "A v, B v, CD v", "A a, B a, CD a"};
    k<0><<<256, 256>>>(d, iters); k<1><<<256, 256>>>(d, iters); k<2><<<256, 256>>>(d, iters); k<3><<<256, 256>>>(d, iters); k<4><<<256, 256>>>(d, iters); k<5><<<256, 256>>>(d, iters);
    hipDeviceSynchronize();
    long long h[6]; hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
    for (int m = 0; m < 6; ++m) printf("%s: %.2f cycles per MFMA\n", names[m], (double)h[m] / (iters * 16.0));
    return 0;
}
