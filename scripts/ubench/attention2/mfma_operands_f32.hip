// v_mfma_f32_32x32x2_f32 issue rate by operand register file, as a DEPENDENT chain (one accumulator, what the PV block of
// savad_attn_pipe.h does per feature block) and with 4 accumulators in rotation.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(long long* out, int iters) {
    asm volatile("" ::: "a0","a15","a16","a31","a32","a47","a48","a63","a128","a129","a192","a193","a255");
    for (int i = 0; i < 1; ++i) asm volatile("v_accvgpr_write_b32 a128, 0\n v_accvgpr_write_b32 a192, 0");
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP16(asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], v8, v12, a[0:15]");) }
        else if (MODE == 1) { REP16(asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], a192, a128, a[0:15]");) }
        else if (MODE == 2) { REP16(asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], a192, v12, a[0:15]");) }
        else if (MODE == 3) { REP16(asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], v8, a128, a[0:15]");) }
        else if (MODE == 4) {
            for (int q = 0; q < 4; ++q) {
                asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], a192, a128, a[0:15]\n v_mfma_f32_32x32x2_f32 a[16:31], a192, a128, a[16:31]\n"
                             "v_mfma_f32_32x32x2_f32 a[32:47], a192, a128, a[32:47]\n v_mfma_f32_32x32x2_f32 a[48:63], a192, a128, a[48:63]");
            }
        } else if (MODE == 5) { REP16(asm volatile("v_mfma_f32_32x32x2_f32 v[16:31], v8, v12, v[16:31]");) }
    }
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" ::: "v8","v12","v16","v31");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[MODE] = t1 - t0;
}
int main() {
    long long* d; hipMalloc(&d, 64);
    const int iters = 2000;
    const char* names[6] = {"chain: A v, B v, CD a", "chain: A a, B a, CD a", "chain: A a, B v, CD a", "chain: A v, B a, CD a", "4 accumulators: A a, B a, CD a", "chain: A v, B v, CD v"};
    k<0><<<256, 256>>>(d, iters); k<1><<<256, 256>>>(d, iters); k<2><<<256, 256>>>(d, iters); k<3><<<256, 256>>>(d, iters); k<4><<<256, 256>>>(d, iters); k<5><<<256, 256>>>(d, iters);
    hipDeviceSynchronize();
    long long h[6]; hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
    for (int m = 0; m < 6; ++m) printf("%s: %.2f cycles per MFMA\n", names[m], (double)h[m] / (iters * 16.0));
    return 0;
}
