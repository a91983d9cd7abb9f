#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned u = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r2 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[128 + threadIdx.x] = r2[0];
    out[192 + threadIdx.x] = r2[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int j = 0; j < 4; ++j) { printf("r%d:", j); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[64 * j + i]); printf("\n"); }
    return 0;
}
