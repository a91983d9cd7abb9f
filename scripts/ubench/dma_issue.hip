// Microbenchmark: what one LDS-DMA piece (global_load_lds_dwordx4, 1 KiB per wave-instruction) costs a wave that is
// busy issuing bf16 MFMAs (one wave per SIMD, 4 waves per workgroup, one workgroup per CU) -- and whether the
// instruction's immediate offset moves the LDS destination along with the global source.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/dma_issue.hip -o scripts/ubench/dma_issue.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define MF(acc) "v_mfma_f32_32x32x16_bf16 a[" acc "], v[0:3], v[4:7], a[" acc "]\n\t"
#define MF4 MF("0:15") MF("16:31") MF("32:47") MF("48:63")
#define MF8 MF4 MF4
// piece forms
#define P_M0NOP(voff, imm) "s_add_u32 m0, %[lds], " imm "\n\ts_nop 0\n\tglobal_load_lds_dwordx4 " voff ", %[src]\n\t"
#define P_OFF(imm) "global_load_lds_dwordx4 %[v0], %[src] offset:" imm "\n\t"
#define SETM0 "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
#define DSR(n) "ds_read_b128 a[" n "], %[la] offset:" #n "*0\n\t"

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const char* src, long long* cyc, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane16 = (threadIdx.x & 63) * 16;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem) + w * 4096;
    const char* s = src + (size_t)blockIdx.x * 65536 + w * 4096;
    const unsigned v1 = lane16 + 1024, v2 = lane16 + 2048, v3 = lane16 + 3072;
    const unsigned la = (unsigned)(size_t)smem + lane16;
    long long r0 = __builtin_amdgcn_s_memrealtime();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const char* sp = s + (size_t)(it & 63) * 16384 * 256;  // walks 256 MB: HBM / MALL traffic as in the real kernel
        if (MODE == 0) {
            asm volatile(MF8 MF8 MF8 MF8 : : : "memory");
        } else if (MODE == 1) {  // the kernel's current form: M0 write + nop + load, pieces behind MFMAs 25, 27, 29, 31
            asm volatile("s_waitcnt vmcnt(12)\n\t" MF8 MF8 MF8 MF("0:15") MF("16:31") P_M0NOP("%[v0]", "0") MF("32:47") MF("48:63")
                         P_M0NOP("%[v1]", "1024") MF("0:15") MF("16:31") P_M0NOP("%[v2]", "2048") MF("32:47") MF("48:63") P_M0NOP("%[v3]", "3072")
                         : : [lds] "s"(lds), [src] "s"(sp), [v0] "v"(lane16), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3) : "memory", "scc");
        } else if (MODE == 2) {  // one M0 write, the piece offset in the instruction
            asm volatile("s_waitcnt vmcnt(12)\n\t" MF8 MF8 MF8 SETM0 MF("0:15") MF("16:31") P_OFF("0") MF("32:47") MF("48:63") P_OFF("1024")
                         MF("0:15") MF("16:31") P_OFF("2048") MF("32:47") MF("48:63") P_OFF("3072")
                         : : [lds] "s"(lds), [src] "s"(sp), [v0] "v"(lane16) : "memory", "scc");
        } else if (MODE == 3) {  // burst: four pieces back to back
            asm volatile("s_waitcnt vmcnt(12)\n\t" MF8 MF8 MF8 MF4 SETM0 P_OFF("0") P_OFF("1024") P_OFF("2048") P_OFF("3072") MF4
                         : : [lds] "s"(lds), [src] "s"(sp), [v0] "v"(lane16) : "memory", "scc");
        } else if (MODE == 5) {  // one M0 write, pieces as far apart as possible
            asm volatile("s_waitcnt vmcnt(12)\n\t" SETM0 MF4 P_OFF("0") MF8 P_OFF("1024") MF8 P_OFF("2048") MF8 P_OFF("3072") MF4
                         : : [lds] "s"(lds), [src] "s"(sp), [v0] "v"(lane16) : "memory", "scc");
        } else if (MODE == 6) {  // MODE 2 + 16 ds_read_b128 spread over the first half (the kernel's operand reads)
#define RD(a, o) "ds_read_b128 a[" a "], %[la] offset:" o "\n\t"
            asm volatile("s_waitcnt vmcnt(12)\n\t"
                         MF("0:15") RD("64:67", "0") MF("16:31") RD("68:71", "1024") MF("32:47") RD("72:75", "2048") MF("48:63") RD("76:79", "3072")
                         MF("0:15") RD("80:83", "4096") MF("16:31") RD("84:87", "5120") MF("32:47") RD("88:91", "6144") MF("48:63") RD("92:95", "7168")
                         MF("0:15") RD("64:67", "8192") MF("16:31") RD("68:71", "9216") MF("32:47") RD("72:75", "10240") MF("48:63") RD("76:79", "11264")
                         MF("0:15") RD("80:83", "12288") MF("16:31") RD("84:87", "13312") MF("32:47") RD("88:91", "14336") MF("48:63") RD("92:95", "15360")
                         MF8 SETM0 MF("0:15") MF("16:31") P_OFF("0") MF("32:47") MF("48:63") P_OFF("1024")
                         MF("0:15") MF("16:31") P_OFF("2048") MF("32:47") MF("48:63") P_OFF("3072") "s_waitcnt lgkmcnt(0)\n\t"
                         : : [lds] "s"(lds), [src] "s"(sp), [v0] "v"(lane16), [la] "v"(la) : "memory", "scc");
        } else if (MODE == 7) {  // MODE 6 without the DMA
            asm volatile(MF("0:15") RD("64:67", "0") MF("16:31") RD("68:71", "1024") MF("32:47") RD("72:75", "2048") MF("48:63") RD("76:79", "3072")
                         MF("0:15") RD("80:83", "4096") MF("16:31") RD("84:87", "5120") MF("32:47") RD("88:91", "6144") MF("48:63") RD("92:95", "7168")
                         MF("0:15") RD("64:67", "8192") MF("16:31") RD("68:71", "9216") MF("32:47") RD("72:75", "10240") MF("48:63") RD("76:79", "11264")
                         MF("0:15") RD("80:83", "12288") MF("16:31") RD("84:87", "13312") MF("32:47") RD("88:91", "14336") MF("48:63") RD("92:95", "15360")
                         MF8 MF8 "s_waitcnt lgkmcnt(0)\n\t"
                         : : [la] "v"(la) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" ::: "a0", "a15", "a31", "a47", "a63", "a95");
    long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
    if (sink && threadIdx.x == 12345) sink[0] = smem[lane16];
}

// where does `offset:` put the data?  one wave: piece with offset 1024 from a source whose dword i holds i
__global__ void where(const unsigned* src, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* l = (unsigned*)smem;
    for (int i = threadIdx.x; i < 2048; i += 64) l[i] = 0xffffffffu;
    __syncthreads();
    const unsigned lane16 = threadIdx.x * 16;
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    asm volatile("s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[src] offset:1024\n\ts_waitcnt vmcnt(0)"
                 : : [lds] "s"(lds), [src] "s"(src), [v0] "v"(lane16) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = l[i];
}

template <int MODE>
void run(const char* name, const char* src, long long* cyc) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 20000;
    k<MODE><<<256, 256, 131072>>>(src, cyc, 200, nullptr);
    hipDeviceSynchronize();
    k<MODE><<<256, 256, 131072>>>(src, cyc, iters, nullptr);
    hipDeviceSynchronize();
    long long c[2] = {0, 0};
    hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-64s %8.1f cycles per 32 MFMAs (%5.2f per MFMA)  clock %.0f MHz\n", name, (double)c[0] / iters, (double)c[0] / iters / 32, (double)c[0] / (double)c[1] * 100.0);
}

int main() {
    char* src;
    long long* cyc;
    hipMalloc(&src, (size_t)512 << 20);
    hipMemset(src, 0, (size_t)512 << 20);
    hipMalloc(&cyc, 16);
    {
        std::vector<unsigned> h(4096);
        for (int i = 0; i < 4096; ++i) h[i] = i;
        unsigned *d, *o;
        hipMalloc(&d, 16384);
        hipMalloc(&o, 8192);
        hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
        where<<<1, 64, 8192>>>(d, o);
        std::vector<unsigned> r(2048);
        hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost);
        int first = -1;
        for (int i = 0; i < 2048; ++i)
            if (r[i] != 0xffffffffu) { first = i; break; }
        printf("offset:1024 -> first LDS dword written = %d (byte %d), holds source dword %u (byte %u)\n", first, first * 4, first >= 0 ? r[first] : 0,
               first >= 0 ? r[first] * 4 : 0);
        hipMemset(src, 0, (size_t)512 << 20);
    }
    run<0>("32 MFMAs alone", src, cyc);
    run<1>("+ 4 x [s_add m0; s_nop; piece], behind MFMAs 25/27/29/31", src, cyc);
    run<2>("+ 1 M0 write, 4 pieces with offset:, same places", src, cyc);
    run<3>("+ 1 M0 write, 4 pieces back to back", src, cyc);
    run<5>("+ 1 M0 write, 4 pieces 8 MFMAs apart", src, cyc);
    run<7>("32 MFMAs + 16 ds_read_b128", src, cyc);
    run<6>("32 MFMAs + 16 ds_read_b128 + 4 pieces with offset:", src, cyc);
    return 0;
}
