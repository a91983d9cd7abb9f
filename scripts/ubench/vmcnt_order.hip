// Hardware check of the assumption every counted s_waitcnt vmcnt(N) in csrc/ (and in hipcc's own code for gfx9) rests on: vector-memory
// operations of one wave retire IN ISSUE ORDER, stores included -- "at most N outstanding" can therefore only be reached once
// everything older than the youngest N has completed.  Each wave requests one dword that has to come from HBM (a line of a 4 GiB
// buffer it has never touched), then issues NST stores to lines that are hot in its L2, then waits with vmcnt(NST) and copies the
// load's destination register at once; vmcnt(0) follows and the register is read again.  If a fast store could retire before the
// slow load, the first copy would still hold the sentinel the register was set to.  Reported: waves whose early copy differs from
// the loaded value (must be 0), and -- as a control that the experiment can see a load in flight at all -- the same with vmcnt(NST + 1).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/vmcnt_order.hip -o scripts/ubench/vmcnt_order.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define NST 24
#define STR2(x) #x
#define STR(x) STR2(x)
#define ST8(o) "global_store_dword %[hot], %[one], off offset:" #o "\n\t"
#define ST8B(o) "global_store_dword %[hot2], %[one], off offset:" #o "\n\t"

template <int EXTRA>
__global__ __launch_bounds__(256) void k(const unsigned* cold, unsigned* hot, unsigned* early, unsigned* late, size_t stride_words) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned lane = threadIdx.x & 63;
    const unsigned* src = cold + (size_t)wave * stride_words + lane;          // one 256-byte stretch per wave, never touched before
    unsigned* h = hot + (size_t)(wave & 1023) * 64 * 64 + lane;               // 24 stores of 256 contiguous bytes each (two lines per instruction: quick to issue)
    unsigned v = 0xdeadbeefu, e, one = 1u;
    for (int i = 0; i < 24; ++i) h[i * 64] = 0;                                // make the store targets resident
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile(
        "global_load_dword %[v], %[src], off\n\t"
        ST8(0) ST8(256) ST8(512) ST8(768) ST8(1024) ST8(1280) ST8(1536) ST8(1792) ST8(2048) ST8(2304) ST8(2560) ST8(2816)
        ST8(3072) ST8(3328) ST8(3584) ST8(3840) ST8B(0) ST8B(256) ST8B(512) ST8B(768) ST8B(1024) ST8B(1280) ST8B(1536) ST8B(1792)
        "s_waitcnt vmcnt(%[n])\n\t"
        "v_mov_b32 %[e], %[v]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        : [v] "+v"(v), [e] "=&v"(e)
        : [src] "v"(src), [hot] "v"(h), [hot2] "v"(h + 1024), [one] "v"(one), [n] "n"(NST + EXTRA)
        : "memory");
    if (lane == 0) {
        early[wave] = e;
        late[wave] = v;
    }
}

int main() {
    const size_t waves = 1 << 16, stride_words = 16384;                        // 64 KiB apart: 4 GiB walked once per launch
    unsigned *cold, *hot, *early, *late;
    hipMalloc(&cold, waves * stride_words * 4);
    hipMalloc(&hot, (size_t)1024 * 64 * 64 * 4);
    hipMalloc(&early, waves * 4);
    hipMalloc(&late, waves * 4);
    std::vector<unsigned> he(waves), hl(waves);
    for (int extra = 0; extra < 2; ++extra) {
        long bad = 0, total = 0;
        for (int rep = 0; rep < 6; ++rep) {
            hipMemset(cold, 0x11 * (rep + 1), waves * stride_words * 4);        // new values, and the lines leave the caches (4 GiB >> MALL)
            hipDeviceSynchronize();
            if (extra) hipLaunchKernelGGL(k<1>, dim3(waves / 4), dim3(256), 0, 0, cold, hot, early, late, stride_words);
            else       hipLaunchKernelGGL(k<0>, dim3(waves / 4), dim3(256), 0, 0, cold, hot, early, late, stride_words);
            hipDeviceSynchronize();
            hipMemcpy(he.data(), early, waves * 4, hipMemcpyDeviceToHost);
            hipMemcpy(hl.data(), late, waves * 4, hipMemcpyDeviceToHost);
            const unsigned want = 0x01010101u * (0x11 * (rep + 1));
            for (size_t i = 0; i < waves; ++i) {
                total++;
                if (hl[i] != want) { printf("late value wrong at wave %zu: %08x\n", i, hl[i]); return 2; }
                bad += he[i] != want;
            }
        }
        printf("vmcnt(%d) after 1 HBM load + %d hot stores: %ld of %ld waves read the register before the load had landed%s\n", NST + extra, NST,
               bad, total, extra ? "   (control: one operation too many allowed)" : "   (must be 0: in-order retirement)");
    }
    return 0;
}
