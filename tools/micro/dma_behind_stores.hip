// Does LDS-DMA data land in LDS while OLDER global stores of the same wave are still un-retired in vmcnt?
// One wave per workgroup: N x 1-KiB stores, then one global_load_lds (16 B / lane) of a known pattern, then poll LDS.
// Reports (a) when the pattern became visible in LDS, (b) when s_waitcnt vmcnt(0) returned.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(uint4* __restrict__ dst, const uint4* __restrict__ pat, unsigned* __restrict__ res, int nstores) {
    __shared__ uint4 lds[64];
    const int lane = threadIdx.x;
    lds[lane] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    uint4* p = dst + (size_t)blockIdx.x * nstores * 64 + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < nstores; ++i) p[(size_t)i * 64] = make_uint4(lane, i, 1, 2);
    __builtin_amdgcn_global_load_lds((gptr_t)(pat + lane), (lptr_t)lds, 16, 0, 0);
    unsigned long long tseen = 0;
    // poll with an inline-asm ds_read: hipcc puts s_waitcnt vmcnt(0) in front of any LDS read it can see while a
    // global_load_lds is in flight, which would hide exactly what this test looks for
    const unsigned faddr = (unsigned)(uintptr_t)(lptr_t)(reinterpret_cast<unsigned*>(&lds[63]) + 3);   // last dword the DMA writes
    for (int it = 0; it < 100000; ++it) {
        unsigned f;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(faddr) : "memory");
        if (__builtin_amdgcn_readfirstlane(f) == 0xABCD0000u + 63) { tseen = __builtin_amdgcn_s_memrealtime(); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tdone = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { res[blockIdx.x * 2] = (unsigned)(tseen - t0); res[blockIdx.x * 2 + 1] = (unsigned)(tdone - t0); }
}
int main() {
    const int blocks = 256;
    uint4 *dst, *pat; unsigned* res;
    hipMalloc(&dst, (size_t)blocks * 64 * 64 * 16); hipMalloc(&pat, 64 * 16); hipMalloc(&res, blocks * 8);
    std::vector<uint4> hp(64);
    for (int i = 0; i < 64; ++i) hp[i] = make_uint4(i, i, i, 0xABCD0000u + i);
    hipMemcpy(pat, hp.data(), 64 * 16, hipMemcpyHostToDevice);
    std::vector<unsigned> h(blocks * 2);
    for (int nst : {0, 8, 32, 64}) {
        for (int rep = 0; rep < 3; ++rep) k<<<blocks, 64>>>(dst, pat, res, nst);
        hipMemcpy(h.data(), res, blocks * 8, hipMemcpyDeviceToHost);
        std::vector<unsigned> a, b;
        for (int i = 0; i < blocks; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        printf("%2d stores before the DMA: pattern visible in LDS after %.2f us (median), vmcnt(0) returned after %.2f us\n", nst,
               a[blocks / 2] * 0.01, b[blocks / 2] * 0.01);
    }
    return 0;
}
