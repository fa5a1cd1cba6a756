// Wave 0 of a workgroup issues N x 1-KiB global stores; wave 1 (another SIMD) meanwhile issues 16 DMA loads one after
// another (each waited).  How much are wave 1's loads slowed by wave 0's store burst?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(uint4* __restrict__ dst, const uint4* __restrict__ src, unsigned* __restrict__ res, int nstores, int storing_waves) {
    __shared__ uint4 lds[8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (wave < storing_waves) {
        uint4* p = dst + ((size_t)blockIdx.x * 8 + wave) * 64 * 64 + lane;
        for (int i = 0; i < nstores; ++i) p[(size_t)i * 64] = make_uint4(lane, i, 1, 2);
    } else if (wave == 7) {
        const uint4* s = src + (size_t)blockIdx.x * 16 * 4096 + lane;
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(s + (size_t)i * 4096), (lptr_t)&lds[wave][0], 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) res[blockIdx.x] = (unsigned)(t1 - t0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
int main() {
    const int blocks = 256;
    uint4 *dst, *src; unsigned* res;
    hipMalloc(&dst, (size_t)blocks * 8 * 64 * 64 * 16); hipMalloc(&src, (size_t)blocks * 16 * 4096 * 16); hipMalloc(&res, blocks * 4);
    hipMemset(src, 1, (size_t)blocks * 16 * 4096 * 16);
    std::vector<unsigned> h(blocks);
    for (int sw : {0, 1, 4, 7}) {
        for (int rep = 0; rep < 3; ++rep) k<<<blocks, 512>>>(dst, src, res, 64, sw);
        hipMemcpy(h.data(), res, blocks * 4, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%d storing waves (64 x 1 KiB each): 16 dependent DMA loads of wave 7 took %.2f us (median) -> %.2f us per load\n", sw,
               h[blocks / 2] * 0.01, h[blocks / 2] * 0.01 / 16);
    }
    return 0;
}
