// How long until a wave's burst of global stores has retired from vmcnt?  Each wave of a 512-thread workgroup writes
// 16 x 1 KiB (its 128x64 bf16 tile equivalent) to fresh memory, then waits vmcnt(0); s_memrealtime (100 MHz) brackets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(uint4* __restrict__ dst, unsigned* __restrict__ ticks, int nstores, size_t wave_stride) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint4* p = dst + ((size_t)blockIdx.x * 8 + wave) * wave_stride + lane;
    const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < nstores; ++i) p[(size_t)i * 64] = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = (unsigned)(t1 - t0);
}
int main() {
    const int maxb = 256 * 8;
    const size_t wave_stride = 64 * 64;   // 64 KiB per wave region (uint4 = 16 B)
    uint4* dst; unsigned* ticks;
    hipMalloc(&dst, (size_t)maxb * 8 * wave_stride * 16); hipMalloc(&ticks, maxb * 8 * 4);
    std::vector<unsigned> h(maxb * 8);
    for (int blocks : {1, 32, 256, 1024}) {
        for (int nst : {16, 32}) {
            for (int rep = 0; rep < 3; ++rep) k<<<blocks, 512>>>(dst, ticks, nst, wave_stride);
            hipMemcpy(h.data(), ticks, blocks * 8 * 4, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + blocks * 8);
            printf("blocks %4d stores/wave %2d (%.1f MB total): retire time median %.2f us, p90 %.2f us, max %.2f us\n", blocks, nst,
                   blocks * 8.0 * nst * 1024 / 1e6, h[blocks * 4] * 0.01, h[(int)(blocks * 8 * 0.9)] * 0.01, h[blocks * 8 - 1] * 0.01);
        }
    }
    return 0;
}
