// Per-CU retire rate of 16-B/lane global stores under different cache-policy bits (one workgroup, 8 waves, 16 stores each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int POL>
__global__ void k(uint4* __restrict__ dst, unsigned* __restrict__ ticks, int nstores) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint4* p = dst + ((size_t)blockIdx.x * 8 + wave) * nstores * 64 + lane;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 7u};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < nstores; ++i) {
        uint4* q = p + (size_t)i * 64;
        if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(q), "v"(v) : "memory");
        if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(q), "v"(v) : "memory");
        if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
        if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(v) : "memory");
        if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(q), "v"(v) : "memory");
        if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(q), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = (unsigned)(t1 - t0);
}
template <int POL>
void run(const char* name, uint4* dst, unsigned* ticks, int blocks) {
    std::vector<unsigned> h(blocks * 8);
    for (int rep = 0; rep < 3; ++rep) k<POL><<<blocks, 512>>>(dst, ticks, 16);
    hipMemcpy(h.data(), ticks, blocks * 8 * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("blocks %3d  %-12s: median retire %.2f us -> %.1f GB/s per CU\n", blocks, name, h[blocks * 4] * 0.01, 8.0 * 16 * 1024 / (h[blocks * 4] * 0.01) / 1e3);
}
int main() {
    uint4* dst; unsigned* ticks;
    hipMalloc(&dst, (size_t)256 * 8 * 16 * 64 * 16); hipMalloc(&ticks, 256 * 8 * 4);
    for (int blocks : {1, 256}) {
        run<0>("plain", dst, ticks, blocks); run<1>("nt", dst, ticks, blocks); run<2>("sc1", dst, ticks, blocks);
        run<3>("sc0 sc1", dst, ticks, blocks); run<4>("sc0", dst, ticks, blocks); run<5>("sc0 sc1 nt", dst, ticks, blocks);
    }
    return 0;
}
