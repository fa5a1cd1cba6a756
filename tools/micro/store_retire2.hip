// Per-CU store throughput: vary the number of storing waves per workgroup, the bytes per lane and the grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int W>   // W = dwords per lane
__global__ void k(unsigned* __restrict__ dst, unsigned* __restrict__ ticks, int nstores, size_t wave_stride, int active_waves) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned* p = dst + ((size_t)blockIdx.x * 16 + wave) * wave_stride + lane * W;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (wave < active_waves) {
        for (int i = 0; i < nstores; ++i) {
            if (W == 4) *reinterpret_cast<uint4*>(p + (size_t)i * 64 * W) = make_uint4(lane, wave, i, 7);
            if (W == 2) *reinterpret_cast<uint2*>(p + (size_t)i * 64 * W) = make_uint2(lane, i);
            if (W == 1) p[(size_t)i * 64 * W] = lane + i;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) ticks[blockIdx.x * 16 + wave] = (unsigned)(t1 - t0);
}
template <int W>
void run(unsigned* dst, unsigned* ticks, int blocks, int threads, int nst, int active) {
    const size_t wave_stride = (size_t)nst * 64 * W + 0;   // contiguous per workgroup
    std::vector<unsigned> h(blocks * 16);
    for (int rep = 0; rep < 3; ++rep) k<W><<<blocks, threads>>>(dst, ticks, nst, wave_stride, active);
    hipMemcpy(h.data(), ticks, blocks * 16 * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned> v;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < active; ++w) v.push_back(h[b * 16 + w]);
    std::sort(v.begin(), v.end());
    const double us = v[v.size() / 2] * 0.01, bytes = (double)active * nst * 64 * W * 4;
    printf("blocks %4d threads %4d storing waves %2d  %2d B/lane x %2d stores: median retire %.2f us -> %.1f GB/s per CU\n", blocks, threads,
           active, W * 4, nst, us, bytes / us / 1e3);
}
int main() {
    unsigned *dst, *ticks;
    hipMalloc(&dst, (size_t)1024 * 16 * 64 * 64 * 4 * 4); hipMalloc(&ticks, 1024 * 16 * 4);
    for (int blocks : {1, 256}) {
        run<4>(dst, ticks, blocks, 512, 16, 1);
        run<4>(dst, ticks, blocks, 512, 16, 2);
        run<4>(dst, ticks, blocks, 512, 16, 4);
        run<4>(dst, ticks, blocks, 512, 16, 8);
        run<4>(dst, ticks, blocks, 1024, 16, 16);
        run<2>(dst, ticks, blocks, 512, 16, 8);
        run<1>(dst, ticks, blocks, 512, 16, 8);
        run<4>(dst, ticks, blocks, 512, 64, 8);
    }
    return 0;
}
