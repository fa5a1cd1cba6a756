// Retire rate of a 128x64 bf16 wave tile (16 KiB, 16 x dwordx4 stores per wave, 8 waves per workgroup = one 256x256
// tile) for different lane -> address maps.  Row pitch 4608 B (N = 2304 bf16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr long PITCH = 4608;
template <int PAT>
__global__ void k(char* __restrict__ base, unsigned* __restrict__ ticks) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wr = wave >> 2, wc = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    // tile origin of this workgroup: 197 x 9 tiles of 256 rows x 512 B
    const long tm = blockIdx.x / 9, tn = blockIdx.x % 9;
    char* t0p = base + (tm * 256 + wr * 128) * PITCH + tn * 512 + wc * 128;      // this wave's 128 rows x 128 B
    const u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 7u};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        char* q;
        if (PAT == 0) q = base + ((long)blockIdx.x * 8 + wave) * 16384 + i * 1024 + lane * 16;            // contiguous 1 KiB
        if (PAT == 1) q = t0p + (long)(i * 8 + (lane >> 3)) * PITCH + (lane & 7) * 16;                      // 8 rows x 128 B
        if (PAT == 2) q = t0p + (long)((i >> 1) * 16 + li) * PITCH + (i & 1) * 64 + lg * 16;                // 16 rows x 64 B
        if (PAT == 3) q = t0p + (long)((i >> 1) * 16 + li) * PITCH + (i & 1) * 64 + (lg & 1) * 32 + (lg >> 1) * 16;  // as epilogue_direct
        if (PAT == 4) q = t0p + (long)((i >> 2) * 32 + (lane >> 1)) * PITCH + (i & 3) * 32 + (lane & 1) * 16;       // 32 rows x 32 B
        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(q), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = (unsigned)(t1 - t0);
}
template <int PAT>
void run(const char* name, char* dst, unsigned* ticks, int blocks) {
    std::vector<unsigned> h(blocks * 8);
    for (int rep = 0; rep < 3; ++rep) k<PAT><<<blocks, 512>>>(dst, ticks);
    hipMemcpy(h.data(), ticks, blocks * 8 * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("blocks %4d  %-34s: median retire %.2f us  p90 %.2f us -> %.1f GB/s per CU\n", blocks, name, h[blocks * 4] * 0.01,
           h[(int)(blocks * 8 * 0.9)] * 0.01, 8.0 * 16 * 1024 / (h[blocks * 4] * 0.01) / 1e3);
}
int main() {
    char* dst; unsigned* ticks;
    hipMalloc(&dst, (size_t)50432 * PITCH + (1 << 20)); hipMalloc(&ticks, 2048 * 8 * 4);
    for (int blocks : {1, 256, 1773}) {
        run<0>("contiguous 1 KiB / instr", dst, ticks, blocks);
        run<1>("8 rows x 128 B (full lines)", dst, ticks, blocks);
        run<2>("16 rows x 64 B", dst, ticks, blocks);
        run<3>("16 rows x 2x32 B (epilogue_direct)", dst, ticks, blocks);
        run<4>("32 rows x 32 B", dst, ticks, blocks);
    }
    return 0;
}
