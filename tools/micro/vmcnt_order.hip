// Does a younger global STORE retire from vmcnt before an older, slower global LOAD on gfx950?
// Each wave: load (HBM miss) -> store (small hot buffer) -> s_waitcnt vmcnt(1) -> look at the load's destination.
// If vmcnt retires strictly in issue order the destination always holds the loaded value; if stores can retire
// early, some lanes still see the sentinel.  Prints the number of early observations.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* __restrict__ big, unsigned* __restrict__ hot, unsigned* __restrict__ out, size_t stride_elems, int mode) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned* src = big + gid * stride_elems;
    unsigned* dst = hot + (gid & 1023);
    unsigned v = 0xDEADBEEFu, one = gid;
    if (mode == 0) {
        asm volatile("global_load_dword %0, %1, off\n\t"
                     "global_store_dword %2, %3, off\n\t"
                     "s_waitcnt vmcnt(1)\n\t"
                     "v_mov_b32 %0, %0" : "+v"(v) : "v"(src), "v"(dst), "v"(one) : "memory");
    } else {   // control: store first, then load, vmcnt(1) must NOT guarantee the load
        asm volatile("global_store_dword %2, %3, off\n\t"
                     "global_load_dword %0, %1, off\n\t"
                     "s_waitcnt vmcnt(1)\n\t"
                     "v_mov_b32 %0, %0" : "+v"(v) : "v"(src), "v"(dst), "v"(one) : "memory");
    }
    const unsigned seen = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[gid] = seen;
}
int main() {
    const int blocks = 2048, threads = 256;
    const size_t n = (size_t)blocks * threads, stride = 64;      // 256 B apart: every lane its own line, 128 MiB footprint
    unsigned *big, *hot, *out;
    hipMalloc(&big, n * stride * 4); hipMalloc(&hot, 4096); hipMalloc(&out, n * 4);
    std::vector<unsigned> h(n * stride);
    for (size_t i = 0; i < n; ++i) h[i * stride] = 0x1000000u + (unsigned)i;
    hipMemcpy(big, h.data(), n * stride * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> o(n);
    for (int mode = 0; mode < 2; ++mode) {
        size_t early = 0, wrong = 0;
        for (int rep = 0; rep < 5; ++rep) {
            k<<<blocks, threads>>>(big, hot, out, stride, mode);
            hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < n; ++i) {
                if (o[i] == 0xDEADBEEFu) ++early;
                else if (o[i] != 0x1000000u + (unsigned)i) ++wrong;
            }
        }
        printf("mode %d (%s): sentinel seen %zu times of %zu, wrong values %zu\n", mode,
               mode == 0 ? "load then store, vmcnt(1)" : "control: store then load, vmcnt(1)", early, n * 5, wrong);
    }
    return 0;
}
