// Does packed f32 VALU (v_pk_fma_f32 / v_pk_mul_f32) double the throughput of a VALU-only section (a GEMM epilogue
// with no MFMA in flight)?  512-thread workgroups (two waves per SIMD, like the GEMM kernels), one per CU.
// Each variant performs the same number of float FMAs per lane; independent chains, registers only.
//   hipcc --offload-arch=gfx950 -O2 -o valu_pk valu_pk.hip && ./valu_pk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    const float m = 0.999f, c = 1e-3f;
    f32x2 m2 = {m, m}, c2 = {c, c};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {                 // 16 scalar FMAs
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MODE == 1) {          // 8 packed FMAs = the same 16 float FMAs
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 v = {a[i], a[i + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(m2), "v"(c2));
                a[i] = v[0];
                a[i + 1] = v[1];
            }
        } else if (MODE == 2) {          // 16 scalar multiplies
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        } else if (MODE == 3) {          // 8 packed multiplies
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 v = {a[i], a[i + 1]};
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(m2));
                a[i] = v[0];
                a[i + 1] = v[1];
            }
        } else if (MODE == 4) {          // 16 v_exp_f32 (transcendental rate)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == 5) {          // 16 v_cvt_pk_bf16_f32 (8 results)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                unsigned r;
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(a[i + 1]));
                a[i] = __uint_as_float(r);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int iters, int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    k<MODE><<<256, threads>>>(out, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 200000;
    const char* names[] = {"16 v_fma_f32", "8 v_pk_fma_f32", "16 v_mul_f32", "8 v_pk_mul_f32", "16 v_exp_f32", "8 v_cvt_pk_bf16_f32"};
    for (int w = 1; w <= 2; ++w) {
        float t[6] = {run<0>(out, iters, w), run<1>(out, iters, w), run<2>(out, iters, w),
                      run<3>(out, iters, w), run<4>(out, iters, w), run<5>(out, iters, w)};
        for (int i = 0; i < 6; ++i)
            printf("%d wave(s)/SIMD  %-22s %8.3f ms  = %.2f ns per iteration (16 floats per lane)\n", w, names[i], t[i],
                   t[i] * 1e6 / iters);
    }
    return 0;
}
