// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the Spatial-CLIP training step.
// gfx950 only: 64-lane waves, MFMA 16x16x32 bf16, ds_read_b64_tr_b16, buffer loads with range check.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define SC_DEVICE __device__ __forceinline__
#define SC_WAVE 64

// ---- error plumbing of the C ABI (defined in sc_api.hip) ----
extern "C" const char* sc_last_error();
void sc_set_error(const char* fmt, ...);
#define SC_CHECK(cond, ...)            \
    do {                               \
        if (!(cond)) {                 \
            sc_set_error(__VA_ARGS__); \
            return -1;                 \
        }                              \
    } while (0)
#define SC_LAUNCH_CHECK()                                              \
    do {                                                               \
        hipError_t e__ = hipGetLastError();                            \
        if (e__ != hipSuccess) {                                       \
            sc_set_error("HIP launch failed: %s", hipGetErrorString(e__)); \
            return -2;                                                 \
        }                                                              \
    } while (0)

// ---- buffer resources: out-of-range loads return 0, which gives free edge handling ----
SC_DEVICE __amdgpu_buffer_rsrc_t sc_make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
SC_DEVICE u32x4 sc_buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}
SC_DEVICE uint32_t sc_clamp_bytes(uint64_t bytes) {
    return bytes > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)bytes;
}

// ---- bf16 helpers ----
SC_DEVICE float sc_bf2f(bf16 x) { return (float)x; }
SC_DEVICE bf16 sc_f2bf(float x) { return (bf16)x; }
SC_DEVICE float sc_bfbits2f(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }

SC_DEVICE bf16x8 sc_as_bf16x8(u32x4 v) {
    union { u32x4 u; bf16x8 b; } c;
    c.u = v;
    return c.b;
}
SC_DEVICE u32x4 sc_as_u32x4(bf16x8 v) {
    union { u32x4 u; bf16x8 b; } c;
    c.b = v;
    return c.u;
}
SC_DEVICE u32x2 sc_pack4(float a, float b, float c, float d) {
    union { u32x2 u; bf16x4 h; } x;
    x.h[0] = (bf16)a; x.h[1] = (bf16)b; x.h[2] = (bf16)c; x.h[3] = (bf16)d;
    return x.u;
}

// ---- LDS transposed read: 4 rows x 16 cols of 16-bit per 16-lane group, delivered column-major ----
// lane 4q+p of a group supplies the address of (row q, cols 4p..4p+3); lane i receives column i, rows 0..3.
SC_DEVICE bf16x4 sc_lds_tr16(const void* lds_ptr) {
    typedef short4v __attribute__((address_space(3))) * lds_s4_ptr;
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(uintptr_t)(lds_ptr));
    union { short4v s; bf16x4 b; } c;
    c.s = v;
    return c.b;
}
SC_DEVICE bf16x8 sc_cat(bf16x4 lo, bf16x4 hi) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// compile-time loop whose body needs the index as a constant (immediate operands of inline asm)
template <int V> struct sc_int { static constexpr int value = V; };
template <int N, int I = 0, class F>
SC_DEVICE void sc_static_for(F&& f) {
    if constexpr (I < N) {
        f(sc_int<I>{});
        sc_static_for<N, I + 1>(f);
    }
}

SC_DEVICE f32x4 sc_mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// exact-erf GELU and its derivative (nn.GELU default)
SC_DEVICE float sc_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
SC_DEVICE float sc_gelu_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

SC_DEVICE float sc_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
SC_DEVICE float sc_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a 1-D block id: consecutive remapped ids share an XCD (and its L2).
SC_DEVICE int sc_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
