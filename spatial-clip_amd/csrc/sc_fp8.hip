// FP8 (OCP e4m3fn) forward path of the big linears -- BASELINE configs[4] "fp8 MFMA" (no reference counterpart: the
// reference's only low-precision linear is the int8 bitsandbytes hook of its legacy loop, src/open_clip_train/main.py:259-271).
//
// Scaling recipe ("per-row, power-of-two, just in time"):
//   * every ROW of an operand (a token's activation vector; an output channel's weight vector) gets its own scale
//     s = 2^floor(log2(448 / amax(row)))  (448 = largest e4m3 value), so the stored row uses the top of the e4m3 range and
//     the scale multiplication is exact in fp32;  amax = 0 -> s = 1;
//   * scales are computed by the kernel that quantises the row (the whole row is in one wave) -- no amax history, no
//     delayed scaling, nothing carried between steps;
//   * the GEMM accumulates the raw e4m3 products in fp32 on the matrix cores (block scale 1.0) and multiplies
//     accumulator (m, n) by a_scale_inv[m] * b_scale_inv[n] before bias / residual / GELU;
//   * master weights stay fp32; weight gradients stay bf16 (they read the bf16 activations, which are still written);
//   * round 3: the quantiser runs INSIDE the kernel that produces a complete row -- LayerNorm forward (A operand of the
//     qkv / c_fc GEMMs) and LayerNorm backward (the residual gradient, A operand of the c_proj / out_proj data-gradient
//     GEMMs, against a per-input-channel e4m3 copy of the transposed weight) -- see sc_norm.hip; operands whose rows are
//     assembled by several workgroups (attention output, GELU output, dU, dqkv) stay bf16: a stand-alone quantiser pass
//     over them costs more than the e4m3 GEMM gives back.
#include "sc_gemm_common.h"

namespace {

SC_DEVICE unsigned pack4_fp8(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// one wave per row; cols % 8 == 0.  scale_inv[row] = 1 / s.
template <bool F32>
__global__ __launch_bounds__(256) void quantize_rows_kernel(const void* __restrict__ src, long long ld_src, int rows, int cols,
                                                            unsigned char* __restrict__ dst, long long ld_dst,
                                                            float* __restrict__ scale_inv, float fixed_scale) {
    const int row = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* sf = reinterpret_cast<const float*>(src) + (long long)row * ld_src;
    const bf16* sb = reinterpret_cast<const bf16*>(src) + (long long)row * ld_src;
    float s = fixed_scale;
    if (!(fixed_scale > 0.f)) {
        float amax = 0.f;
        for (int c = lane * 8; c < cols; c += 512) {
            if (F32) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(sf + c), v1 = *reinterpret_cast<const f32x4*>(sf + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(v0[e]), fabsf(v1[e])));
            } else {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(sb + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf((float)v[e]));
            }
        }
        amax = sc_wave_max(amax);
        s = amax > 0.f ? exp2f(floorf(log2f(448.0f / amax))) : 1.0f;
    }
    if (lane == 0 && scale_inv) scale_inv[row] = 1.0f / s;
    unsigned char* d = dst + (long long)row * ld_dst;
    for (int c = lane * 8; c < cols; c += 512) {
        float v[8];
        if (F32) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sf + c), v1 = *reinterpret_cast<const f32x4*>(sf + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
        } else {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(sb + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(v[e] * s, -448.f), 448.f);
        u32x2 o;
        o[0] = pack4_fp8(v[0], v[1], v[2], v[3]);
        o[1] = pack4_fp8(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<u32x2*>(d + c) = o;
    }
}

// The same for MANY matrices in one launch (round 5): the e4m3 copies of every Linear weight are refreshed after each optimiser
// step -- at ViT-L/14 that was 225 launches of ~7 us, 1.6 ms per step, most of it launch latency.  desc[i] = {src, src_is_f32,
// ld_src, rows, cols, dst, ld_dst, scale_inv} (8 x int64), block_prefix[i] = first block (4 rows each) of matrix i.
__global__ __launch_bounds__(256) void quantize_rows_batched_kernel(const long long* __restrict__ desc,
                                                                    const int* __restrict__ block_prefix, int n) {
    int lo = 0, hi = n;
    const int b = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (block_prefix[mid] <= b) lo = mid; else hi = mid;
    }
    const long long* dsc = desc + (long long)lo * 8;
    const bool f32 = dsc[1] != 0;
    const long long ld_src = dsc[2], ld_dst = dsc[6];
    const int rows = (int)dsc[3], cols = (int)dsc[4];
    const int row = (b - block_prefix[lo]) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* sf = reinterpret_cast<const float*>(dsc[0]) + (long long)row * ld_src;
    const bf16* sb = reinterpret_cast<const bf16*>(dsc[0]) + (long long)row * ld_src;
    float amax = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
        if (f32) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sf + c), v1 = *reinterpret_cast<const f32x4*>(sf + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(v0[e]), fabsf(v1[e])));
        } else {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(sb + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf((float)v[e]));
        }
    }
    amax = sc_wave_max(amax);
    const float s = amax > 0.f ? exp2f(floorf(log2f(448.0f / amax))) : 1.0f;
    float* scale_inv = reinterpret_cast<float*>(dsc[7]);
    if (lane == 0 && scale_inv) scale_inv[row] = 1.0f / s;
    unsigned char* d = reinterpret_cast<unsigned char*>(dsc[5]) + (long long)row * ld_dst;
    for (int c = lane * 8; c < cols; c += 512) {
        float v[8];
        if (f32) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sf + c), v1 = *reinterpret_cast<const f32x4*>(sf + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
        } else {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(sb + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(v[e] * s, -448.f), 448.f);
        u32x2 o;
        o[0] = pack4_fp8(v[0], v[1], v[2], v[3]);
        o[1] = pack4_fp8(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<u32x2*>(d + c) = o;
    }
}

// One thread per tensor: scale <- 2^(floor(log2(448 / amax)) - margin_bits) from the maximum the epilogues recorded during the
// step that has just finished (left alone while nothing was recorded), then clear the 64 slots for the next step.
__global__ void fp8_scale_update_kernel(float* __restrict__ amax_slots, float* __restrict__ scale,
                                        float* __restrict__ scale_inv, int n, int margin_bits) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float amax = 0.f;
    for (int k = 0; k < 64; ++k) {
        amax = fmaxf(amax, amax_slots[t * 64 + k]);
        amax_slots[t * 64 + k] = 0.f;
    }
    if (amax > 0.f && amax < 3.0e38f) {
        const float s = exp2f(floorf(log2f(448.0f / amax)) - (float)margin_bits);
        scale[t] = s;
        scale_inv[t] = 1.0f / s;
    }
}

// The same with a short history (advisor, round 3): this step's maximum goes to hist[slot][t] and the scale is taken from the
// maximum over all hist_len recorded steps, so that one quiet step does not leave the next, louder one to the clamp at +-448.
__global__ void fp8_scale_update_hist_kernel(float* __restrict__ amax_slots, float* __restrict__ hist, int hist_len, int slot,
                                             float* __restrict__ scale, float* __restrict__ scale_inv, int n, int margin_bits) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float amax = 0.f;
    for (int k = 0; k < 64; ++k) {
        amax = fmaxf(amax, amax_slots[t * 64 + k]);
        amax_slots[t * 64 + k] = 0.f;
    }
    if (!(amax < 3.0e38f)) amax = 0.f;          // inf / nan maxima are not history
    hist[(long long)slot * n + t] = amax;
    float hmax = 0.f;
    for (int k = 0; k < hist_len; ++k) hmax = fmaxf(hmax, hist[(long long)k * n + t]);
    if (hmax > 0.f) {
        const float s = exp2f(floorf(log2f(448.0f / hmax)) - (float)margin_bits);
        scale[t] = s;
        scale_inv[t] = 1.0f / s;
    }
}

}  // namespace

extern "C" int sc_fp8_scale_update_hist(float* amax_slots, float* hist, int hist_len, int slot, float* scale, float* scale_inv,
                                        int n, int margin_bits, void* stream) {
    SC_CHECK(n > 0 && amax_slots && hist && scale && scale_inv && hist_len >= 1 && hist_len <= 64 && slot >= 0 && slot < hist_len &&
             margin_bits >= 0 && margin_bits <= 8, "sc_fp8_scale_update_hist: bad arguments");
    fp8_scale_update_hist_kernel<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(amax_slots, hist, hist_len, slot, scale, scale_inv, n,
                                                                                margin_bits);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_fp8_scale_update(float* amax_slots, float* scale, float* scale_inv, int n, int margin_bits, void* stream) {
    SC_CHECK(n > 0 && amax_slots && scale && scale_inv && margin_bits >= 0 && margin_bits <= 8, "sc_fp8_scale_update: bad arguments");
    fp8_scale_update_kernel<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(amax_slots, scale, scale_inv, n, margin_bits);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_quantize_rows_fp8(const void* src, int src_is_f32, long long ld_src, int rows, int cols, void* dst_fp8,
                                    long long ld_dst, float* scale_inv, float fixed_scale, void* stream) {
    SC_CHECK(rows > 0 && cols > 0 && (cols % 8) == 0 && ld_src >= cols && ld_dst >= cols && (ld_dst % 8) == 0,
             "sc_quantize_rows_fp8: bad shape rows=%d cols=%d", rows, cols);
    SC_CHECK((ld_src % (src_is_f32 ? 4 : 8)) == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst_fp8 % 8) == 0,
             "sc_quantize_rows_fp8: alignment");
    const int blocks = (rows + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
    if (src_is_f32) quantize_rows_kernel<true><<<blocks, 256, 0, st>>>(src, ld_src, rows, cols, (unsigned char*)dst_fp8, ld_dst, scale_inv, fixed_scale);
    else quantize_rows_kernel<false><<<blocks, 256, 0, st>>>(src, ld_src, rows, cols, (unsigned char*)dst_fp8, ld_dst, scale_inv, fixed_scale);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_quantize_rows_fp8_batched(const long long* desc, const int* block_prefix, int n, int total_blocks,
                                            void* stream) {
    SC_CHECK(desc != nullptr && block_prefix != nullptr && n > 0 && total_blocks > 0, "sc_quantize_rows_fp8_batched: empty plan");
    quantize_rows_batched_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(desc, block_prefix, n);
    SC_LAUNCH_CHECK();
    return 0;
}

static int gemm_fp8_impl(int epi, const void* A8, int lda, const float* a_scale_inv, int a_scale_scalar, const void* B8, int ldb,
                         const float* b_scale_inv, int M, int N, int K, void* C, int ldc, void* C2, int ldc2,
                         const float* bias, const void* res, int ldres, const void* aux, int ldaux, void* q8_out,
                         long long ldq8, const float* q8_scale, float* q8_amax, void* stream) {
    const int act = sc_epi_act(epi);          // SC_EPI_QGELU_*: the erf twin's kernel instance with the activation flag set
    epi = sc_epi_base(epi);
    SC_CHECK(M > 0 && N > 0 && K > 0 && (K % 128) == 0, "sc_gemm_fp8: K (%d) must be a positive multiple of 128", K);
    SC_CHECK((lda % 16) == 0 && (ldb % 16) == 0 && ((uintptr_t)A8 % 16) == 0 && ((uintptr_t)B8 % 16) == 0,
             "sc_gemm_fp8: operand rows must be 16-byte aligned (lda=%d ldb=%d)", lda, ldb);
    const bool f32out = (epi == SC_EPI_F32 || epi == SC_EPI_F32_BIAS_RES);
    SC_CHECK(!sc_epi_aux_mul(epi) || (aux != nullptr && (ldaux % 8) == 0), "sc_gemm_fp8: the GELU' epilogues need aux (ldaux %% 8 == 0)");
    SC_CHECK((N % (f32out ? 4 : 8)) == 0 && (ldc % 4) == 0 && ((uintptr_t)C % 16) == 0, "sc_gemm_fp8: N=%d ldc=%d", N, ldc);
    GemmArgs g;
    g.A = (const bf16*)A8; g.B = (const bf16*)B8; g.M = M; g.N = N; g.K = K / 2; g.lda = lda / 2; g.ldb = ldb / 2;
    g.C = C; g.ldc = ldc; g.C2 = C2; g.ldc2 = ldc2; g.bias = bias; g.res = (const float*)res; g.ldres = ldres;
    g.aux = (const bf16*)aux; g.ldaux = ldaux; g.colsum = nullptr; g.tile_offset = 0;
    g.a_scale = a_scale_inv; g.b_scale = b_scale_inv; g.a_scale_scalar = a_scale_scalar;
    g.act = act;
    if (q8_out != nullptr) {
        SC_CHECK(sc_epi_gelu_fwd(epi) || sc_epi_aux_mul(epi), "sc_gemm_fp8_q: the e4m3 second output exists for the GELU pair / GELU' epilogues");
        SC_CHECK(q8_scale != nullptr && q8_amax != nullptr && ldq8 >= N && (ldq8 % 8) == 0 && ((uintptr_t)q8_out % 8) == 0,
                 "sc_gemm_fp8_q: q8 output needs its scale, 64 amax slots and an 8-byte aligned row stride (ldq8=%lld)", ldq8);
        g.q8 = (unsigned char*)q8_out; g.ldq8 = ldq8; g.q8_scale = q8_scale; g.q8_amax = q8_amax;
    }
    const int took = sc_gemm8p_fp8(epi, g, (hipStream_t)stream);
    SC_CHECK(took == 1, "sc_gemm_fp8: shape not supported (M=%d N=%d K=%d epi=%d)", M, N, K, epi);
    return 0;
}

extern "C" int sc_gemm_fp8(int epi, const void* A8, int lda, const float* a_scale_inv, const void* B8, int ldb,
                           const float* b_scale_inv, int M, int N, int K, void* C, int ldc, void* C2, int ldc2,
                           const float* bias, const void* res, int ldres, const void* aux, int ldaux, void* stream) {
    return gemm_fp8_impl(epi, A8, lda, a_scale_inv, 0, B8, ldb, b_scale_inv, M, N, K, C, ldc, C2, ldc2, bias, res, ldres, aux,
                         ldaux, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int sc_gemm_fp8_q(int epi, const void* A8, int lda, const float* a_scale_inv, int a_scale_scalar, const void* B8,
                             int ldb, const float* b_scale_inv, int M, int N, int K, void* C, int ldc, void* C2, int ldc2,
                             const float* bias, const void* res, int ldres, const void* aux, int ldaux, void* q8_out,
                             long long ldq8, const float* q8_scale, float* q8_amax, void* stream) {
    return gemm_fp8_impl(epi, A8, lda, a_scale_inv, a_scale_scalar, B8, ldb, b_scale_inv, M, N, K, C, ldc, C2, ldc2, bias, res,
                         ldres, aux, ldaux, q8_out, ldq8, q8_scale, q8_amax, stream);
}
