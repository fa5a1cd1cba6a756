// HBM-bound row kernels of the ViT tower: LayerNorm fwd/bwd on the fp32 residual stream (bf16 out for
// the next GEMM), L2-normalise fwd/bwd of the embedding heads, column sums (bias grads), casts.
// One 64-lane wave owns one row; every access is a 16-byte (fp32x4) or 8-byte (bf16x4) vector.
#include "sc_common.h"
#include <stdlib.h>
#include <mutex>
#include "sc_kernels.h"
#include "sc_gemm_common.h"   // sc_gelu_fast: the GELU of the GEMM epilogues

namespace {

constexpr int MAXV = 8;  // float4 slots per lane -> d <= 2048

SC_DEVICE f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
SC_DEVICE void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
SC_DEVICE f32x4 ldbf4(const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
SC_DEVICE void stbf4(bf16* p, f32x4 v) {
    bf16x4 o;
    o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}

// ------------------------------------------------------------------ LayerNorm forward
// fp8 path (sc_fp8.hip's recipe, fused): the row is complete in this wave's registers, so the kernel that normalises it
// also emits its e4m3 copy with the row's power-of-two scale -- the forward GEMM's A operand -- and the separate
// quantiser pass (read bf16 + write fp8 of the whole activation) disappears.
SC_DEVICE unsigned ln_pack4_fp8(f32x4 v, float s) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[0] * s, -448.f), 448.f), fminf(fmaxf(v[1] * s, -448.f), 448.f), 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[2] * s, -448.f), 448.f), fminf(fmaxf(v[3] * s, -448.f), 448.f), w, true);
    return (unsigned)w;
}
SC_DEVICE float ln_row_scale(float amax_lane) {
    const float amax = sc_wave_max(amax_lane);
    return amax > 0.f ? exp2f(floorf(log2f(448.0f / amax))) : 1.0f;
}
// Second e4m3 copy with ONE scale for the whole tensor (round 4): the operand of the e4m3 WEIGHT-gradient GEMM, whose reduction
// runs over the token rows -- a per-row scale cannot be pulled out of it.  Delayed scaling as for the GELU epilogues' copies
// (sc_fp8.hip): quantise with the previous steps' scale *t_scale, max-reduce |value| of this launch into t_amax[64].
struct LnT8 {
    unsigned char* t8 = nullptr;
    long long ldt8 = 0;
    const float* t_scale = nullptr;
    float* t_amax = nullptr;
};
SC_DEVICE void ln_t8_amax(float row_amax, float* slots, int row) {     // row_amax: wave-uniform, non-negative
    unsigned* s = reinterpret_cast<unsigned*>(slots) + (row & 63);
    if ((threadIdx.x & 63) == 0 && row_amax > __uint_as_float(*s)) atomicMax(s, __float_as_uint(row_amax));
}

// XB: the input rows are bf16 (the residual stream kept in bf16, SC_EPI_BF16_BIAS_RES); x then points at bf16 data
template <int NV, bool Q8, bool XB = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     bf16* __restrict__ y, long long ldy, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, int d, float eps,
                                                     unsigned char* __restrict__ y8, long long ldy8,
                                                     float* __restrict__ scale_inv, const LnT8 t8 = LnT8()) {
    const int lane = threadIdx.x & 63;
    // Rows are walked LAST ROW FIRST: the producer in front (a GEMM walking its tiles upwards) wrote the high rows last, so
    // they are the ones still in the 256-MB Infinity Cache; and the rows this kernel writes last are the low ones the next
    // GEMM reads first.  Same-box A/B of the whole step: -0.19 ms (profiles/r03_ln_row_order_ab.txt).
    const int row = rows - 1 - (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));     // wave-uniform: scalar row pointers
    if (row < 0) return;
    const float* xr = x + (long long)row * ldx;
    const bf16* xrb = reinterpret_cast<const bf16*>(x) + (long long)row * ldx;
    const int nv = d >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = i * 64 + lane;
        if (e < nv) {
            v[i] = XB ? ldbf4(xrb + e * 4) : ld4(xr + e * 4);
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mu = sc_wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = i * 64 + lane;
        if (e < nv) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float t = v[i][c] - mu; q += t * t; }
        }
    }
    const float rs = rsqrtf(sc_wave_sum(q) / (float)d + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
    bf16* yr = y + (long long)row * ldy;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = i * 64 + lane;
        if (e < nv) {
            const f32x4 g = ld4(gamma + e * 4), b = ld4(beta + e * 4);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (v[i][c] - mu) * rs * g[c] + b[c];
            stbf4(yr + e * 4, o);
            if (Q8) {
                v[i] = o;
#pragma unroll
                for (int c = 0; c < 4; ++c) amax = fmaxf(amax, fabsf(o[c]));
            }
        }
    }
    if (Q8) {
        const float sc = ln_row_scale(amax);
        if (lane == 0) scale_inv[row] = 1.0f / sc;
        unsigned char* y8r = y8 + (long long)row * ldy8;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = i * 64 + lane;
            if (e < nv) *reinterpret_cast<unsigned*>(y8r + e * 4) = ln_pack4_fp8(v[i], sc);
        }
        if (t8.t8 != nullptr) {              // per-tensor copy for the weight-gradient GEMM
            const float ts = *t8.t_scale;
            unsigned char* tr = t8.t8 + (long long)row * t8.ldt8;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = i * 64 + lane;
                if (e < nv) *reinterpret_cast<unsigned*>(tr + e * 4) = ln_pack4_fp8(v[i], ts);
            }
            ln_t8_amax(sc_wave_max(amax), t8.t_amax, row);
        }
    }
}

// ------------------------------------------------------------------ LayerNorm backward
// dres_new = (accumulate ? dres : 0) + LNbwd(dy) (accumulate = -P: dres only holds rows r % P == 0); also emits the bf16 copy of dres_new (the A operand of
// the next dgrad / wgrad GEMMs) and per-block partials of dgamma, dbeta and colsum(dres_new).
// gin != nullptr: the residual gradient travels in bf16 (the reference's own precision for it: under its bf16 autocast the residual
// stream and therefore its gradient are bf16 tensors): the incoming gradient is read from ``gin`` (bf16) instead of the fp32
// ``dres``, and ``dres`` is written only when ``write_f32`` asks for it (the last hop in front of the stem) -- 10 instead of
// 16 bytes per element cross HBM.  The sparse form (accumulate = -P) still reads its few class-token rows from ``dres``.
// GIN: the incoming gradient is the bf16 stream `gin` for every row (accumulate > 0): its loads are issued together with
// dy / x instead of behind the two wave reductions (one exposed memory round trip per row less; gamma is re-read per row from
// L1 so that the kernel stays at 128 VGPRs = 4 waves per SIMD).
template <int NV, bool Q8, bool XB = false, bool GIN = false, bool LEAN = false>
__global__ __launch_bounds__(256, (LEAN && NV == 3 && !Q8) ? 5 : 1) void ln_bwd_kernel(const bf16* __restrict__ dy, long long lddy,
                                                     const float* __restrict__ x, long long ldx,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, float* __restrict__ dres,
                                                     long long lddres, bf16* __restrict__ dres_bf, long long lddbf,
                                                     float* __restrict__ partial, int rows, int d, int accumulate,
                                                     unsigned char* __restrict__ d8, long long ldd8,
                                                     float* __restrict__ scale_inv, const bf16* __restrict__ gin = nullptr,
                                                     long long ldgin = 0, int write_f32 = 1, const LnT8 t8 = LnT8(),
                                                     int nominal_blocks = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: row pointers live in SGPRs
    const int nv = d >> 2;
    f32x4 ag[NV], ab[NV], ac[NV], gm[GIN ? 1 : NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = ab[i] = ac[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int e = i * 64 + lane;
        if (!GIN) gm[i] = e < nv ? ld4(gamma + e * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int row_ = blockIdx.x * 4 + wave; row_ < rows; row_ += gridDim.x * 4) {
        const int row = rows - 1 - row_;              // last row first: see ln_fwd_kernel
        const float mu = mean[row], rs = rstd[row];
        const bf16* dyr = dy + (long long)row * lddy;
        const float* xr = x + (long long)row * ldx;
        const bf16* xrb = reinterpret_cast<const bf16*>(x) + (long long)row * ldx;
        if constexpr (LEAN) {
            // Wide rows (d = 1024: NV = 4) with bf16 rows and a bf16 gradient stream: the row's three inputs stay in their
            // packed bf16 form (24 registers instead of the 40 of g / x_hat in fp32) and dy * gamma, x_hat are formed twice,
            // gamma re-read from L1 -- 140 -> under 128 VGPRs, the fourth wave per SIMD (same arithmetic, same order).  With
            // the e4m3 copies (Q8) the new gradient is formed a THIRD time once the row's scale is known, instead of being kept.
            static_assert(GIN && XB, "lean row body: bf16 rows, bf16 gradient stream");
            bf16x4 dyp[NV], xp[NV], gp[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = i * 64 + lane;
                if (e < nv) {
                    gp[i] = *reinterpret_cast<const bf16x4*>(gin + (long long)row * ldgin + e * 4);
                    dyp[i] = *reinterpret_cast<const bf16x4*>(dyr + e * 4);
                    xp[i] = *reinterpret_cast<const bf16x4*>(xrb + e * 4);
                }
            }
            float s1 = 0.f, s2 = 0.f, amax_l = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = i * 64 + lane;
                if (e < nv) {
                    const f32x4 gmv = ld4(gamma + e * 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float dyv = (float)dyp[i][c];
                        const float xhv = __fmul_rn((float)xp[i][c] - mu, rs);      // (rounded products: the second pass must
                        const float gv = __fmul_rn(dyv, gmv[c]);                     //  form the very same values, uncontracted)
                        s1 += gv;
                        s2 += gv * xhv;
                        ag[i][c] += dyv * xhv;
                        ab[i][c] += dyv;
                    }
                }
            }
            s1 = sc_wave_sum(s1) / (float)d;
            s2 = sc_wave_sum(s2) / (float)d;
            float* dr = dres + (long long)row * lddres;
            bf16* db = dres_bf ? dres_bf + (long long)row * lddbf : nullptr;
#pragma unroll
            for (int i = 0; i < NV; ++i) {       // opaque to the optimiser: otherwise it keeps the fp32 conversions of the first pass alive
                union { bf16x4 v; u32x2 u; } a, b;
                a.v = dyp[i]; b.v = xp[i];
                asm volatile("" : "+v"(a.u), "+v"(b.u));
                dyp[i] = a.v; xp[i] = b.v;
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = i * 64 + lane;
                if (e < nv) {
                    const f32x4 gmv = ld4(gamma + e * 4);
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float xhv = __fmul_rn((float)xp[i][c] - mu, rs);
                        const float gv = __fmul_rn((float)dyp[i][c], gmv[c]);
                        o[c] = __fmaf_rn(rs, __fmaf_rn(-xhv, s2, __fsub_rn(gv, s1)), (float)gp[i][c]);     // (pinned: formed again below)
                        ac[i][c] += o[c];
                        if (Q8) amax_l = fmaxf(amax_l, fabsf(o[c]));
                    }
                    if (write_f32) st4(dr + e * 4, o);
                    if (db) stbf4(db + e * 4, o);
                }
            }
            if constexpr (Q8) {
                const float sc = ln_row_scale(amax_l);
                if (lane == 0) scale_inv[row] = 1.0f / sc;
                unsigned char* d8r = d8 + (long long)row * ldd8;
                const bool tt = t8.t8 != nullptr;
                const float ts = tt ? *t8.t_scale : 1.0f;
                unsigned char* tr = tt ? t8.t8 + (long long)row * t8.ldt8 : nullptr;
#pragma unroll
                for (int i = 0; i < NV; ++i) {   // opaque again: or the second pass's values stay alive for this one
                    union { bf16x4 v; u32x2 u; } a, b, cc;
                    a.v = dyp[i]; b.v = xp[i]; cc.v = gp[i];
                    asm volatile("" : "+v"(a.u), "+v"(b.u), "+v"(cc.u));
                    dyp[i] = a.v; xp[i] = b.v; gp[i] = cc.v;
                }
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int e = i * 64 + lane;
                    if (e < nv) {
                        const f32x4 gmv = ld4(gamma + e * 4);
                        f32x4 o;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float xhv = __fmul_rn((float)xp[i][c] - mu, rs);
                            const float gv = __fmul_rn((float)dyp[i][c], gmv[c]);
                            o[c] = __fmaf_rn(rs, __fmaf_rn(-xhv, s2, __fsub_rn(gv, s1)), (float)gp[i][c]);
                        }
                        *reinterpret_cast<unsigned*>(d8r + e * 4) = ln_pack4_fp8(o, sc);
                        if (tt) *reinterpret_cast<unsigned*>(tr + e * 4) = ln_pack4_fp8(o, ts);
                    }
                }
                if (tt) ln_t8_amax(sc_wave_max(amax_l), t8.t_amax, row);
            }
            continue;
        }
        f32x4 g[NV], xh[NV];
        bf16x4 gin_raw[GIN ? NV : 1];
        float s1 = 0.f, s2 = 0.f;
        if (GIN) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = i * 64 + lane;
                if (e < nv) gin_raw[i] = *reinterpret_cast<const bf16x4*>(gin + (long long)row * ldgin + e * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = i * 64 + lane;
            if (e < nv) {
                const f32x4 dyv = ldbf4(dyr + e * 4);
                const f32x4 xv = XB ? ldbf4(xrb + e * 4) : ld4(xr + e * 4);
                const f32x4 gmv = GIN ? ld4(gamma + e * 4) : gm[GIN ? 0 : i];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    xh[i][c] = (xv[c] - mu) * rs;
                    g[i][c] = dyv[c] * gmv[c];
                    s1 += g[i][c];
                    s2 += g[i][c] * xh[i][c];
                    ag[i][c] += dyv[c] * xh[i][c];
                    ab[i][c] += dyv[c];
                }
            }
        }
        s1 = sc_wave_sum(s1) / (float)d;
        s2 = sc_wave_sum(s2) / (float)d;
        float* dr = dres + (long long)row * lddres;
        bf16* db = dres_bf ? dres_bf + (long long)row * lddbf : nullptr;
        float amax = 0.f;
        // accumulate > 0: every row of dres holds an incoming residual gradient; accumulate = -P: only rows r with
        // r % P == 0 do (the class-token rows after a class-token-only block), the others start from zero and are not read
        const bool acc_row = accumulate > 0 || (accumulate < 0 && (row % (-accumulate)) == 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = i * 64 + lane;
            if (e < nv) {
                f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (GIN) o = (f32x4){(float)gin_raw[i][0], (float)gin_raw[i][1], (float)gin_raw[i][2], (float)gin_raw[i][3]};
                else if (acc_row) o = ld4(dr + e * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    o[c] += rs * (g[i][c] - s1 - xh[i][c] * s2);
                    ac[i][c] += o[c];
                }
                if (write_f32) st4(dr + e * 4, o);
                if (db) stbf4(db + e * 4, o);
                if (Q8) {
                    g[i] = o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) amax = fmaxf(amax, fabsf(o[c]));
                }
            }
        }
        if (Q8) {       // e4m3 copy of the new residual gradient + its row scale: the A operand of the fp8 dgrad GEMMs
            const float sc = ln_row_scale(amax);
            if (lane == 0) scale_inv[row] = 1.0f / sc;
            unsigned char* d8r = d8 + (long long)row * ldd8;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = i * 64 + lane;
                if (e < nv) *reinterpret_cast<unsigned*>(d8r + e * 4) = ln_pack4_fp8(g[i], sc);
            }
            if (t8.t8 != nullptr) {          // per-tensor copy of the new residual gradient: dY of the e4m3 c_proj weight gradient
                const float ts = *t8.t_scale;
                unsigned char* tr = t8.t8 + (long long)row * t8.ldt8;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int e = i * 64 + lane;
                    if (e < nv) *reinterpret_cast<unsigned*>(tr + e * 4) = ln_pack4_fp8(g[i], ts);
                }
                ln_t8_amax(sc_wave_max(amax), t8.t_amax, row);
            }
        }
    }
    // block reduce of the 3 column vectors: waves 1..3 through smem[wave - 1][3][d], wave 0 adds them to its registers in wave
    // order (w0 + w1 + w2 + w3) -- 36 KiB at d = 1024, so that four blocks fit a CU's LDS
    float* sm = reinterpret_cast<float*>(smem);
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = i * 64 + lane;
            if (e < nv) {
                st4(sm + ((wave - 1) * 3 + 0) * d + e * 4, ag[i]);
                st4(sm + ((wave - 1) * 3 + 1) * d + e * 4, ab[i]);
                st4(sm + ((wave - 1) * 3 + 2) * d + e * 4, ac[i]);
            }
        }
    }
    __syncthreads();
    float* pout = partial + (long long)blockIdx.x * 3 * d;
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = i * 64 + lane;
            if (e < nv) {
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    ag[i] += ld4(sm + (w * 3 + 0) * d + e * 4);
                    ab[i] += ld4(sm + (w * 3 + 1) * d + e * 4);
                    ac[i] += ld4(sm + (w * 3 + 2) * d + e * 4);
                }
                st4(pout + 0 * d + e * 4, ag[i]);
                st4(pout + 1 * d + e * 4, ab[i]);
                st4(pout + 2 * d + e * 4, ac[i]);
            }
        }
    }
    // the finalising pass sums `nominal_blocks` slots (a function of the row count alone: it may run from another entry point,
    // on another stream); a grid capped at the resident blocks leaves the others as zeros
    for (int sl = blockIdx.x + gridDim.x; sl < nominal_blocks; sl += gridDim.x) {
        float* z = partial + (long long)sl * 3 * d;
        for (int e = threadIdx.x; e < 3 * d; e += 256) z[e] = 0.f;
    }
}

// out_k[c] = sum_b partial[b][k][c], k = 0..nvec-1 ; outputs may be null
// block = 64 columns x 16 partial groups (fixed summation order -> deterministic)
__global__ __launch_bounds__(1024) void colvec_finalize_kernel(const float* __restrict__ partial, int nblk, int nvec,
                                                               int d, float* __restrict__ o0, float* __restrict__ o1,
                                                               float* __restrict__ o2) {
    __shared__ float sm[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (e < nvec * d) {
        // eight independent loads in flight per thread (the loop is latency-bound: 36 workgroups read ~9 MB); the order of
        // the additions is fixed, so the result stays deterministic
        float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = ty;
        for (; b + 7 * 16 < nblk; b += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] += partial[(long long)(b + u * 16) * nvec * d + e];
        }
        for (int u = 0; b < nblk; b += 16, ++u) p[u & 7] += partial[(long long)b * nvec * d + e];
        s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    }
    sm[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && e < nvec * d) {
        s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += sm[k][tx];
        const int k = e / d, c = e - k * d;
        float* o = k == 0 ? o0 : (k == 1 ? o1 : o2);
        if (o) o[c] = s;
    }
}

// ------------------------------------------------------------------ column sums of a bf16 matrix
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, long long ld, int rows, int n,
                                                     float* __restrict__ partial) {
    // block = 64 column-quads x 4 row lanes; grid.x over column chunks of 256, grid.y over row slices
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + cq) * 4;
    __shared__ f32x4 sm[4][64];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (col < n) {
        for (int r = blockIdx.y * 4 + rl; r < rows; r += gridDim.y * 4) acc += ldbf4(x + (long long)r * ld + col);
    }
    sm[rl][cq] = acc;
    __syncthreads();
    if (rl == 0 && col < n) {
        const f32x4 s = sm[0][cq] + sm[1][cq] + sm[2][cq] + sm[3][cq];
        st4(partial + (long long)blockIdx.y * n + col, s);
    }
}


// ------------------------------------------------------------------ h = gelu(u), bf16 -> bf16
// The activation-recomputation mode (SpatialClipNet.set_grad_checkpointing) does not keep the GELU output of a block for
// the backward; the c_proj weight gradient gets it back from the saved pre-activation u with the epilogue's own formula
// on the epilogue's own input (the bf16-rounded u), i.e. bit-identical to what the forward wrote.
__global__ __launch_bounds__(256) void gelu_bf16_kernel(const bf16* __restrict__ u, bf16* __restrict__ h, long long n8, int act) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(u + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)sc_act((float)v[e], act);
        *reinterpret_cast<bf16x8*>(h + i * 8) = o;
    }
}

// ------------------------------------------------------------------ L2 normalise (F.normalize, eps 1e-12)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         bf16* __restrict__ ybf, float* __restrict__ inv, int rows,
                                                         int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long long)row * d;
    float s = 0.f;
    for (int e = lane; e < d; e += 64) s += xr[e] * xr[e];
    const float iv = 1.0f / fmaxf(sqrtf(sc_wave_sum(s)), 1e-12f);
    if (lane == 0 && inv) inv[row] = iv;
    for (int e = lane; e < d; e += 64) {
        const float o = xr[e] * iv;
        y[(long long)row * d + e] = o;
        if (ybf) ybf[(long long)row * d + e] = (bf16)o;
    }
}

// dx = (dy - y * <y,dy>) * inv   (y = normalised output); written as bf16 for the dgrad/wgrad GEMMs
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv, bf16* __restrict__ dx,
                                                         int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= rows) return;
    const float* yr = y + (long long)row * d;
    const float* gr = dy + (long long)row * d;
    float s = 0.f;
    for (int e = lane; e < d; e += 64) s += yr[e] * gr[e];
    s = sc_wave_sum(s);
    const float iv = inv[row];
    for (int e = lane; e < d; e += 64) dx[(long long)row * d + e] = (bf16)((gr[e] - yr[e] * s) * iv);
}

// ------------------------------------------------------------------ casts
__global__ void cast_pad_kernel(const float* __restrict__ src, long long lds_, bf16* __restrict__ dst, long long ldd,
                                int rows, int cols, int cols_pad) {
    const long long nq = (long long)rows * (cols_pad >> 2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nq;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / (cols_pad >> 2));
        const int c = (int)(i - (long long)r * (cols_pad >> 2)) * 4;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c + 3 < cols) {
            v = ld4(src + r * lds_ + c);
        } else {
            for (int k = 0; k < 4; ++k)
                if (c + k < cols) v[k] = src[r * lds_ + c + k];
        }
        stbf4(dst + r * ldd + c, v);
    }
}

// dst[c][r] = bf16(src[r][c]) through a 64x64 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                             int rows, int cols, long long ldd) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(long long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[(long long)c * ldd + r] = (bf16)tile[tx][i];
    }
}

// one launch for every transposed bf16 weight copy: desc[i] = {src offset (floats), dst pointer, rows, cols, ld_dst},
// tile_prefix[i] = first 64x64 tile of matrix i (tile_prefix[n] = total); block -> matrix by binary search
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(const float* __restrict__ master,
                                                                     const long long* __restrict__ desc,
                                                                     const int* __restrict__ tile_prefix, int n) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n;
    const int b = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tile_prefix[mid] <= b) lo = mid; else hi = mid;
    }
    const long long* dsc = desc + (long long)lo * 5;
    const float* src = master + dsc[0];
    bf16* dst = reinterpret_cast<bf16*>(dsc[1]);
    const int rows = (int)dsc[2], cols = (int)dsc[3];
    const long long ldd = dsc[4];
    const int tb = b - tile_prefix[lo];
    const int tcols = (cols + 63) >> 6;
    const int r0 = (tb / tcols) * 64, c0 = (tb % tcols) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(long long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[(long long)c * ldd + r] = (bf16)tile[tx][i];
    }
}

// The same plan on the bf16 MIRROR of the master weights (the AdamW kernel has just written it): 2 instead of 4 bytes read
// per element, 16-byte loads and stores on interior tiles (the fp32 version moves 4-byte loads and 2-byte stores:
// 160 us per step at ViT-B/16).  bf16(master) either way: the copies are bit-identical.
__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const bf16* __restrict__ mirror,
                                                                     const long long* __restrict__ desc,
                                                                     const int* __restrict__ tile_prefix, int n) {
    __shared__ __attribute__((aligned(16))) bf16 tile[64][72];
    int lo = 0, hi = n;
    const int b = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tile_prefix[mid] <= b) lo = mid; else hi = mid;
    }
    const long long* dsc = desc + (long long)lo * 5;
    const bf16* src = mirror + dsc[0];
    bf16* dst = reinterpret_cast<bf16*>(dsc[1]);
    const int rows = (int)dsc[2], cols = (int)dsc[3];
    const long long ldd = dsc[4];
    const int tb = b - tile_prefix[lo];
    const int tcols = (cols + 63) >> 6;
    const int r0 = (tb / tcols) * 64, c0 = (tb % tcols) * 64;
    const bool interior = r0 + 64 <= rows && c0 + 64 <= cols && (cols & 7) == 0 && (ldd & 7) == 0 &&
                          ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    if (interior) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = threadIdx.x + 256 * j, row = q >> 3, ch = q & 7;
            *reinterpret_cast<u32x4*>(&tile[row][ch * 8]) =
                *reinterpret_cast<const u32x4*>(src + (long long)(r0 + row) * cols + c0 + ch * 8);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = threadIdx.x + 256 * j, c = q >> 3, rc = q & 7;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = tile[rc * 8 + e][c];
            *reinterpret_cast<bf16x8*>(dst + (long long)(c0 + c) * ldd + r0 + rc * 8) = o;
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(long long)r * cols + c] : (bf16)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[(long long)c * ldd + r] = tile[tx][i];
    }
}

}  // namespace

// u = bf16(x + bias), h = bf16(gelu_erf(u)) : the epilogue of a split-K forward Linear (gene fc1, K = 20k)
__global__ void bias_gelu_pair_kernel(const float* __restrict__ x, const float* __restrict__ bias, bf16* __restrict__ u,
                                      bf16* __restrict__ h, long long total4, int n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)((i * 4) % n);
        const f32x4 v = ld4(x + i * 4) + ld4(bias + c);
        bf16x4 uo, ho;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uo[k] = (bf16)v[k];
            const float uf = (float)uo[k];
            ho[k] = (bf16)(0.5f * uf * (1.0f + erff(uf * 0.70710678118654752f)));
        }
        *reinterpret_cast<bf16x4*>(u + i * 4) = uo;
        *reinterpret_cast<bf16x4*>(h + i * 4) = ho;
    }
}

extern "C" int sc_bias_gelu_pair(const float* x, const float* bias, void* u, void* h, int rows, int n, void* stream) {
    SC_CHECK(rows > 0 && n > 0 && (n % 4) == 0, "sc_bias_gelu_pair: bad shape");
    const long long total4 = (long long)rows * n / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    bias_gelu_pair_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, bias, (bf16*)u, (bf16*)h, total4, n);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cast_transpose_batched(const float* master, const void* mirror_bf16, const long long* desc,
                                         const int* tile_prefix, int n, int total_tiles, void* stream) {
    SC_CHECK(n > 0 && total_tiles > 0, "sc_cast_transpose_batched: nothing to do");
    SC_CHECK(master != nullptr || mirror_bf16 != nullptr, "sc_cast_transpose_batched: no source");
    if (mirror_bf16 != nullptr)
        transpose_bf16_batched_kernel<<<total_tiles, 256, 0, (hipStream_t)stream>>>((const bf16*)mirror_bf16, desc,
                                                                                    tile_prefix, n);
    else
        cast_transpose_batched_kernel<<<total_tiles, 256, 0, (hipStream_t)stream>>>(master, desc, tile_prefix, n);
    SC_LAUNCH_CHECK();
    return 0;
}

static int ln_fwd_launch(const float* x, long long ldx, const float* gamma, const float* beta, void* y, long long ldy,
                         float* mean, float* rstd, int rows, int d, float eps, void* y8, long long ldy8, float* scale_inv,
                         void* stream, bool xb = false, const LnT8 t8 = LnT8()) {
    SC_CHECK(rows > 0 && d > 0 && (d % 4) == 0 && d <= MAXV * 256, "sc_layernorm_fwd: bad shape rows=%d d=%d", rows, d);
    SC_CHECK((ldx % 4) == 0 && (ldy % 4) == 0, "sc_layernorm_fwd: row strides must be multiples of 4");
    SC_CHECK(y8 == nullptr || (scale_inv != nullptr && (ldy8 % 4) == 0 && ldy8 >= d),
             "sc_layernorm_fwd_q8: fp8 output needs scale_inv and a row stride that is a multiple of 4 (ldy8=%lld)", ldy8);
    const int nvv = (d / 4 + 63) / 64;
#define SC_LN_FWD(NV)                                                                                                     \
    do {                                                                                                                  \
        if (xb) {                                                                                                         \
            if (y8) ln_fwd_kernel<NV, true, true><<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(                       \
                x, ldx, gamma, beta, (bf16*)y, ldy, mean, rstd, rows, d, eps, (unsigned char*)y8, ldy8, scale_inv, t8);   \
            else ln_fwd_kernel<NV, false, true><<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(                         \
                x, ldx, gamma, beta, (bf16*)y, ldy, mean, rstd, rows, d, eps, nullptr, 0, nullptr);                       \
        } else if (y8) ln_fwd_kernel<NV, true><<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(                          \
            x, ldx, gamma, beta, (bf16*)y, ldy, mean, rstd, rows, d, eps, (unsigned char*)y8, ldy8, scale_inv);           \
        else ln_fwd_kernel<NV, false><<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(                                   \
            x, ldx, gamma, beta, (bf16*)y, ldy, mean, rstd, rows, d, eps, nullptr, 0, nullptr);                           \
    } while (0)
    if (nvv <= 1) SC_LN_FWD(1); else if (nvv == 2) SC_LN_FWD(2); else if (nvv == 3) SC_LN_FWD(3);
    else if (nvv == 4) SC_LN_FWD(4); else SC_LN_FWD(8);
#undef SC_LN_FWD
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_layernorm_fwd(const float* x, long long ldx, const float* gamma, const float* beta, void* y,
                                long long ldy, float* mean, float* rstd, int rows, int d, float eps, void* stream) {
    return ln_fwd_launch(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, d, eps, nullptr, 0, nullptr, stream);
}

extern "C" int sc_layernorm_fwd_q8(const float* x, long long ldx, const float* gamma, const float* beta, void* y,
                                   long long ldy, void* y_fp8, long long ldy8, float* scale_inv, float* mean, float* rstd,
                                   int rows, int d, float eps, void* stream) {
    SC_CHECK(y_fp8 != nullptr, "sc_layernorm_fwd_q8: fp8 output required");
    return ln_fwd_launch(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, d, eps, y_fp8, ldy8, scale_inv, stream);
}

extern "C" int sc_layernorm_fwd_x16(const void* x_bf16, long long ldx, const float* gamma, const float* beta, void* y,
                                    long long ldy, void* y_fp8, long long ldy8, float* scale_inv, float* mean, float* rstd,
                                    int rows, int d, float eps, void* stream) {
    SC_CHECK(x_bf16 != nullptr, "sc_layernorm_fwd_x16: input required");
    return ln_fwd_launch((const float*)x_bf16, ldx, gamma, beta, y, ldy, mean, rstd, rows, d, eps, y_fp8, ldy8, scale_inv, stream,
                         true);
}

extern "C" int sc_layernorm_fwd_x16_t8(const void* x_bf16, long long ldx, const float* gamma, const float* beta, void* y,
                                       long long ldy, void* y_fp8, long long ldy8, float* scale_inv, void* y_t8, long long ldt8,
                                       const float* t_scale, float* t_amax, float* mean, float* rstd, int rows, int d, float eps,
                                       void* stream) {
    SC_CHECK(x_bf16 != nullptr && y_fp8 != nullptr, "sc_layernorm_fwd_x16_t8: bf16 input rows and the per-row e4m3 output are required");
    SC_CHECK(y_t8 != nullptr && t_scale != nullptr && t_amax != nullptr && (ldt8 % 4) == 0 && ldt8 >= d,
             "sc_layernorm_fwd_x16_t8: per-tensor e4m3 output needs its scale, 64 amax slots and a row stride %% 4 == 0 (ldt8=%lld)", ldt8);
    LnT8 t8;
    t8.t8 = (unsigned char*)y_t8; t8.ldt8 = ldt8; t8.t_scale = t_scale; t8.t_amax = t_amax;
    return ln_fwd_launch((const float*)x_bf16, ldx, gamma, beta, y, ldy, mean, rstd, rows, d, eps, y_fp8, ldy8, scale_inv, stream,
                         true, t8);
}

// Partial-sum slots of a LayerNorm backward launch = its nominal block count, a function of the shape alone (the finalising
// pass may run from another entry point).  Rows up to 768 wide fit FIVE blocks per CU (<= 96 VGPRs, 27 KiB of LDS): 1280.
static int ln_nominal_blocks(int rows, int d) {
    const int cap = d <= 768 ? 1280 : 1024;
    const int nblk = (rows + 3) / 4;
    return nblk > cap ? cap : nblk;
}

extern "C" long long sc_layernorm_bwd_ws_floats(int rows, int d) {
    return (long long)ln_nominal_blocks(rows, d) * 3 * d;
}

// The row loop of the LayerNorm kernels is persistent (a wave walks rows with a grid stride), so a grid larger than what is
// resident at once only adds a second, thinly occupied round: at d = 1024 the backward needs 140 VGPRs = 3 waves per SIMD and
// 48 KiB of LDS = 3 blocks per CU, i.e. 768 of the 1024 blocks launched ran first and the other 256 afterwards with one block
// per CU (ViT-L/14: 190 us for 540 MB).  The grid is capped at the kernel's resident blocks (occupancy query, once per kernel
// and LDS size).
static int ln_resident_blocks(const void* fn, size_t lds, int nblk) {
#ifdef SC_LN_GRID_UNCAPPED
    return nblk;
#else
    struct Slot { const void* fn; size_t lds; int blocks; };
    static Slot slots[64];
    static int nslots = 0;
    static std::mutex mu;                       // (backward runs on the autograd engine's thread; tools call from the main one)
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < nslots; ++i)
        if (slots[i].fn == fn && slots[i].lds == lds) return nblk < slots[i].blocks ? nblk : slots[i].blocks;
    int per_cu = 0, dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return nblk;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess || per_cu <= 0) return nblk;
    const int blocks = per_cu * (p.multiProcessorCount > 0 ? p.multiProcessorCount : 256);
    if (nslots < 64) { slots[nslots].fn = fn; slots[nslots].lds = lds; slots[nslots].blocks = blocks; ++nslots; }
    return nblk < blocks ? nblk : blocks;
#endif
}

static int ln_bwd_launch(const void* dy, long long lddy, const float* x, long long ldx, const float* mean,
                         const float* rstd, const float* gamma, float* dres, long long lddres, void* dres_bf16,
                         long long lddbf, int accumulate, float* dgamma, float* dbeta, float* colsum, float* ws, int rows,
                         int d, void* d8, long long ldd8, float* scale_inv, void* stream,
                         const void* gin = nullptr, long long ldgin = 0, int write_f32 = 1, bool xb = false,
                         const LnT8 t8 = LnT8()) {
    SC_CHECK(rows > 0 && d > 0 && (d % 4) == 0 && d <= MAXV * 256, "sc_layernorm_bwd: bad shape rows=%d d=%d", rows, d);
    SC_CHECK(ws != nullptr, "sc_layernorm_bwd: workspace required");
    SC_CHECK(d8 == nullptr || (scale_inv != nullptr && (ldd8 % 4) == 0 && ldd8 >= d),
             "sc_layernorm_bwd_q8: fp8 output needs scale_inv and a row stride that is a multiple of 4 (ldd8=%lld)", ldd8);
    const int nblk = ln_nominal_blocks(rows, d);
    const size_t lds = (size_t)3 * 3 * d * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int nvv = (d / 4 + 63) / 64;
#define SC_LN_BWD_QG(NV, Q, G, I)                                                                                       \
    do {                                                                                                                \
        if (lds > 48 * 1024)                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_kernel<NV, Q, G, I>),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        const int grid = ln_resident_blocks(reinterpret_cast<const void*>(&ln_bwd_kernel<NV, Q, G, I>), lds, nblk);     \
        ln_bwd_kernel<NV, Q, G, I><<<grid, 256, lds, st>>>((const bf16*)dy, lddy, x, ldx, mean, rstd, gamma, dres,      \
                                                           lddres, (bf16*)dres_bf16, lddbf, ws, rows, d, accumulate,    \
                                                           (unsigned char*)d8, ldd8, scale_inv, (const bf16*)gin,       \
                                                           ldgin, write_f32, t8, nblk);                                 \
    } while (0)
#define SC_LN_BWD_Q(NV, Q, G)                                                                                           \
    do {                                                                                                                \
        if (gin != nullptr && accumulate > 0) SC_LN_BWD_QG(NV, Q, G, true); else SC_LN_BWD_QG(NV, Q, G, false);         \
    } while (0)
#define SC_LN_BWD(NV)                                                                                                   \
    do {                                                                                                                \
        if (xb) { if (d8) SC_LN_BWD_Q(NV, true, true); else SC_LN_BWD_Q(NV, false, true); }                             \
        else { if (d8) SC_LN_BWD_Q(NV, true, false); else SC_LN_BWD_Q(NV, false, false); }                              \
    } while (0)
    // lean row body (bf16 rows + bf16 gradient stream).  With the wave index in an SGPR (row pointers are scalar) the plain body
    // needs 128 VGPRs at d = 1024 = four waves per SIMD by itself and is the faster one there (111 vs 117 us); the lean body stays
    // the default for the QUANTISING instances at d = 1024 (130 VGPRs otherwise; 171 vs 209 us with both e4m3 copies).
    // SC_LN_BWD_LEAN=0: never; =4: also without e4m3 copies at d = 1024; =3: everywhere it exists (d = 768 too: no gain measured).
    const char* lean_env = getenv("SC_LN_BWD_LEAN");      // read per call: the tests run both bodies in one process
    const int lean_mode = lean_env ? atoi(lean_env) : -1;
    const bool lean_base = xb && gin != nullptr && accumulate > 0 && lean_mode != 0;
    const bool lean4 = lean_base && nvv == 4 && (d8 != nullptr || lean_mode == 4 || lean_mode == 3);
    const bool lean3 = lean_base && nvv == 3 && lean_mode == 3;
#define SC_LN_BWD_LEAN_Q(NV, Q)                                                                                         \
    do {                                                                                                                \
        if (lds > 48 * 1024)                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_kernel<NV, Q, true, true, true>),           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        const int grid = ln_resident_blocks(reinterpret_cast<const void*>(&ln_bwd_kernel<NV, Q, true, true, true>),     \
                                            lds, nblk);                                                                 \
        ln_bwd_kernel<NV, Q, true, true, true><<<grid, 256, lds, st>>>(                                                 \
            (const bf16*)dy, lddy, x, ldx, mean, rstd, gamma, dres, lddres, (bf16*)dres_bf16, lddbf, ws, rows, d,       \
            accumulate, (unsigned char*)d8, ldd8, scale_inv, (const bf16*)gin, ldgin, write_f32, t8, nblk);             \
    } while (0)
#define SC_LN_BWD_LEAN(NV) do { if (d8) SC_LN_BWD_LEAN_Q(NV, true); else SC_LN_BWD_LEAN_Q(NV, false); } while (0)
    if (lean4) SC_LN_BWD_LEAN(4);
    else if (lean3) SC_LN_BWD_LEAN(3);
    else if (nvv <= 1) SC_LN_BWD(1); else if (nvv == 2) SC_LN_BWD(2); else if (nvv == 3) SC_LN_BWD(3);
    else if (nvv == 4) SC_LN_BWD(4); else SC_LN_BWD(8);
#undef SC_LN_BWD_LEAN
#undef SC_LN_BWD_LEAN_Q
#undef SC_LN_BWD
#undef SC_LN_BWD_Q
#undef SC_LN_BWD_QG
    SC_LAUNCH_CHECK();
    if (dgamma == nullptr) return 0;      // deferred: the caller runs sc_layernorm_bwd_reduce (possibly on another stream)
    colvec_finalize_kernel<<<(3 * d + 63) / 64, 1024, 0, st>>>(ws, nblk, 3, d, dgamma, dbeta, colsum);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_layernorm_bwd(const void* dy, long long lddy, const float* x, long long ldx, const float* mean,
                                const float* rstd, const float* gamma, float* dres, long long lddres, void* dres_bf16,
                                long long lddbf, int accumulate, float* dgamma, float* dbeta, float* colsum,
                                float* ws, int rows, int d, void* stream) {
    return ln_bwd_launch(dy, lddy, x, ldx, mean, rstd, gamma, dres, lddres, dres_bf16, lddbf, accumulate, dgamma, dbeta,
                         colsum, ws, rows, d, nullptr, 0, nullptr, stream);
}

extern "C" int sc_layernorm_bwd_q8(const void* dy, long long lddy, const float* x, long long ldx, const float* mean,
                                   const float* rstd, const float* gamma, float* dres, long long lddres, void* dres_bf16,
                                   long long lddbf, void* dres_fp8, long long ldd8, float* scale_inv, int accumulate,
                                   float* dgamma, float* dbeta, float* colsum, float* ws, int rows, int d, void* stream) {
    SC_CHECK(dres_fp8 != nullptr, "sc_layernorm_bwd_q8: fp8 output required");
    return ln_bwd_launch(dy, lddy, x, ldx, mean, rstd, gamma, dres, lddres, dres_bf16, lddbf, accumulate, dgamma, dbeta,
                         colsum, ws, rows, d, dres_fp8, ldd8, scale_inv, stream);
}

extern "C" int sc_layernorm_bwd_g16(const void* dy, long long lddy, const float* x, long long ldx, const float* mean,
                                    const float* rstd, const float* gamma, const void* gin_bf16, long long ldgin,
                                    float* dres, long long lddres, int write_f32, void* gout_bf16, long long ldgout,
                                    void* gout_fp8, long long ldd8, float* scale_inv, int accumulate, float* dgamma,
                                    float* dbeta, float* colsum, float* ws, int rows, int d, void* stream) {
    SC_CHECK(gout_bf16 != nullptr, "sc_layernorm_bwd_g16: bf16 output required");
    SC_CHECK(accumulate <= 0 || (gin_bf16 != nullptr && (ldgin % 4) == 0), "sc_layernorm_bwd_g16: bf16 input gradient required");
    SC_CHECK((accumulate >= 0 && !write_f32) || dres != nullptr, "sc_layernorm_bwd_g16: fp32 buffer required (sparse input or fp32 output)");
    return ln_bwd_launch(dy, lddy, x, ldx, mean, rstd, gamma, dres, lddres, gout_bf16, ldgout, accumulate, dgamma, dbeta,
                         colsum, ws, rows, d, gout_fp8, ldd8, scale_inv, stream, gin_bf16, ldgin, write_f32);
}

extern "C" int sc_layernorm_bwd_x16(const void* dy, long long lddy, const void* x_bf16, long long ldx, const float* mean,
                                    const float* rstd, const float* gamma, const void* gin_bf16, long long ldgin,
                                    float* dres, long long lddres, int write_f32, void* gout_bf16, long long ldgout,
                                    void* gout_fp8, long long ldd8, float* scale_inv, int accumulate, float* dgamma,
                                    float* dbeta, float* colsum, float* ws, int rows, int d, void* stream) {
    SC_CHECK(x_bf16 != nullptr && (ldx % 4) == 0, "sc_layernorm_bwd_x16: bf16 input rows with a stride that is a multiple of 4");
    SC_CHECK((accumulate == 0 && !write_f32) || gin_bf16 != nullptr || dres != nullptr,
             "sc_layernorm_bwd_x16: an incoming gradient or an fp32 output needs its buffer");
    return ln_bwd_launch(dy, lddy, (const float*)x_bf16, ldx, mean, rstd, gamma, dres, lddres, gout_bf16, ldgout, accumulate,
                         dgamma, dbeta, colsum, ws, rows, d, gout_fp8, ldd8, scale_inv, stream, gin_bf16, ldgin, write_f32, true);
}

extern "C" int sc_layernorm_bwd_x16_t8(const void* dy, long long lddy, const void* x_bf16, long long ldx, const float* mean,
                                       const float* rstd, const float* gamma, const void* gin_bf16, long long ldgin,
                                       float* dres, long long lddres, int write_f32, void* gout_bf16, long long ldgout,
                                       void* gout_fp8, long long ldd8, float* scale_inv, void* gout_t8, long long ldt8,
                                       const float* t_scale, float* t_amax, int accumulate, float* dgamma, float* dbeta,
                                       float* colsum, float* ws, int rows, int d, void* stream) {
    SC_CHECK(x_bf16 != nullptr && (ldx % 4) == 0 && gout_fp8 != nullptr, "sc_layernorm_bwd_x16_t8: bf16 input rows and the per-row e4m3 output are required");
    SC_CHECK(gout_t8 != nullptr && t_scale != nullptr && t_amax != nullptr && (ldt8 % 4) == 0 && ldt8 >= d,
             "sc_layernorm_bwd_x16_t8: per-tensor e4m3 output needs its scale, 64 amax slots and a row stride %% 4 == 0 (ldt8=%lld)", ldt8);
    SC_CHECK((accumulate == 0 && !write_f32) || gin_bf16 != nullptr || dres != nullptr,
             "sc_layernorm_bwd_x16_t8: an incoming gradient or an fp32 output needs its buffer");
    LnT8 t8;
    t8.t8 = (unsigned char*)gout_t8; t8.ldt8 = ldt8; t8.t_scale = t_scale; t8.t_amax = t_amax;
    return ln_bwd_launch(dy, lddy, (const float*)x_bf16, ldx, mean, rstd, gamma, dres, lddres, gout_bf16, ldgout, accumulate,
                         dgamma, dbeta, colsum, ws, rows, d, gout_fp8, ldd8, scale_inv, stream, gin_bf16, ldgin, write_f32, true, t8);
}

extern "C" int sc_layernorm_bwd_reduce(const float* ws, int rows, int d, float* dgamma, float* dbeta, float* colsum,
                                       void* stream) {
    SC_CHECK(rows > 0 && d > 0 && ws != nullptr && dgamma != nullptr && dbeta != nullptr,
             "sc_layernorm_bwd_reduce: bad arguments rows=%d d=%d", rows, d);
    const int nblk = ln_nominal_blocks(rows, d);
    colvec_finalize_kernel<<<(3 * d + 63) / 64, 1024, 0, (hipStream_t)stream>>>(ws, nblk, 3, d, dgamma, dbeta, colsum);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" long long sc_colsum_ws_floats(int rows, int n) {
    int ny = (rows + 63) / 64;
    if (ny > 256) ny = 256;
    if (ny < 1) ny = 1;
    return (long long)ny * n;
}

extern "C" int sc_colsum_bf16(const void* x, long long ld, int rows, int n, float* out, float* ws, void* stream) {
    SC_CHECK(rows > 0 && n > 0 && (n % 4) == 0 && (ld % 4) == 0, "sc_colsum_bf16: bad shape rows=%d n=%d", rows, n);
    int ny = (rows + 63) / 64;
    if (ny > 256) ny = 256;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((n + 255) / 256, ny);
    colsum_kernel<<<grid, 256, 0, st>>>((const bf16*)x, ld, rows, n, ws);
    SC_LAUNCH_CHECK();
    colvec_finalize_kernel<<<(n + 63) / 64, 1024, 0, st>>>(ws, ny, 1, n, out, nullptr, nullptr);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_l2norm_fwd(const float* x, float* y, void* y_bf16, float* inv_norm, int rows, int d, void* stream) {
    SC_CHECK(rows > 0 && d > 0, "sc_l2norm_fwd: bad shape");
    l2norm_fwd_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, y, (bf16*)y_bf16, inv_norm, rows, d);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, void* dx_bf16, int rows, int d,
                             void* stream) {
    SC_CHECK(rows > 0 && d > 0, "sc_l2norm_bwd: bad shape");
    l2norm_bwd_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(dy, y, inv_norm, (bf16*)dx_bf16, rows, d);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cast_pad_bf16(const float* src, long long ld_src, void* dst, long long ld_dst, int rows, int cols,
                                int cols_pad, void* stream) {
    SC_CHECK(rows > 0 && cols > 0 && cols_pad >= cols && (cols_pad % 4) == 0 && (ld_dst % 4) == 0,
             "sc_cast_pad_bf16: bad shape rows=%d cols=%d pad=%d", rows, cols, cols_pad);
    const long long nq = (long long)rows * (cols_pad / 4);
    int blocks = (int)((nq + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    cast_pad_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(src, ld_src, (bf16*)dst, ld_dst, rows, cols, cols_pad);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cast_transpose_bf16(const float* src, void* dst, int rows, int cols, long long ld_dst, void* stream) {
    SC_CHECK(rows > 0 && cols > 0 && ld_dst >= rows, "sc_cast_transpose_bf16: bad shape");
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    cast_transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, (bf16*)dst, rows, cols, ld_dst);
    SC_LAUNCH_CHECK();
    return 0;
}

static int act_bf16_impl(const void* u, void* h, long long n, int act, void* stream) {
    SC_CHECK(n > 0 && (n % 8) == 0, "sc_gelu_bf16: element count must be a positive multiple of 8 (n=%lld)", n);
    const long long n8 = n / 8;
    long long blocks = (n8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    gelu_bf16_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>((const bf16*)u, (bf16*)h, n8, act);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_gelu_bf16(const void* u, void* h, long long n, void* stream) { return act_bf16_impl(u, h, n, 0, stream); }

extern "C" int sc_quick_gelu_bf16(const void* u, void* h, long long n, void* stream) { return act_bf16_impl(u, h, n, 1, stream); }
