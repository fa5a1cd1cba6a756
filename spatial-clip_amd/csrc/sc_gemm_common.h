// Shared pieces of the MFMA GEMM kernels (128x128 general kernel in sc_gemm.hip, 256x256 LDS-DMA kernel in
// sc_gemm256.hip, phase-interleaved kernels in sc_gemm8p.hip): argument block and the fused epilogue that drains a
// wave's 64x64 fp32 tile from LDS.
//
// Epilogue memory-level parallelism: the extra epilogue INPUT of a 64x64 sub-tile (fp32 residual rows, or the bf16
// pre-GELU tensor for GELU') is fetched by ONE burst of loads into registers (sc_epi_load) before the accumulators
// take their trip through LDS, and the registers are refilled for the next sub-tile as they are consumed -- a wave
// keeps 8-16 KiB in flight instead of waiting for HBM once per row group.
#pragma once
#include "sc_common.h"
#include "sc_kernels.h"

struct GemmArgs {
    const bf16* A;
    const bf16* B;
    int M, N, K;
    int lda, ldb;
    void* C;
    int ldc;
    void* C2;
    int ldc2;
    const float* bias;
    const float* res;
    int ldres;
    const bf16* aux;
    int ldaux;
    int splitk;
    int k_per_split;
    long long slab_stride;
    int ntm, ntn;
    int tile_offset;      // first logical tile of this launch (tail launches of sc_gemm256)
    float* colsum;        // TN + EPI_F32 only: [splitk][M] partial column sums of the At operand (bias gradient), or null
    const float* a_scale = nullptr;   // fp8 kernels only: per-row dequantisation factors of A [M] and B [N]
    const float* b_scale = nullptr;
    // fp8 kernels only -- e4m3 copy of the epilogue's bf16 output for the NEXT GEMM (GELU pair: h, A operand of c_proj;
    // GELU': dU, A operand of the c_fc data gradient) with ONE scale for the whole tensor, taken from the previous step's
    // maximum ("delayed scaling": a tile cannot know its rows' maxima, sc_fp8.hip): q8 = e4m3(value * *q8_scale), and the
    // maximum of |value| over the launch is max-reduced into q8_amax[64] (slot = workgroup & 63; non-negative floats
    // compared as integers) for sc_fp8_scale_update.  a_scale_scalar: a_scale points at ONE factor, not at M of them.
    unsigned char* q8 = nullptr;
    long long ldq8 = 0;
    const float* q8_scale = nullptr;
    float* q8_amax = nullptr;
    int a_scale_scalar = 0;
    // GELU-pair epilogues of gemm8p_kernel: table of (gelu'(u) << 16 | gelu(u)) bf16 pairs indexed by the bf16 bits of u (below)
    const unsigned* gelu_lut = nullptr;
    // activation of the GELU epilogues: 0 = exact-erf GELU (nn.GELU), 1 = QuickGELU x * sigmoid(1.702 x) (the ABI's
    // SC_EPI_QGELU_* values select the same template instance with act = 1; wave-uniform, so the choice is a scalar branch)
    int act = 0;
    // tile walk of gemm8p_kernel (round 5): 0 = row-major over the whole matrix (an XCD's consecutive tiles sweep all ntn
    // column tiles, i.e. ALL of B, every ~2.7 tile rows); Gc > 0 = an XCD's band of `band_rows` tile rows is walked column
    // group by column group (Gc column tiles: a B sub-panel that stays in the XCD's 4-MB L2 while the band's rows stream by)
    int col_group = 0;
    int band_rows = 0;
};

// (tm, tn) of remapped tile index idx under the column-group walk (splitk == 1)
SC_DEVICE void sc_tile_colgroup(int idx, const GemmArgs& g, int& tm, int& tn) {
    const int band_tiles = g.band_rows * g.ntn;
    const int band = idx / band_tiles;
    const int r = idx - band * band_tiles;
    const int row0 = band * g.band_rows;
    const int rows = min(g.band_rows, g.ntm - row0);
    const int nfg = g.ntn / g.col_group;                       // full groups; the last one may be narrower
    const int full = rows * g.col_group * nfg;
    if (r < full) {
        const int gsz = rows * g.col_group;
        const int grp = r / gsz, rr = r - grp * gsz;
        tm = row0 + rr / g.col_group;
        tn = grp * g.col_group + rr % g.col_group;
    } else {
        const int rem = g.ntn - nfg * g.col_group, r2 = r - full;
        tm = row0 + r2 / rem;
        tn = nfg * g.col_group + r2 % rem;
    }
}

// ---- GELU by table (round 4) ----
// The GELU epilogues evaluate gelu / gelu' of u AFTER u has been rounded to bf16, and store bf16: both are functions of 16 bits.
// A table indexed by those bits therefore returns exactly what the formula returns -- it is filled BY the formula
// (sc_gelu_lut_fill_kernel) -- at ~8 VALU + one LDS read per element instead of ~22 VALU + v_rcp + v_exp.  The table covers
// 2^-20 <= |u| < 32 (25 binades x 128 mantissas x 2 signs = 6400 entries, 25 KiB of LDS: the operand ring is dead by the time
// the non-persistent kernel's epilogue runs); a chunk with any value outside (zeros, |u| >= 32: rare) takes the formula.
constexpr int SC_GELU_LUT_LO = 107 << 7;                // bf16 bits of 2^-20
constexpr int SC_GELU_LUT_HALF = 25 * 128;              // entries per sign
constexpr int SC_GELU_LUT_N = 2 * SC_GELU_LUT_HALF;
const unsigned* sc_gelu_lut_device(hipStream_t st, int act = 0);     // sc_gemm8p.hip: the device copy (built on first use per device and activation), or null

constexpr int SC_EPI_LD = 68;  // floats per staged epilogue row (64 + 4 pad: conflict-free b128 writes and reads)

// exact-erf GELU to 1.5e-7 (Abramowitz-Stegun 7.1.26) sharing one exp between erf and the Gaussian pdf, arranged for the
// fewest VALU issue slots (the GELU epilogues are VALU-bound while the matrix pipe idles: profiles/r03_gelu_epilogue_decomposition.txt):
//   y = |x| sqrt(log2(e) / 2);  e = 2^(-y^2) = exp(-x^2 / 2);  t = 1 / (1 + p' y);  q(t) = poly(t) / 2
//   erf(|x| / sqrt 2) = 1 - 2 q e   =>   Phi(x) = 1/2 + copysign(1/2 - q e, x)          (the reference's 0.5 (1 + erf(x / sqrt 2)))
//                                        gelu(x) = x Phi(x);   gelu'(x) = Phi(x) + x e / sqrt(2 pi)
// 12 (gelu) / 13 (gelu') / 14 (both) plain VALU operations + v_rcp + v_exp per element; abs / neg ride on source modifiers.
// Round 4: ONE definition of Phi for every path (GELU-pair epilogues, sc_gelu_bf16, the GELU' epilogue, and the forward
// epilogue that stores gelu'(u) next to gelu(u)), so that the paths agree bit for bit by construction.
SC_DEVICE void sc_gelu_qe(float x, float& q, float& e) {
    const float y = fabsf(x) * 0.84932180028801904f;                       // sqrt(log2(e) / 2)
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.2727374808792225f, y, 1.0f));   // 0.3275911 / sqrt(log2 e)
    e = __builtin_amdgcn_exp2f(-(y * y));
    float r = __builtin_fmaf(t, 0.5307027145f, -0.7265760135f);            // poly / 2, Horner from the top
    r = __builtin_fmaf(r, t, 0.7107068705f);
    r = __builtin_fmaf(r, t, -0.142248368f);
    r = __builtin_fmaf(r, t, 0.127414796f);
    q = r * t;
}
SC_DEVICE float sc_gelu_phi(float x, float& e) {
    float q;
    sc_gelu_qe(x, q, e);
    return 0.5f + copysignf(__builtin_fmaf(-q, e, 0.5f), x);
}
SC_DEVICE float sc_gelu_fast(float x) {
    float e;
    return x * sc_gelu_phi(x, e);
}
SC_DEVICE float sc_gelu_grad_fast(float x) {
    float e;
    const float phi = sc_gelu_phi(x, e);
    return __builtin_fmaf(x * e, 0.3989422804014327f, phi);
}
// h = gelu(x) and g = gelu'(x) from one evaluation of q, e (the forward epilogue SC_EPI_GELU_GRAD_PAIR)
SC_DEVICE void sc_gelu_both(float x, float& h, float& g) {
    float e;
    const float phi = sc_gelu_phi(x, e);
    h = x * phi;
    g = __builtin_fmaf(x * e, 0.3989422804014327f, phi);
}
// the GELU' factor as every backward path applies it: rounded to bf16 first, because the default path reads it back
// from the bf16 tensor the forward epilogue stored (SC_EPI_BF16_MUL_AUX) -- recomputation mode must give the same bits
SC_DEVICE float sc_gelu_grad_bf16(float x) { return (float)(bf16)sc_gelu_grad_fast(x); }

// ---- QuickGELU (the OpenAI-pretrained towers: src/open_clip/transformer.py:32-35, model.py:142-145 act_layer = QuickGELU) ----
//   s = sigmoid(1.702 x) = 1 / (1 + 2^(-1.702 log2(e) x));  h = x s;  h' = s + 1.702 x s (1 - s) = s + [1.702 (1 - s)] h
// (in this order no intermediate overflows for any finite x: s -> 0 or 1 exactly at the ends, and 0 * h stays 0)
SC_DEVICE void sc_qgelu_both(float x, float& h, float& g) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));
    h = x * s;
    g = __builtin_fmaf(1.702f * (1.0f - s), h, s);
}
// activation-generic forms used by every epilogue: act is a kernel argument (wave-uniform)
SC_DEVICE float sc_act(float x, int act) {
    if (act) { float h, g; sc_qgelu_both(x, h, g); return h; }
    return sc_gelu_fast(x);
}
SC_DEVICE void sc_act_both(float x, int act, float& h, float& g) {
    if (act) sc_qgelu_both(x, h, g); else sc_gelu_both(x, h, g);
}
SC_DEVICE float sc_act_grad_bf16(float x, int act) {
    if (act) { float h, g; sc_qgelu_both(x, h, g); return (float)(bf16)g; }
    return sc_gelu_grad_bf16(x);
}

// ABI epilogue values 9..11 = the GELU epilogues with QuickGELU: same kernels, GemmArgs::act = 1
constexpr int sc_epi_act(int e) { return (e == SC_EPI_QGELU_PAIR || e == SC_EPI_BF16_DQGELU || e == SC_EPI_QGELU_GRAD_PAIR) ? 1 : 0; }
constexpr int sc_epi_base(int e) {
    return e == SC_EPI_QGELU_PAIR ? SC_EPI_GELU_PAIR : e == SC_EPI_BF16_DQGELU ? SC_EPI_BF16_DGELU
         : e == SC_EPI_QGELU_GRAD_PAIR ? SC_EPI_GELU_GRAD_PAIR : e;
}
constexpr bool sc_epi_gelu_fwd(int e) { return e == SC_EPI_GELU_PAIR || e == SC_EPI_GELU_GRAD_PAIR; }   // two bf16 outputs
constexpr bool sc_epi_aux_mul(int e) { return e == SC_EPI_BF16_DGELU || e == SC_EPI_BF16_MUL_AUX; }      // bf16 input tile, product

// eight values -> eight e4m3 bytes (value * s, saturating at +-448)
SC_DEVICE u32x2 sc_pack8_fp8(const float (&v)[8], float s) {
    float c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = fminf(fmaxf(v[k] * s, -448.f), 448.f);
    int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w0, true);
    int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], 0, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], w1, true);
    return (u32x2){(unsigned)w0, (unsigned)w1};
}
// launch-wide maximum of non-negative floats: one atomic per wave into one of 64 slots
SC_DEVICE void sc_amax_publish(float amax_lane, float* slots) {
    float a = amax_lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a = fmaxf(a, __shfl_xor(a, off, 64));
    if ((threadIdx.x & 63) == 0 && a > 0.f)
        atomicMax(reinterpret_cast<unsigned*>(slots) + (blockIdx.x & 63), __float_as_uint(a));
}

// Stage one MFMA accumulator block (swapped orientation: lane owns C[m = li][n = 4*lg .. +3]) into the wave's LDS tile
SC_DEVICE void sc_epi_put(float* ep, int row16, int col16, int li, int lg, f32x4 acc) {
    *reinterpret_cast<f32x4*>(ep + (row16 * 16 + li) * SC_EPI_LD + col16 * 16 + lg * 4) = acc;
}

template <int EPI>
struct EpiRegs {
    static constexpr bool kRes = (EPI == SC_EPI_F32_BIAS_RES);
    static constexpr bool kAux = (sc_epi_aux_mul(EPI) || EPI == SC_EPI_BF16_BIAS_RES);    // a bf16 input tile
    f32x4 r[kRes ? 16 : 1];
    bf16x8 a[kAux ? 8 : 1];
};

// issue the loads of the epilogue input of the 64x64 sub-tile at (gm0, gn0)
template <int EPI>
SC_DEVICE void sc_epi_load(EpiRegs<EPI>& e, int gm0, int gn0, int lane, const GemmArgs& g, int mrows = 64) {
    const int mlim = min(g.M, gm0 + mrows);
    if (EPI == SC_EPI_F32_BIAS_RES) {
        const int gn = gn0 + (lane & 15) * 4;
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
            const int gm = gm0 + ps * 4 + (lane >> 4);
            e.r[ps] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (g.res && gm < mlim && gn < g.N) e.r[ps] = *reinterpret_cast<const f32x4*>(g.res + (size_t)gm * g.ldres + gn);
        }
    }
    if (sc_epi_aux_mul(EPI)) {
        const int gn = gn0 + (lane & 7) * 8;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int gm = gm0 + ps * 8 + (lane >> 3);
            if (gm < mlim && gn < g.N) e.a[ps] = *reinterpret_cast<const bf16x8*>(g.aux + (size_t)gm * g.ldaux + gn);
        }
    }
    if (EPI == SC_EPI_BF16_BIAS_RES) {            // the residual stream in bf16: g.res points at bf16 data, ldres in elements
        const int gn = gn0 + (lane & 7) * 8;
        const bf16* res = reinterpret_cast<const bf16*>(g.res);
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int gm = gm0 + ps * 8 + (lane >> 3);
#pragma unroll
            for (int k = 0; k < 8; ++k) e.a[ps][k] = (bf16)0.0f;
            if (res && gm < mlim && gn < g.N) e.a[ps] = *reinterpret_cast<const bf16x8*>(res + (size_t)gm * g.ldres + gn);
        }
    }
}

// Drain a wave-private 64x64 fp32 tile (row stride SC_EPI_LD) to global memory with full-row-segment accesses.
// If next_gm0 >= 0 the input registers are refilled for the sub-tile at (next_gm0, gn0) as they are consumed.
// hi_col_skip: the tile's columns 32..63 sit that many columns further right in C (two 32-column blocks; fp32 only).
template <int EPI, bool Q8 = false>
SC_DEVICE void sc_epilogue_store(const float* ep, EpiRegs<EPI>& e, int gm0, int gn0, int lane, const GemmArgs& g, int z,
                                 int next_gm0 = -1, int mrows = 64, int hi_col_skip = 0, float* amax_lane = nullptr) {
    const int mlim = min(g.M, gm0 + mrows);
    if (EPI == SC_EPI_F32 || EPI == SC_EPI_F32_BIAS_RES) {
        float* C = reinterpret_cast<float*>(g.C) + (size_t)z * g.slab_stride;
        const int col = (lane & 15) * 4;
        const int gn = gn0 + col + ((col >> 5) ? hi_col_skip : 0);
        f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (EPI == SC_EPI_F32_BIAS_RES && g.bias && gn < g.N) bv = *reinterpret_cast<const f32x4*>(g.bias + gn);
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
            const int row = ps * 4 + (lane >> 4);
            const int gm = gm0 + row;
            f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * SC_EPI_LD + col);
            if (EPI == SC_EPI_F32_BIAS_RES) {
                v += bv + e.r[ps];
                if (next_gm0 >= 0) {
                    const int gm2 = next_gm0 + row;
                    e.r[ps] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (g.res && gm2 < g.M && gn < g.N)
                        e.r[ps] = *reinterpret_cast<const f32x4*>(g.res + (size_t)gm2 * g.ldres + gn);
                }
            }
            if (gm < mlim && gn < g.N) *reinterpret_cast<f32x4*>(C + (size_t)gm * g.ldc + gn) = v;
        }
    } else {
        bf16* C = reinterpret_cast<bf16*>(g.C);
        const int col = (lane & 7) * 8;
        const int gn = gn0 + col;
        float bv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) bv[k] = 0.f;
        if ((EPI == SC_EPI_BF16_BIAS || sc_epi_gelu_fwd(EPI) || EPI == SC_EPI_BF16_BIAS_RES) && g.bias && gn < g.N) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias + gn);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(g.bias + gn + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { bv[k] = b0[k]; bv[4 + k] = b1[k]; }
        }
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 8 + (lane >> 3);
            const int gm = gm0 + row;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + row * SC_EPI_LD + col);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(ep + row * SC_EPI_LD + col + 4);
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = v0[k] + bv[k]; v[4 + k] = v1[k] + bv[4 + k]; }
            if (sc_epi_aux_mul(EPI)) {
#pragma unroll
                for (int k = 0; k < 8; ++k)          // aux = the pre-GELU tensor u (DGELU) or the stored factor gelu'(u) (MUL_AUX)
                    v[k] *= (EPI == SC_EPI_BF16_DGELU) ? sc_act_grad_bf16((float)e.a[ps][k], g.act) : (float)e.a[ps][k];
                if (next_gm0 >= 0) {
                    const int gm2 = next_gm0 + row;
                    if (gm2 < g.M && gn < g.N)
                        e.a[ps] = *reinterpret_cast<const bf16x8*>(g.aux + (size_t)gm2 * g.ldaux + gn);
                }
            }
            if (EPI == SC_EPI_BF16_BIAS_RES) {      // x_new = bf16(acc + bias + x): one rounding, from the fp32 staging tile
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += (float)e.a[ps][k];
                if (next_gm0 >= 0) {
                    const int gm2 = next_gm0 + row;
                    const bf16* res = reinterpret_cast<const bf16*>(g.res);
#pragma unroll
                    for (int k = 0; k < 8; ++k) e.a[ps][k] = (bf16)0.0f;
                    if (res && gm2 < g.M && gn < g.N)
                        e.a[ps] = *reinterpret_cast<const bf16x8*>(res + (size_t)gm2 * g.ldres + gn);
                }
            }
            if (gm < mlim && gn < g.N) {
                bf16x8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (bf16)v[k];
                if (EPI != SC_EPI_GELU_GRAD_PAIR) *reinterpret_cast<bf16x8*>(C + (size_t)gm * g.ldc + gn) = o;
                if (Q8 && sc_epi_aux_mul(EPI) && g.q8) {          // e4m3 copy of dU for the c_fc data-gradient GEMM
                    float r[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { r[k] = (float)o[k]; *amax_lane = fmaxf(*amax_lane, fabsf(r[k])); }
                    *reinterpret_cast<u32x2*>(g.q8 + (size_t)gm * g.ldq8 + gn) = sc_pack8_fp8(r, *g.q8_scale);
                }
                if (EPI == SC_EPI_GELU_PAIR) {
                    bf16x8 h;
#pragma unroll
                    for (int k = 0; k < 8; ++k) h[k] = (bf16)sc_act((float)o[k], g.act);
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(g.C2) + (size_t)gm * g.ldc2 + gn) = h;
                }
                if (EPI == SC_EPI_GELU_GRAD_PAIR) {          // C = gelu'(u), C2 = gelu(u), u = bf16(acc + bias) never stored
                    bf16x8 h, gd;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float hv, gv;
                        sc_act_both((float)o[k], g.act, hv, gv);
                        h[k] = (bf16)hv;
                        gd[k] = (bf16)gv;
                    }
                    *reinterpret_cast<bf16x8*>(C + (size_t)gm * g.ldc + gn) = gd;
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(g.C2) + (size_t)gm * g.ldc2 + gn) = h;
                }
            }
        }
    }
}

// 256x256 LDS-DMA kernel (sc_gemm256.hip): returns 1 if it took the problem, 0 if not eligible, <0 on error
int sc_gemm256_try(int mode, int epi, GemmArgs& g, int splitk_req, float* slabs, float* c_final, hipStream_t st);
// 256x256x64 phase-interleaved (ping-pong) kernel, NT only (sc_gemm8p.hip)
int sc_gemm8p_try(int mode, int epi, GemmArgs& g, int splitk_req, float* slabs, hipStream_t st);
// grouped TN launch of the same kernel (several weight gradients over one token axis): plan the common split-K, then launch
int sc_gemm8p_tn_group_plan(const GemmArgs* g, int n, int splitk_req, int* splitk_out, int* k_per_split);
int sc_gemm8p_tn_group_launch(const GemmArgs* g, int n, hipStream_t st);
// fp8 (e4m3) TN weight gradient (per-tensor scales; lda / ldb in bytes); 1 = launched, 0 = shape outside its range
int sc_gemm8p_tn_fp8(GemmArgs& g, int splitk_req, float* slabs, hipStream_t st);
// fp8 (e4m3) NT variant of the same kernel; g.K / lda / ldb in 2-byte units, g.a_scale / g.b_scale set
int sc_gemm8p_fp8(int epi, GemmArgs& g, hipStream_t st);
