// Shared pieces of the two MFMA GEMM kernels (128x128 general kernel in sc_gemm.hip, 256x256 LDS-DMA kernel in
// sc_gemm256.hip): argument block and the fused epilogue that drains a wave's 64x64 fp32 tile from LDS.
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
};

constexpr int SC_EPI_LD = 68;  // floats per staged epilogue row (64 + 4 pad: conflict-free b128 writes and reads)

// Stage one MFMA accumulator block (swapped orientation: lane owns C[m = li][n = 4*lg .. +3]) into the wave's LDS tile
SC_DEVICE void sc_epi_put(float* ep, int row16, int col16, int li, int lg, f32x4 acc) {
    *reinterpret_cast<f32x4*>(ep + (row16 * 16 + li) * SC_EPI_LD + col16 * 16 + lg * 4) = acc;
}

// Drain a wave-private 64x64 fp32 tile (row stride SC_EPI_LD) to global memory with full-row-segment accesses.
template <int EPI>
SC_DEVICE void sc_epilogue_store(const float* ep, int gm0, int gn0, int lane, const GemmArgs& g, int z) {
    if (EPI == SC_EPI_F32 || EPI == SC_EPI_F32_BIAS_RES) {
        float* C = reinterpret_cast<float*>(g.C) + (size_t)z * g.slab_stride;
        const int col = (lane & 15) * 4;
        const int gn = gn0 + col;
        f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (EPI == SC_EPI_F32_BIAS_RES && g.bias && gn < g.N) bv = *reinterpret_cast<const f32x4*>(g.bias + gn);
#pragma unroll 4
        for (int ps = 0; ps < 16; ++ps) {
            const int row = ps * 4 + (lane >> 4);
            const int gm = gm0 + row;
            if (gm < g.M && gn < g.N) {
                f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * SC_EPI_LD + col);
                if (EPI == SC_EPI_F32_BIAS_RES) {
                    v += bv;
                    if (g.res) v += *reinterpret_cast<const f32x4*>(g.res + (size_t)gm * g.ldres + gn);
                }
                *reinterpret_cast<f32x4*>(C + (size_t)gm * g.ldc + gn) = v;
            }
        }
    } else {
        bf16* C = reinterpret_cast<bf16*>(g.C);
        const int col = (lane & 7) * 8;
        const int gn = gn0 + col;
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = 0.f;
        if ((EPI == SC_EPI_BF16_BIAS || EPI == SC_EPI_GELU_PAIR) && g.bias && gn < g.N) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias + gn);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(g.bias + gn + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
        }
#pragma unroll 4
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 8 + (lane >> 3);
            const int gm = gm0 + row;
            if (gm < g.M && gn < g.N) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + row * SC_EPI_LD + col);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(ep + row * SC_EPI_LD + col + 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = v0[e] + bv[e]; v[4 + e] = v1[e] + bv[4 + e]; }
                if (EPI == SC_EPI_BF16_DGELU) {
                    const bf16x8 u = *reinterpret_cast<const bf16x8*>(g.aux + (size_t)gm * g.ldaux + gn);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= sc_gelu_grad((float)u[e]);
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                *reinterpret_cast<bf16x8*>(C + (size_t)gm * g.ldc + gn) = o;
                if (EPI == SC_EPI_GELU_PAIR) {
                    bf16x8 h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (bf16)sc_gelu((float)o[e]);
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(g.C2) + (size_t)gm * g.ldc2 + gn) = h;
                }
            }
        }
    }
}

// 256x256 LDS-DMA kernel (sc_gemm256.hip): returns 1 if it took the problem, 0 if not eligible, <0 on error
int sc_gemm256_try(int mode, int epi, GemmArgs& g, int splitk_req, float* slabs, float* c_final, hipStream_t st);
