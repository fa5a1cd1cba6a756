// bf16 MFMA GEMM for gfx950 with fused epilogues.  Two operand layouts:
//   NT : C[M,N] = A[M,K] . B[N,K]^T      (both operands K-contiguous: forward linears and dgrads,
//                                          the latter against a transposed bf16 weight copy)
//   TN : C[M,N] = At[K,M]^T . Bt[K,N]     (both operands reduction-major: weight gradients
//                                          dW = dY^T X straight from the row-major activations,
//                                          fragments fetched with ds_read_b64_tr_b16)
// 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles, fp32 accumulate.
// Register-staged double buffering (global -> VGPR -> swizzled LDS), one barrier per K tile.
// Edge handling: buffer loads return 0 beyond the operand (outer dims), stores are guarded.
// The accumulators are produced "swapped" (D = B.A^T) so that every lane owns 4 consecutive columns
// of C; they are then bounced through LDS once so that global stores / residual loads are full
// 128..256-byte row segments.
#include "sc_gemm_common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;          // 16 KiB per operand tile
constexpr int EPI_LD = SC_EPI_LD;
constexpr int LDS_BYTES = 4 * 64 * EPI_LD * 4;     // 69632 >= 4 * TILE_BYTES

template <int MODE, bool KTAIL>
SC_DEVICE void stage_load(u32x4 (&ra)[4], u32x4 (&rb)[4], __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB,
                          int lda, int ldb, int k0, int t, int kend) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = p * 256 + t;
        if (MODE == SC_GEMM_NT) {
            const int row = c >> 3, kc = c & 7;
            ra[p] = sc_buf_load16(rsA, (uint32_t)(row * lda + k0 + kc * 8) * 2u);
            rb[p] = sc_buf_load16(rsB, (uint32_t)(row * ldb + k0 + kc * 8) * 2u);
            if (KTAIL && k0 + kc * 8 >= kend) {   // K not a multiple of 64: zero the chunks past the row end
                ra[p] = (u32x4){0u, 0u, 0u, 0u};
                rb[p] = (u32x4){0u, 0u, 0u, 0u};
            }
        } else {
            const int krow = c >> 4, mc = c & 15;
            ra[p] = sc_buf_load16(rsA, (uint32_t)((k0 + krow) * lda + mc * 8) * 2u);
            rb[p] = sc_buf_load16(rsB, (uint32_t)((k0 + krow) * ldb + mc * 8) * 2u);
        }
    }
}

template <int MODE>
SC_DEVICE void stage_store(const u32x4 (&ra)[4], const u32x4 (&rb)[4], char* sA, char* sB, int t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = p * 256 + t;
        int off;
        if (MODE == SC_GEMM_NT) {
            const int row = c >> 3, kc = c & 7;
            off = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
        } else {
            const int krow = c >> 4, mc = c & 15;
            const int s = (krow & 3) | (((krow >> 3) & 1) << 2);
            off = krow * 256 + ((((mc >> 1) ^ s)) << 5) + ((mc & 1) << 4);
        }
        *reinterpret_cast<u32x4*>(sA + off) = ra[p];
        *reinterpret_cast<u32x4*>(sB + off) = rb[p];
    }
}

template <int MODE, int EPI, bool KTAIL>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;

    int idx = sc_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = idx % g.ntn;
    idx /= g.ntn;
    const int tm = idx % g.ntm;
    const int z = idx / g.ntm;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nt = (kend - kbeg + BK - 1) / BK;

    __amdgpu_buffer_rsrc_t rsA, rsB;
    if (MODE == SC_GEMM_NT) {
        rsA = sc_make_rsrc(g.A + (size_t)m0 * g.lda, sc_clamp_bytes((uint64_t)max(g.M - m0, 0) * g.lda * 2));
        rsB = sc_make_rsrc(g.B + (size_t)n0 * g.ldb, sc_clamp_bytes((uint64_t)max(g.N - n0, 0) * g.ldb * 2));
    } else {
        // base at (kbeg, m0); everything past the last reduction row is out of range -> 0
        rsA = sc_make_rsrc(g.A + (size_t)kbeg * g.lda + m0,
                           sc_clamp_bytes(((uint64_t)(g.K - kbeg) * g.lda - m0) * 2));
        rsB = sc_make_rsrc(g.B + (size_t)kbeg * g.ldb + n0,
                           sc_clamp_bytes(((uint64_t)(g.K - kbeg) * g.ldb - n0) * 2));
    }
    const int kb = (MODE == SC_GEMM_NT) ? kbeg : 0;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x4 ra[4], rb[4];
    if (nt > 0) {
        stage_load<MODE, KTAIL>(ra, rb, rsA, rsB, g.lda, g.ldb, kb, t, kend);
        stage_store<MODE>(ra, rb, smem, smem + TILE_BYTES, t);
    }
    __syncthreads();

    for (int it = 0; it < nt; ++it) {
        const int cur = it & 1;
        char* sA = smem + cur * 2 * TILE_BYTES;
        char* sB = sA + TILE_BYTES;
        const bool more = (it + 1 < nt);
        if (more) stage_load<MODE, KTAIL>(ra, rb, rsA, rsB, g.lda, g.ldb, kb + (it + 1) * BK, t, kend);

#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
            if (MODE == SC_GEMM_NT) {
                const int sw = (li >> 1) & 7;
                const int coff = ((kk * 4 + lg) ^ sw) << 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rowa = wm * 64 + i * 16 + li;
                    const int rowb = wn * 64 + i * 16 + li;
                    af[i] = *reinterpret_cast<const bf16x8*>(sA + rowa * 128 + coff);
                    bfr[i] = *reinterpret_cast<const bf16x8*>(sB + rowb * 128 + coff);
                }
            } else {
                const int q = li >> 2, p = li & 3;
                const int krow = kk * 32 + lg * 8 + q;
                const int s = q | ((lg & 1) << 2);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ca = ((wm * 4 + i) ^ s) << 5;
                    const int cb = ((wn * 4 + i) ^ s) << 5;
                    const char* pa = sA + krow * 256 + ca + p * 8;
                    const char* pb = sB + krow * 256 + cb + p * 8;
                    af[i] = sc_cat(sc_lds_tr16(pa), sc_lds_tr16(pa + 4 * 256));
                    bfr[i] = sc_cat(sc_lds_tr16(pb), sc_lds_tr16(pb + 4 * 256));
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = sc_mfma16(bfr[j], af[i], acc[i][j]);
        }
        if (more) {
            char* nA = smem + (cur ^ 1) * 2 * TILE_BYTES;
            stage_store<MODE>(ra, rb, nA, nA + TILE_BYTES, t);
        }
        __syncthreads();
    }

    // ---------------- epilogue: bounce the wave's 64x64 tile through LDS ----------------
    EpiRegs<EPI> er;
    sc_epi_load<EPI>(er, m0 + wm * 64, n0 + wn * 64, lane, g);
    float* ep = reinterpret_cast<float*>(smem) + wave * 64 * EPI_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(ep + (i * 16 + li) * EPI_LD + j * 16 + lg * 4) = acc[i][j];
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): a wave only reads back its own region
    __builtin_amdgcn_wave_barrier();

    sc_epilogue_store<EPI>(ep, er, m0 + wm * 64, n0 + wn * 64, lane, g, z);
}

__global__ void reduce_slabs_kernel(float* __restrict__ out, const float* __restrict__ slabs, int nslab,
                                    long long slab_stride, long long n4, float* __restrict__ cs_out,
                                    const float* __restrict__ cs_part, int cs_n) {
    if (slabs != nullptr) {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
             i += (long long)gridDim.x * blockDim.x) {
            f32x4 s = reinterpret_cast<const f32x4*>(slabs)[i];
            for (int z = 1; z < nslab; ++z) s += reinterpret_cast<const f32x4*>(slabs + z * slab_stride)[i];
            reinterpret_cast<f32x4*>(out)[i] = s;
        }
    }
    if (cs_out != nullptr) {        // partial column sums of the fused bias gradient: [nslab][cs_n] -> [cs_n]
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cs_n; i += gridDim.x * blockDim.x) {
            float s = 0.f;
            for (int z = 0; z < nslab; ++z) s += cs_part[(long long)z * cs_n + i];
            cs_out[i] = s;
        }
    }
}

// reduce_slabs_kernel for several (output, slabs) pairs in one launch: pair p owns blocks [first[p], first[p + 1])
struct ReduceGroup {
    float* out[SC_WGRAD_GROUP_MAX];
    const float* slabs[SC_WGRAD_GROUP_MAX];        // null: the GEMM wrote `out` itself (split-K 1)
    long long slab_stride[SC_WGRAD_GROUP_MAX];
    long long n4[SC_WGRAD_GROUP_MAX];
    float* cs_out[SC_WGRAD_GROUP_MAX];             // bias gradient, or null
    const float* cs_part[SC_WGRAD_GROUP_MAX];
    int cs_n[SC_WGRAD_GROUP_MAX];
    int first[SC_WGRAD_GROUP_MAX + 1];
    int n, nslab;
};
__global__ void reduce_slabs_group_kernel(const ReduceGroup r) {
    int p = 0;
    while (p + 1 < r.n && (int)blockIdx.x >= r.first[p + 1]) ++p;
    const long long b = blockIdx.x - r.first[p], nb = r.first[p + 1] - r.first[p];
    const float* slabs = r.slabs[p];
    if (slabs != nullptr) {
        float* out = r.out[p];
        const long long stride = r.slab_stride[p], n4 = r.n4[p];
        for (long long i = b * blockDim.x + threadIdx.x; i < n4; i += nb * blockDim.x) {
            f32x4 s = reinterpret_cast<const f32x4*>(slabs)[i];
            for (int z = 1; z < r.nslab; ++z) s += reinterpret_cast<const f32x4*>(slabs + z * stride)[i];
            reinterpret_cast<f32x4*>(out)[i] = s;
        }
    }
    if (r.cs_out[p] != nullptr) {
        const float* part = r.cs_part[p];
        const int n = r.cs_n[p];
        for (long long i = b * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) {
            float s = 0.f;
            for (int z = 0; z < r.nslab; ++z) s += part[(long long)z * n + i];
            r.cs_out[p][i] = s;
        }
    }
}

template <int MODE, int EPI, bool KTAIL>
int launch1(const GemmArgs& g, int nblocks, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<MODE, EPI, KTAIL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done = true;
    }
    gemm_kernel<MODE, EPI, KTAIL><<<nblocks, 256, LDS_BYTES, st>>>(g);
    SC_LAUNCH_CHECK();
    return 0;
}
template <int MODE, int EPI>
int launch(const GemmArgs& g, int nblocks, hipStream_t st) {
    if (MODE == SC_GEMM_NT && (g.K % BK) != 0) return launch1<MODE, EPI, true>(g, nblocks, st);
    return launch1<MODE, EPI, false>(g, nblocks, st);
}

}  // namespace

extern "C" int sc_gemm_bf16(int mode, int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                            void* C, int ldc, void* C2, int ldc2, const float* bias, const void* res, int ldres,
                            const void* aux, int ldaux, int splitk, float* slabs, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int act = sc_epi_act(epi);          // SC_EPI_QGELU_*: the erf twin's kernel instance with the activation flag set
    epi = sc_epi_base(epi);
    SC_CHECK(mode == SC_GEMM_NT || mode == SC_GEMM_TN, "sc_gemm_bf16: bad mode %d", mode);
    SC_CHECK(M > 0 && N > 0 && K > 0, "sc_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
    const bool f32out = (epi == SC_EPI_F32 || epi == SC_EPI_F32_BIAS_RES);
    SC_CHECK((N % (f32out ? 4 : 8)) == 0 && (ldc % 4) == 0,
             "sc_gemm_bf16: N (%d) must be a multiple of %d, ldc (%d) of 4", N, f32out ? 4 : 8, ldc);
    SC_CHECK((lda % 8) == 0 && (ldb % 8) == 0, "sc_gemm_bf16: lda/ldb (%d,%d) must be multiples of 8", lda, ldb);
    SC_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0,
             "sc_gemm_bf16: operands must be 16-byte aligned");
    if (mode == SC_GEMM_NT) SC_CHECK((K % 8) == 0, "sc_gemm_bf16: NT needs K %% 8 == 0 (K=%d)", K);
    if (splitk < 1) splitk = 1;
    SC_CHECK(splitk == 1 || (epi == SC_EPI_F32 && slabs != nullptr), "sc_gemm_bf16: split-K needs EPI_F32 + slabs");
    SC_CHECK(!sc_epi_aux_mul(epi) || (aux != nullptr && (ldaux % 8) == 0), "sc_gemm_bf16: epilogue %d needs aux (ldaux %% 8 == 0)", epi);
    SC_CHECK(!sc_epi_gelu_fwd(epi) || (C2 != nullptr && (ldc2 % 8) == 0), "sc_gemm_bf16: epilogue %d needs the second output C2", epi);
    GemmArgs g;
    g.A = (const bf16*)A; g.B = (const bf16*)B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.C = C; g.ldc = ldc; g.C2 = C2; g.ldc2 = ldc2; g.bias = bias; g.res = (const float*)res; g.ldres = ldres;
    g.aux = (const bf16*)aux; g.ldaux = ldaux;
    g.colsum = nullptr; g.tile_offset = 0;
    g.act = act;
    // kernel choice: 256x256 phase-interleaved kernel (NT) -> 256x256 two-stage LDS-DMA kernel (TN, and NT when
    // pinned) -> 128x128 general kernel.  SC_GEMM_FORCE = 128 | 256 pins one kernel for A/B benchmarking.
    static const char* force = getenv("SC_GEMM_FORCE");
    int took = 0;
    if (!force) took = sc_gemm8p_try(mode, epi, g, splitk, slabs, st);
    if (took == 0 && (!force || force[0] == '2')) {
        g.C = C;
        took = sc_gemm256_try(mode, epi, g, splitk, slabs, (float*)C, st);
    }
    if (took < 0) return took;
    if (took == 1) {
        splitk = g.splitk;
        if (splitk > 1) {
            const long long n4 = (long long)M * N / 4;
            int blocks = (int)((n4 + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            reduce_slabs_kernel<<<blocks, 256, 0, st>>>((float*)C, slabs, splitk, g.slab_stride, n4, nullptr, nullptr, 0);
            SC_LAUNCH_CHECK();
        }
        return 0;
    }
    g.C = C;
    g.ntm = (M + BM - 1) / BM; g.ntn = (N + BN - 1) / BN;
    int ktiles = (K + BK - 1) / BK;
    if (splitk > ktiles) splitk = ktiles;
    int tiles_per = (ktiles + splitk - 1) / splitk;
    splitk = (ktiles + tiles_per - 1) / tiles_per;
    g.splitk = splitk; g.k_per_split = tiles_per * BK;
    g.slab_stride = 0;
    if (splitk > 1) {
        SC_CHECK(ldc == N, "sc_gemm_bf16: split-K needs a dense C (ldc == N)");
        g.C = slabs; g.slab_stride = (long long)M * N;
    }
    const int nblocks = g.ntm * g.ntn * splitk;
    int rc = -1;
#define SC_CASE(MODE, EPI) \
    if (mode == MODE && epi == EPI) rc = launch<MODE, EPI>(g, nblocks, st);
    SC_CASE(SC_GEMM_NT, SC_EPI_BF16)
    SC_CASE(SC_GEMM_NT, SC_EPI_BF16_BIAS)
    SC_CASE(SC_GEMM_NT, SC_EPI_F32_BIAS_RES)
    SC_CASE(SC_GEMM_NT, SC_EPI_GELU_PAIR)
    SC_CASE(SC_GEMM_NT, SC_EPI_BF16_DGELU)
    SC_CASE(SC_GEMM_NT, SC_EPI_BF16_BIAS_RES)
    SC_CASE(SC_GEMM_NT, SC_EPI_GELU_GRAD_PAIR)
    SC_CASE(SC_GEMM_NT, SC_EPI_BF16_MUL_AUX)
    SC_CASE(SC_GEMM_NT, SC_EPI_F32)
    SC_CASE(SC_GEMM_TN, SC_EPI_F32)
    SC_CASE(SC_GEMM_TN, SC_EPI_BF16)
#undef SC_CASE
    SC_CHECK(rc != -1 || false, "sc_gemm_bf16: unsupported (mode=%d, epi=%d)", mode, epi);
    if (rc != 0) return rc;
    if (splitk > 1) {
        const long long n4 = (long long)M * N / 4;
        int blocks = (int)((n4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        reduce_slabs_kernel<<<blocks, 256, 0, st>>>((float*)C, slabs, splitk, g.slab_stride, n4, nullptr, nullptr, 0);
        SC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" long long sc_gemm_slab_floats(int M, int N, int K, int splitk) {
    int ktiles = (K + BK - 1) / BK;
    if (splitk < 1) splitk = 1;
    if (splitk > ktiles) splitk = ktiles;
    return splitk > 1 ? (long long)splitk * M * N : 0;
}


// Weight gradient + bias gradient of one Linear in one pass: dW[M,N] = dY[K,M]^T . X[K,N] (fp32) and
// dbias[M] = column sums of dY, fused into the TN kernel when the 256x256 kernel takes the problem.
extern "C" long long sc_gemm_wgrad_ws_floats(int M, int N, int K, int splitk) {
    long long a = sc_gemm_slab_floats(M, N, K, splitk);
    int sk = splitk < 1 ? 1 : splitk;
    long long b = (long long)sk * M + sc_colsum_ws_floats(K, M);
    return a + b + 64;
}

extern "C" int sc_gemm_wgrad_bias(const void* dY, int lddy, const void* X, int ldx, int M, int N, int K, float* dW,
                                  int ldw, float* dbias, int splitk, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SC_CHECK(M > 0 && N > 0 && K > 0 && (M % 4) == 0 && (N % 4) == 0, "sc_gemm_wgrad_bias: bad shape M=%d N=%d K=%d", M, N, K);
    SC_CHECK(ws != nullptr && dbias != nullptr, "sc_gemm_wgrad_bias: workspace / dbias required");
    static const char* force = getenv("SC_GEMM_FORCE");
    if (splitk < 1) splitk = 1;
    const long long slab_floats = sc_gemm_slab_floats(M, N, K, splitk);
    float* slabs = ws;
    float* cs_part = ws + ((slab_floats + 15) / 16) * 16;
    GemmArgs g;
    g.A = (const bf16*)dY; g.B = (const bf16*)X; g.M = M; g.N = N; g.K = K; g.lda = lddy; g.ldb = ldx;
    g.C = dW; g.ldc = ldw; g.C2 = nullptr; g.ldc2 = 0; g.bias = nullptr; g.res = nullptr; g.ldres = 0;
    g.aux = nullptr; g.ldaux = 0; g.tile_offset = 0;
    g.colsum = cs_part;
    int took = 0;
    if (!force) took = sc_gemm8p_try(SC_GEMM_TN, SC_EPI_F32, g, splitk, slab_floats ? slabs : nullptr, st);
    if (took == 0 && (!force || force[0] == '2')) {
        g.C = dW;
        took = sc_gemm256_try(SC_GEMM_TN, SC_EPI_F32, g, splitk, slab_floats ? slabs : nullptr, dW, st);
    }
    if (took < 0) return took;
    if (took == 1) {
        const long long n4 = (long long)M * N / 4;
        int blocks = (int)((n4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < (M + 255) / 256) blocks = (M + 255) / 256;
        reduce_slabs_kernel<<<blocks, 256, 0, st>>>(dW, g.splitk > 1 ? slabs : nullptr, g.splitk, g.slab_stride, n4, dbias,
                                                    cs_part, M);
        SC_LAUNCH_CHECK();
        return 0;
    }
    // general path: separate GEMM and column-sum kernels
    int rc = sc_gemm_bf16(SC_GEMM_TN, SC_EPI_F32, dY, lddy, X, ldx, M, N, K, dW, ldw, nullptr, 0, nullptr, nullptr, 0,
                          nullptr, 0, slab_floats ? splitk : 1, slab_floats ? slabs : nullptr, stream);
    if (rc != 0) return rc;
    return sc_colsum_bf16(dY, lddy, K, M, dbias, cs_part, stream);
}


// ---- e4m3 weight (+ bias) gradient: dW[M,N] = s_dy s_x sum_k dY8[k,m] X8[k,n], per-tensor scales (include/spatial_clip_hip.h) ----
extern "C" int sc_gemm_wgrad_fp8(const void* dY8, long long lddy, const float* dy_scale_inv, const void* X8, long long ldx,
                                 const float* x_scale_inv, int M, int N, int K, float* dW, int ldw, float* dbias, int splitk,
                                 float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SC_CHECK(M > 0 && N > 0 && K > 0 && dY8 && X8 && dW && ws && dy_scale_inv && x_scale_inv, "sc_gemm_wgrad_fp8: null / empty argument");
    SC_CHECK((lddy % 16) == 0 && (ldx % 16) == 0 && ((uintptr_t)dY8 % 16) == 0 && ((uintptr_t)X8 % 16) == 0 && ((uintptr_t)dW % 16) == 0,
             "sc_gemm_wgrad_fp8: operand rows must be 16-byte aligned (lddy=%lld ldx=%lld)", lddy, ldx);
    SC_CHECK(ldw == N, "sc_gemm_wgrad_fp8: dW must be dense (ldw == N)");
    if (splitk < 1) splitk = 1;
    const long long slab_floats = sc_gemm_slab_floats(M, N, K, splitk);
    float* slabs = ws;
    float* cs_part = ws + ((slab_floats + 15) / 16) * 16;
    GemmArgs g;
    g.A = (const bf16*)dY8; g.B = (const bf16*)X8; g.M = M; g.N = N; g.K = K; g.lda = (int)lddy; g.ldb = (int)ldx;
    g.C = dW; g.ldc = ldw; g.C2 = nullptr; g.ldc2 = 0; g.bias = nullptr; g.res = nullptr; g.ldres = 0;
    g.aux = nullptr; g.ldaux = 0; g.tile_offset = 0;
    g.colsum = dbias ? cs_part : nullptr;
    g.a_scale = dy_scale_inv; g.b_scale = x_scale_inv;
    const int took = sc_gemm8p_tn_fp8(g, splitk, slab_floats ? slabs : nullptr, st);
    if (took < 0) return took;
    SC_CHECK(took == 1, "sc_gemm_wgrad_fp8: shape outside the kernel's range (M=%d N=%d K=%d: M >= 256, N >= 192, M %% 16 == N %% 16 == 0, K %% 128 == 0)", M, N, K);
    if (g.splitk > 1 || dbias != nullptr) {
        const long long n4 = (long long)M * N / 4;
        int blocks = (int)((n4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < (M + 255) / 256) blocks = (M + 255) / 256;
        reduce_slabs_kernel<<<blocks, 256, 0, st>>>(dW, g.splitk > 1 ? slabs : nullptr, g.splitk, g.slab_stride, n4, dbias, cs_part,
                                                    dbias ? M : 0);
        SC_LAUNCH_CHECK();
    }
    return 0;
}

// ---- several weight (+ bias) gradients over one token axis in one launch (include/spatial_clip_hip.h) ----
static long long wgrad_group_layout(const sc_wgrad_desc* d, int n, int K, int splitk, long long* slab_off, long long* cs_off) {
    // workspace = per problem [splitk slabs of M x N | splitk partial column sums of M], 16-float aligned; the plain
    // per-problem path (sc_gemm_wgrad_bias) must fit too
    long long off = 0, worst_single = 0;
    const int sk = splitk < 1 ? 1 : splitk;
    for (int p = 0; p < n; ++p) {
        if (slab_off) slab_off[p] = off;
        off += ((long long)sk * d[p].M * d[p].N + 15) / 16 * 16;
        if (cs_off) cs_off[p] = off;
        off += ((long long)sk * d[p].M + 15) / 16 * 16;
        const long long single = sc_gemm_wgrad_ws_floats(d[p].M, d[p].N, K, splitk);
        if (single > worst_single) worst_single = single;
    }
    return (off > worst_single ? off : worst_single) + 64;
}

extern "C" long long sc_gemm_wgrad_group_ws_floats(const sc_wgrad_desc* descs, int n, int K, int splitk) {
    if (descs == nullptr || n < 1 || n > SC_WGRAD_GROUP_MAX) return 0;
    return wgrad_group_layout(descs, n, K, splitk, nullptr, nullptr);
}

extern "C" int sc_gemm_wgrad_group(const sc_wgrad_desc* descs, int n, int K, int splitk, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SC_CHECK(descs != nullptr && n >= 1 && n <= SC_WGRAD_GROUP_MAX, "sc_gemm_wgrad_group: 1..%d problems (n=%d)", SC_WGRAD_GROUP_MAX, n);
    SC_CHECK(K > 0 && ws != nullptr, "sc_gemm_wgrad_group: K=%d, workspace required", K);
    GemmArgs g[SC_WGRAD_GROUP_MAX];
    for (int p = 0; p < n; ++p) {
        const sc_wgrad_desc& d = descs[p];
        SC_CHECK(d.M > 0 && d.N > 0 && (d.M % 4) == 0 && (d.N % 4) == 0 && d.dY && d.X && d.dW,
                 "sc_gemm_wgrad_group: problem %d: bad shape M=%d N=%d", p, d.M, d.N);
        SC_CHECK((d.lddy % 8) == 0 && (d.ldx % 8) == 0 && ((uintptr_t)d.dY % 16) == 0 && ((uintptr_t)d.X % 16) == 0 &&
                 ((uintptr_t)d.dW % 16) == 0, "sc_gemm_wgrad_group: problem %d: operand alignment", p);
        GemmArgs& a = g[p];
        a.A = (const bf16*)d.dY; a.B = (const bf16*)d.X; a.M = d.M; a.N = d.N; a.K = K; a.lda = (int)d.lddy; a.ldb = (int)d.ldx;
        a.C = d.dW; a.ldc = d.N; a.C2 = nullptr; a.ldc2 = 0; a.bias = nullptr; a.res = nullptr; a.ldres = 0;
        a.aux = nullptr; a.ldaux = 0; a.tile_offset = 0; a.colsum = nullptr;
    }
    static const char* force = getenv("SC_GEMM_FORCE");
    int sk = 1, kps = K;
    if (!force && n > 1 && sc_gemm8p_tn_group_plan(g, n, splitk, &sk, &kps)) {
        long long slab_off[SC_WGRAD_GROUP_MAX], cs_off[SC_WGRAD_GROUP_MAX];
        (void)wgrad_group_layout(descs, n, K, sk, slab_off, cs_off);
        ReduceGroup r;
        r.n = n; r.nslab = sk;
        int total = 0;
        for (int p = 0; p < n; ++p) {
            GemmArgs& a = g[p];
            a.ntm = (a.M + 255) / 256; a.ntn = (a.N + 255) / 256;
            a.splitk = sk; a.k_per_split = kps; a.slab_stride = 0;
            if (sk > 1) { a.C = ws + slab_off[p]; a.slab_stride = (long long)a.M * a.N; }
            a.colsum = descs[p].dbias ? ws + cs_off[p] : nullptr;
            const long long n4 = (long long)a.M * a.N / 4;
            int blocks = (int)((n4 + 255) / 256);
            if (blocks > 1024) blocks = 1024;
            r.out[p] = descs[p].dW; r.slabs[p] = sk > 1 ? ws + slab_off[p] : nullptr; r.slab_stride[p] = a.slab_stride;
            r.n4[p] = n4; r.cs_out[p] = descs[p].dbias; r.cs_part[p] = ws + cs_off[p]; r.cs_n[p] = a.M;
            r.first[p] = total;
            total += blocks;
        }
        for (int p = n; p <= SC_WGRAD_GROUP_MAX; ++p) r.first[p] = total;
        for (int p = n; p < SC_WGRAD_GROUP_MAX; ++p) { r.out[p] = nullptr; r.slabs[p] = nullptr; r.cs_out[p] = nullptr; }
        const int rc = sc_gemm8p_tn_group_launch(g, n, st);
        if (rc != 1) return rc < 0 ? rc : -1;
        bool need = sk > 1;
        for (int p = 0; p < n; ++p) need = need || descs[p].dbias != nullptr;
        if (need) {
            reduce_slabs_group_kernel<<<total, 256, 0, st>>>(r);
            SC_LAUNCH_CHECK();
        }
        return 0;
    }
    // outside the 256x256 kernel's range (toy shapes) or a single problem: one by one through the per-Linear entry points
    for (int p = 0; p < n; ++p) {
        const sc_wgrad_desc& d = descs[p];
        int rc;
        if (d.dbias) {
            rc = sc_gemm_wgrad_bias(d.dY, (int)d.lddy, d.X, (int)d.ldx, d.M, d.N, K, d.dW, d.N, d.dbias, splitk, ws, stream);
        } else {
            const long long sf = sc_gemm_slab_floats(d.M, d.N, K, splitk);
            rc = sc_gemm_bf16(SC_GEMM_TN, SC_EPI_F32, d.dY, (int)d.lddy, d.X, (int)d.ldx, d.M, d.N, K, d.dW, d.N, nullptr, 0,
                              nullptr, nullptr, 0, nullptr, 0, sf ? splitk : 1, sf ? ws : nullptr, stream);
        }
        if (rc != 0) return rc;
    }
    return 0;
}
