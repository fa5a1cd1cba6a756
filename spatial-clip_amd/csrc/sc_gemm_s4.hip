// 256x256x32 bf16 MFMA GEMM with a 4-stage LDS-DMA ring (prefetch distance 3), gfx950.  Experimental variant of
// sc_gemm256.hip (same tile / wave layout / epilogue): BK = 32 so that 4 stages fit the 128 KiB staging budget; the
// K loop keeps up to 2 younger stages in flight behind a counted s_waitcnt vmcnt(8) and a raw s_barrier per 32-deep
// step, instead of draining to vmcnt(0) every 64-deep step.  Selected with SC_GEMM_FORCE=s4 (A/B benchmarking).
#include "sc_gemm_common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 256, BN = 256, BK = 32, NST = 4;
constexpr int TA = BM * BK * 2;                      // 16 KiB
constexpr int TB = BN * BK * 2;                      // 16 KiB
constexpr int STAGE = TA + TB;                       // 32 KiB
constexpr int EPI_BYTES = 8 * 64 * SC_EPI_LD * 4;    // 139264
constexpr int LDS_BYTES = NST * STAGE > EPI_BYTES ? NST * STAGE : EPI_BYTES;   // 139264

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

SC_DEVICE void dma16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// 4 wave-instructions per wave per stage: 2 for A (16 groups of 1 KiB over 8 waves), 2 for B
template <int MODE>
SC_DEVICE void stage_tile(const GemmArgs& g, char* sA, char* sB, int m0, int n0, int k0, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int grp = p * 8 + wave;
        if (MODE == SC_GEMM_NT) {
            const int r = grp * 16 + (lane >> 2);                 // rows of 64 B (4 chunks); one DMA = 16 rows
            const int lc = (lane & 3) ^ (((r >> 3) & 1) << 1);
            dma16(g.A + (size_t)min(m0 + r, g.M - 1) * g.lda + k0 + lc * 8, sA + grp * 1024);
            dma16(g.B + (size_t)min(n0 + r, g.N - 1) * g.ldb + k0 + lc * 8, sB + grp * 1024);
        } else {
            const int kr = grp * 2 + (lane >> 5);                 // [32 k][256 cols]: 512-B rows, one DMA = 2 rows
            const int s = (kr & 3) | (((kr >> 3) & 1) << 2);
            const int lc = (lane & 31) ^ (s << 1);
            dma16(g.A + (size_t)(k0 + kr) * g.lda + min(m0 + lc * 8, g.M - 8), sA + grp * 1024);
            dma16(g.B + (size_t)(k0 + kr) * g.ldb + min(n0 + lc * 8, g.N - 8), sB + grp * 1024);
        }
    }
}

template <int MODE, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_s4_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, lg = lane >> 4;

    int idx = sc_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = idx % g.ntn;
    idx /= g.ntn;
    const int tm = idx % g.ntm;
    const int z = idx / g.ntm;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nt = (kend - kbeg) / BK;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nt > 0) stage_tile<MODE>(g, smem, smem + TA, m0, n0, kbeg, wave, lane);
    if (nt > 1) stage_tile<MODE>(g, smem + STAGE, smem + STAGE + TA, m0, n0, kbeg + BK, wave, lane);
    if (nt > 2) stage_tile<MODE>(g, smem + 2 * STAGE, smem + 2 * STAGE + TA, m0, n0, kbeg + 2 * BK, wave, lane);

    int cur = 0;                                   // stage index of tile `it`
    for (int it = 0; it < nt; ++it) {
        if (it + 2 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // tile `it` landed, 2 younger may fly
        else if (it + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + 3 < nt) {
            const int nxt = (cur + 3) & 3;
            char* nA = smem + nxt * STAGE;
            stage_tile<MODE>(g, nA, nA + TA, m0, n0, kbeg + (it + 3) * BK, wave, lane);
        }
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + TA;
        bf16x8 af[8], bfr[4];
        if (MODE == SC_GEMM_NT) {
            const int coff = (lg ^ (((li >> 3) & 1) << 1)) << 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(sB + (wn * 64 + j * 16 + li) * 64 + coff);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(sA + (wm * 128 + i * 16 + li) * 64 + coff);
        } else {
            const int q = li >> 2, p = li & 3;
            const int kr = lg * 8 + q;
            const int s = q | ((lg & 1) << 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* pb = sB + kr * 512 + (((wn * 4 + j) ^ s) << 5) + p * 8;
                bfr[j] = sc_cat(sc_lds_tr16(pb), sc_lds_tr16(pb + 4 * 512));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const char* pa = sA + kr * 512 + (((wm * 8 + i) ^ s) << 5) + p * 8;
                af[i] = sc_cat(sc_lds_tr16(pa), sc_lds_tr16(pa + 4 * 512));
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = sc_mfma16(bfr[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        cur = (cur + 1) & 3;
    }
    __syncthreads();          // every wave is done with the staging buffers before they become epilogue space

    float* ep = reinterpret_cast<float*>(smem) + wave * 64 * SC_EPI_LD;
    EpiRegs<EPI> er;
    sc_epi_load<EPI>(er, m0 + wm * 128, n0 + wn * 64, lane, g);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sc_epi_put(ep, i, j, li, lg, acc[h * 4 + i][j]);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        sc_epilogue_store<EPI>(ep, er, m0 + wm * 128 + h * 64, n0 + wn * 64, lane, g, z,
                               h == 0 ? m0 + wm * 128 + 64 : -1);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

template <int MODE, int EPI>
int launch(const GemmArgs& g, int nblocks, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_s4_kernel<MODE, EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done = true;
    }
    gemm_s4_kernel<MODE, EPI><<<nblocks, 512, LDS_BYTES, st>>>(g);
    SC_LAUNCH_CHECK();
    return 1;
}

}  // namespace

int sc_gemm_s4_try(int mode, int epi, GemmArgs& g, int splitk_req, float* slabs, hipStream_t st) {
    if (g.M < 256 || g.N < 192 || (g.K % BK) != 0) return 0;
    if (mode == SC_GEMM_TN && ((g.M % 8) != 0 || (g.N % 8) != 0)) return 0;
    if ((long long)g.M * g.N < 256LL * 256 * 8) return 0;
    g.ntm = (g.M + BM - 1) / BM;
    g.ntn = (g.N + BN - 1) / BN;
    const int ktiles = g.K / BK;
    int splitk = splitk_req < 1 ? 1 : splitk_req;
    if (epi != SC_EPI_F32 || slabs == nullptr) splitk = 1;
    if (splitk > ktiles) splitk = ktiles;
    int tiles_per = (ktiles + splitk - 1) / splitk;
    splitk = (ktiles + tiles_per - 1) / tiles_per;
    g.splitk = splitk;
    g.k_per_split = tiles_per * BK;
    g.slab_stride = 0;
    if (splitk > 1) {
        if (g.ldc != g.N) return 0;
        g.C = slabs;
        g.slab_stride = (long long)g.M * g.N;
    }
    const int nblocks = g.ntm * g.ntn * splitk;
    int rc = 0;
#define SC_CASE(MODE, EPI) \
    if (mode == MODE && epi == EPI) rc = launch<MODE, EPI>(g, nblocks, st);
    SC_CASE(SC_GEMM_NT, SC_EPI_BF16)
    SC_CASE(SC_GEMM_NT, SC_EPI_BF16_BIAS)
    SC_CASE(SC_GEMM_NT, SC_EPI_F32_BIAS_RES)
    SC_CASE(SC_GEMM_NT, SC_EPI_GELU_PAIR)
    SC_CASE(SC_GEMM_NT, SC_EPI_BF16_DGELU)
    SC_CASE(SC_GEMM_NT, SC_EPI_F32)
    SC_CASE(SC_GEMM_TN, SC_EPI_F32)
#undef SC_CASE
    return rc;
}
