// 256x256x64 bf16 MFMA GEMM for the big regular shapes of the ViT tower (M = B*197 tokens), gfx950.
//   8 waves (2 x 4), each wave a 128x64 output tile = 8x4 MFMA 16x16x32 tiles (128 accumulator VGPRs),
//   operands staged global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging
//   VGPRs), double-buffered 2 x 64 KiB, one barrier per K tile; bank-conflict-free XOR swizzle applied on the
//   per-lane SOURCE address (the DMA destination is lane-linear) and again on the fragment read;
//   one workgroup per CU (139 KiB LDS), grid order XCD-aware so that the tiles sharing an A row-panel are
//   neighbours in one L2.
// Same operand layouts / epilogues as the 128x128 kernel (sc_gemm.hip), which remains the general fallback
// (ragged K, tiny problems).  Edges in M/N are handled by clamping the source row/column (the duplicated
// results are never stored); K (per split) must be a multiple of 64.
#include "sc_gemm_common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE = 256 * 64 * 2;                  // 32 KiB per operand tile
constexpr int STAGE = 2 * TILE;                     // A + B
constexpr int EPI_BYTES = 8 * 64 * SC_EPI_LD * 4;   // 139264
constexpr int LDS_BYTES = EPI_BYTES > 2 * STAGE ? EPI_BYTES : 2 * STAGE;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

SC_DEVICE void dma16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// ds_read_b64_tr_b16 through inline asm.  In front of the builtin form hipcc places `s_waitcnt vmcnt(0)` whenever an
// LDS-DMA is in flight (it cannot see that the DMA fills the OTHER stage), which serialises the prefetch of K tile
// i+1 with the fragment reads of tile i; plain ds_read_b128 (the NT path) does not get that wait.  The asm form is
// invisible to the waitcnt pass, so the lgkmcnt waits it would have placed are written by hand (tr_wait).
template <int OFF>
SC_DEVICE u32x2 tr16_asm(unsigned lds_addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF) : "memory");
    return r;
}
template <int CNT>
SC_DEVICE void tr_wait(u32x2& a, u32x2& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(CNT) : "memory");
}
template <int CNT>
SC_DEVICE void tr_wait_b(u32x2 (&lo)[4], u32x2 (&hi)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])
                 : "n"(CNT) : "memory");
}
SC_DEVICE bf16x8 tr_cat(u32x2 lo, u32x2 hi) {
    union { u32x4 u; bf16x8 b; } c;
    c.u = (u32x4){lo[0], lo[1], hi[0], hi[1]};
    return c.b;
}

// Issue the LDS-DMA of one K tile (A and B) : 8 wave-instructions per wave.
template <int MODE>
SC_DEVICE void stage_tile(const GemmArgs& g, char* sA, char* sB, int m0, int n0, int k0, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int grp = p * 8 + wave;                       // 32 groups of 1 KiB per operand tile
        if (MODE == SC_GEMM_NT) {
            const int r = grp * 8 + (lane >> 3);            // tile row, 128 B per row
            const int lc = (lane & 7) ^ ((r >> 1) & 7);     // logical 16-byte chunk stored at physical lane&7
            const int ra = min(m0 + r, g.M - 1), rb = min(n0 + r, g.N - 1);
            dma16(g.A + (size_t)ra * g.lda + k0 + lc * 8, sA + grp * 1024);
            dma16(g.B + (size_t)rb * g.ldb + k0 + lc * 8, sB + grp * 1024);
        } else {
            const int kr = grp * 2 + (lane >> 5);           // reduction row of the tile, 512 B per row
            const int s = (kr & 3) | (((kr >> 3) & 1) << 2);
            const int lc = (lane & 31) ^ (s << 1);
            const int ca = min(m0 + lc * 8, g.M - 8), cb = min(n0 + lc * 8, g.N - 8);
            dma16(g.A + (size_t)(k0 + kr) * g.lda + ca, sA + grp * 1024);
            dma16(g.B + (size_t)(k0 + kr) * g.ldb + cb, sB + grp * 1024);
        }
    }
}

// One 32-deep K step of the TN path: 8 + 2*NI transposed reads issued back to back, the MFMAs of row-fragment i
// start as soon as the B fragments and A fragment i have landed (LDS returns in order).
template <int NI, int KK, bool COLSUM>
SC_DEVICE void tn_step(unsigned sbase, const unsigned (&off_a)[NI], const unsigned (&off_b)[4], f32x4 (&acc)[NI][4],
                       bool do_cs, int wn, float (&cs)[2]) {
    constexpr int K0 = KK * 32 * 512, K1 = K0 + 4 * 512;
    u32x2 bl[4], bh[4], al[NI], ah[NI];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bl[j] = tr16_asm<K0>(sbase + off_b[j]);
        bh[j] = tr16_asm<K1>(sbase + off_b[j]);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        al[i] = tr16_asm<K0>(sbase + off_a[i]);
        ah[i] = tr16_asm<K1>(sbase + off_a[i]);
    }
    __builtin_amdgcn_s_setprio(1);
    tr_wait_b<(2 * NI > 15 ? 15 : 2 * NI)>(bl, bh);          // lgkmcnt is a 4-bit counter
    bf16x8 bfr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = tr_cat(bl[j], bh[j]);
    sc_static_for<NI>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        tr_wait<2 * (NI - 1 - i)>(al[i], ah[i]);
        const bf16x8 af = tr_cat(al[i], ah[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = sc_mfma16(bfr[j], af, acc[i][j]);
        if (COLSUM && do_cs && (i >> 1) == wn) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[i & 1] += (float)af[e];
        }
    });
    __builtin_amdgcn_s_setprio(0);
}

// NI = MFMA row-fragments per wave: 8 -> the workgroup owns a full 256-row tile; 4 / 2 -> it owns one half / quarter
// of a tile (128 / 64 rows).  The split variants run the LAST, partially filled round of a launch: the remaining
// tiles are cut so that every CU gets a (shorter) piece instead of most CUs idling for a whole tile time.
template <int MODE, int EPI, int NI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, lg = lane >> 4;

    constexpr int MSPLIT = 8 / NI;
    int idx = sc_xcd_remap(blockIdx.x, gridDim.x);
    const int part = idx % MSPLIT;
    idx = idx / MSPLIT + g.tile_offset;
    const int tn = idx % g.ntn;
    idx /= g.ntn;
    const int tm = idx % g.ntm;
    const int z = idx / g.ntm;
    const int m0 = tm * BM, n0 = tn * BN;
    const int mw = part * (BM / MSPLIT) + wm * (16 * NI);      // this wave's first row inside the tile
    const int kbeg = z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nt = (kend - kbeg) / BK;

    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fused bias gradient of a weight-gradient GEMM: column sums of the At operand (dY) over this K range, taken
    // from the fragments the wn == 0 waves of the tn == 0 workgroups read anyway (VALU work hidden under MFMA)
    constexpr bool kColsum = (MODE == SC_GEMM_TN && EPI == SC_EPI_F32 && NI == 8);
    const bool do_cs = kColsum && g.colsum != nullptr && tn == 0;
    float cs[2] = {0.f, 0.f};          // each of the 4 wn-waves sums 2 of the 8 row-fragments (balanced VALU work)

    // TN: byte offsets (inside a stage) of this lane's transposed-read fragments, first 4-row group of kk = 0
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    unsigned tr_off_a[NI], tr_off_b[4];
    if (MODE == SC_GEMM_TN) {
        const int q = li >> 2, p = li & 3;
        const int s = q | ((lg & 1) << 2);
        const int row = (lg * 8 + q) * 512 + p * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) tr_off_b[j] = TILE + row + (((wn * 4 + j) ^ s) << 5);
#pragma unroll
        for (int i = 0; i < NI; ++i) tr_off_a[i] = row + ((((mw >> 4) + i) ^ s) << 5);
    }

    if (nt > 0) stage_tile<MODE>(g, smem, smem + TILE, m0, n0, kbeg, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int it = 0; it < nt; ++it) {
        const int cur = it & 1;
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + TILE;
        if (it + 1 < nt) {
            char* nA = smem + (cur ^ 1) * STAGE;
            stage_tile<MODE>(g, nA, nA + TILE, m0, n0, kbeg + (it + 1) * BK, wave, lane);
        }
        if (MODE == SC_GEMM_NT) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[NI], bfr[4];
                const int coff = ((kk * 4 + lg) ^ ((li >> 1) & 7)) << 4;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bfr[j] = *reinterpret_cast<const bf16x8*>(sB + (wn * 64 + j * 16 + li) * 128 + coff);
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(sA + (mw + i * 16 + li) * 128 + coff);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = sc_mfma16(bfr[j], af[i], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
            }
        } else {
            const unsigned sbase = lds0 + cur * STAGE;
            tn_step<NI, 0, kColsum>(sbase, tr_off_a, tr_off_b, acc, do_cs, wn, cs);
            tn_step<NI, 1, kColsum>(sbase, tr_off_a, tr_off_b, acc, do_cs, wn, cs);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (kColsum && do_cs) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float v = cs[k];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int m = m0 + mw + (2 * wn + k) * 16 + li;
            if (lg == 0 && m < g.M) g.colsum[(size_t)z * g.M + m] = v;
        }
    }
    // ---------------- epilogue: two 64-row halves of the wave's 128x64 tile through a private LDS region ----------------
    float* ep = reinterpret_cast<float*>(smem) + wave * 64 * SC_EPI_LD;
    constexpr int NH = NI > 4 ? 2 : 1;               // 64-row passes of the wave's tile
    constexpr int IH = NI > 4 ? 4 : NI;              // row-fragments per pass
    EpiRegs<EPI> er;
    sc_epi_load<EPI>(er, m0 + mw, n0 + wn * 64, lane, g, 16 * IH);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int i = 0; i < IH; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sc_epi_put(ep, i, j, li, lg, acc[h * 4 + i][j]);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        sc_epilogue_store<EPI>(ep, er, m0 + mw + h * 64, n0 + wn * 64, lane, g, z,
                               (h + 1 < NH) ? m0 + mw + 64 : -1, 16 * IH);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

template <int MODE, int EPI, int NI>
int launch1(const GemmArgs& g, int nblocks, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<MODE, EPI, NI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done = true;
    }
    gemm256_kernel<MODE, EPI, NI><<<nblocks, 512, LDS_BYTES, st>>>(g);
    SC_LAUNCH_CHECK();
    return 1;
}

// full rounds with whole tiles, then the remainder cut into halves / quarters so that it still covers the chip
template <int MODE, int EPI>
int launch(GemmArgs& g, int ntiles, hipStream_t st) {
    constexpr int CUS = 256;
    // measured (interleaved A/B on one MI355X, ViT-B/16 step): splitting the last round is 0.4 ms/step SLOWER than
    // leaving it whole -- workgroups are not in lockstep rounds, the hardware backfills; opt-in only (SC_GEMM_TAIL=1)
    static const bool tail_split = (getenv("SC_GEMM_TAIL") && getenv("SC_GEMM_TAIL")[0] == '1');
    const int rem = ntiles % CUS;
    int msplit = 1;
    if (tail_split && g.splitk == 1 && g.colsum == nullptr && ntiles > CUS && rem > 0) {
        if (rem * 4 <= CUS) msplit = 4;
        else if (rem * 2 <= CUS + 64) msplit = 2;
    }
    g.tile_offset = 0;
    if (msplit == 1) return launch1<MODE, EPI, 8>(g, ntiles, st);
    const int nfull = ntiles - rem;
    int rc = launch1<MODE, EPI, 8>(g, nfull, st);
    if (rc != 1) return rc;
    g.tile_offset = nfull;
    return msplit == 4 ? launch1<MODE, EPI, 2>(g, rem * 4, st) : launch1<MODE, EPI, 4>(g, rem * 2, st);
}

}  // namespace

int sc_gemm256_try(int mode, int epi, GemmArgs& g, int splitk_req, float* slabs, float* c_final, hipStream_t st) {
    // eligibility: enough work for 256-tiles, K a multiple of 64 per split, inner dims allow clamped 16-byte chunks
    if (g.M < 256 || g.N < 192 || (g.K % BK) != 0) return 0;
    if (mode == SC_GEMM_TN && ((g.M % 8) != 0 || (g.N % 8) != 0)) return 0;
    const long long work = (long long)g.M * g.N;
    if (work < 256LL * 256 * 8) return 0;
    if (mode == SC_GEMM_NT && splitk_req <= 1) {      // few tiles: the 128x128 general kernel (see sc_gemm8p_try)
        static const int small_tiles = getenv("SC_GEMM_SMALL_TILES") ? atoi(getenv("SC_GEMM_SMALL_TILES")) : 100;
        if (work < 256LL * 256 * small_tiles) return 0;
    }
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
    SC_CASE(SC_GEMM_NT, SC_EPI_BF16_BIAS_RES)
    SC_CASE(SC_GEMM_NT, SC_EPI_GELU_GRAD_PAIR)
    SC_CASE(SC_GEMM_NT, SC_EPI_BF16_MUL_AUX)
    SC_CASE(SC_GEMM_NT, SC_EPI_F32)
    SC_CASE(SC_GEMM_TN, SC_EPI_F32)
#undef SC_CASE
    return rc;
}
