// 256x256x64 bf16 MFMA GEMM with a phase-interleaved ("ping-pong") main loop, gfx950.  NT layout only.
//
//   8 waves (2 x 4), each wave a 128x64 output tile (128 accumulator VGPRs) cut into four 64x32 quadrants.
//   A K tile (64 deep) is four phases, one quadrant each: {fragment ds_reads + 2 LDS-DMA of a later half-tile ->
//   s_barrier -> 16 MFMA -> s_barrier}.  Waves 4-7 run one barrier behind waves 0-3, and wave w shares its SIMD
//   with wave w+4, so on every SIMD one wave is in its MFMA section while its partner reads LDS / issues DMA:
//   the matrix pipe never waits for a fragment read, and the DMA never waits for the matrix pipe.
//
//   Operand tiles live in LDS as HALF-TILES of 128 rows x 64 k (16 KiB, 128-B rows, same XOR swizzle as
//   sc_gemm256.hip): A half i holds, for each of the two M-waves, rows [64 i, 64 i + 64) of its 128 rows; B half j
//   holds, for each of the four N-waves, columns [32 j, 32 j + 32) of its 64.  Ring = 2 K tiles x 4 half-tiles
//   = 128 KiB.  Half-tiles are staged in the order they are consumed (A0, B0, B1, A1), six half-tiles ahead of the
//   phase that issues them, with a counted `s_waitcnt vmcnt(8)` (four half-tiles stay in flight across the barriers).
//
//   Ordering rules the schedule is built on:
//     RAW: a half-tile is read one phase after the phase whose load section waited for it (every wave waits for its
//          own DMA, the barrier that follows publishes it to the other waves -- including the staggered group);
//     WAR: a ring slot is restaged at the earliest two phases after the phase that read it (the staggered group
//          retires its reads one barrier later than the leading group).
//   Epilogue: bf16 outputs without an extra input tile take a bf16 LDS strip (epilogue_bf16_lds); the others the
//   fp32 LDS staging shared with sc_gemm256.hip (sc_gemm_common.h).
#include "sc_gemm_common.h"
#include <stdlib.h>
#include <map>
#include <mutex>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 128 * 64 * 2;                  // 16 KiB half-tile
constexpr int RING = 8 * HALF;                      // 128 KiB
constexpr int EPI_BYTES = 8 * 64 * SC_EPI_LD * 4;   // 139264 (LDS-staged epilogues)
constexpr int LDS_BYTES = EPI_BYTES > RING ? EPI_BYTES : RING;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

SC_DEVICE void dma16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// bf16 epilogue through a small wave-private LDS strip (4 KiB: 32 rows x 128 B), four passes per 128x64 wave tile.
// Why: a store instruction that writes whole 128-B lines (8 rows x 128 B) retires 3.7x faster than the 16 rows x 64 B
// an accumulator-layout store touches (tools/micro/store_pattern.hip: 1.0 vs 3.8 us per 256x256 bf16 tile on one CU),
// and an un-retired store holds back every later LDS-DMA of the same wave (vmcnt retires in order).  The accumulators
// are packed to bf16 BEFORE the LDS trip (half the LDS bytes of the fp32 staging in sc_gemm_common.h), 16-B chunks
// XOR-swizzled by row so the 8-B writes (2-way at worst) and 16-B reads spread over the banks.
//   BF16 / BF16_BIAS: C = bf16(acc (+ bias));  GELU_PAIR: C = u = bf16(acc + bias), C2 = bf16(gelu(float(u)));
//   GELU_GRAD_PAIR: C = bf16(gelu'(float(u))), C2 as before, u itself is not stored.
template <int EPI, bool Q8 = false, bool LUT = false>
SC_DEVICE void epilogue_bf16_lds(f32x4 (&acc)[8][4], const GemmArgs& g, char* strip, int row0, int col0, int lane,
                                 const unsigned* lut = nullptr) {
    float amax_lane = 0.f;
    const float q8s = (Q8 && sc_epi_gelu_fwd(EPI) && g.q8) ? *g.q8_scale : 0.f;
    const int li = lane & 15, lg = lane >> 4;
    constexpr bool kBias = (EPI == SC_EPI_BF16_BIAS || sc_epi_gelu_fwd(EPI));
    f32x4 bj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bj[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int c = col0 + j * 16 + lg * 4;
        if (kBias && g.bias && c < g.N) bj[j] = *reinterpret_cast<const f32x4*>(g.bias + c);
    }
    bf16* C = reinterpret_cast<bf16*>(g.C);
    bf16* C2 = reinterpret_cast<bf16*>(g.C2);
    const int rr = lane >> 3, rc = lane & 7;                      // read side: row (+ 8 s), 16-B chunk
    const int gcol = col0 + rc * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int r = ib * 16 + li;                           // strip row
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = acc[p * 2 + ib][j] + bj[j];
                const int chunk = (j * 2 + (lg >> 1)) ^ (r & 7);
                *reinterpret_cast<u32x2*>(strip + r * 128 + chunk * 16 + (lg & 1) * 8) = sc_pack4(v[0], v[1], v[2], v[3]);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int r = s4 * 8 + rr;
            const u32x4 u = *reinterpret_cast<const u32x4*>(strip + r * 128 + ((rc ^ (r & 7)) << 4));
            const int grow = row0 + p * 32 + r;
            if (grow < g.M && gcol < g.N) {
                if (EPI != SC_EPI_GELU_GRAD_PAIR) *reinterpret_cast<u32x4*>(C + (size_t)grow * g.ldc + gcol) = u;
                if (sc_epi_gelu_fwd(EPI)) {
                    union { u32x4 w; bf16x8 h; } x;
                    x.w = u;
                    bf16x8 o;
                    if (EPI == SC_EPI_GELU_PAIR) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16)sc_act((float)x.h[e], g.act);
                    } else {                             // the backward's factor gelu'(u) instead of u
                        bf16x8 gd;
                        bool formula = true;
                        if (LUT) {                       // both values by table (SC_GELU_LUT_*): same bits as the formula
                            unsigned ent[8];
                            bool inside = true;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const unsigned bits = (u[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                                const unsigned rel = (bits & 0x7FFFu) - (unsigned)SC_GELU_LUT_LO;
                                inside = inside && rel < (unsigned)SC_GELU_LUT_HALF;
                                const unsigned idx = min(rel, (unsigned)SC_GELU_LUT_HALF - 1u) + (bits >> 15) * (unsigned)SC_GELU_LUT_HALF;
                                ent[e] = lut[idx];
                            }
                            formula = __builtin_amdgcn_ballot_w64(!inside) != 0;      // wave-uniform
                            if (!formula) {
                                union { u32x4 w; bf16x8 h; } lo, hi;
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    lo.w[q] = __builtin_amdgcn_perm(ent[2 * q + 1], ent[2 * q], 0x05040100u);
                                    hi.w[q] = __builtin_amdgcn_perm(ent[2 * q + 1], ent[2 * q], 0x07060302u);
                                }
                                o = lo.h;
                                gd = hi.h;
                            }
                        }
                        if (formula) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float hv, gv;
                                sc_act_both((float)x.h[e], g.act, hv, gv);
                                o[e] = (bf16)hv;
                                gd[e] = (bf16)gv;
                            }
                        }
                        *reinterpret_cast<bf16x8*>(C + (size_t)grow * g.ldc + gcol) = gd;
                    }
                    *reinterpret_cast<bf16x8*>(C2 + (size_t)grow * g.ldc2 + gcol) = o;
                    if (Q8 && g.q8) {                    // e4m3 copy of h for the c_proj forward GEMM (GemmArgs::q8)
                        float r[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) { r[e] = (float)o[e]; amax_lane = fmaxf(amax_lane, fabsf(r[e])); }
                        *reinterpret_cast<u32x2*>(g.q8 + (size_t)grow * g.ldq8 + gcol) = sc_pack8_fp8(r, q8s);
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    if (Q8 && sc_epi_gelu_fwd(EPI) && g.q8) sc_amax_publish(amax_lane, g.q8_amax);
}

// ring slot of half-tile q (0: A half 0, 1: B half 0, 2: B half 1, 3: A half 1) of the K tile with parity D
constexpr int slot(int D, int q) { return D * 4 * HALF + q * HALF; }

struct Stager {
    const bf16* src[4][2];      // [q][p] : this lane's source of the two 1-KiB pieces it copies per half-tile
    int nt;
    int wave;
};

// One phase of K tile `t` (ring parity D):  PH = 1..4  <->  quadrant (0,0) (0,1) (1,1) (1,0).
template <int D, int PH>
SC_DEVICE void phase(char* smem, const Stager& S, int t, const int (&a_off)[2], const int (&b_off)[2], bf16x8 (&a)[8],
                     bf16x8 (&b0)[4], bf16x8 (&b1)[4], f32x4 (&acc)[8][4]) {
    // ---- load section: fragments of this quadrant that are not in registers yet ----
    if (PH == 1) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                b0[kk * 2 + jj] = *reinterpret_cast<const bf16x8*>(smem + slot(D, 1) + b_off[kk] + jj * 2048);
    }
    if (PH == 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                b1[kk * 2 + jj] = *reinterpret_cast<const bf16x8*>(smem + slot(D, 2) + b_off[kk] + jj * 2048);
    }
    if (PH == 1 || PH == 3) {
        constexpr int sl = slot(D, PH == 1 ? 0 : 3);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                a[kk * 4 + ii] = *reinterpret_cast<const bf16x8*>(smem + sl + a_off[kk] + ii * 2048);
    }
    // ---- stage the half-tile six positions ahead (consumption order A0 B0 B1 A1) ----
    constexpr int q = (PH + 1) & 3;
    constexpr int DS = PH <= 2 ? (D ^ 1) : D;
    const int ts = t + (PH <= 2 ? 1 : 2);
    if (ts < S.nt) {
        dma16(S.src[q][0] + (size_t)ts * BK, smem + slot(DS, q) + S.wave * 1024);
        dma16(S.src[q][1] + (size_t)ts * BK, smem + slot(DS, q) + (8 + S.wave) * 1024);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // retires the half-tile read in the NEXT phase
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ring is draining: fewer than four in flight
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- MFMA section ----
    constexpr int mi = PH >= 3 ? 1 : 0;
    constexpr int nj = (PH == 2 || PH == 3) ? 1 : 0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                acc[mi * 4 + ii][nj * 2 + jj] =
                    sc_mfma16(nj ? b1[kk * 2 + jj] : b0[kk * 2 + jj], a[kk * 4 + ii], acc[mi * 4 + ii][nj * 2 + jj]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int D>
SC_DEVICE void ktile(char* smem, const Stager& S, int t, const int (&a_off)[2], const int (&b_off)[2], bf16x8 (&a)[8],
                     bf16x8 (&b0)[4], bf16x8 (&b1)[4], f32x4 (&acc)[8][4]) {
    phase<D, 1>(smem, S, t, a_off, b_off, a, b0, b1, acc);
    phase<D, 2>(smem, S, t, a_off, b_off, a, b0, b1, acc);
    phase<D, 3>(smem, S, t, a_off, b_off, a, b0, b1, acc);
    phase<D, 4>(smem, S, t, a_off, b_off, a, b0, b1, acc);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 15, lg = lane >> 4;

    int idx = sc_xcd_remap(blockIdx.x, gridDim.x);
    int tn, tm, z;
    if (g.col_group > 0) {                                       // column-group walk (splitk == 1): sc_gemm_common.h
        sc_tile_colgroup(idx, g, tm, tn);
        z = 0;
    } else {
        tn = idx % g.ntn;
        idx /= g.ntn;
        tm = idx % g.ntm;
        z = idx / g.ntm;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    Stager S;
    S.nt = (kend - kbeg) / BK;
    S.wave = wave;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = (p * 8 + wave) * 8 + (lane >> 3);          // row of the half-tile image, 128 B per row
        const int lc = (lane & 7) ^ ((r >> 1) & 7);              // logical 16-byte chunk stored at physical lane&7
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ga = min(m0 + (r >> 6) * 128 + h * 64 + (r & 63), g.M - 1);
            const int gb = min(n0 + (r >> 5) * 64 + h * 32 + (r & 31), g.N - 1);
            S.src[h ? 3 : 0][p] = g.A + (size_t)ga * g.lda + kbeg + lc * 8;
            S.src[h ? 2 : 1][p] = g.B + (size_t)gb * g.ldb + kbeg + lc * 8;
        }
    }
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int coff = ((kk * 4 + lg) ^ ((li >> 1) & 7)) << 4;
        a_off[kk] = (wr * 64 + li) * 128 + coff;
        b_off[kk] = (wc * 32 + li) * 128 + coff;
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: the first six half-tiles (all of K tile 0, A0 and B0 of K tile 1)
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int ts = s >> 2, q = s & 3;
        if (ts < S.nt) {
            dma16(S.src[q][0] + (size_t)ts * BK, smem + slot(ts & 1, q) + wave * 1024);
            dma16(S.src[q][1] + (size_t)ts * BK, smem + slot(ts & 1, q) + (8 + wave) * 1024);
        }
    }
    if (S.nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                   // waves 4-7 run one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    bf16x8 a[8], b0[4], b1[4];
    for (int kt = 0; kt < S.nt; kt += 2) {
        ktile<0>(smem, S, kt, a_off, b_off, a, b0, b1, acc);
        if (kt + 1 < S.nt) ktile<1>(smem, S, kt + 1, a_off, b_off, a, b0, b1, acc);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();                   // re-align the two wave groups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // Epilogue: bf16 outputs without an extra input tile go through the bf16 LDS strip (full-line stores, half the LDS
    // bytes); the fp32-residual, GELU' and fp32 epilogues keep the fp32 staging shared with sc_gemm256.hip.
    if (EPI == SC_EPI_GELU_GRAD_PAIR && g.gelu_lut != nullptr) {
        // the table moves into the dead operand ring (behind the eight 4-KiB strips) while the first strip pass is packed
        unsigned* lut = reinterpret_cast<unsigned*>(smem + 8 * 4096);
        for (int c = t; c < SC_GELU_LUT_N / 4; c += 512)
            reinterpret_cast<u32x4*>(lut)[c] = reinterpret_cast<const u32x4*>(g.gelu_lut)[c];
        __syncthreads();
        epilogue_bf16_lds<EPI, false, true>(acc, g, smem + wave * 4096, m0 + wr * 128, n0 + wc * 64, lane, lut);
    } else if (EPI == SC_EPI_BF16 || EPI == SC_EPI_BF16_BIAS || sc_epi_gelu_fwd(EPI)) {
        epilogue_bf16_lds<EPI>(acc, g, smem + wave * 4096, m0 + wr * 128, n0 + wc * 64, lane);
    } else {
        const int mw = wr * 128;
        float* ep = reinterpret_cast<float*>(smem) + wave * 64 * SC_EPI_LD;
        EpiRegs<EPI> er;
        sc_epi_load<EPI>(er, m0 + mw, n0 + wc * 64, lane, g, 64);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) sc_epi_put(ep, i, j, li, lg, acc[h * 4 + i][j]);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            sc_epilogue_store<EPI>(ep, er, m0 + mw + h * 64, n0 + wc * 64, lane, g, z, (h + 1 < 2) ? m0 + mw + 64 : -1, 64);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__global__ void sc_gelu_lut_fill_kernel(unsigned* lut, int act) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SC_GELU_LUT_N) return;
    const unsigned bits = (unsigned)(SC_GELU_LUT_LO + i % SC_GELU_LUT_HALF) | (i >= SC_GELU_LUT_HALF ? 0x8000u : 0u);
    const float u = __uint_as_float(bits << 16);
    float hv, gv;
    sc_act_both(u, act, hv, gv);
    union { bf16 b; unsigned short s; } h, gq;
    h.b = (bf16)hv;
    gq.b = (bf16)gv;
    lut[i] = ((unsigned)gq.s << 16) | (unsigned)h.s;
}

template <int EPI>
int launch(const GemmArgs& g, int nblocks, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_kernel<EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done = true;
    }
    gemm8p_kernel<EPI><<<nblocks, 512, LDS_BYTES, st>>>(g);
    SC_LAUNCH_CHECK();
    return 1;
}


// =====================================================================================================================
// Persistent NT variant: one workgroup per CU walks a list of output tiles (tile = first + r * gridDim).  The staging
// stream never stops at a tile boundary -- the first six half-tiles of the next tile are issued during the last K
// tiles of the current one -- so a new tile starts without the DMA round trip, without a workgroup launch and without
// waiting for the previous tile's stores to drain.  At a boundary the two wave groups re-align (one extra barrier
// for waves 0-3), every wave writes its 128x64 result through its 4-KiB LDS strip behind the ring (epilogue_bf16_lds), and the
// stagger is restored (one extra barrier for waves 4-7).  The epilogue's stores enter the vmcnt stream between two half-tile DMAs: the
// three phases that follow allow for them in their counted waits (kind POSTEPI, interior tiles: every store is
// issued); tiles that touch the M / N edge, where a fully masked store may be skipped, drain to zero instead.
enum { PW_STEADY = 0, PW_POSTEPI = 1, PW_DRAIN0 = 2 };      // counted-wait flavour of the K tile that follows an epilogue

struct PStager {
    const bf16* src[4][2];
    long long koff;             // element offset of the K tile being staged
    int sk;                     // K tile (within the staging tile) being staged
    int stile;                  // staging tile index, >= total when the list is exhausted
    int nt, wave, total, stride;
};

SC_DEVICE void pstage_set_tile(PStager& S, const GemmArgs& g, int lane) {
    const int tn = S.stile % g.ntn, tm = S.stile / g.ntn;
    const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = (p * 8 + S.wave) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ga = min(m0 + (r >> 6) * 128 + h * 64 + (r & 63), g.M - 1);
            const int gb = min(n0 + (r >> 5) * 64 + h * 32 + (r & 31), g.N - 1);
            S.src[h ? 3 : 0][p] = g.A + (size_t)ga * g.lda + lc * 8;
            S.src[h ? 2 : 1][p] = g.B + (size_t)gb * g.ldb + lc * 8;
        }
    }
}
// move the staging cursor to the next K tile (possibly the first one of the next tile of this workgroup's list)
SC_DEVICE void pstage_advance(PStager& S, const GemmArgs& g, int lane) {
    S.sk += 1;
    S.koff += BK;
    if (S.sk == S.nt) {
        S.sk = 0;
        S.koff = 0;
        S.stile += S.stride;
        if (S.stile < S.total) pstage_set_tile(S, g, lane);
    }
}

// One phase; `doff` = byte offset of the ring half (parity) that holds the K tile being computed.
template <int PH, int ESTORES>
SC_DEVICE void pphase(char* smem, int doff, PStager& S, const GemmArgs& g, int lane, int pw, const int (&a_off)[2],
                      const int (&b_off)[2], bf16x8 (&a)[8], bf16x8 (&b0)[4], bf16x8 (&b1)[4], f32x4 (&acc)[8][4]) {
    const char* cur = smem + doff;
    if (PH == 1) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                b0[kk * 2 + jj] = *reinterpret_cast<const bf16x8*>(cur + 1 * HALF + b_off[kk] + jj * 2048);
    }
    if (PH == 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                b1[kk * 2 + jj] = *reinterpret_cast<const bf16x8*>(cur + 2 * HALF + b_off[kk] + jj * 2048);
    }
    if (PH == 1 || PH == 3) {
        constexpr int sl = (PH == 1 ? 0 : 3) * HALF;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                a[kk * 4 + ii] = *reinterpret_cast<const bf16x8*>(cur + sl + a_off[kk] + ii * 2048);
    }
    // staging: the K tile under the cursor is (compute K tile + 1) in phases 1-2 and (+ 2) in phases 3-4
    constexpr int q = (PH + 1) & 3;
    char* dst = smem + (PH <= 2 ? (doff ^ (4 * HALF)) : doff) + q * HALF;
    if (PH == 3) pstage_advance(S, g, lane);
    if (S.stile < S.total) {
        dma16(S.src[q][0] + S.koff, dst + S.wave * 1024);
        dma16(S.src[q][1] + S.koff, dst + (8 + S.wave) * 1024);
        if (PH <= 3 && pw == PW_POSTEPI) {
            constexpr int N = 8 + ESTORES > 63 ? 63 : 8 + ESTORES;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
        } else if (PH == 1 && pw == PW_DRAIN0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    constexpr int mi = PH >= 3 ? 1 : 0;
    constexpr int nj = (PH == 2 || PH == 3) ? 1 : 0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                acc[mi * 4 + ii][nj * 2 + jj] =
                    sc_mfma16(nj ? b1[kk * 2 + jj] : b0[kk * 2 + jj], a[kk * 4 + ii], acc[mi * 4 + ii][nj * 2 + jj]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm8pp_kernel(const GemmArgs g, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    // stores one wave issues per interior tile (every lane in bounds): 4 passes x 4 full-line stores, two tensors -> 32
    constexpr int ESTORES = sc_epi_gelu_fwd(EPI) ? 32 : 16;

    PStager S;
    S.nt = g.K / BK;
    S.wave = wave;
    S.total = total_tiles;
    S.stride = gridDim.x;
    S.stile = sc_xcd_remap(blockIdx.x, gridDim.x);
    S.sk = 0;
    S.koff = 0;
    pstage_set_tile(S, g, lane);
    int ctile = S.stile;                                         // tile being computed

    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int coff = ((kk * 4 + lg) ^ ((li >> 1) & 7)) << 4;
        a_off[kk] = (wr * 64 + li) * 128 + coff;
        b_off[kk] = (wc * 32 + li) * 128 + coff;
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: K tile 0 (four half-tiles) and A0, B0 of K tile 1 (the launcher guarantees nt >= 3)
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int q = s & 3;
        if (s == 4) pstage_advance(S, g, lane);
        dma16(S.src[q][0] + S.koff, smem + slot(s >> 2, q) + wave * 1024);
        dma16(S.src[q][1] + S.koff, smem + slot(s >> 2, q) + (8 + wave) * 1024);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    bf16x8 a[8], b0[4], b1[4];
    int k = 0;                                                   // K tile inside the tile being computed
    int pw = PW_STEADY;
    int doff = 0;                                                // ring half of the K tile being computed
    while (ctile < total_tiles) {
        pphase<1, ESTORES>(smem, doff, S, g, lane, pw, a_off, b_off, a, b0, b1, acc);
        pphase<2, ESTORES>(smem, doff, S, g, lane, pw, a_off, b_off, a, b0, b1, acc);
        pphase<3, ESTORES>(smem, doff, S, g, lane, pw, a_off, b_off, a, b0, b1, acc);
        pphase<4, ESTORES>(smem, doff, S, g, lane, pw, a_off, b_off, a, b0, b1, acc);
        doff ^= 4 * HALF;
        pw = PW_STEADY;
        if (++k == S.nt) {
            k = 0;
            const int tn = ctile % g.ntn, tm = ctile / g.ntn;
            const int m0 = tm * BM, n0 = tn * BN;
            if (wr == 0) __builtin_amdgcn_s_barrier();           // waves 0-3 wait for 4-7: groups aligned
            __builtin_amdgcn_sched_barrier(0);
            epilogue_bf16_lds<EPI>(acc, g, smem + RING + wave * 4096, m0 + wr * 128, n0 + wc * 64, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
            pw = ((m0 + BM <= g.M) && (n0 + BN <= g.N)) ? PW_POSTEPI : PW_DRAIN0;
            ctile += gridDim.x;
            if (ctile < total_tiles && wr == 1) __builtin_amdgcn_s_barrier();   // restore the stagger
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int EPI>
int launch_persistent(const GemmArgs& g, int total_tiles, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8pp_kernel<EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, RING + 8 * 4096);
        attr_done = true;
    }
    const int grid = total_tiles < 256 ? total_tiles : 256;
    gemm8pp_kernel<EPI><<<grid, 512, RING + 8 * 4096, st>>>(g, total_tiles);
    SC_LAUNCH_CHECK();
    return 1;
}

// =====================================================================================================================
// TN layout (weight gradients): C[m,n] = sum_k At[k,m] Bt[k,n], both operands k-major in global memory.
//   Half-tile image = [64 k][128 columns] bf16 (256-B rows); A half i = tile columns [128 i, 128 i + 128), of which
//   M-wave wr owns [64 wr, 64 wr + 64); B half j likewise with N-wave wc owning [32 wc, 32 wc + 32) -- so a wave's
//   128x64 output is rows {128 i + 64 wr + ..} x columns {128 j + 32 wc + ..} (2 x 2 blocks of 64 x 32), and every
//   DMA row is one contiguous 256-B run of the source.  Fragments come out of LDS through ds_read_b64_tr_b16 (inline
//   asm + hand-placed lgkmcnt: see sc_gemm256.hip), 32-B chunk ^= (k & 3) | ((k >> 3) & 1) << 2 keeps the 8 rows a
//   32-lane half touches on distinct banks.  Phase / ring schedule identical to the NT kernel above.
template <int OFF>
SC_DEVICE u32x2 tr16_asm(unsigned lds_addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF) : "memory");
    return r;
}
SC_DEVICE bf16x8 tr_cat(u32x2 lo, u32x2 hi) {
    union { u32x4 u; bf16x8 b; } c;
    c.u = (u32x4){lo[0], lo[1], hi[0], hi[1]};
    return c.b;
}

struct FragTN {                 // one operand fragment set: [kk][f] as two 64-bit halves
    u32x2 lo, hi;
};
// The asm reads above are invisible to hipcc's waitcnt pass, so the wait is written by hand -- and it must NAME the
// registers it retires ("+v"): a clobber-only `s_waitcnt` orders memory, not registers, and the compiler may copy or
// consume an asm-loaded register in front of it (seen in the attention kernels).  Same pattern as
// tr_wait / tr_wait_b in sc_gemm256.hip.
SC_DEVICE void tn_wait4(FragTN (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo),
                   "+v"(f[3].hi)
                 :: "memory");
}
SC_DEVICE void tn_wait8(FragTN (&f)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo),
                   "+v"(f[3].hi), "+v"(f[4].lo), "+v"(f[4].hi), "+v"(f[5].lo), "+v"(f[5].hi), "+v"(f[6].lo), "+v"(f[6].hi),
                   "+v"(f[7].lo), "+v"(f[7].hi)
                 :: "memory");
}

template <int NF>
SC_DEVICE void tn_read(unsigned base, const unsigned (&off)[NF], FragTN (&f)[2 * NF]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int x = 0; x < NF; ++x) {
            if (kk == 0) {
                f[x].lo = tr16_asm<0>(base + off[x]);
                f[x].hi = tr16_asm<1024>(base + off[x]);
            } else {
                f[NF + x].lo = tr16_asm<8192>(base + off[x]);
                f[NF + x].hi = tr16_asm<8192 + 1024>(base + off[x]);
            }
        }
}

struct StagerTN {
    // wave-uniform base (SGPRs) + 32-bit per-lane byte offset: the DMA takes the scalar-base addressing form (no address VALU, 4
    // VGPRs instead of 16); piece 1 of a half-tile is 32 k rows behind piece 0 = a scalar addend (round 4, from the e4m3 kernel)
    const char* base[4];
    unsigned off[4];
    long long step[4];          // BYTES per K tile (64 source rows) for each half-tile kind
    int nt;
    int wave;
};

template <int D, int PH>
SC_DEVICE void phase_tn(char* smem, unsigned lds0, const StagerTN& S, int t, const unsigned (&a_off)[4],
                        const unsigned (&b_off)[2], FragTN (&a)[8], FragTN (&b0)[4], FragTN (&b1)[4], f32x4 (&acc)[8][4],
                        bool do_cs, int wc, float (&cs)[2]) {
    if (PH == 1) tn_read<2>(lds0 + slot(D, 1), b_off, b0);
    if (PH == 2) tn_read<2>(lds0 + slot(D, 2), b_off, b1);
    if (PH == 1) tn_read<4>(lds0 + slot(D, 0), a_off, a);
    if (PH == 3) tn_read<4>(lds0 + slot(D, 3), a_off, a);
    constexpr int q = (PH + 1) & 3;
    constexpr int DS = PH <= 2 ? (D ^ 1) : D;
    const int ts = t + (PH <= 2 ? 1 : 2);
    if (ts < S.nt) {
        const char* kb = S.base[q] + ts * S.step[q];
        unsigned o0 = S.off[q];
        asm volatile("" : "+v"(o0));                      // keep (scalar base + 32-bit offset): no hoisted 64-bit sums
        dma16(kb + o0, smem + slot(DS, q) + S.wave * 1024);
        dma16(kb + (S.step[q] >> 1) + o0, smem + slot(DS, q) + (8 + S.wave) * 1024);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // every fragment register this phase's reads wrote is tied to the wait that retires it
    if (PH == 1) { tn_wait4(b0); tn_wait8(a); }
    if (PH == 2) tn_wait4(b1);
    if (PH == 3) tn_wait8(a);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int mi = PH >= 3 ? 1 : 0;
    constexpr int nj = (PH == 2 || PH == 3) ? 1 : 0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const bf16x8 af = tr_cat(a[kk * 4 + ii].lo, a[kk * 4 + ii].hi);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const FragTN& bf = nj ? b1[kk * 2 + jj] : b0[kk * 2 + jj];
                acc[mi * 4 + ii][nj * 2 + jj] = sc_mfma16(tr_cat(bf.lo, bf.hi), af, acc[mi * 4 + ii][nj * 2 + jj]);
            }
            // fused bias gradient: column sums of At over this K tile, one 16-column fragment per wave (ii == wc),
            // taken from the A fragments when they are first used (PH 1: half 0, PH 3: half 1); the VALU adds sit
            // between the MFMAs so they issue in the matrix pipe's shadow
            if ((PH == 1 || PH == 3) && do_cs && ii == wc) {
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[mi] += (float)af[e];
            }
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int D>
SC_DEVICE void ktile_tn(char* smem, unsigned lds0, const StagerTN& S, int t, const unsigned (&a_off)[4],
                        const unsigned (&b_off)[2], FragTN (&a)[8], FragTN (&b0)[4], FragTN (&b1)[4], f32x4 (&acc)[8][4],
                        bool do_cs, int wc, float (&cs)[2]) {
    phase_tn<D, 1>(smem, lds0, S, t, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
    phase_tn<D, 2>(smem, lds0, S, t, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
    phase_tn<D, 3>(smem, lds0, S, t, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
    phase_tn<D, 4>(smem, lds0, S, t, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
}

// one 256x256 output tile (or split-K slab tile) of the TN product described by g; idx = (z * ntm + tm) * ntn + tn
SC_DEVICE void tn_tile(const GemmArgs& g, int idx, char* smem) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 15, lg = lane >> 4;

    const int tn = idx % g.ntn;
    idx /= g.ntn;
    const int tm = idx % g.ntm;
    const int z = idx / g.ntm;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    StagerTN S;
    S.nt = (kend - kbeg) / BK;
    S.wave = wave;
    S.step[0] = S.step[3] = (long long)BK * g.lda * 2;
    S.step[1] = S.step[2] = (long long)BK * g.ldb * 2;
    {
        // piece p = k rows [32 p + 4 wave, + 4): rows kr and kr + 32 share the swizzle term (bits 0, 1, 3 of k)
        const int kr = wave * 4 + (lane >> 4);                          // k row of the half-tile image, 256 B per row
        const int s = (kr & 3) | (((kr >> 3) & 1) << 2);
        const int c = ((((lane & 15) >> 1) ^ s) << 4) + (lane & 1) * 8;  // logical column held at physical lane&15
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ca = min(m0 + h * 128 + c, g.M - 8), cb = min(n0 + h * 128 + c, g.N - 8);
            S.off[h ? 3 : 0] = ((unsigned)kr * (unsigned)g.lda + (unsigned)ca) * 2u;
            S.off[h ? 2 : 1] = ((unsigned)kr * (unsigned)g.ldb + (unsigned)cb) * 2u;
        }
    }
    S.base[0] = S.base[3] = reinterpret_cast<const char*>(g.A + (size_t)kbeg * g.lda);
    S.base[1] = S.base[2] = reinterpret_cast<const char*>(g.B + (size_t)kbeg * g.ldb);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    unsigned a_off[4], b_off[2];
    {
        const int q = li >> 2, p = li & 3;
        const int s = q | ((lg & 1) << 2);
        const int row = (lg * 8 + q) * 256 + p * 8;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) a_off[ii] = row + (((wr * 4 + ii) ^ s) << 5);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) b_off[jj] = row + (((wc * 2 + jj) ^ s) << 5);
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_cs = g.colsum != nullptr && tn == 0;
    float cs[2] = {0.f, 0.f};

#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int ts = s >> 2, q = s & 3;
        if (ts < S.nt) {
            const char* kb = S.base[q] + ts * S.step[q];
            dma16(kb + S.off[q], smem + slot(ts & 1, q) + wave * 1024);
            dma16(kb + (S.step[q] >> 1) + S.off[q], smem + slot(ts & 1, q) + (8 + wave) * 1024);
        }
    }
    if (S.nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    FragTN a[8], b0[4], b1[4];
    for (int kt = 0; kt < S.nt; kt += 2) {
        ktile_tn<0>(smem, lds0, S, kt, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
        if (kt + 1 < S.nt) ktile_tn<1>(smem, lds0, S, kt + 1, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    if (do_cs) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v = cs[h];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int m = m0 + h * 128 + wr * 64 + wc * 16 + li;
            if (lg == 0 && m < g.M) g.colsum[(size_t)z * g.M + m] = v;
        }
    }
    float* ep = reinterpret_cast<float*>(smem) + wave * 64 * SC_EPI_LD;
    EpiRegs<SC_EPI_F32> er;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sc_epi_put(ep, i, j, li, lg, acc[h * 4 + i][j]);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        sc_epilogue_store<SC_EPI_F32>(ep, er, m0 + h * 128 + wr * 64, n0 + wc * 32, lane, g, z, -1, 64, 128 - 32);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(512, 2) void gemm8p_tn_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tn_tile(g, sc_xcd_remap(blockIdx.x, gridDim.x), smem);
}

// Several weight gradients that share the reduction length (one Linear's token axis) in ONE launch: problem p owns the
// remapped block ids [first[p], first[p + 1]).  Why (round 4): the four weight gradients of a transformer block were four
// launches of about one round of workgroups each; out_proj (9 output tiles) needed split-K 28 to fill the chip -- 66 MB of
// fp32 slabs for a 2.4 MB result and 28-K-tile workgroups that are mostly prologue and epilogue (820 TFLOP/s against
// 1 100-1 190 for its siblings).  Grouped, every problem runs at the group's split-K.
struct GemmGroup {
    GemmArgs g[SC_WGRAD_GROUP_MAX];
    int first[SC_WGRAD_GROUP_MAX + 1];
    int n;
};
__global__ __launch_bounds__(512, 2) void gemm8p_tn_group_kernel(const GemmGroup gg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int idx = sc_xcd_remap(blockIdx.x, gridDim.x);
    int p = 0;
    while (p + 1 < gg.n && idx >= gg.first[p + 1]) ++p;
    tn_tile(gg.g[p], idx - gg.first[p], smem);
}

int launch_tn(const GemmArgs& g, int nblocks, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_tn_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done = true;
    }
    gemm8p_tn_kernel<<<nblocks, 512, LDS_BYTES, st>>>(g);
    SC_LAUNCH_CHECK();
    return 1;
}

// =====================================================================================================================
// FP8 (OCP e4m3) TN variant (round 4): weight gradients dW[m, n] = sum_k A8[k, m] B8[k, n] with both operands token-major bytes
// quantised with ONE scale per tensor (the reduction runs over the tokens, so a per-token scale cannot be pulled out of it).
//   K tile = 128 tokens; half-tile image = [128 k][128 columns] bytes (16 KiB, 128-B rows): the ring, the DMA piece count and the
//   phase schedule are byte-for-byte those of the bf16 TN kernel above, with twice the k per tile.
//   Fragments: the MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) wants, in lane (g = lane >> 4, r = lane & 15), the 32 consecutive k
//   [32 g, 32 g + 32) of output row r, one per byte.  ds_read_b64_tr_b8 (layout probed with tools/micro/tr_b8_layout.hip: lane
//   2 q + p of a 16-lane group supplies the address of (row q, byte columns 8 p .. 8 p + 7); lane i receives column i of rows
//   0 .. 7) delivers 8 of them: four reads per fragment at k offsets 0, 8, 16, 24 (+1024 B each in the image).
//   Swizzle: 16-byte chunk ^= ((k >> 1) & 3) | ((k >> 5) & 1) << 2 -- the four same-parity rows of an 8-row block and the two
//   row blocks a 32-lane half reads land on eight different 16-byte bank windows; the per-lane swizzle term does not depend on
//   which of the four reads it is, so they are immediate offsets off one address.
//   C = a_scale_inv * b_scale_inv * sum (fp32 slabs as above); the fused bias gradient sums the e4m3 A fragments (x a_scale_inv).
// Block scales of the f8f6f4 MFMA.  Unit scales two ways: E8M0 127 (= 2^0) in the scale registers of the SCALED opcode
// (v_mfma_scale_..., a 16-byte encoding that loads the scales in front of every MFMA), or the constant 0, for which the compiler
// selects the UNSCALED opcode v_mfma_f32_16x16x128_f8f6f4 (8-byte encoding, no scale load; the scales are implicitly one).
// -DSC_F8_SCALED_OPCODE keeps the first form (A/B; tests/test_gpu_fp8.py's exact-integer tests pin the arithmetic of either).
#ifdef SC_F8_SCALED_OPCODE
#define SC_F8_UNIT_SCALE 0x7F7F7F7F
#else
#define SC_F8_UNIT_SCALE 0
#endif
typedef __attribute__((ext_vector_type(8))) int i32x8t;
struct FragT8 {
    union {
        i32x8t v;               // the MFMA operand: 8 consecutive registers
        u32x2 q[4];             // k blocks 0..7, 8..15, 16..23, 24..31 of this lane's group (one transposed read each)
    };
};
template <int OFF>
SC_DEVICE u32x2 tr8_asm(unsigned lds_addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF) : "memory");
    return r;
}
// Q = half-tile index inside one ring parity: its byte offset (<= 48 KiB) rides in the instruction's 16-bit offset field, so a
// fragment needs ONE address register per ring parity instead of one per ring slot (with the slot folded into the address the
// compiler keeps ~24 hoisted address registers and spills)
template <int Q>
SC_DEVICE void t8_read(unsigned addr, FragT8& f) {
    f.q[0] = tr8_asm<Q * HALF>(addr);
    f.q[1] = tr8_asm<Q * HALF + 1024>(addr);
    f.q[2] = tr8_asm<Q * HALF + 2048>(addr);
    f.q[3] = tr8_asm<Q * HALF + 3072>(addr);
}
SC_DEVICE void t8_wait2(FragT8 (&f)[2]) {          // retire every register the asm reads above wrote (see tn_wait4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].v), "+v"(f[1].v) :: "memory");
}
SC_DEVICE void t8_wait4(FragT8 (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].v), "+v"(f[1].v), "+v"(f[2].v), "+v"(f[3].v) :: "memory");
}
SC_DEVICE i32x8t t8_cat(const FragT8& f) { return f.v; }
SC_DEVICE float t8_sum(const FragT8& f) {          // sum of this lane's 32 e4m3 values
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int v = (int)f.q[j][w];
            s += __builtin_amdgcn_cvt_f32_fp8(v, 0) + __builtin_amdgcn_cvt_f32_fp8(v, 1) + __builtin_amdgcn_cvt_f32_fp8(v, 2) +
                 __builtin_amdgcn_cvt_f32_fp8(v, 3);
        }
    return s;
}

struct StagerT8 {
    // wave-uniform base (SGPRs) + 32-bit per-lane offset: the DMA takes the scalar-base addressing form and the eight source
    // addresses cost 8 VGPRs instead of 16 (with 64-bit pointers the kernel spilled them and reloaded one before every DMA)
    const unsigned char* base[4];
    unsigned off[4];            // piece 0 of each half-tile kind; piece 1 is 64 k rows further: a scalar addend (step / 2)
    long long step[4];          // bytes per K tile (128 source rows) for each half-tile kind
    int nt;
    int wave;
};

template <int D, int PH>
SC_DEVICE void phase_t8(char* smem, unsigned lds0, const StagerT8& S, int t, const unsigned (&a_off)[4], const unsigned (&b_off)[2],
                        FragT8 (&a)[4], FragT8 (&b0)[2], FragT8 (&b1)[2], f32x4 (&acc)[8][4], bool do_cs, int wc, float (&cs)[2]) {
    constexpr unsigned pd = D * 4 * HALF;               // ring parity base, part of the (hoisted) address
    if (PH == 1) { t8_read<1>(lds0 + pd + b_off[0], b0[0]); t8_read<1>(lds0 + pd + b_off[1], b0[1]); }
    if (PH == 2) { t8_read<2>(lds0 + pd + b_off[0], b1[0]); t8_read<2>(lds0 + pd + b_off[1], b1[1]); }
    if (PH == 1) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) t8_read<0>(lds0 + pd + a_off[ii], a[ii]);
    }
    if (PH == 3) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) t8_read<3>(lds0 + pd + a_off[ii], a[ii]);
    }
    constexpr int q = (PH + 1) & 3;
    constexpr int DS = PH <= 2 ? (D ^ 1) : D;
    const int ts = t + (PH <= 2 ? 1 : 2);
    if (ts < S.nt) {
        const unsigned char* kb = S.base[q] + ts * S.step[q];
        unsigned o0 = S.off[q];
        asm volatile("" : "+v"(o0));                      // keep (scalar base + 32-bit offset): a hoisted 64-bit sum costs 2 VGPRs each
        dma16(kb + o0, smem + slot(DS, q) + S.wave * 1024);
        dma16(kb + (S.step[q] >> 1) + o0, smem + slot(DS, q) + (8 + S.wave) * 1024);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (PH == 1) { t8_wait2(b0); t8_wait4(a); }
    if (PH == 2) t8_wait2(b1);
    if (PH == 3) t8_wait4(a);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int mi = PH >= 3 ? 1 : 0;
    constexpr int nj = (PH == 2 || PH == 3) ? 1 : 0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const i32x8t af = t8_cat(a[ii]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            acc[mi * 4 + ii][nj * 2 + jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                t8_cat(nj ? b1[jj] : b0[jj]), af, acc[mi * 4 + ii][nj * 2 + jj], 0, 0, 0, SC_F8_UNIT_SCALE, 0, SC_F8_UNIT_SCALE);
        if ((PH == 1 || PH == 3) && do_cs && ii == wc) cs[mi] += t8_sum(a[ii]);     // fused bias gradient, as in the bf16 kernel
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int D>
SC_DEVICE void ktile_t8(char* smem, unsigned lds0, const StagerT8& S, int t, const unsigned (&a_off)[4], const unsigned (&b_off)[2],
                        FragT8 (&a)[4], FragT8 (&b0)[2], FragT8 (&b1)[2], f32x4 (&acc)[8][4], bool do_cs, int wc, float (&cs)[2]) {
    phase_t8<D, 1>(smem, lds0, S, t, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
    phase_t8<D, 2>(smem, lds0, S, t, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
    phase_t8<D, 3>(smem, lds0, S, t, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
    phase_t8<D, 4>(smem, lds0, S, t, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
}

// g.A / g.B: e4m3 bytes [K tokens][M] / [K][N], lda / ldb in BYTES; g.K tokens (k_per_split a multiple of 128);
// g.a_scale / g.b_scale: ONE dequantisation factor each (device scalars)
__global__ __launch_bounds__(512, 2) void gemm8p_tn_f8_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK8 = 128;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 15, lg = lane >> 4;

    int idx = sc_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = idx % g.ntn;
    idx /= g.ntn;
    const int tm = idx % g.ntm;
    const int z = idx / g.ntm;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const unsigned char* A8 = reinterpret_cast<const unsigned char*>(g.A);
    const unsigned char* B8 = reinterpret_cast<const unsigned char*>(g.B);

    StagerT8 S;
    S.nt = (kend - kbeg) / BK8;
    S.wave = wave;
    S.step[0] = S.step[3] = (long long)BK8 * g.lda;
    S.step[1] = S.step[2] = (long long)BK8 * g.ldb;
    {
        // piece p of a half-tile = k rows [8 (8 p + wave), + 8): rows kr and kr + 64 have the same swizzle term (bits 1, 2, 5 of k)
        const int kr = wave * 8 + (lane >> 3);                           // k row of the half-tile image, 128 B per row
        const int sw = ((kr >> 1) & 3) | (((kr >> 5) & 1) << 2);
        const int c = ((lane & 7) ^ sw) << 4;                            // logical byte column held at physical chunk lane & 7
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ca = min(m0 + h * 128 + c, g.M - 16), cb = min(n0 + h * 128 + c, g.N - 16);
            S.off[h ? 3 : 0] = (unsigned)kr * (unsigned)g.lda + (unsigned)ca;
            S.off[h ? 2 : 1] = (unsigned)kr * (unsigned)g.ldb + (unsigned)cb;
        }
    }
    S.base[0] = S.base[3] = A8 + (size_t)kbeg * g.lda;
    S.base[1] = S.base[2] = B8 + (size_t)kbeg * g.ldb;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    unsigned a_off[4], b_off[2];
    {
        const int sw = ((li >> 2) & 3) | ((lg & 1) << 2);                // swizzle term of rows 32 lg + 8 j + (li >> 1), any j
        const int row = (32 * lg + (li >> 1)) * 128 + (li & 1) * 8;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) a_off[ii] = row + (((wr * 4 + ii) ^ sw) << 4);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) b_off[jj] = row + (((wc * 2 + jj) ^ sw) << 4);
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_cs = g.colsum != nullptr && tn == 0;
    float cs[2] = {0.f, 0.f};

#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int ts = s >> 2, q = s & 3;
        if (ts < S.nt) {
            const unsigned char* kb = S.base[q] + ts * S.step[q];
            dma16(kb + S.off[q], smem + slot(ts & 1, q) + wave * 1024);
            dma16(kb + (S.step[q] >> 1) + S.off[q], smem + slot(ts & 1, q) + (8 + wave) * 1024);
        }
    }
    if (S.nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    FragT8 a[4], b0[2], b1[2];
    for (int kt = 0; kt < S.nt; kt += 2) {
        ktile_t8<0>(smem, lds0, S, kt, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
        if (kt + 1 < S.nt) ktile_t8<1>(smem, lds0, S, kt + 1, a_off, b_off, a, b0, b1, acc, do_cs, wc, cs);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    const float sa = g.a_scale ? *g.a_scale : 1.0f, sb = g.b_scale ? *g.b_scale : 1.0f;
    if (do_cs) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v = cs[h];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int m = m0 + h * 128 + wr * 64 + wc * 16 + li;
            if (lg == 0 && m < g.M) g.colsum[(size_t)z * g.M + m] = v * sa;
        }
    }
    const float sab = sa * sb;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] *= sab;
    float* ep = reinterpret_cast<float*>(smem) + wave * 64 * SC_EPI_LD;
    EpiRegs<SC_EPI_F32> er;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sc_epi_put(ep, i, j, li, lg, acc[h * 4 + i][j]);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        sc_epilogue_store<SC_EPI_F32>(ep, er, m0 + h * 128 + wr * 64, n0 + wc * 32, lane, g, z, -1, 64, 128 - 32);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

// =====================================================================================================================
// FP8 (OCP e4m3) NT variant: the same tile, ring, phase schedule and epilogues on one-byte operands.  A half-tile row
// is still 128 bytes, i.e. 128 k values instead of 64, so the staging stream, swizzle and waits are byte-for-byte the
// bf16 kernel's; the caller passes K / 2, lda / 2, ldb / 2 ("bf16 elements") and the MFMA section issues ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 per fragment pair where the bf16 kernel issues two 16x16x32: half the MFMA count
// for twice the k per tile = 2x the matrix rate.  Operand layout (probed with integer data, tools/micro/
// mfma_fp8_layout.hip): lane (g = lane >> 4, r = lane & 15) supplies row r and the 32 consecutive k of bytes
// [32 g, 32 g + 32) of the 128-byte row = the two 16-byte chunks 2g, 2g + 1; the block scales are E8M0 1.0 -- the real
// scales are per-row floats applied to the accumulators before the epilogue:
//     C[m][n] = a_scale[m] * b_scale[n] * sum_k A8[m][k] B8[n][k]   (+ bias, residual, GELU as in the bf16 kernel).
typedef __attribute__((ext_vector_type(8))) int i32x8;
struct Frag8 {
    union { i32x8 v; u32x4 h[2]; };
};

template <int D, int PH>
SC_DEVICE void phase_f8(char* smem, const Stager& S, int t, const int (&a_off)[2], const int (&b_off)[2], Frag8 (&a)[4],
                        Frag8 (&b0)[2], Frag8 (&b1)[2], f32x4 (&acc)[8][4]) {
    if (PH == 1) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                b0[jj].h[kk] = *reinterpret_cast<const u32x4*>(smem + slot(D, 1) + b_off[kk] + jj * 2048);
    }
    if (PH == 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                b1[jj].h[kk] = *reinterpret_cast<const u32x4*>(smem + slot(D, 2) + b_off[kk] + jj * 2048);
    }
    if (PH == 1 || PH == 3) {
        constexpr int sl = slot(D, PH == 1 ? 0 : 3);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                a[ii].h[kk] = *reinterpret_cast<const u32x4*>(smem + sl + a_off[kk] + ii * 2048);
    }
    constexpr int q = (PH + 1) & 3;
    constexpr int DS = PH <= 2 ? (D ^ 1) : D;
    const int ts = t + (PH <= 2 ? 1 : 2);
    if (ts < S.nt) {
        dma16(S.src[q][0] + (size_t)ts * BK, smem + slot(DS, q) + S.wave * 1024);
        dma16(S.src[q][1] + (size_t)ts * BK, smem + slot(DS, q) + (8 + S.wave) * 1024);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    constexpr int mi = PH >= 3 ? 1 : 0;
    constexpr int nj = (PH == 2 || PH == 3) ? 1 : 0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            acc[mi * 4 + ii][nj * 2 + jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                nj ? b1[jj].v : b0[jj].v, a[ii].v, acc[mi * 4 + ii][nj * 2 + jj], 0, 0, 0, SC_F8_UNIT_SCALE, 0, SC_F8_UNIT_SCALE);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int D>
SC_DEVICE void ktile_f8(char* smem, const Stager& S, int t, const int (&a_off)[2], const int (&b_off)[2], Frag8 (&a)[4],
                        Frag8 (&b0)[2], Frag8 (&b1)[2], f32x4 (&acc)[8][4]) {
    phase_f8<D, 1>(smem, S, t, a_off, b_off, a, b0, b1, acc);
    phase_f8<D, 2>(smem, S, t, a_off, b_off, a, b0, b1, acc);
    phase_f8<D, 3>(smem, S, t, a_off, b_off, a, b0, b1, acc);
    phase_f8<D, 4>(smem, S, t, a_off, b_off, a, b0, b1, acc);
}

// g.K / lda / ldb are in 2-byte units (see above); g.a_scale [M] and g.b_scale [N] are the dequantisation factors
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm8p_f8_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 15, lg = lane >> 4;

    int idx = sc_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = idx % g.ntn;
    const int tm = idx / g.ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    Stager S;
    S.nt = g.K / BK;
    S.wave = wave;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = (p * 8 + wave) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ga = min(m0 + (r >> 6) * 128 + h * 64 + (r & 63), g.M - 1);
            const int gb = min(n0 + (r >> 5) * 64 + h * 32 + (r & 31), g.N - 1);
            S.src[h ? 3 : 0][p] = g.A + (size_t)ga * g.lda + lc * 8;
            S.src[h ? 2 : 1][p] = g.B + (size_t)gb * g.ldb + lc * 8;
        }
    }
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        // Which two 16-byte chunks of the 128-byte row lane group lg takes is free -- the MFMA sums over all 128 k, and A and B
        // fragments use the same map -- but not for the LDS: a ds_read_b128 is served in groups of 16 lanes
        // ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS table) that mix rows of lg and lg + 1, and with the staging
        // swizzle chunk ^ ((row >> 1) & 7) the obvious map 2 lg + kk put both halves of such a group on the same bank windows:
        // SQ_LDS_BANK_CONFLICT 29.5 M cycles against the bf16 kernel's 4.2 M on the same bytes (profiles/r06_fp8_ktile_probe.txt).
        // Conflict-free needs c(lg, kk) ^ c(lg ^ 1, kk) in {1, 6, 7}: c = 4 (lg >> 1) + 2 kk + (lg & 1).
        const int coff = (((lg >> 1) * 4 + 2 * kk + (lg & 1)) ^ ((li >> 1) & 7)) << 4;
        a_off[kk] = (wr * 64 + li) * 128 + coff;
        b_off[kk] = (wc * 32 + li) * 128 + coff;
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int ts = s >> 2, q = s & 3;
        if (ts < S.nt) {
            dma16(S.src[q][0] + (size_t)ts * BK, smem + slot(ts & 1, q) + wave * 1024);
            dma16(S.src[q][1] + (size_t)ts * BK, smem + slot(ts & 1, q) + (8 + wave) * 1024);
        }
    }
    if (S.nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    Frag8 a[4], b0[2], b1[2];
    for (int kt = 0; kt < S.nt; kt += 2) {
        ktile_f8<0>(smem, S, kt, a_off, b_off, a, b0, b1, acc);
        if (kt + 1 < S.nt) ktile_f8<1>(smem, S, kt + 1, a_off, b_off, a, b0, b1, acc);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // dequantise: lane owns C[m = .. + 16 i + li][n = .. + 16 j + 4 lg .. + 3]
    {
        const int mrow = m0 + wr * 128 + li, ncol = n0 + wc * 64 + lg * 4;
        f32x4 sb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sb[j] = (f32x4){1.f, 1.f, 1.f, 1.f};
            if (g.b_scale && ncol + j * 16 < g.N) sb[j] = *reinterpret_cast<const f32x4*>(g.b_scale + ncol + j * 16);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float sa = g.a_scale ? g.a_scale[g.a_scale_scalar ? 0 : min(mrow + i * 16, g.M - 1)] : 1.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= sb[j] * sa;
        }
    }
    if (EPI == SC_EPI_GELU_GRAD_PAIR && g.gelu_lut != nullptr) {     // GELU by table, as in gemm8p_kernel
        unsigned* lut = reinterpret_cast<unsigned*>(smem + 8 * 4096);
        for (int c = t; c < SC_GELU_LUT_N / 4; c += 512)
            reinterpret_cast<u32x4*>(lut)[c] = reinterpret_cast<const u32x4*>(g.gelu_lut)[c];
        __syncthreads();
        epilogue_bf16_lds<EPI, true, true>(acc, g, smem + wave * 4096, m0 + wr * 128, n0 + wc * 64, lane, lut);
    } else if (EPI == SC_EPI_BF16 || EPI == SC_EPI_BF16_BIAS || sc_epi_gelu_fwd(EPI)) {
        epilogue_bf16_lds<EPI, true>(acc, g, smem + wave * 4096, m0 + wr * 128, n0 + wc * 64, lane);
    } else {
        const int mw = wr * 128;
        float* ep = reinterpret_cast<float*>(smem) + wave * 64 * SC_EPI_LD;
        EpiRegs<EPI> er;
        float amax_lane = 0.f;
        sc_epi_load<EPI>(er, m0 + mw, n0 + wc * 64, lane, g, 64);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) sc_epi_put(ep, i, j, li, lg, acc[h * 4 + i][j]);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            sc_epilogue_store<EPI, true>(ep, er, m0 + mw + h * 64, n0 + wc * 64, lane, g, 0, (h + 1 < 2) ? m0 + mw + 64 : -1,
                                         64, 0, &amax_lane);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
        if (sc_epi_aux_mul(EPI) && g.q8) sc_amax_publish(amax_lane, g.q8_amax);
    }
}

template <int EPI>
int launch_f8(const GemmArgs& g, int nblocks, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_f8_kernel<EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done = true;
    }
    gemm8p_f8_kernel<EPI><<<nblocks, 512, LDS_BYTES, st>>>(g);
    SC_LAUNCH_CHECK();
    return 1;
}

}  // namespace

// fp8 NT GEMM: g.A / g.B point at e4m3 bytes, g.K / lda / ldb already halved ("2-byte units"); 1 = launched
int sc_gemm8p_fp8(int epi, GemmArgs& g, hipStream_t st) {
    if (g.M < 1 || g.N < 8 || (g.K % BK) != 0) return 0;
    g.ntm = (g.M + BM - 1) / BM;
    g.ntn = (g.N + BN - 1) / BN;
    g.splitk = 1;
    g.k_per_split = g.K;
    g.slab_stride = 0;
    const int nblocks = g.ntm * g.ntn;
    if (epi == SC_EPI_GELU_GRAD_PAIR) {
        const char* sw = getenv("SC_GELU_LUT");
        if (!(sw && sw[0] == '0')) g.gelu_lut = sc_gelu_lut_device(st, g.act);
    }
    if (epi == SC_EPI_BF16) return launch_f8<SC_EPI_BF16>(g, nblocks, st);
    if (epi == SC_EPI_BF16_BIAS) return launch_f8<SC_EPI_BF16_BIAS>(g, nblocks, st);
    if (epi == SC_EPI_F32_BIAS_RES) return launch_f8<SC_EPI_F32_BIAS_RES>(g, nblocks, st);
    if (epi == SC_EPI_GELU_PAIR) return launch_f8<SC_EPI_GELU_PAIR>(g, nblocks, st);
    if (epi == SC_EPI_BF16_DGELU) return launch_f8<SC_EPI_BF16_DGELU>(g, nblocks, st);
    if (epi == SC_EPI_F32) return launch_f8<SC_EPI_F32>(g, nblocks, st);
    if (epi == SC_EPI_BF16_BIAS_RES) return launch_f8<SC_EPI_BF16_BIAS_RES>(g, nblocks, st);
    if (epi == SC_EPI_GELU_GRAD_PAIR) return launch_f8<SC_EPI_GELU_GRAD_PAIR>(g, nblocks, st);
    if (epi == SC_EPI_BF16_MUL_AUX) return launch_f8<SC_EPI_BF16_MUL_AUX>(g, nblocks, st);
    return 0;
}

// Grouped TN launch (sc_gemm_wgrad_group): every g[p] describes dW_p[M_p, N_p] = A_p[K, M_p]^T . B_p[K, N_p] with the SAME K;
// C / slab_stride / colsum are set by the caller for the split-K factor returned by sc_gemm8p_tn_group_plan.
// 1 = launched, 0 = some problem is outside the 256x256 kernel's range (the caller runs the problems one by one).
int sc_gemm8p_tn_group_plan(const GemmArgs* g, int n, int splitk_req, int* splitk_out, int* k_per_split) {
    if (n < 1 || n > SC_WGRAD_GROUP_MAX) return 0;
    const int K = g[0].K;
    if ((K % BK) != 0) return 0;
    for (int p = 0; p < n; ++p) {
        if (g[p].K != K || g[p].M < 256 || g[p].N < 192 || (g[p].M % 8) != 0 || (g[p].N % 8) != 0) return 0;
        if ((long long)g[p].M * g[p].N < 256LL * 256 * 8 || g[p].ldc != g[p].N) return 0;
    }
    const int ktiles = K / BK;
    int splitk = splitk_req < 1 ? 1 : splitk_req;
    if (splitk > ktiles) splitk = ktiles;
    const int tiles_per = (ktiles + splitk - 1) / splitk;
    *splitk_out = (ktiles + tiles_per - 1) / tiles_per;
    *k_per_split = tiles_per * BK;
    return 1;
}
int sc_gemm8p_tn_group_launch(const GemmArgs* g, int n, hipStream_t st) {
    GemmGroup gg;
    gg.n = n;
    int total = 0;
    for (int p = 0; p < n; ++p) {
        gg.g[p] = g[p];
        gg.first[p] = total;
        total += g[p].ntm * g[p].ntn * g[p].splitk;
    }
    for (int p = n; p <= SC_WGRAD_GROUP_MAX; ++p) gg.first[p] = total;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_tn_group_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done = true;
    }
    gemm8p_tn_group_kernel<<<total, 512, LDS_BYTES, st>>>(gg);
    SC_LAUNCH_CHECK();
    return 1;
}

// fp8 TN weight gradient: g.A / g.B e4m3 bytes [K][M] / [K][N] (lda / ldb in bytes), g.a_scale / g.b_scale device scalars, g.C fp32;
// split-K through `slabs` exactly as the bf16 TN path.  1 = launched, 0 = shape outside the kernel's range.
int sc_gemm8p_tn_fp8(GemmArgs& g, int splitk_req, float* slabs, hipStream_t st) {
    if (g.M < 256 || g.N < 192 || (g.K % 128) != 0 || (g.M % 16) != 0 || (g.N % 16) != 0) return 0;
    if ((g.lda % 16) != 0 || (g.ldb % 16) != 0) return 0;
    g.ntm = (g.M + BM - 1) / BM;
    g.ntn = (g.N + BN - 1) / BN;
    const int ktiles = g.K / 128;
    int splitk = splitk_req < 1 ? 1 : splitk_req;
    if (slabs == nullptr) splitk = 1;
    if (splitk > ktiles) splitk = ktiles;
    const int tiles_per = (ktiles + splitk - 1) / splitk;
    splitk = (ktiles + tiles_per - 1) / tiles_per;
    g.splitk = splitk;
    g.k_per_split = tiles_per * 128;
    g.slab_stride = 0;
    if (splitk > 1) {
        if (g.ldc != g.N) return 0;
        g.C = slabs;
        g.slab_stride = (long long)g.M * g.N;
    }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_tn_f8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_BYTES);
        attr_done = true;
    }
    gemm8p_tn_f8_kernel<<<g.ntm * g.ntn * splitk, 512, LDS_BYTES, st>>>(g);
    SC_LAUNCH_CHECK();
    return 1;
}

// Which launches take the column-group walk by default (measured per launch class: profiles/r05_gemm_colgroup.txt).
static int sc_colgroup_default(int epi, const GemmArgs& g) {
    (void)epi; (void)g;
    return 0;
}

int sc_gemm8p_try(int mode, int epi, GemmArgs& g, int splitk_req, float* slabs, hipStream_t st) {
    if (g.M < 256 || g.N < 192 || (g.K % BK) != 0) return 0;
    if (mode == SC_GEMM_TN && (epi != SC_EPI_F32 || (g.M % 8) != 0 || (g.N % 8) != 0)) return 0;
    if ((long long)g.M * g.N < 256LL * 256 * 8) return 0;
    // NT launches with fewer 256x256 tiles than ~0.4 of the chip's CUs go to the 128x128 general kernel: four times the
    // workgroups, two of them per CU.  Measured (tools/bench_small_m.py, profiles/r06_small_m_kernel_choice.txt; the token
    // counts of ViT-B-32 + CLIP text tower at the reference's batch 32): 20-84 tiles 1.08-1.52x faster on the small kernel,
    // 150 tiles and more 1.1-1.5x faster here.  SC_GEMM_SMALL_TILES=<n> moves the threshold (0: never).
    if (mode == SC_GEMM_NT && splitk_req <= 1) {
        static const int small_tiles = getenv("SC_GEMM_SMALL_TILES") ? atoi(getenv("SC_GEMM_SMALL_TILES")) : 100;
        if ((long long)g.M * g.N < 256LL * 256 * small_tiles) return 0;
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
    if (mode == SC_GEMM_TN) {
        return launch_tn(g, nblocks, st);
    }
    // GELU by table: only the non-persistent kernel has LDS to spare for it (the persistent one fills all 160 KiB)
    if (epi == SC_EPI_GELU_GRAD_PAIR) {
        const char* sw = getenv("SC_GELU_LUT");                  // read per call: A/B switch
        if (!(sw && sw[0] == '0')) g.gelu_lut = sc_gelu_lut_device(st, g.act);
    }
    // persistent walk of the tile list for the store-only bf16 epilogues once there is more than one round of tiles
    static const bool persist = !(getenv("SC_GEMM_PERSIST") && getenv("SC_GEMM_PERSIST")[0] == '0');
    if (g.gelu_lut == nullptr && persist && splitk == 1 && nblocks >= 1024 && ktiles >= 3) {      // >= 4 rounds of tiles (measured: +7 % at 7 rounds, -3 % at 2.3)
        if (epi == SC_EPI_BF16) return launch_persistent<SC_EPI_BF16>(g, nblocks, st);
        if (epi == SC_EPI_BF16_BIAS) return launch_persistent<SC_EPI_BF16_BIAS>(g, nblocks, st);
        if (epi == SC_EPI_GELU_PAIR) return launch_persistent<SC_EPI_GELU_PAIR>(g, nblocks, st);      // +1.5 % at 9.2 rounds
        if (epi == SC_EPI_GELU_GRAD_PAIR) return launch_persistent<SC_EPI_GELU_GRAD_PAIR>(g, nblocks, st);
    }
    // Tile walk of the non-persistent kernel: SC_GEMM_COLGROUP="<epi>:<Gc>[,<epi>:<Gc>...]" (A/B switch, read per call)
    // walks the named epilogues' launches in column groups of Gc tiles inside per-XCD row bands (sc_tile_colgroup).
    g.col_group = 0;
    if (splitk == 1 && g.ntn > 1) {
        int gc = sc_colgroup_default(epi, g);
        if (const char* sw = getenv("SC_GEMM_COLGROUP")) {
            for (const char* p = sw; *p;) {
                char* e = nullptr;
                const long ep = strtol(p, &e, 10);
                if (e == p || *e != ':') break;
                const long v = strtol(e + 1, &e, 10);
                if (ep == epi || ep == -1) gc = (int)v;
                p = (*e == ',') ? e + 1 : e;
                if (*e != ',') break;
            }
        }
        if (gc > 0 && gc < g.ntn) {
            g.col_group = gc;
            g.band_rows = (g.ntm + 7) / 8;                       // an XCD's contiguous share of the remapped tile list
        }
    }
    int rc = 0;
#define SC_CASE(EPI) \
    if (epi == EPI) rc = launch<EPI>(g, nblocks, st);
    SC_CASE(SC_EPI_BF16)
    SC_CASE(SC_EPI_BF16_BIAS)
    SC_CASE(SC_EPI_F32_BIAS_RES)
    SC_CASE(SC_EPI_GELU_PAIR)
    SC_CASE(SC_EPI_BF16_DGELU)
    SC_CASE(SC_EPI_F32)
    SC_CASE(SC_EPI_BF16_BIAS_RES)
    SC_CASE(SC_EPI_GELU_GRAD_PAIR)
    SC_CASE(SC_EPI_BF16_MUL_AUX)
#undef SC_CASE
    return rc;
}

// device copy of the GELU table, one per device and activation, filled on first use by the formula itself (sc_gemm_common.h)
const unsigned* sc_gelu_lut_device(hipStream_t st, int act) {
    static std::mutex mu;
    static std::map<int, unsigned*> all;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const int key = dev * 2 + (act ? 1 : 0);
    std::lock_guard<std::mutex> lock(mu);
    auto it = all.find(key);
    if (it != all.end()) return it->second;
    unsigned* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), SC_GELU_LUT_N * sizeof(unsigned)) != hipSuccess) p = nullptr;
    if (p) {
        sc_gelu_lut_fill_kernel<<<(SC_GELU_LUT_N + 255) / 256, 256, 0, st>>>(p, act ? 1 : 0);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {     // once per device: later launches may use any stream
            (void)hipFree(p);
            p = nullptr;
        }
    }
    all[key] = p;
    return p;
}
