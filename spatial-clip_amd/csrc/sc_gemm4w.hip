// 256x256x64 bf16 MFMA GEMM for the TN layout (weight gradients) with FOUR waves per workgroup, gfx950.
//
// Why a second TN kernel.  Both operands of a weight gradient are token-major, so every MFMA fragment leaves LDS through
// ds_read_b64_tr_b16 -- 8 bytes per lane, the widest 16-bit transposed read.  The 8-wave kernel (sc_gemm8p.hip,
// 128x64 per wave) issues 48 such reads per wave and 64-deep K tile, 384 per CU, for 2 x 1024 cycles of MFMA per SIMD:
// the LDS pipeline, not the matrix pipe, sets the pace (PMC: MFMA pipe 61 % busy against 76 % for the NT kernel).
// Here a wave owns 128x128 of the tile (2 x 2 waves): 64 reads per wave, 256 per CU for the same MFMA work (-33 %).
// The 256 accumulator registers fit because a 4-wave workgroup runs one wave per SIMD and each wave may use the whole
// 512-entry register file of its lane.
//
// With one wave per SIMD nothing else hides a wave's LDS latency, so the loop is software-pipelined inside the wave:
// a K tile is two 32-deep slices; while the 64 MFMAs of a slice run, the 32 transposed reads of the NEXT slice are
// issued between them (one read per two MFMAs) into the other fragment set.  One s_barrier per K tile, at the start of
// its second slice: by then the wave has read all of the current tile (so its ring slot may be restaged two tiles
// ahead) and has waited for its own DMA pieces of the next tile (so the first slice of the next tile can be read behind
// the barrier).  Operand images, swizzle and DMA source addressing are those of gemm8p_tn_kernel.
// Status: opt-in (SC_GEMM_TN4W=1).  1 174 vs 1 024 TFLOP/s at 4096^3 and +5 % on one ViT-B/16 weight-gradient shape in a
// loop, but no gain on the step's own sequence (cold operands) and -0.35 ms/step in the step: see sc_gemm8p.hip's dispatcher.
// A variant with two barriers per K tile and half-image staging groups (1.5-tile DMA lead) was slower (973 at 4096^3):
// with one wave per SIMD a barrier stalls the matrix pipe outright.
//   reference: autograd of nn.Linear (weight gradient dW = dY^T X), src/open_clip/transformer.py:238-300.
#include "sc_gemm_common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 128 * 64 * 2;                  // 16 KiB half-tile image: [64 k][128 columns]
constexpr int RING = 8 * HALF;                      // two K tiles x four half-tiles
constexpr int EPI4 = 4 * 64 * SC_EPI_LD * 4;        // fp32 staging of the epilogue, one 64x64 block per wave
constexpr int LDS4 = RING > EPI4 ? RING : EPI4;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

SC_DEVICE void dma16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}
// ring slot of half-tile q (0: A half 0, 1: B half 0, 2: B half 1, 3: A half 1) of the K tile with parity D
constexpr int slot(int D, int q) { return D * 4 * HALF + q * HALF; }

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
struct Frag {
    u32x2 lo, hi;
};

struct Stager4 {
    const bf16* src[4];         // [half-tile kind q]: this lane's source of piece `wave` of the image; pieces wave + 4 pp
                                // sit 16 pp source rows further down (same swizzle: the row offset is a multiple of 16)
    long long step[4];          // elements per K tile for each half-tile kind
    long long pstep[4];         // elements per 16 source rows
    int nt;
    int wave;
};

SC_DEVICE void stage_half(char* smem, const Stager4& S, int ts, int D, int q) {
#pragma unroll
    for (int pp = 0; pp < 4; ++pp)
        dma16(S.src[q] + ts * S.step[q] + pp * S.pstep[q], smem + slot(D, q) + (pp * 4 + S.wave) * 1024);
}

// The 256 accumulators are pinned to the AGPR half of the register file by an asm MFMA with a "+a" operand: left to the
// register allocator (builtin MFMA), accumulators and fragments migrated between the two halves -- ~900 v_accvgpr moves
// and 26 spills in the loop.  Volatile asm also fixes the issue order below exactly as written.
SC_DEVICE void mfma_acc(f32x4& acc, bf16x8 x, bf16x8 y) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));
}

// one 32-deep slice: 64 MFMAs on fragment set `cur`.  The 32 reads of the next slice go out during the FIRST 32 MFMAs
// (one read per MFMA: fragments x and x + 4 of both operands while m-tile x is computed), so that the last of them has
// 32 MFMAs (512 cycles) to land before the lgkmcnt(0) that opens the next slice.  KK = which slice of its K tile `nxt`
// is (0: rows 0-31 of the image, 1: rows 32-63 = +8192 bytes).  STAGE: also issue the 16 DMA pieces of K tile `ts`
// into ring half Ds, two behind each of the eight m-tiles.
template <int KK, bool STAGE>
SC_DEVICE void slice(unsigned abase, unsigned bbase, const unsigned (&off)[8],
                     const Frag (&ca)[8], const Frag (&cb)[8], Frag (&na)[8], Frag (&nb)[8], f32x4 (&acc)[8][8],
                     char* smem, const Stager4& S, int ts, int Ds, bool do_stage, unsigned cs_mask, float (&cs)[8]) {
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
        const bf16x8 af = tr_cat(ca[ii].lo, ca[ii].hi);
        // fused bias gradient: column sums of At over this slice for the 16-column groups this wave owns (cs_mask;
        // VALU work between the MFMAs)
        if (cs_mask & (1u << ii)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[ii] += (float)af[e];
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            if (ii < 4) {
                const int x = ii + (jj >> 2) * 4;
                if ((jj & 3) == 0) na[x].lo = tr16_asm<KK * 8192>(abase + off[x]);
                if ((jj & 3) == 1) na[x].hi = tr16_asm<KK * 8192 + 1024>(abase + off[x]);
                if ((jj & 3) == 2) nb[x].lo = tr16_asm<KK * 8192>(bbase + off[x]);
                if ((jj & 3) == 3) nb[x].hi = tr16_asm<KK * 8192 + 1024>(bbase + off[x]);
            }
            mfma_acc(acc[ii][jj], tr_cat(cb[jj].lo, cb[jj].hi), af);
        }
        if (STAGE && do_stage) {                         // two 1-KiB pieces behind each m-tile
            const int q = ii >> 1, pp = (ii & 1) * 2;
            dma16(S.src[q] + ts * S.step[q] + pp * S.pstep[q], smem + slot(Ds, q) + (pp * 4 + S.wave) * 1024);
            dma16(S.src[q] + ts * S.step[q] + (pp + 1) * S.pstep[q], smem + slot(Ds, q) + ((pp + 1) * 4 + S.wave) * 1024);
        }
    }
}

__global__ __launch_bounds__(256) void gemm4w_tn_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lg = lane >> 4;

    int idx = sc_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = idx % g.ntn;
    idx /= g.ntn;
    const int tm = idx % g.ntm;
    const int z = idx / g.ntm;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    Stager4 S;
    S.nt = (kend - kbeg) / BK;
    S.wave = wave;
    S.step[0] = S.step[3] = (long long)BK * g.lda;
    S.step[1] = S.step[2] = (long long)BK * g.ldb;
    S.pstep[0] = S.pstep[3] = 16LL * g.lda;
    S.pstep[1] = S.pstep[2] = 16LL * g.ldb;
    {
        // piece P = 4 pp + wave of an image = k rows [4 P', 4 P' + 4) with P' = P & 7 in the image half P >> 3, i.e. source
        // rows wave * 4 + 16 pp + (lane >> 4); the swizzle key of a row ignores multiples of 16
        const int kr = wave * 4 + (lane >> 4);
        const int s = (kr & 3) | (((kr >> 3) & 1) << 2);
        const int c = ((((lane & 15) >> 1) ^ s) << 4) + (lane & 1) * 8;  // logical column held at physical lane&15
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ca = min(m0 + h * 128 + c, g.M - 8), cb = min(n0 + h * 128 + c, g.N - 8);
            S.src[h ? 3 : 0] = g.A + (size_t)(kbeg + kr) * g.lda + ca;
            S.src[h ? 2 : 1] = g.B + (size_t)(kbeg + kr) * g.ldb + cb;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    unsigned off[8];                                     // fragment x of either operand inside its half-tile image
    {
        const int q = li >> 2, p = li & 3;
        const int s = q | ((lg & 1) << 2);
        const int row = (lg * 8 + q) * 256 + p * 8;
#pragma unroll
        for (int x = 0; x < 8; ++x) off[x] = row + ((x ^ s) << 5);
    }
    const int qa = wr ? 3 : 0, qb = wc ? 2 : 1;          // my A / B half-tile images

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // bias-gradient column sums: every (row-tile, 16-column group) has ONE owner among the workgroups of the tile row and
    // their waves -- group x of a wave row belongs to tile column x % ntn and to wave column (x / ntn) & 1 -- so the work
    // is spread over all workgroups instead of loading the tn == 0 ones (a single round: the slowest workgroup is the
    // kernel's time)
    unsigned cs_mask = 0;
    if (g.colsum != nullptr) {
        const int nn = g.ntn < 8 ? g.ntn : 8;
#pragma unroll
        for (int x = 0; x < 8; ++x)
            if (x % nn == tn && ((x / nn) & 1) == wc) cs_mask |= 1u << x;
    }
    cs_mask = __builtin_amdgcn_readfirstlane(cs_mask);
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // prologue: K tiles 0 and 1 on their way, tile 0 landed, its first slice in registers
#pragma unroll
    for (int q = 0; q < 4; ++q) stage_half(smem, S, 0, 0, q);
    if (S.nt > 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) stage_half(smem, S, 1, 1, q);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    Frag a0[8], b0[8], a1[8], b1[8];                     // slice 0 / slice 1 fragment sets
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        a0[x].lo = tr16_asm<0>(lds0 + slot(0, qa) + off[x]);
        a0[x].hi = tr16_asm<1024>(lds0 + slot(0, qa) + off[x]);
        b0[x].lo = tr16_asm<0>(lds0 + slot(0, qb) + off[x]);
        b0[x].hi = tr16_asm<1024>(lds0 + slot(0, qb) + off[x]);
    }

    for (int kt = 0; kt < S.nt; ++kt) {
        const int D = kt & 1;
        const unsigned cur_a = lds0 + slot(D, qa), cur_b = lds0 + slot(D, qb);
        // after the last K tile there is no next slice: the reads go to the current tile again (results unused) so that the
        // loop body has no branch around its 32 reads
        const int Dn = kt + 1 < S.nt ? (D ^ 1) : D;
        const unsigned nxt_a = lds0 + slot(Dn, qa), nxt_b = lds0 + slot(Dn, qb);
        // ---- slice 0 of tile kt; reads slice 1 of tile kt
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        slice<1, false>(cur_a, cur_b, off, a0, b0, a1, b1, acc, smem, S, 0, 0, false, cs_mask, cs);
        // ---- slice 1 of tile kt; behind the barrier: stage tile kt + 2 into this tile's slot, read slice 0 of tile kt + 1
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // my pieces of tile kt + 1 (issued one tile ago); tile kt read
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        slice<0, true>(nxt_a, nxt_b, off, a1, b1, a0, b0, acc, smem, S, kt + 2, D, kt + 2 < S.nt, cs_mask, cs);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");     // + the last MFMAs have written back
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                        // the ring is dead: its LDS becomes the epilogue staging
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int x = 0; x < 8; ++x) {
        if (cs_mask & (1u << x)) {
            float v = cs[x];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int m = m0 + wr * 128 + x * 16 + li;
            if (lg == 0 && m < g.M) g.colsum[(size_t)z * g.M + m] = v;
        }
    }
    // epilogue: four 64x64 fp32 blocks per wave through the wave's LDS staging (full-row-segment stores), slab z
    float* ep = reinterpret_cast<float*>(smem) + wave * 64 * SC_EPI_LD;
    EpiRegs<SC_EPI_F32> er;
#pragma unroll
    for (int hm = 0; hm < 2; ++hm)
#pragma unroll
        for (int hn = 0; hn < 2; ++hn) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) sc_epi_put(ep, i, j, li, lg, acc[hm * 4 + i][hn * 4 + j]);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            sc_epilogue_store<SC_EPI_F32>(ep, er, m0 + wr * 128 + hm * 64, n0 + wc * 128 + hn * 64, lane, g, z, -1, 64);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
}

}  // namespace

// TN, fp32 slabs / fp32 output, optional fused bias-gradient column sums: returns 1 if launched
int sc_gemm4w_tn(const GemmArgs& g, int nblocks, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
        attr_done = true;
    }
    gemm4w_tn_kernel<<<nblocks, 256, LDS4, st>>>(g);
    return hipGetLastError() == hipSuccess ? 1 : -2;
}
