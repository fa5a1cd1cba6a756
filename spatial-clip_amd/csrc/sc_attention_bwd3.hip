// Single-pass multi-head attention backward, round 4: dQ reduced by MFMA chains instead of an fp32 LDS accumulator, and
// the next head's Q / dO rows streamed in block by block behind the sweep.  L <= 224, dh = 64, non-causal, gfx950.
//
// What sc_attention_bwd1.hip does and what bounded it: wave w owns 32 keys and sweeps the query blocks; its
// partial dQ of a block (8 MFMA) is added into an fp32 LDS image [L][64] behind an ordered hand-off between the seven key
// waves.  That accumulation is 16 KiB of LDS read-modify-write per wave and block (784 KiB of a head's 1.8 MB of LDS
// traffic), sits on every step's dependency chain (dS -> LDS trip -> transposed read -> wait for the previous
// contributor -> MFMA -> read-add-write), needs a staggered block order, and the staggered order keeps every Q / dO row
// in use until the head ends, so the next head's images can only be requested at the very end of a head.
//
// Here:
//   * all key waves walk the query blocks in the SAME order 0, 1, 2, ...; per block a wave computes S, dP (16 MFMA), P and
//     dS, dV^T += dO^T P and dK^T += Q^T dS (16 MFMA) and leaves its dS tile [32 keys][32 queries] (bf16, 2 KiB) in a ring
//     of three block slots -- nothing else of dQ is on the sweep's critical path;
//   * block b's dQ is formed ONCE, by wave b, one step later: dQ^T[d][q] = sum over the seven key tiles of K^T . dS^T as
//     ONE accumulation chain per output fragment (56 MFMA, K^T fragments by transposed reads of the K image, no fp32
//     LDS traffic, fixed summation order = bit-reproducible), then straight to global memory through the wave's staging
//     tile (whole 128-byte rows per store instruction).  Every wave reduces exactly one block per head;
//   * the helper wave (s_setprio 3) refills the Q / dO rows of block j with the NEXT head's rows as soon as all key waves
//     have finished block j (counter ready[j]), streams the next head's K into the other of two K images and computes
//     delta = rowsum(dO * O), -lse log2e for it.  One workgroup barrier per head; the last block's refill is allowed to
//     land after that barrier (flag `tail`).
// LDS (L = 197): Q, dO images 2 x 28 KiB, K images 2 x 28 KiB, dS ring 3 x 14 KiB, statistics 3.5 KiB = 157.6 KiB.
// Synchronisation counters (LDS words, monotonic over the heads a workgroup walks, i = head index within the walk):
//   ready[j] += 1 by every key wave after block j          (NW (i + 1) = all done)      -> reducer of j, helper refill of j
//   done[j]  += 1 by the reducer of block j                (i + 1)                      -> producers of block j + 3 (ring slot)
//   tail     += 1 by the helper when the last block's rows of the next head have landed -> key waves before their last block
//   reference: autograd of nn.MultiheadAttention's SDPA, src/open_clip/transformer.py:253,272-287.
#include "sc_attn_common.h"
#include <stdlib.h>

namespace {

constexpr int BDH = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int RING = 3;                       // dS ring depth in query blocks

#ifdef SC_BWD3_CLOCK
// Diagnostic build only (tools/attn_bwd3_clock.py, tools/build_variant.py; never in libspatialclip_hip.so): per workgroup,
// Delta s_memtime (shader clocks) and Delta s_memrealtime (100 MHz) around the key waves' walk over the heads, in a buffer of
// their own that nothing else reads -- in-kernel clock = Delta memtime / Delta realtime x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
__device__ unsigned long long sc_bwd3_stamps[4 * 1024];
#endif

template <int NB>
__global__ __launch_bounds__(512) void attn_bwd3_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                        const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                        float* __restrict__ delta, bf16* __restrict__ dqkv, int L, int H,
                                                        int nheads, float scale, unsigned dq_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DH = BDH, KS = DH / 32, DT = DH / 16;
    constexpr int Lp = NB * 32;
    constexpr int IMG = Lp * DH * 2;
    constexpr int NW = NB;                              // key waves: one per 32-key block; wave NB = helper
    constexpr int SLOT = NW * 2048;                     // one ring slot: NW dS tiles [key 32][q 32] bf16
    char* Qimg = smem;
    char* Gimg = smem + IMG;
    char* Kimg0 = smem + 2 * IMG;                       // two K images: head i reads image i & 1
    char* ring = smem + 4 * IMG;
    float* stats = reinterpret_cast<float*>(ring + RING * SLOT);        // [2 heads][2: lse2, delta][Lp]
    const unsigned ctr0 = (unsigned)(uintptr_t)(lptr_t)smem + 4 * IMG + RING * SLOT + 4 * Lp * 4;
    // ready[j] at ctr0 + 4 j, done[j] at ctr0 + 4 NB + 4 j, tail at ctr0 + 8 NB
    const unsigned tailc = ctr0 + 8 * NB;

    const int t = threadIdx.x, lane = t & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int d = H * DH;
    const long long rs = 3LL * d;
    const float c2 = scale * LOG2E;
    const int prow = lane >> 3, pch = lane & 7;

    // 1-KiB pieces [first, last) of an image: 8 rows x 128 B each, rows >= L are finite copies of row L - 1 (masked later)
    auto dma_rows = [&](const bf16* src0, long long stride, char* img, int first, int last) {
        for (int pp = first; pp < last; ++pp) {
            const int row = pp * 8 + prow, rowc = min(row, L - 1);
            dma16(src0 + (long long)rowc * stride + (pch ^ Img<DH>::swz(row)) * 8, img + pp * 1024);
        }
    };
    if (t < 2 * NB + 1) asm volatile("ds_write_b32 %0, %1" ::"v"(ctr0 + 4 * t), "v"(0u) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_barrier();                                       // counters are zero before any wave bumps or polls one

    if (wave == NW) {
        // ------------------------------------------------------------------ helper wave
        __builtin_amdgcn_s_setprio(3);
        auto prepare = [&](int head, int buf) {          // K image + statistics of `head`
            const int b = head / H, h = head % H;
            dma_rows(qkv + (long long)b * L * rs + h * DH + d, rs, Kimg0 + buf * IMG, 0, Lp / 8);
            const bf16* gbase = dout + (long long)b * L * d + h * DH;
            const bf16* obase = out + (long long)b * L * d + h * DH;
            const float* lrow = lse + ((long long)b * H + h) * L;
            float* sl = stats + buf * 2 * Lp;
            constexpr int GRP = 4;                       // two lanes per row, 32 rows per trip, four trips in flight
#pragma unroll
            for (int t0 = 0; t0 < NB; t0 += GRP) {
                bf16x8 g8[GRP][4], o8[GRP][4];
                float lv[GRP];
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    if (t0 + u < NB) {
                        const int rc = min((t0 + u) * 32 + (lane >> 1), L - 1), half = lane & 1;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            g8[u][c] = *reinterpret_cast<const bf16x8*>(gbase + (long long)rc * d + half * 32 + c * 8);
                            o8[u][c] = *reinterpret_cast<const bf16x8*>(obase + (long long)rc * d + half * 32 + c * 8);
                        }
                        lv[u] = lrow[rc];
                    }
                }
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    if (t0 + u < NB) {
                        const int r = (t0 + u) * 32 + (lane >> 1), half = lane & 1;
                        float acc = 0.f;
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc += (float)g8[u][c][e] * (float)o8[u][c][e];
                        acc += __shfl_xor(acc, 1, 64);
                        if (half == 0) {
                            sl[r] = -lv[u] * LOG2E;
                            sl[Lp + r] = acc;
                            if (r < L) delta[((long long)b * H + h) * L + r] = acc;
                        }
                    }
                }
            }
        };
        auto qg_block = [&](int head, int j) {           // Q and dO rows of query block j (four pieces per image)
            const int b = head / H, h = head % H;
            dma_rows(qkv + (long long)b * L * rs + h * DH, rs, Qimg, 4 * j, 4 * j + 4);
            dma_rows(dout + (long long)b * L * d + h * DH, d, Gimg, 4 * j, 4 * j + 4);
        };
        int head = blockIdx.x;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (head < nheads) {
            prepare(head, 0);
            for (int j = 0; j < NB; ++j) qg_block(head, j);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_bump(tailc);                                  // head 0's last block is there with everything else
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_barrier();                                                    // A(0)
        for (int i = 0; head < nheads; ++i, head += gridDim.x) {
            const int next = head + gridDim.x;
            if (next < nheads) {
                // K image (i + 1) & 1 was last read by the reducers of head i - 1: all of them are behind barrier A(i)
                prepare(next, (i + 1) & 1);
                for (int j = 0; j < NB; ++j) {
                    lds_wait_ge(ctr0 + 4 * j, (unsigned)NW * (unsigned)(i + 1));     // every key wave is done with block j
                    qg_block(next, j);
                }
                // everything but the last block's 8 pieces must have landed before the barrier publishes it
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            wg_barrier();                                                // A(i + 1)
            if (next < nheads) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) lds_bump(tailc);                          // tail == i + 2: last block of head i + 1 landed
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- key waves
    const int kb = wave * 32;                           // this wave's keys
    const __amdgpu_buffer_rsrc_t dq_rsrc = sc_make_rsrc(dqkv, dq_bytes);

    auto load_vf = [&](int head, bf16x8 (&vf)[2][KS]) {
        const int b = head / H, h = head % H;
        const bf16* vbase = qkv + (long long)b * L * rs + h * DH + 2 * d;
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
            const int key = min(kb + bt * 16 + li, L - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) vf[bt][ks] = *reinterpret_cast<const bf16x8*>(vbase + (long long)key * rs + ks * 32 + lg * 8);
        }
    };
    // 16 rows x 64 columns of fp32 accumulators (lane = row li, registers = columns 16 dt + 4 lg + r), scaled, to bf16 rows
    // of dqkv at (row0 + r, column offset col0): through a 2-KiB LDS tile so that a store instruction writes 8 whole 128-B rows
    auto store_rows16 = [&](char* tile, const f32x4 (&v)[DT], float mul, int row0, long long col0, int b) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f32x4 x = v[dt] * mul;
            *reinterpret_cast<u32x2*>(tile + stage_off(li, dt * 2 + (lg >> 1)) + (lg & 1) * 8) = sc_pack4(x[0], x[1], x[2], x[3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int r = hf * 8 + (lane >> 3), ch = lane & 7;
            const u32x4 u = *reinterpret_cast<const u32x4*>(tile + stage_off(r, ch));
            const int row = row0 + r;
            const unsigned off = row < L ? (unsigned)((((long long)b * L + row) * rs + col0 + ch * 8) * 2) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b128(u, dq_rsrc, off, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };

    int head = blockIdx.x;
    bf16x8 vf[2][KS];
    if (head < nheads) load_vf(head, vf);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    wg_barrier();                                                       // A(0)
#ifdef SC_BWD3_CLOCK
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int i = 0; head < nheads; ++i, head += gridDim.x) {
        const int b = head / H, h = head % H;
        const char* Kimg = Kimg0 + (i & 1) * IMG;
        const float* slse = stats + (i & 1) * 2 * Lp;
        const float* sdel = slse + Lp;
        const unsigned u1 = (unsigned)(i + 1);

        // dQ of the query rows [32 bq + 16 a0, 32 bq + 16 a1) (this wave is their reducer): one MFMA chain per output fragment
        // over the NW key tiles
        auto reduce = [&](int bq, int a0, int a1) {
            lds_wait_ge(ctr0 + 4 * bq, (unsigned)NW * u1);                // every key wave has left its dS tile of block bq
            const char* slot = ring + (bq % RING) * SLOT;
            f32x4 dq[2][DT];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) dq[a][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int kw = 0; kw < NW; ++kw) {          // not unrolled: 24 fragment registers per key tile, not 24 NW
                bf16x8 dst[2], ktr[DT];
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    if (a >= a0 && a < a1) dst[a] = frag_tr_ds(slot + kw * 2048, a * 16, li, lg);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) ktr[dt] = frag_tr<DH>(Kimg, kw * 32, dt * 16, li, lg);
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    if (a >= a0 && a < a1) {
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) dq[a][dt] = sc_mfma16(ktr[dt], dst[a], dq[a][dt]);
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // staging area.  A whole-block reducer is the only reader of the slot until done[bq] moves: its own tile there.  The two
            // half-block reducers of the last block read EACH OTHER's tiles, so they stage in the slot a block NB would take
            // (free once block NB - RING is reduced; the same tile serves this wave's dK / dV stores afterwards)
            const bool whole = (a1 - a0 == 2);
            if (!whole && NB >= RING) lds_wait_ge(ctr0 + 4 * NB + 4 * (NB - RING), u1);
            char* tile = ring + ((whole ? bq : NB) % RING) * SLOT + wave * 2048;
#pragma unroll
            for (int a = 0; a < 2; ++a)
                if (a >= a0 && a < a1) store_rows16(tile, dq[a], scale, bq * 32 + a * 16, (long long)h * DH, b);
            if (lane == 0 && a1 - a0 == 2) lds_bump(ctr0 + 4 * NB + 4 * bq);     // done[bq] = i + 1 (only whole-block reducers are waited for)
        };

        // ---------------- this wave's K row fragments (B operand of S) stay in registers for the head
        bf16x8 kf[2][KS];
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[bt][ks] = frag_row<DH>(Kimg, kb + bt * 16, ks, li, lg);
        f32x4 dk[2][DT], dv[2][DT];
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dk[bt][dt] = dv[bt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int j = 0; j < NB; ++j) {
            const int q0 = j * 32;
            if (j == NB - 1) lds_wait_ge(tailc, u1);                     // the last block's rows were allowed to land late
            // S and dP: q rows in the accumulator registers (row 4 lg + r of tile a), key on the lane
            f32x4 s[2][2], p[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bt = 0; bt < 2; ++bt) s[a][bt] = p[a][bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qa = frag_row<DH>(Qimg, q0 + a * 16, ks, li, lg);
                    const bf16x8 ga = frag_row<DH>(Gimg, q0 + a * 16, ks, li, lg);
#pragma unroll
                    for (int bt = 0; bt < 2; ++bt) {
                        s[a][bt] = sc_mfma16(qa, kf[bt][ks], s[a][bt]);
                        p[a][bt] = sc_mfma16(ga, vf[bt][ks], p[a][bt]);
                    }
                }
            // the ring slot of block j held block j - RING: its reducer must be through
            if (j >= RING) lds_wait_ge(ctr0 + 4 * NB + 4 * (j - RING), u1);
            char* tile = ring + (j % RING) * SLOT + wave * 2048;
            const bool edge = (q0 + 32 > L) || (kb + 32 > L);
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) {
                f32x4 pr[2], ds[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const f32x4 l2 = *reinterpret_cast<const f32x4*>(slse + q0 + a * 16 + 4 * lg);
                    const f32x4 dl = *reinterpret_cast<const f32x4*>(sdel + q0 + a * 16 + 4 * lg);
                    f32x4 e;
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[r] = fast_exp2(fmaf(s[a][bt][r], c2, l2[r]));
                    f32x4 dd = e * (p[a][bt] - dl);
                    if (edge) {
                        const int key = kb + bt * 16 + li;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int q = q0 + a * 16 + 4 * lg + r;
                            const bool m = (q >= L) || (key >= L);
                            e[r] = m ? 0.f : e[r];
                            dd[r] = m ? 0.f : dd[r];
                        }
                    }
                    pr[a] = e;
                    ds[a] = dd;
                }
                pf[bt] = pack8(pr[0], pr[1]);
                dsf[bt] = pack8(ds[0], ds[1]);
                // dS tile for the dQ product: row = key, 4 consecutive queries = 8 bytes (ds_tile_off: conflict-free)
                union { bf16x8 v; u32x2 h[2]; } u;
                u.v = dsf[bt];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int row = bt * 16 + li, qc = a * 16 + 4 * lg;
                    *reinterpret_cast<u32x2*>(tile + ds_tile_off(row, qc >> 3) + ((qc >> 2) & 1) * 8) = u.h[a];
                }
            }
            // dV^T += dO^T . P ,  dK^T += Q^T . dS   (contraction over the 32 queries of the block)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 gtr = frag_tr<DH>(Gimg, q0, dt * 16, li, lg);
                const bf16x8 qtr = frag_tr<DH>(Qimg, q0, dt * 16, li, lg);
#pragma unroll
                for (int bt = 0; bt < 2; ++bt) {
                    dv[bt][dt] = sc_mfma16(gtr, pf[bt], dv[bt][dt]);
                    dk[bt][dt] = sc_mfma16(qtr, dsf[bt], dk[bt][dt]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // rows of block j read, dS tile written
            if (lane == 0) lds_bump(ctr0 + 4 * j);                        // ready[j]
            // this wave reduces block `wave` one step behind (the other key waves are through it by then).  The last block has
            // no later step: its reduction is the serial tail of the head, so two waves share it (16 query rows each)
            if (j == wave + 1) reduce(wave, 0, 2);
            if (j == NB - 1) {
                if (NB == 1) reduce(0, 0, 2);
                else if (wave == NB - 1) reduce(NB - 1, 0, 1);
                else if (wave == 0) reduce(NB - 1, 1, 2);
            }
        }
        // ---------------- next head's V fragments first, then this head's dK / dV stores (they drain under the loads)
        const int next = head + gridDim.x;
        bf16x8 vfn[2][KS];
        if (next < nheads) load_vf(next, vfn);
        asm volatile("" ::: "memory");                     // no store may move above the loads: the counted wait below relies on it
        // staging tile: the ring slot a block NB would take (free once block NB - RING is reduced; never used when NB < RING)
        if (NB >= RING) lds_wait_ge(ctr0 + 4 * NB + 4 * (NB - RING), u1);
        char* stile = ring + (NB % RING) * SLOT + wave * 2048;
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
            store_rows16(stile, dk[bt], scale, kb + bt * 16, (long long)d + h * DH, b);
            store_rows16(stile, dv[bt], 1.0f, kb + bt * 16, 2LL * d + h * DH, b);
        }
        // the next head's V fragments have landed; exactly the 8 younger dK / dV stores may still fly
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) vf[bt][ks] = vfn[bt][ks];
        wg_barrier();                                      // A(i + 1)
    }
#ifdef SC_BWD3_CLOCK
    if (wave == 0 && lane == 0 && blockIdx.x < 1024) {
        sc_bwd3_stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - clk0;
        sc_bwd3_stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
        sc_bwd3_stamps[4 * blockIdx.x + 2] = (unsigned long long)((nheads - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
    }
#endif
}

template <int NB>
void launch_bwd3(int grid, size_t lds, hipStream_t st, const bf16* qkv, const bf16* out, const bf16* dout, const float* lse,
                 float* delta, bf16* dqkv, int L, int H, int nheads, float scale, unsigned dqb) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd3_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attn_bwd3_kernel<NB><<<grid, (NB + 1) * 64, lds, st>>>(qkv, out, dout, lse, delta, dqkv, L, H, nheads, scale, dqb);
}

}  // namespace

// returns 1 if the kernel took the launch, 0 if the shape is outside its range (caller falls back)
int sc_attn_bwd_ring(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
                     int L, int Lq, int H, int dh, int causal, hipStream_t st) {
    if (dh != BDH || L > 224 || Lq != L || causal) return 0;
    const int NB = (L + 31) / 32;
    const int Lp = NB * 32;
    const size_t lds = (size_t)4 * Lp * dh * 2 + (size_t)RING * NB * 2048 + (size_t)4 * Lp * 4 + (size_t)(2 * NB + 1) * 4 + 60;
    const long long dqb = (long long)B * L * 3 * H * dh * 2;
    if (dqb >= 0xFFFFFFF0ll) return 0;
    if (lds > 160 * 1024) return 0;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        ncu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    const int nheads = B * H;
    int grid = nheads < ncu ? nheads : ncu;
    if (const char* e = getenv("SC_ATTN_GRID")) { const int gcap = atoi(e); if (gcap > 0 && gcap < grid) grid = gcap; }   // measurement: fewer workgroups

    const float scale = 1.0f / sqrtf((float)dh);
    const bf16 *q = (const bf16*)qkv, *o = (const bf16*)out, *g = (const bf16*)dout;
    bf16* dq = (bf16*)dqkv;
    switch (NB) {
        case 1: launch_bwd3<1>(grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 2: launch_bwd3<2>(grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 3: launch_bwd3<3>(grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 4: launch_bwd3<4>(grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 5: launch_bwd3<5>(grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 6: launch_bwd3<6>(grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 7: launch_bwd3<7>(grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        default: return 0;
    }
    return 1;
}

#ifdef SC_BWD3_CLOCK
extern "C" int sc_debug_bwd3_stamps(unsigned long long* host, int n_words) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sc_bwd3_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#endif
