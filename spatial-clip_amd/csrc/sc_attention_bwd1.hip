// Single-pass multi-head attention backward for short sequences (L <= 224, dh = 64) on gfx950.
//
// The two-pass backward of sc_attention.hip (one pass per output: dQ with the query on the MFMA column, dK / dV with the
// key on it) recomputes S = Q.K^T, dP = dO.V^T and the exponentials twice: 56 MFMAs and 16 v_exp per 32x32 block of the
// score matrix.  Here ONE sweep produces all three gradients: 40 MFMAs and 8 v_exp per 32x32 block.
//
// Work split.  A persistent workgroup walks (batch, head) pairs; wave w owns the 32 keys [32w, 32w+32) of the head:
// their K / V row fragments and the transposed K fragments stay in registers for the whole head, dK^T and dV^T
// accumulate in 64 VGPRs.  The wave sweeps the query blocks (32 queries each); per block:
//     S  [q][key] = Q . K^T        A = Q rows (LDS image),   B = K row fragments (registers)          8 MFMA
//     dP [q][key] = dO . V^T       A = dO rows (LDS image),  B = V row fragments (registers)          8 MFMA
//     P = exp2(S c2 - lse2[q]),  dS = P (dP - delta[q])       (q in the accumulator registers, key on the lane)
//     dV^T[d][key] += dO^T . P     A = dO by ds_read_b64_tr_b16, B = P  (accumulator as operand)      8 MFMA
//     dK^T[d][key] += Q^T . dS     A = Q  by ds_read_b64_tr_b16, B = dS (accumulator as operand)      8 MFMA
//     dQ^T[d][q]    = K^T . dS^T   needs dS with the KEY in the k-slots: dS takes one trip through a 2-KiB wave-private
//                                  LDS tile ([key][q], written as 8-byte runs, read back transposed)   8 MFMA
// dQ is summed over the seven key waves in an fp32 LDS accumulator [L][64]: wave w visits query block (w + step) mod nqb
// at step `step`, and an s_barrier separates the steps, so no two waves touch the same rows at the same time and every
// row receives its seven contributions in a fixed order -- plain read-add-write, no float atomics, bit-reproducible.
// LDS (L = 197): Q, K, dO images 3 x 28 KiB, dQ accumulator 56 KiB, dS tiles 14 KiB, lse / delta 1.75 KiB = 155.8 KiB.
// The k-slot convention of "accumulator as operand" and of the transposed reads is the one of sc_attention.hip.
// Measured (B = 256, L = 197, H = 12, one MI355X): 256 us per layer against 283-296 us for the two-pass kernel on the
// same box; memory pipeline alone 153 us, compute alone 197 us (157 without the dQ accumulation): with 7 compute waves
// of 246 VGPRs per CU the sweep is latency-bound (35 LDS / MFMA waits per step and wave), not pipe-bound.
//   reference: autograd of nn.MultiheadAttention's SDPA, src/open_clip/transformer.py:253,272-287; mask :1080-1086.
#include "sc_attn_common.h"
#include <stdlib.h>

namespace {

constexpr int BDH = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int NSTORE = 2 * 2 * 2 + 4;        // per compute wave and head: dK / dV stores (2 tensors x 2 key tiles x 2 halves) + 4 dQ flush stores

// Overlap (the kernel is memory-heavy: 618 MB per ViT-B/16 layer = ~100 us at the HBM rate, against ~130 us of compute):
//   * wave NB is a HELPER (s_setprio 3): while the compute waves sweep head i it streams K of head i+1 into the K image
//     (free once the compute waves hold their K fragments: an arrival counter says when) and computes
//     delta = rowsum(dO * O) and -lse*log2e of head i+1 into the other half of a double-buffered statistics array;
//   * at the end of a head every compute wave first loads its V fragments of head i+1 and issues its share of the Q / dO
//     image DMA of head i+1, THEN issues its dK / dV stores and the dQ flush, and waits with a COUNTED vmcnt that leaves
//     exactly those NSTORE younger stores in flight: the stores of head i drain while head i+1's images arrive (all
//     epilogue stores are range-checked buffer stores, so their count does not depend on L);
//   * the dQ accumulation needs no workgroup barrier: query block j has a turn counter; the wave that adds the s-th
//     contribution to block j waits for turn[j] == s -- an ordered hand-off that blocks only when the previous
//     contributor is late, and fixes the summation order (bit-reproducible).
template <int NB, bool CAUSAL>
__global__ __launch_bounds__(512) void attn_bwd1_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                        const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                        float* __restrict__ delta, bf16* __restrict__ dqkv, int L, int H,
                                                        int nheads, float scale, unsigned dq_bytes) {
    static_assert(!CAUSAL, "the first-contributor overwrite of the dQ accumulator assumes every (wave, block) step is live");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DH = BDH, KS = DH / 32, DT = DH / 16;
    constexpr int Lp = NB * 32;
    constexpr int IMG = Lp * DH * 2;
    constexpr int PIECES = Lp / 8;                      // 1-KiB DMA pieces per image
    constexpr int NW = NB;                              // compute waves: one per 32-key block; wave NB = helper
    char* Qimg = smem;
    char* Kimg = smem + IMG;
    char* Gimg = smem + 2 * IMG;
    float* dQacc = reinterpret_cast<float*>(smem + 3 * IMG);            // [Lp][64] fp32, 16-byte chunks XOR (row & 15)
    char* scratch0 = smem + 3 * IMG + Lp * DH * 4;                      // NW x 2 KiB: dS tiles [key 32][q 32] bf16
    float* stats = reinterpret_cast<float*>(scratch0 + NW * 2048);      // [2 heads][2: lse2, delta][Lp]
    const unsigned ctr0 = (unsigned)(uintptr_t)(lptr_t)smem + 3 * IMG + Lp * DH * 4 + NW * 2048 + 4 * Lp * 4;
    const unsigned khoist = ctr0;                                       // compute waves that hold their K fragments (monotonic)
    // turn[j] at ctr0 + 4 + 4 j : contributions added to query block j (monotonic over heads)

    const int t = threadIdx.x, lane = t & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int d = H * DH;
    const long long rs = 3LL * d;
    const float c2 = scale * LOG2E;
    const int prow = lane >> 3, pch = lane & 7;

    auto dma_image = [&](const bf16* src0, long long stride, char* img, int first, int stride_p) {
        for (int pp = first; pp < PIECES; pp += stride_p) {
            const int row = pp * 8 + prow, rowc = min(row, L - 1);      // rows >= L: finite copies of row L-1 (masked)
            dma16(src0 + (long long)rowc * stride + (pch ^ Img<DH>::swz(row)) * 8, img + pp * 1024);
        }
    };

    if (wave == NW) {
        // ------------------------------------------------------------------ helper wave
        __builtin_amdgcn_s_setprio(3);
        auto prepare = [&](int head, int buf) {          // K image + statistics of `head`
            const int b = head / H, h = head % H;
            const bf16* base = qkv + (long long)b * L * rs + h * DH;
            dma_image(base + d, rs, Kimg, 0, 1);
            const bf16* gbase = dout + (long long)b * L * d + h * DH;
            const bf16* obase = out + (long long)b * L * d + h * DH;
            const float* lrow = lse + ((long long)b * H + h) * L;
            float* sl = stats + buf * 2 * Lp;
            // two lanes per row, 32 rows per trip; the loads of up to four trips are in flight before the first reduction
            // (the helper has the kernel's VGPR budget to itself): two memory round trips per head instead of seven
            constexpr int GRP = 4;
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
        int head = blockIdx.x;
        if (head < nheads) prepare(head, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        wg_barrier();                                                   // A(0)
        for (int i = 0; head < nheads; ++i, head += gridDim.x) {
            const int next = head + gridDim.x;
            if (next < nheads) {
                lds_wait_ge(khoist, (unsigned)NW * (unsigned)(i + 1));  // the K image of this head is no longer read
                prepare(next, (i + 1) & 1);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            wg_barrier();                                               // B(i): compute waves are done with the Q / dO images
            wg_barrier();                                               // A(i+1): next head's images, statistics, zeroed accumulator
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const int kb = wave * 32;                           // this wave's keys
    char* scratch = scratch0 + wave * 2048;
    const __amdgpu_buffer_rsrc_t dq_rsrc = sc_make_rsrc(dqkv, dq_bytes);
    // the counters start at zero; the dQ accumulator needs no fill (the first contributor of a block overwrites)
    if (t < 1 + NB) asm volatile("ds_write_b32 %0, %1" ::"v"(ctr0 + 4 * t), "v"(0u) : "memory");

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
    auto issue_qg = [&](int head) {
        const int b = head / H, h = head % H;
        dma_image(qkv + (long long)b * L * rs + h * DH, rs, Qimg, wave, NW);
        dma_image(dout + (long long)b * L * d + h * DH, d, Gimg, wave, NW);
    };

    int head = blockIdx.x;
    bf16x8 vf[2][KS];
    if (head < nheads) { load_vf(head, vf); issue_qg(head); }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    wg_barrier();                                                       // A(0)
    for (int i = 0; head < nheads; ++i, head += gridDim.x) {
        const int b = head / H, h = head % H;
        const float* slse = stats + (i & 1) * 2 * Lp;
        const float* sdel = slse + Lp;
        // ---------------- hoist this wave's K fragments: rows (B of S) and transposed (A of dQ^T); release the K image
        bf16x8 kf[2][KS], ktr[DT];
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[bt][ks] = frag_row<DH>(Kimg, kb + bt * 16, ks, li, lg);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ktr[dt] = frag_tr<DH>(Kimg, kb, dt * 16, li, lg);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_bump(khoist);
        f32x4 dk[2][DT], dv[2][DT];
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dk[bt][dt] = dv[bt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int step = 0; step < NB; ++step) {
            int j = wave + step;
            if (j >= NB) j -= NB;
            const int q0 = j * 32;
            const unsigned turn = ctr0 + 4 + 4 * j;
            const bool live = !CAUSAL || (q0 + 31 >= kb);          // causal: some query of the block sees some key of mine
            if (live) {
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
                const bool edge = (q0 + 32 > L) || (kb + 32 > L) || CAUSAL;
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
                                const bool m = (q >= L) || (key >= L) || (CAUSAL && key > q);
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
                        *reinterpret_cast<u32x2*>(scratch + ds_tile_off(row, qc >> 3) + ((qc >> 2) & 1) * 8) = u.h[a];
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
                // dQ^T[d][q] = K^T . dS^T over this wave's 32 keys: fragments before the hand-off, MFMAs + adds inside it
                bf16x8 dst[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) dst[a] = frag_tr_ds(scratch, a * 16, li, lg);
                // ordered hand-off: I am contributor number `step` of query block j
                lds_wait_ge(turn, (unsigned)NB * (unsigned)i + (unsigned)step);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int q = q0 + a * 16 + li;
                    float* arow = dQacc + q * DH;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const f32x4 dq = sc_mfma16(ktr[dt], dst[a], (f32x4){0.f, 0.f, 0.f, 0.f});
                        f32x4* cell = reinterpret_cast<f32x4*>(arow + (((dt * 4 + lg) ^ (q & 15)) << 2));
                        if (step == 0) *cell = dq;                  // first contribution of the head: no read, no zero-fill
                        else *cell = *cell + dq;
                    }
                }
            } else {
                lds_wait_ge(turn, (unsigned)NB * (unsigned)i + (unsigned)step);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_bump(turn);
        }
        wg_barrier();                                      // B(i): every contribution is in; Q / dO images are free
        // ---------------- next head's V fragments and image DMA FIRST, then this head's stores (they drain under the loads)
        const int next = head + gridDim.x;
        bf16x8 vfn[2][KS];
        if (next < nheads) { load_vf(next, vfn); issue_qg(next); }
        asm volatile("" ::: "memory");                     // no store may move above the DMA issue: the counted wait below relies on it
        // dK, dV through the wave's 2-KiB LDS tile so that every store instruction writes whole 128-byte rows (8 rows x
        // 128 B per instruction; the accumulator layout would touch 16 rows x 32 B per instruction, which retires several
        // times slower and holds back everything queued behind it)
#pragma unroll
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const f32x4 v = which ? dv[bt][dt] : dk[bt][dt] * scale;
                    *reinterpret_cast<u32x2*>(scratch + stage_off(li, dt * 2 + (lg >> 1)) + (lg & 1) * 8) =
                        sc_pack4(v[0], v[1], v[2], v[3]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int r = hf * 8 + (lane >> 3), ch = lane & 7;
                    const u32x4 u = *reinterpret_cast<const u32x4*>(scratch + stage_off(r, ch));
                    const int key = kb + bt * 16 + r;
                    const unsigned off = key < L ? (unsigned)((((long long)b * L + key) * rs + (which + 1) * d + h * DH + ch * 8) * 2)
                                                 : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b128(u, dq_rsrc, off, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // flush dQ (bf16, x scale): 4 trips of 16-byte stores per lane
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * (NW * 64) + t;
            const int row = idx >> 3, c2x = (idx & 7) * 2;      // two 16-byte chunks = 8 floats -> 8 bf16 = one 16-byte store
            f32x4* c0 = reinterpret_cast<f32x4*>(dQacc + row * DH + ((c2x ^ (row & 15)) << 2));
            f32x4* c1 = reinterpret_cast<f32x4*>(dQacc + row * DH + (((c2x + 1) ^ (row & 15)) << 2));
            const f32x4 v0 = *c0, v1 = *c1;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (bf16)(v0[e] * scale); o[4 + e] = (bf16)(v1[e] * scale); }
            const unsigned off = row < L ? (unsigned)((((long long)b * L + row) * rs + h * DH + c2x * 4) * 2) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b128(sc_as_u32x4(o), dq_rsrc, off, 0, 0);
        }
        // my image pieces and V fragments of the next head have landed; exactly NSTORE younger stores may still fly
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTORE) : "memory");
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) vf[bt][ks] = vfn[bt][ks];
        wg_barrier();                                      // A(i+1)
    }
}

template <typename K>
void set_lds_b(K kern, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int NB>
void launch_bwd1(bool causal, int grid, size_t lds, hipStream_t st, const bf16* qkv, const bf16* out, const bf16* dout,
                 const float* lse, float* delta, bf16* dqkv, int L, int H, int nheads, float scale, unsigned dqb) {
    (void)causal;       // causal instances exceed the 256-VGPR budget (mask arithmetic on every block): the text tower
                        // keeps the two-pass kernel
    set_lds_b(attn_bwd1_kernel<NB, false>, lds);
    attn_bwd1_kernel<NB, false><<<grid, (NB + 1) * 64, lds, st>>>(qkv, out, dout, lse, delta, dqkv, L, H, nheads, scale, dqb);
}

}  // namespace

// returns 1 if the single-pass kernel took the launch, 0 if the shape is outside its range (caller falls back)
int sc_attn_bwd_single_pass(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                            int B, int L, int Lq, int H, int dh, int causal, hipStream_t st) {
    if (dh != BDH || L > 224 || Lq != L || causal) return 0;
    const int NB = (L + 31) / 32;
    const int Lp = NB * 32;
    const size_t lds = (size_t)3 * Lp * dh * 2 + (size_t)Lp * dh * 4 + (size_t)NB * 2048 + (size_t)4 * Lp * 4 + 64;
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
    const int grid = nheads < ncu ? nheads : ncu;
    const float scale = 1.0f / sqrtf((float)dh);
    const bf16 *q = (const bf16*)qkv, *o = (const bf16*)out, *g = (const bf16*)dout;
    bf16* dq = (bf16*)dqkv;
    switch (NB) {
        case 1: launch_bwd1<1>(causal, grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 2: launch_bwd1<2>(causal, grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 3: launch_bwd1<3>(causal, grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 4: launch_bwd1<4>(causal, grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 5: launch_bwd1<5>(causal, grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 6: launch_bwd1<6>(causal, grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        case 7: launch_bwd1<7>(causal, grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb); break;
        default: return 0;
    }
    return 1;
}
