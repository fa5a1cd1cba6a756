// Single-pass multi-head attention backward for 224 < L <= 257, dh = 64, non-causal (round 5): ViT-L/14's 257 tokens
// (src/open_clip/model_configs/ViT-L-14.json; autograd of the SDPA inside nn.MultiheadAttention,
// src/open_clip/transformer.py:253,272-287).
//
// sc_attention_bwd3.hip (the dS ring) stops at 224 tokens: it has one 255-register key wave per 32-key block plus a helper
// wave, and a workgroup holds eight such waves.  257 tokens are 256 + 1: EIGHT key waves cover keys 0..255, so the helper
// wave has to go and the 257th token (key AND query row 256) needs a home.  This kernel keeps the ring design --
//   * every key wave owns 32 keys (K / V row fragments in registers, dK^T / dV^T accumulators) and sweeps the query blocks in
//     the same order; per block S, dP (16 MFMA), P, dS, dV^T += dO^T P, dK^T += Q^T dS (16 MFMA), its dS tile into a ring of
//     three block slots;
//   * block b's dQ is formed one step later as one MFMA chain per output fragment over the eight key tiles (fixed order:
//     bit-reproducible) by TWO waves, 16 query rows each (waves b and 7 - b): a whole-block reduction takes longer than a
//     sweep step, and a wave that runs one falls behind the ring; results go from the accumulators straight to global memory
//     (four 8-byte pieces per row and lane) --
// and distributes what the helper did:
//   * the K image is SINGLE (LDS: 3 x 33 KiB of images + 48 KiB of ring leave no room for a second one); the key waves take
//     their K / V row fragments of the next head from GLOBAL memory at the very end of a head (they never needed the image),
//     so only the reducers read the image, from the second step on: its DMA goes out behind the barrier that ends the head
//     and is not waited for there -- every wave reports its pieces at the second step of the next head (counter `kready`);
//   * the Q / dO rows of block b are refilled with the next head's rows two steps later (all key waves are past the block:
//     counter ready[b]) by a wave that is NOT reducing at that step;
//   * delta = rowsum(dO O) and -lse log2(e) of the next head: every wave for its own 32 rows, double-buffered; at L = 257 the
//     rows are requested at the end of the second-to-last step and turned into statistics at the end of the last one;
//   * key 256 (L = 257 only): given lse and delta its contributions are additive.  At the start of a head wave w runs it
//     against query block w as a third 16-key tile of the sweep's products -- S, dP (8 MFMA), p and dS on the lanes that hold
//     key 0, dV_256 += dO^T p and dK_256 += Q^T dS (8 MFMA) into per-wave partial sums in LDS; the block's reducers add
//     dQ += dS k_256 (rank one).  Query 256: a last, single-row sweep step outside the ring (its dQ as per-wave partial sums:
//     one MFMA per 16 features on the wave's own dS tile and K rows).  Key 256 x query 256 is one element, done by one wave
//     as two dot products.  All partial sums are added in wave order behind the end-of-head barrier.
// Every wave issues LDS-DMA, global loads and stores, so NOTHING here may look like an LDS-DMA or a transposed LDS read to
// the compiler: behind its builtins hipcc waits `vmcnt(0)` before every LDS read that might alias a DMA destination, i.e. for
// the wave's own refills and for the acknowledgement of its dQ / dK / dV stores, several times per head (b4_dma16, b4_tr).
// One workgroup barrier per head.  Counters (LDS words, monotonic over the heads a workgroup walks): ready[j] += 1 by every
// key wave after block j; done[j] += 1 by each of the two reducers of block j (2 (i + 1): ring slot j % 3 is free for block
// j + 3); kready += 1 by every wave at its second step (its K pieces of this head have landed); psum: the partial sums of the
// previous head have been added up.  Every wait is bounded by work that does not depend on the waiter (the reducers of block
// b have themselves finished block b; producers of block j wait for the reducers of block j - 3, which ran two steps earlier).
// LDS (L = 257): Q, dO images 2 x 33 KiB (264 rows), K image 32 KiB, ring 48 KiB, statistics 4.1 KiB, partial sums 6 KiB,
// scratch = 158.8 KiB.
#include "sc_attn_common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int B4DH = 64;
constexpr float B4LOG2E = 1.4426950408889634f;
constexpr int B4RING = 3;
constexpr int B4NW = 8;                        // key waves = all waves of the workgroup
constexpr int B4XW = 2;                        // L = 257: the wave that fetches rows 256 of the next head (phase stamps: the first to finish a head)

// What a wave requests for the next head at the end of a head (plain loads: the compiler places the wait in front of the
// first use.  An inline-asm variant with a hand-placed counted wait was built and withdrawn: under this kernel's register
// pressure the compiler spilled the asm-loaded registers BEFORE the wait, i.e. stored registers whose loads were in flight).
struct B4Next {
    bf16x8 g8[4], o8[4];        // dO / O: half a row (32 features) of one of its 32 statistics rows
    float lv;                   // lse of that row
    u32x4 v4, k4;               // L = 257, one wave: V / K / dO / O row 256, 16 bytes per lane (lanes 0..7)
    bf16x8 g1, o1;
    float lvs;
};

// Transposed LDS fragments by inline asm.  Every wave of this kernel issues LDS-DMA (global_load_lds) itself, and hipcc puts a
// full `s_waitcnt vmcnt(0)` in front of its ds_read_tr builtin whenever such a DMA -- or anything else the counter counts: the
// dQ / dK / dV stores -- may be in flight (it cannot see that the DMA fills other rows).  With the builtin every wave waited for
// its refills, for its dQ stores' acknowledgement and, at the start of a head, for the K image it had just requested: 5-6 us
// of a 27-us head (rocprof phase stamps + the waits in the ISA).  The asm reads are invisible to the compiler's wait pass, so
// the waits are written by hand and NAME the registers they retire (sc_gemm8p.hip, tn_wait4).  LDS operations return in
// order: `lgkmcnt(n)` with n = the number of younger LDS operations of this wave retires everything older.
struct TrF { u32x2 lo, hi; };
// One address register per fragment column group, everything else in the instruction's 16-bit offset field (the builtin let the
// compiler fold constant offsets; asm operands do not fold, and one address register per read spills this kernel).
template <int OFF>
SC_DEVICE u32x2 b4_tr(unsigned lds_addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF) : "memory");
    return r;
}
template <int OFF, int HI>                       // HI: byte distance of the fragment's second half (16 rows further down)
SC_DEVICE void b4_frag(TrF& f, unsigned lds_addr) {
    f.lo = b4_tr<OFF>(lds_addr);
    f.hi = b4_tr<OFF + HI>(lds_addr);
}
// lane part of frag_tr<64>'s address (sc_attn_common.h) for the 16 columns [16 dt, +16) of a [rows][64] image, relative to a
// block start row0 that is a multiple of 8 (the swizzle then depends on the lane's row alone); second half: + 16 rows = + 2048
SC_DEVICE unsigned b4_img_lane(int dt, int lane) {
    const int li = lane & 15, lg = lane >> 4, q = li >> 2, p = li & 3;
    return (unsigned)(Img<B4DH>::off(4 * lg + q, 2 * dt + (p >> 1)) + ((p & 1) << 3));
}
// lane part of frag_tr_ds's address for the 16 queries [16 a, +16) of a dS tile; second half: + 16 key rows = + 1024
SC_DEVICE unsigned b4_ds_lane(int a, int lane) {
    const int li = lane & 15, lg = lane >> 4, q = li >> 2, p = li & 3;
    return (unsigned)(ds_tile_off(4 * lg + q, 2 * a + (p >> 1)) + ((p & 1) << 3));
}
// LDS-DMA by inline asm for the same reason: behind the builtin the compiler waits `vmcnt(0)` in front of EVERY later LDS read
// that might alias the DMA's destination -- the plain row reads of the next sweep step included, i.e. a refilling wave sat out
// its own refill's latency and a reducing wave its dQ stores' acknowledgement.  The landing of a DMA is waited for where the
// protocol needs it: `s_waitcnt vmcnt(0)` before `kready` is bumped and before the end-of-head barrier.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 is the instruction's implicit LDS-base operand; nothing else in this kernel uses it
SC_DEVICE void b4_dma16(const void* src, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int... I, class F>
SC_DEVICE void b4_static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
SC_DEVICE bf16x8 b4_cat(const TrF& f) {
    union { u32x4 u; bf16x8 b; } c;
    c.u = (u32x4){f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return c.b;
}
template <int YOUNGER>
SC_DEVICE void b4_wait4(TrF (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi)
                 : "n"(YOUNGER) : "memory");
}
template <int YOUNGER>
SC_DEVICE void b4_wait5(TrF& d, TrF (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%10)"
                 : "+v"(d.lo), "+v"(d.hi), "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi),
                   "+v"(f[3].lo), "+v"(f[3].hi)
                 : "n"(YOUNGER) : "memory");
}

#ifdef SC_ATTN_TRACE
// debug build only (tools/attn_bwd4_trace.py): s_memrealtime stamps of workgroup 0, heads 1..3: [head 4][wave 8][slot 32]
__device__ unsigned long long g_trace4[4 * 8 * 32];
#define TR4(slot) do { if (blockIdx.x == 0 && i < 4 && (threadIdx.x & 63) == 0) g_trace4[(i * 8 + wave) * 32 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TR4(slot) do {} while (0)
#endif

template <int NBQ>                             // query blocks: 8 (224 < L <= 256) or 9 (L = 257: block 8 = the single row 256)
__global__ __launch_bounds__(512) void attn_bwd4_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                        const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                        float* __restrict__ delta, bf16* __restrict__ dqkv, int L, int H,
                                                        int nheads, float scale, unsigned dq_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool STRAY = (NBQ == 9);
    constexpr int DH = B4DH, KS = DH / 32, DT = DH / 16, NW = B4NW, RING = B4RING;
    constexpr int ROWS = STRAY ? 264 : 256;             // rows of an image (8-row DMA pieces; row 256 is the only real one of the last)
    constexpr int IMG = ROWS * DH * 2;                  // Q and dO images
    constexpr int KIMG = 256 * DH * 2;                  // K image: keys 0..255 (the stray key's row travels on its own)
    constexpr int SLOT = NW * 2048;                     // one ring slot: NW dS tiles [key 32][q 32] bf16
    char* Qimg = smem;
    char* Gimg = smem + IMG;
    char* Kimg = smem + 2 * IMG;
    char* ring = smem + 2 * IMG + KIMG;
    float* stats0 = reinterpret_cast<float*>(ring + RING * SLOT);      // [2 heads][2: lse2, delta][ROWS]: head i uses buffer i & 1
    float* pR = stats0 + 4 * ROWS;                                       // [NW][2: dV, dK][64] stray partial sums
    char* vR0 = reinterpret_cast<char*>(pR + NW * 128);                 // [2 heads][V row 256 | K row 256] (64 bf16 each)
    float* scr = reinterpret_cast<float*>(vR0 + 512);                   // [NW][64] dS of the stray key for the wave's query block(s)
    float* qR = scr + NW * 64;                                           // [NW][64] query row 256: the waves' partial sums of its dQ
    const unsigned ctr0 = (unsigned)(uintptr_t)(lptr_t)smem + 2 * IMG + KIMG + RING * SLOT + 4 * ROWS * 4 + NW * 128 * 4 + 512 +
                          NW * 64 * 4 + (STRAY ? NW * 64 * 4 : 0);
    // ready[j] at ctr0 + 4 j, done[j] at ctr0 + 4 NBQ + 4 j, kready at ctr0 + 8 NBQ
    const unsigned lds_q = (unsigned)(uintptr_t)(lptr_t)smem, lds_k = lds_q + 2 * IMG, lds_ring = lds_k + KIMG;     // Q (dO: + IMG), K, ring
    const unsigned kready = ctr0 + 8 * NBQ;             // += 1 by every wave once its pieces of the head's K image have landed
    const unsigned psum = kready + 4;                   // += 1 by wave 0 once it has added up the head's stray partial sums

    const int t = threadIdx.x, lane = t & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int d = H * DH;
    const long long rs = 3LL * d;
    const float c2 = scale * B4LOG2E;
    const int kb = wave * 32;                           // this wave's keys (and its rows of the statistics)
    // Lane-dependent addresses of the helper duties (DMA sources, statistics, next-head fragments) are derived from a
    // LAUNDERED copy of the lane id at each use: hoisted out of the head / block loops they would sit beside the key wave's
    // ~250 live registers and come back as scratch traffic (the first build of this kernel: 384 bytes of scratch per lane)
    auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    const __amdgpu_buffer_rsrc_t dq_rsrc = sc_make_rsrc(dqkv, dq_bytes);

    // 1-KiB pieces [first, last) of an image: 8 rows x 128 B each, rows >= L are finite copies of row L - 1 (masked later)
    auto dma_rows = [&](const bf16* src0, long long stride, char* img, int first, int last) {
        const int ln = fresh_lane(), prow = ln >> 3, pch = ln & 7;
        for (int pp = first; pp < last; ++pp) {
            const int row = pp * 8 + prow, rowc = min(row, L - 1);
            b4_dma16(src0 + (long long)rowc * stride + (pch ^ Img<DH>::swz(row)) * 8, (unsigned)(uintptr_t)(lptr_t)img + pp * 1024);
        }
    };
    // Q and dO rows of query block bq of `head` (block 8: the one piece that holds rows 256..263; the fragment reads of rows
    // 264..287 land in the image behind -- finite values that only ever meet masked probabilities)
    auto refill = [&](int head, int bq) {
        const int b = head / H, h = head % H;
        const int first = 4 * bq, last = (STRAY && bq == 8) ? 4 * bq + 1 : 4 * bq + 4;
        dma_rows(qkv + (long long)b * L * rs + h * DH, rs, Qimg, first, last);
        dma_rows(dout + (long long)b * L * d + h * DH, d, Gimg, first, last);
    };
    // this wave's share of the K image of `head`: its own 32 rows (the reducers read keys 0..255 from it)
    auto k_image = [&](int head) {
        const int b = head / H, h = head % H;
        const bf16* kbase = qkv + (long long)b * L * rs + h * DH + d;
        dma_rows(kbase, rs, Kimg, 4 * wave, 4 * wave + 4);
    };
    // What this wave needs of `head` from global memory, requested in one go:
    //   * K and V row fragments of its 32 keys (B operands of S and dP);
    //   * for the statistics of rows kb .. kb + 31 (two lanes per row): half a row of dO and of O, and the row's lse;
    //   * L = 257, wave NW - 2, lanes 0..7: rows 256 of V, K, dO, O (16 bytes per lane) and lse[256].
    auto issue_kv = [&](int head, bf16x8 (&kf)[2][KS], bf16x8 (&vf)[2][KS]) {
        const int b = head / H, h = head % H;
        const int ln = fresh_lane(), li = ln & 15, lg = ln >> 4;
        const bf16* kbase = qkv + (long long)b * L * rs + h * DH + d;
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
            const int key = min(kb + bt * 16 + li, L - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kf[bt][ks] = *reinterpret_cast<const bf16x8*>(kbase + (long long)key * rs + ks * 32 + lg * 8);
                vf[bt][ks] = *reinterpret_cast<const bf16x8*>(kbase + d + (long long)key * rs + ks * 32 + lg * 8);
            }
        }
    };
    auto issue_stats = [&](int head, B4Next& x) {
        const int b = head / H, h = head % H;
        const int ln = fresh_lane();
        const int rc = min(kb + (ln >> 1), L - 1), half = ln & 1;
        const bf16* gbase = dout + ((long long)b * L + rc) * d + h * DH + half * 32;
        const bf16* obase = out + ((long long)b * L + rc) * d + h * DH + half * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            x.g8[c] = *reinterpret_cast<const bf16x8*>(gbase + c * 8);
            x.o8[c] = *reinterpret_cast<const bf16x8*>(obase + c * 8);
        }
        x.lv = lse[((long long)b * H + h) * L + rc];
        x.v4 = x.k4 = (u32x4){0u, 0u, 0u, 0u};
        x.g1 = x.o1 = bf16x8{};
        x.lvs = 0.f;
        if (STRAY && wave == B4XW && ln < 8) {
            const bf16* row = qkv + ((long long)b * L + 256) * rs + h * DH + ln * 8;
            x.k4 = *reinterpret_cast<const u32x4*>(row + d);
            x.v4 = *reinterpret_cast<const u32x4*>(row + 2 * d);
            x.g1 = *reinterpret_cast<const bf16x8*>(dout + ((long long)b * L + 256) * d + h * DH + ln * 8);
            x.o1 = *reinterpret_cast<const bf16x8*>(out + ((long long)b * L + 256) * d + h * DH + ln * 8);
            x.lvs = lse[((long long)b * H + h) * L + 256];
        }
    };
    auto issue_next = [&](int head, B4Next& x, bf16x8 (&kf)[2][KS], bf16x8 (&vf)[2][KS]) {
        issue_kv(head, kf, vf);
        issue_stats(head, x);
    };
    // ... and, once they have landed: statistics (-lse log2 e, delta =
    // rowsum(dO O)) into LDS and delta to global memory, rows 256 of V / K into LDS
    auto finish_next = [&](int head, const B4Next& x, int buf) {
        float* stats = stats0 + buf * 2 * ROWS;
        char* vR = vR0 + buf * 256;
        char* kR = vR + 128;
        const int b = head / H, h = head % H;
        const int ln = fresh_lane();
        const int r = kb + (ln >> 1);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bf16x8 g = x.g8[c], o = x.o8[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (float)g[e] * (float)o[e];
        }
        acc += __shfl_xor(acc, 1, 64);
        if ((ln & 1) == 0) {
            stats[r] = -x.lv * B4LOG2E;
            stats[ROWS + r] = acc;
            if (r < L) delta[((long long)b * H + h) * L + r] = acc;
        }
        if (STRAY && wave == B4XW) {
            float a1 = 0.f;
            if (ln < 8) {
                *reinterpret_cast<u32x4*>(vR + ln * 16) = x.v4;
                *reinterpret_cast<u32x4*>(kR + ln * 16) = x.k4;
                const bf16x8 g = x.g1, o = x.o1;
#pragma unroll
                for (int e = 0; e < 8; ++e) a1 += (float)g[e] * (float)o[e];
            }
            a1 += __shfl_xor(a1, 1, 64);
            a1 += __shfl_xor(a1, 2, 64);
            a1 += __shfl_xor(a1, 4, 64);
            if (ln == 0) {
                stats[256] = -x.lvs * B4LOG2E;
                stats[ROWS + 256] = a1;
                delta[((long long)b * H + h) * L + 256] = a1;
            }
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
        const int ln = fresh_lane();                 // (row offsets hoisted out of the head loop cost 64-bit register pairs)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int r = hf * 8 + (ln >> 3), ch = ln & 7;
            const u32x4 u = *reinterpret_cast<const u32x4*>(tile + stage_off(r, ch));
            const int row = row0 + r;
            const unsigned off = row < L ? (unsigned)((((long long)b * L + row) * rs + col0 + ch * 8) * 2) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b128(u, dq_rsrc, off, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };

    // ---------------------------------------------------------------------- prologue: everything of the first head
    if (t < 2 * NBQ + 2) asm volatile("ds_write_b32 %0, %1" ::"v"(ctr0 + 4 * t), "v"(0u) : "memory");
    if (STRAY && t < 28) {                              // rows 257..263 of both buffers are never valid queries
        const int bufi = t / 14, which = (t % 14) / 7, r = 257 + t % 7;
        stats0[bufi * 2 * ROWS + which * ROWS + r] = 0.f;
    }
    int head = blockIdx.x;
    bf16x8 kf[2][KS], vf[2][KS];
    {
        B4Next x;
        issue_next(head, x, kf, vf);
        refill(head, wave);
        if (STRAY && wave == 0) refill(head, 8);
        k_image(head);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        finish_next(head, x, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    wg_barrier();

    float* sds = scr + wave * 64;                       // reducer scratch: dS of the stray key for the block's 32 queries

    for (int i = 0; head < nheads; ++i, head += gridDim.x) {
        const int b = head / H, h = head % H;
        const unsigned u1 = (unsigned)(i + 1);
        const int next = head + gridDim.x;
        const float* slse = stats0 + (i & 1) * 2 * ROWS;        // this head's statistics and stray rows
        const float* sdel = slse + ROWS;
        const char* vR = vR0 + (i & 1) * 256;
        const char* kR = vR + 128;
        if (STRAY) {                                    // stray key: dV_256[f], dK_256[f] partial sums of this wave, f = lane
            const int ln = fresh_lane();
            lds_wait_ge(psum, (unsigned)i);             // wave 0 has added up the previous head's partial sums
            pR[wave * 128 + ln] = 0.f;
            pR[wave * 128 + 64 + ln] = 0.f;
        }

        // The stray key (row 256) against the 32 (na = 2) or 16 queries of block bq, done once per head by the wave that will
        // reduce the block: S and dP against a 16-key tile whose only real row is the stray key (8 MFMA), p and dS on the
        // lanes that hold key 0 -> dS of the block's queries into `sdst` (the reducer's dQ term), and the key's own
        // gradients as a third key tile of the sweep's products: dV_256 += dO^T p, dK_256 += Q^T dS (8 MFMA), of which only
        // key lane 0 is real: those four lanes add 16 features each into this wave's partial sums.
        auto stray_block = [&](int bq, int na, float* sdst) {
            const int lane = fresh_lane(), li = lane & 15, lg = lane >> 4;
            const int q0 = bq * 32;
            f32x4 s8[2], p8[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) s8[a] = p8[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 kf8 = {}, vf8 = {};
                if (li == 0) {
                    kf8 = *reinterpret_cast<const bf16x8*>(kR + ks * 64 + lg * 16);
                    vf8 = *reinterpret_cast<const bf16x8*>(vR + ks * 64 + lg * 16);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    if (a < na) {
                        s8[a] = sc_mfma16(frag_row<DH>(Qimg, q0 + a * 16, ks, li, lg), kf8, s8[a]);
                        p8[a] = sc_mfma16(frag_row<DH>(Gimg, q0 + a * 16, ks, li, lg), vf8, p8[a]);
                    }
            }
            f32x4 pe[2], pd[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                f32x4 e = (f32x4){0.f, 0.f, 0.f, 0.f}, dd = e;
                if (a < na) {
                    const f32x4 l2 = *reinterpret_cast<const f32x4*>(slse + q0 + a * 16 + 4 * lg);
                    const f32x4 dl = *reinterpret_cast<const f32x4*>(sdel + q0 + a * 16 + 4 * lg);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = (li == 0) && (q0 + a * 16 + 4 * lg + r < L);
                        const float ev = fast_exp2(fmaf(s8[a][r], c2, l2[r]));
                        e[r] = ok ? ev : 0.f;
                        dd[r] = ok ? ev * (p8[a][r] - dl[r]) : 0.f;
                    }
                }
                pe[a] = e;
                pd[a] = dd;
                if (li == 0) *reinterpret_cast<f32x4*>(sdst + a * 16 + 4 * lg) = dd;
            }
            const bf16x8 pf8 = pack8(pe[0], pe[1]), dsf8 = pack8(pd[0], pd[1]);
            float* pv = pR + wave * 128 + 4 * lg;
            TrF gtr[DT], qtr[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const unsigned ad = lds_q + q0 * 128 + b4_img_lane(dt, lane);
                b4_frag<IMG, 2048>(gtr[dt], ad);
                b4_frag<0, 2048>(qtr[dt], ad);
            }
            b4_wait4<0>(gtr);
            b4_wait4<0>(qtr);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 dv8 = sc_mfma16(b4_cat(gtr[dt]), pf8, z);
                if (li == 0) *reinterpret_cast<f32x4*>(pv + 16 * dt) += dv8;
                const f32x4 dk8 = sc_mfma16(b4_cat(qtr[dt]), dsf8, z);
                if (li == 0) *reinterpret_cast<f32x4*>(pv + 64 + 16 * dt) += dk8;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        if (STRAY) {
            stray_block(wave, 2, sds);
            if (wave == 4) {
                // key 256 against query 256, one element: S and dP are two 64-term dot products (lane = feature), its gradients
                // three rank-one terms -- dV_256 += p dO_256 and dK_256 += dS q_256 into this wave's partial sums, dS kept for
                // dQ_256 += dS k_256 in the final sum.  (As a 16 x 16 MFMA tile this cost the wave 1.5 us at the start of a head.)
                const int ln = fresh_lane();
                const float q = (float)reinterpret_cast<const bf16*>(Qimg + 256 * 128)[ln];
                const float g = (float)reinterpret_cast<const bf16*>(Gimg + 256 * 128)[ln];
                const float kx = (float)reinterpret_cast<const bf16*>(kR)[ln];
                const float vx = (float)reinterpret_cast<const bf16*>(vR)[ln];
                const float sdot = sc_wave_sum(q * kx), pdot = sc_wave_sum(g * vx);
                const float pss = fast_exp2(fmaf(sdot, c2, slse[256]));
                const float dss = pss * (pdot - sdel[256]);
                // the MFMA path rounds p and dS to bf16 before the products (they are MFMA operands there): keep that
                const float pb = (float)(bf16)pss, db = (float)(bf16)dss;
                pR[wave * 128 + ln] += pb * g;
                pR[wave * 128 + 64 + ln] += db * q;
                if (ln == 0) sds[32] = dss;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }

        // dQ of the 16 query rows [32 bq + 16 a, +16): one MFMA chain per output fragment over the NW key tiles, plus the stray
        // key's rank-one term, straight from the accumulators to global memory (lane = query row, four 8-byte pieces of its
        // 128-byte row: the four pieces of a row leave in consecutive instructions and meet in L2).  A block is reduced by TWO
        // waves, 16 rows each (block j - 1 at step j by waves j - 1 and 8 - j): a whole-block reduction takes longer than a sweep
        // step, and the wave that runs it falls behind the ring (phase stamps, tools/attn_bwd4_trace.py: the first block's
        // reducer held everybody at step 3, the last block's was the serial tail of the head).
        auto reduce_half = [&](int bq, int a) {
            const int lane = fresh_lane(), li = lane & 15, lg = lane >> 4;       // (shadowing: see fresh_lane)
            lds_wait_ge(ctr0 + 4 * bq, (unsigned)NW * u1);               // every key wave has left its dS tile of block bq
            lds_wait_ge(kready, (unsigned)NW * u1);                      // ... and the head's K image is complete
            const char* slot = ring + (bq % RING) * SLOT;
            f32x4 dq[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // two fragment sets: key tile kw + 1 is requested before tile kw is waited for (10 younger reads stay in flight)
            TrF dst[2], ktr[2][DT];
            unsigned ka[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) ka[dt] = lds_k + b4_img_lane(dt, lane);
            const unsigned da = lds_ring + (bq % RING) * SLOT + b4_ds_lane(a, lane);
            b4_static_for(std::make_integer_sequence<int, NW>{}, [&](auto kc) __attribute__((always_inline)) {
                constexpr int kw = decltype(kc)::value, c = kw & 1, n = c ^ 1;
                if constexpr (kw == 0) {
                    b4_frag<0, 1024>(dst[0], da);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) b4_frag<0, 2048>(ktr[0][dt], ka[dt]);
                }
                if constexpr (kw + 1 < NW) {
                    b4_frag<(kw + 1) * 2048, 1024>(dst[n], da);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) b4_frag<(kw + 1) * 4096, 2048>(ktr[n][dt], ka[dt]);
                    b4_wait5<10>(dst[c], ktr[c]);
                } else {
                    b4_wait5<0>(dst[c], ktr[c]);
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) dq[dt] = sc_mfma16(b4_cat(ktr[c][dt]), b4_cat(dst[c]), dq[dt]);
            });
            if (STRAY) {
                // the stray key's rank-one term: dQ[q][:] += dS_q k_256, dS_q from the pre-pass of wave bq (stray_block), lane =
                // query li, registers = features 16 dt + 4 lg + r
                const float dsq = scr[bq * 64 + a * 16 + li];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x4 kr = *reinterpret_cast<const bf16x4*>(kR + (16 * dt + 4 * lg) * 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dq[dt][r] = fmaf(dsq, (float)kr[r], dq[dt][r]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_bump(ctr0 + 4 * NBQ + 4 * bq);            // done[bq] += 1 (2 (i + 1) = both halves: the slot is free)
            const int row = bq * 32 + a * 16 + li;
            const unsigned off0 = row < L ? (unsigned)((((long long)b * L + row) * rs + (long long)h * DH + 4 * lg) * 2) : 0xFFFFFFF0u;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const f32x4 x = dq[dt] * scale;
                __builtin_amdgcn_raw_buffer_store_b64(sc_pack4(x[0], x[1], x[2], x[3]), dq_rsrc, row < L ? off0 + 32 * dt : off0, 0, 0);
            }
        };

        f32x4 dk[2][DT], dv[2][DT];
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dk[bt][dt] = dv[bt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        TR4(0);
        B4Next x;              // what the next head needs from global memory (requested near the end of the sweep, consumed behind it)

        // One step of the sweep.  PEEL = the last step of the L = 257 form as its own copy of the code: it requests the next
        // head's fragments half-way through, and loads into registers that the loop's next trip would read make the compiler put
        // a full memory wait in front of every trip's first MFMA.
        auto sweep_step = [&](const int j, auto peel_tag) __attribute__((always_inline)) {
            constexpr int PEEL = decltype(peel_tag)::value;      // 0: a trip of the loop; 1 / 2: the last two steps of the L = 257 form
            const int q0 = j * 32;
            if (j == 1) {       // this wave's pieces of the head's K image (requested a step ago or more) have landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) lds_bump(kready);
            }
            // Q / dO rows of block j - 2 are free once every key wave is past it: the next head's rows go in, issued by a wave
            // that is not reducing at this step (a reducer's extra time is on the sweep's critical path, a refiller's is not)
            if (j >= 2 && wave == ((j + 3) & 7) && next < nheads) {
                lds_wait_ge(ctr0 + 4 * (j - 2), (unsigned)NW * u1);
                refill(next, j - 2);
            }
            // S and dP: q rows in the accumulator registers (row 4 lg + r of tile a), key on the lane
            f32x4 s[2][2], p[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bt = 0; bt < 2; ++bt) s[a][bt] = p[a][bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int na = (STRAY && j == NBQ - 1) ? 1 : 2;      // block 8 is the single row 256: one 16-query tile
#pragma unroll
            for (int a = 0; a < 2; ++a)
                if (a < na) {
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
                }
            // the ring slot of block j held block j - RING: its reducers must be through
            if (j >= RING) lds_wait_ge(ctr0 + 4 * NBQ + 4 * (j - RING), 2u * u1);
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
            {
                TrF gtr[DT], qtr[DT];
                const int ln = fresh_lane();
                unsigned ad[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) ad[dt] = lds_q + q0 * 128 + b4_img_lane(dt, ln);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) b4_frag<IMG, 2048>(gtr[dt], ad[dt]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) b4_frag<0, 2048>(qtr[dt], ad[dt]);
                b4_wait4<8>(gtr);          // (the dS tile's ds_writes above are older: retired with them)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int bt = 0; bt < 2; ++bt) dv[bt][dt] = sc_mfma16(b4_cat(gtr[dt]), pf[bt], dv[bt][dt]);
                b4_wait4<0>(qtr);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int bt = 0; bt < 2; ++bt) dk[bt][dt] = sc_mfma16(b4_cat(qtr[dt]), dsf[bt], dk[bt][dt]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // rows of block j read, dS tile written
            if (lane == 0) lds_bump(ctr0 + 4 * j);                        // ready[j]
            if (STRAY && j == NBQ - 1) {
                // The single query row 256 does not go through the ring: its dQ is a sum over ALL keys, and this wave's share
                // (its 32 keys) is one MFMA per 16 features on its own dS tile, read back transposed, against its own rows of
                // the K image; the eight partial sums are added in wave order behind the end-of-head barrier.
                const int ln = fresh_lane(), li2 = ln & 15, lg2 = ln >> 4;
                TrF dst0, ktr0[DT];
                b4_frag<0, 1024>(dst0, lds_ring + (j % RING) * SLOT + wave * 2048 + b4_ds_lane(0, ln));
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) b4_frag<0, 2048>(ktr0[dt], lds_k + kb * 128 + b4_img_lane(dt, ln));
                b4_wait5<0>(dst0, ktr0);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const f32x4 part = sc_mfma16(b4_cat(ktr0[dt]), b4_cat(dst0), z);
                    if (li2 == 0) *reinterpret_cast<f32x4*>(qR + wave * 64 + 16 * dt + 4 * lg2) = part;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // the next head's rows were requested half a step ago: turn them into its statistics now (double-buffered: they
                // can be written while slower waves still sweep), so that their 34 registers are free for the last reductions
                // (no younger load is in flight here: the compiler's wait in front of the first use ends at vmcnt(0) whatever else
                // it could count, so the next head's K / V fragments are requested only behind this -- at the very end of the head)
                if (PEEL == 2 && next < nheads) finish_next(next, x, (i + 1) & 1);
            }
            TR4(1 + j);
            // L = 257, second-to-last step: the next head's statistics rows are requested here -- a whole step (and two waves'
            // reductions) ahead of their use at the end of the last step; requested inside that step they were waited for in the
            // open (phase stamps: the last step took 2.2-4.9 us instead of ~1)
            if (PEEL == 1 && next < nheads) issue_stats(next, x);
            // block j - 1 is reduced one step behind (every key wave is through it by then): rows 0..15 by wave j - 1, rows
            // 16..31 by wave 8 - j
            if (j == wave + 1) { TR4(12); reduce_half(wave, 0); TR4(13); }
            if (j == NW - wave) { TR4(18); reduce_half(NW - 1 - wave, 1); TR4(19); }
        };
#pragma unroll 1
        for (int j = 0; j < (STRAY ? NBQ - 2 : NBQ); ++j) sweep_step(j, std::integral_constant<int, 0>{});
        if (STRAY) {
            sweep_step(NBQ - 2, std::integral_constant<int, 1>{});
            sweep_step(NBQ - 1, std::integral_constant<int, 2>{});
        }
        // What the next head needs from global memory is requested NOW (the key fragments are dead: the next head's go straight
        // into their registers), in front of the work that is left, whose time hides the loads' latency
        if (!STRAY && next < nheads) issue_next(next, x, kf, vf);
        // the block that has no later step: NBQ = 8 -> block 7 by waves 7 and 0 (NBQ = 9: they reduced it at step 8, and the
        // single row of "block" 8 is not a ring block: see the partial sums above)
        if (!STRAY && wave == NW - 1) reduce_half(NW - 1, 0);
        if (!STRAY && wave == 0) reduce_half(NW - 1, 1);
        if (next < nheads) {    // the last two blocks' rows (waves that reduce nothing near the end of a head)
            if (wave == 3) { lds_wait_ge(ctr0 + 4 * (NBQ - 2), (unsigned)NW * u1); refill(next, NBQ - 2); }
            if (wave == 0) { lds_wait_ge(ctr0 + 4 * (NBQ - 1), (unsigned)NW * u1); refill(next, NBQ - 1); }
        }
        TR4(14);

        // ---------------- end of the sweep: ONE workgroup barrier per head.  The next head's loads are consumed BEFORE this
        // head's dK / dV stores are issued: the compiler puts a full `vmcnt(0)` in front of the first use of a loaded value
        // whatever is in flight (it does not count through this kernel's loops and inline asm), and here that drain only covers
        // what is old anyway (refills, dQ stores).  Statistics and the stray rows are double-buffered (head i: buffer i & 1),
        // so they can be written while slower waves still sweep; the stores then drain under the barrier and the next head.
        // (explicit wait: the barrier below publishes this wave's refills and, L <= 256, the loads are consumed here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!STRAY && next < nheads) finish_next(next, x, (i + 1) & 1);
        // L = 257: the next head's K / V fragments go straight into their (dead) registers now; the staging stores, the barrier and
        // the next head's pre-pass (~3 us) are between this and their first use
        if (STRAY && next < nheads) issue_kv(next, kf, vf);
        TR4(15);
        // staging tile: the ring slot a block NBQ would take (free once block NBQ - RING is reduced)
        lds_wait_ge(ctr0 + 4 * NBQ + 4 * (NBQ - RING), 2u * u1);
        {
            char* stile = ring + (NBQ % RING) * SLOT + wave * 2048;
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) {
                store_rows16(stile, dk[bt], scale, kb + bt * 16, (long long)d + h * DH, b);
                store_rows16(stile, dv[bt], 1.0f, kb + bt * 16, 2LL * d + h * DH, b);
            }
        }
        TR4(16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_barrier();          // every sweep and reduction of this head is done; every refill, statistic and fragment of the next has landed
        TR4(17);
        if (next < nheads) k_image(next);                  // the K image is free; not waited for: the next head's reducers look at `kready`
        if (STRAY && wave == 3) {                          // row 256 of dV / dK: the waves' partial sums in wave order
            const int ln = fresh_lane();
            float sv = 0.f, sk = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                sv += pR[w * 128 + ln];
                sk += pR[w * 128 + 64 + ln];
                sq += qR[w * 64 + ln];
            }
            // ... and dQ of query row 256: the partial sums over keys 0..255 plus the stray key's own term dS(256, 256) k_256
            // (wave 4's pre-pass left that dS in its scratch)
            sq = fmaf(scr[4 * 64 + 32], (float)reinterpret_cast<const bf16*>(kR)[ln], sq);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ln == 0) lds_bump(psum);                   // the waves may zero their partial sums for the next head
            sk *= scale;
            sq *= scale;
            const float sv1 = __shfl_down(sv, 1, 64), sk1 = __shfl_down(sk, 1, 64), sq1 = __shfl_down(sq, 1, 64);
            if ((ln & 1) == 0) {
                union { unsigned u; bf16 hh[2]; } pv, pk, pq;
                pv.hh[0] = (bf16)sv; pv.hh[1] = (bf16)sv1;
                pk.hh[0] = (bf16)sk; pk.hh[1] = (bf16)sk1;
                pq.hh[0] = (bf16)sq; pq.hh[1] = (bf16)sq1;
                bf16* row = dqkv + ((long long)b * L + 256) * rs + h * DH + ln;
                *reinterpret_cast<unsigned*>(row) = pq.u;
                *reinterpret_cast<unsigned*>(row + d) = pk.u;
                *reinterpret_cast<unsigned*>(row + 2 * d) = pv.u;
            }
        }
    }
}

template <int NBQ>
void launch_bwd4(int grid, size_t lds, hipStream_t st, const bf16* qkv, const bf16* out, const bf16* dout, const float* lse,
                 float* delta, bf16* dqkv, int L, int H, int nheads, float scale, unsigned dqb) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd4_kernel<NBQ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attn_bwd4_kernel<NBQ><<<grid, B4NW * 64, lds, st>>>(qkv, out, dout, lse, delta, dqkv, L, H, nheads, scale, dqb);
}

}  // namespace

#ifdef SC_ATTN_TRACE
extern "C" int sc_debug_attn_trace4(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_trace4), sizeof(g_trace4)) == hipSuccess ? 0 : -1;
}
#endif

// returns 1 if the kernel took the launch, 0 if the shape is outside its range (caller falls back)
int sc_attn_bwd_ring8(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
                      int L, int Lq, int H, int dh, int causal, hipStream_t st) {
    if (dh != B4DH || L <= 224 || L > 257 || Lq != L || causal) return 0;
    const int NBQ = L > 256 ? 9 : 8;
    const int rows = NBQ == 9 ? 264 : 256;
    const size_t lds = (size_t)2 * rows * dh * 2 + (size_t)256 * dh * 2 + (size_t)B4RING * B4NW * 2048 + (size_t)4 * rows * 4 +
                       (size_t)B4NW * 128 * 4 + 512 + (size_t)B4NW * 64 * 4 + (NBQ == 9 ? (size_t)B4NW * 64 * 4 : 0) +
                       (size_t)(2 * NBQ + 2) * 4 + 56;
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
    if (NBQ == 9) launch_bwd4<9>(grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb);
    else launch_bwd4<8>(grid, lds, st, q, o, g, lse, delta, dq, L, H, nheads, scale, (unsigned)dqb);
    return 1;
}
