// Persistent two-pass multi-head attention backward for short sequences (L <= 224, dh = 64) on gfx950.
//
// Same arithmetic as attn_bwd_fused_kernel of sc_attention.hip (pass A: wave = 16-query tile, dQ and delta; pass B:
// wave = 16-key tile, dK and dV; Q, K, V, dO of the head as swizzled LDS images), re-plumbed so that no compute wave ever
// waits on HBM.  The one-workgroup-per-head form spends about half of its time with the CU idle: 112 KiB of LDS leave
// room for one workgroup per CU, so its image loads (head start) and its stores (head end) are exposed -- ~150 us of a
// 285 us ViT-B/16 layer.  Here a persistent workgroup walks (batch, head) pairs and two LOADER waves (s_setprio 3) feed
// the images by LDS-DMA one phase ahead of their use:
//
//     barrier A(i)   K, V of head i have landed; nobody reads the Q / dO images any more
//       loaders:     DMA Q, dO of head i                         compute: pass A(i) on the K / V images
//                                                                 (its tile's q, dO, O rows and lse: registers, prefetched)
//     barrier B(i)   Q, dO of head i have landed; every delta / lse2 is in LDS
//       compute:     hoist the tile's K / V row fragments into registers, bump an LDS arrival counter, pass B(i)
//       loaders:     wait for the counter (K / V images are free), DMA K, V of head i+1
//       compute:     after pass B: plain loads of the next head's pass-A tile operands, then the dK / dV stores
//     barrier A(i+1)
//
// Output rows are staged through a 2-KiB wave-private LDS tile so that every store instruction writes whole 128-byte
// rows (the accumulator layout would touch 16 rows x 32 B per instruction, which retires several times slower).
// LDS at L = 197: 4 x 28 KiB images + 14 x 2 KiB staging + 1.75 KiB statistics = 141.8 KiB.
//   reference: autograd of nn.MultiheadAttention's SDPA, src/open_clip/transformer.py:253,272-287; mask :1080-1086.
#include "sc_attn_common.h"
#include <stdlib.h>

namespace {

constexpr int PDH = 64;
constexpr int NLOADER = 2;
constexpr float LOG2E = 1.4426950408889634f;

// Opaque copy of a lane-dependent value.  The compiler otherwise hoists every lane-dependent LDS address of both passes
// out of the head loop (~30 VGPRs of invariants) and pays for them with spills; deriving them from a laundered lane id
// at the top of each section keeps them live only where they are used, at the price of a few VALU instructions per head.
SC_DEVICE int launder(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

#ifdef SC_ATTN_TRACE
// debug build only (tools/attn_phase_trace.py): cycle stamps of workgroup 0, first 8 heads; [role][head][slot]
__device__ unsigned long long g_trace[2 * 8 * 8];
#define TR(role, slot) do { if (blockIdx.x == 0 && i < 8 && (threadIdx.x & 63) == 0) g_trace[(role) * 64 + i * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TR(role, slot) do {} while (0)
#endif

template <int NB, bool CAUSAL>
__global__ __launch_bounds__(1024) void attn_bwd2_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                         const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                         float* __restrict__ delta, bf16* __restrict__ dqkv, int L, int H,
                                                         int nheads, float scale, unsigned dq_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DH = PDH, KS = DH / 32, DT = DH / 16;
    const int t = threadIdx.x, lane = t & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int NW = (int)(blockDim.x >> 6) - NLOADER;       // compute waves = 16-row tiles of the head
    constexpr int Lp = NB * 32;                            // compile-time image geometry: LDS addresses fold into immediates
    constexpr int isz = Lp * DH * 2;
    constexpr int pieces = Lp / 8;                         // 1-KiB DMA pieces per image
    constexpr int NWMAX = 2 * NB;
    const int d = H * DH;
    const long long rs = 3LL * d;
    char* Qimg = smem;
    char* Kimg = smem + isz;
    char* Vimg = smem + 2 * isz;
    char* Gimg = smem + 3 * isz;
    char* scratch0 = smem + 4 * isz;                       // 2 KiB per compute wave
    float* slse = reinterpret_cast<float*>(scratch0 + NWMAX * 2048);
    float* sdel = slse + Lp;
    const unsigned khoist = (unsigned)(uintptr_t)(lptr_t)smem + 4 * isz + NWMAX * 2048 + 2 * Lp * 4;   // monotonic arrival counter
    const float c2 = scale * LOG2E;

    if (wave >= NW) {
        // ------------------------------------------------------------------ loader waves
        __builtin_amdgcn_s_setprio(3);
        const int hw = wave - NW, prow = lane >> 3, pch = lane & 7;
        auto dma_image = [&](const bf16* src0, long long stride, char* img) {
            for (int pp = hw; pp < pieces; pp += NLOADER) {
                const int row = pp * 8 + prow, rowc = min(row, L - 1);      // rows >= L: finite copies of row L-1 (masked)
                dma16(src0 + (long long)rowc * stride + (pch ^ Img<DH>::swz(row)) * 8, img + pp * 1024);
            }
        };
        int head = blockIdx.x;
        if (head < nheads) {
            const bf16* base = qkv + (long long)(head / H) * L * rs + (head % H) * DH;
            dma_image(base + d, rs, Kimg);
            dma_image(base + 2 * d, rs, Vimg);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();                                                       // A(0)
        for (int i = 0; head < nheads; ++i, head += gridDim.x) {
            const int b = head / H, h = head % H;
            if (hw == 0) TR(1, 0);
            dma_image(qkv + (long long)b * L * rs + h * DH, rs, Qimg);
            dma_image(dout + (long long)b * L * d + h * DH, d, Gimg);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (hw == 0) TR(1, 1);
            wg_barrier();                                                   // B(i)
            if (hw == 0) TR(1, 2);
            const int next = head + gridDim.x;
            if (next < nheads) {
                lds_wait_ge(khoist, (unsigned)NW * (unsigned)(i + 1));      // every compute wave holds its K / V fragments
                if (hw == 0) TR(1, 3);
                const bf16* base = qkv + (long long)(next / H) * L * rs + (next % H) * DH;
                dma_image(base + d, rs, Kimg);
                dma_image(base + 2 * d, rs, Vimg);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (hw == 0) TR(1, 4);
            }
            wg_barrier();                                                   // A(i+1)
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves: tile `wave` in both passes
    // Every global access of these waves goes through a buffer resource with the head's base in an SGPR offset and a
    // per-lane 32-bit offset that does not change from head to head: no 64-bit per-lane pointers to keep alive (the
    // kernel sits at the 128-VGPR budget of 16 waves per CU, and a spill reload between two stores serialises them).
    char* scratch = scratch0 + wave * 2048;
    const __amdgpu_buffer_rsrc_t dq_rsrc = sc_make_rsrc(dqkv, dq_bytes);
    const __amdgpu_buffer_rsrc_t qkv_rsrc = sc_make_rsrc(qkv, dq_bytes);
    const __amdgpu_buffer_rsrc_t g_rsrc = sc_make_rsrc(dout, dq_bytes / 3);
    const __amdgpu_buffer_rsrc_t o_rsrc = sc_make_rsrc(out, dq_bytes / 3);
    const __amdgpu_buffer_rsrc_t lse_rsrc = sc_make_rsrc(lse, (unsigned)nheads * (unsigned)L * 4u);
    const __amdgpu_buffer_rsrc_t del_rsrc = sc_make_rsrc(delta, (unsigned)nheads * (unsigned)L * 4u);
    const int row0 = wave * 16;
    const int q = row0 + li, qc = min(q, L - 1);           // pass A: my query; pass B: my key
    const unsigned urs = 3u * (unsigned)d;
    for (int i = NW * 16 + t; i < Lp; i += NW * 64) { slse[i] = 0.f; sdel[i] = 0.f; }   // rows no tile owns: stay 0 (masked)
    if (t == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(khoist), "v"(0u) : "memory");

    // one 16-row x 64-column tile out of the accumulator layout (column = lane&15 -> row, 4 lg + r -> d within dt):
    // through the staging tile, out as 2 x (8 rows x 128 B)
    auto store_tile = [&](const f32x4 (&acc)[DT], float mul, unsigned sbase) {
        const int lane = launder((int)(threadIdx.x & 63)), li = lane & 15, lg = lane >> 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f32x4 v = acc[dt] * mul;
            *reinterpret_cast<u32x2*>(scratch + stage_off(li, dt * 2 + (lg >> 1)) + (lg & 1) * 8) = sc_pack4(v[0], v[1], v[2], v[3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int r = hf * 8 + (lane >> 3), ch = lane & 7;
            const u32x4 u = *reinterpret_cast<const u32x4*>(scratch + stage_off(r, ch));
            // the head's base goes into the VGPR offset, not the SGPR offset field: with a register soffset the compiler
            // assumes there is no "store data > 64 bits, then VALU write of the data registers" hazard and places no
            // wait state; on gfx950 the next VALU write did clobber the first data dword of some lanes (seen in the
            // causal instances, where register allocation put an address computation right behind the store)
            const unsigned off = row0 + r < L ? sbase + ((unsigned)(row0 + r) * urs + (unsigned)ch * 8u) * 2u : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b128(u, dq_rsrc, off, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };

    struct TileOps { bf16x8 qf[KS], dof[KS], of[KS]; float l; };
    const unsigned vq = ((unsigned)qc * urs + (unsigned)lg * 8u) * 2u, vg = ((unsigned)qc * (unsigned)d + (unsigned)lg * 8u) * 2u;
    auto load_ops = [&](int head, TileOps& o) {
        const unsigned b = (unsigned)(head / H), h = (unsigned)(head % H);
        const unsigned sq = __builtin_amdgcn_readfirstlane((b * (unsigned)L * urs + h * DH) * 2u);
        const unsigned sg = __builtin_amdgcn_readfirstlane((b * (unsigned)L * (unsigned)d + h * DH) * 2u);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            o.qf[ks] = sc_as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(qkv_rsrc, vq + ks * 64, sq, 0));
            o.dof[ks] = sc_as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(g_rsrc, vg + ks * 64, sg, 0));
            o.of[ks] = sc_as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(o_rsrc, vg + ks * 64, sg, 0));
        }
        o.l = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                  lse_rsrc, (unsigned)qc * 4u, __builtin_amdgcn_readfirstlane((unsigned)head * (unsigned)L * 4u), 0));
    };

    int head = blockIdx.x;
    TileOps ops;
    if (head < nheads) load_ops(head, ops);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(ops.qf[ks]), "+v"(ops.dof[ks]), "+v"(ops.of[ks]));   // see the loop tail
    asm volatile("" : "+v"(ops.l));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_barrier();                                                           // A(0)
    for (int i = 0; head < nheads; ++i, head += gridDim.x) {
        const unsigned b = (unsigned)(head / H), h = (unsigned)(head % H);
        const unsigned sdq = __builtin_amdgcn_readfirstlane((b * (unsigned)L * urs + h * DH) * 2u);     // byte offset of this head's dQ columns
        if (wave == 0) TR(0, 0);
        // ---------------- pass A: dQ of my 16 queries (+ delta, lse2 into LDS for pass B)
        {
            const int lane = launder((int)(threadIdx.x & 63)), li = lane & 15, lg = lane >> 4, q = row0 + li;
            float dl = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += (float)ops.dof[ks][e] * (float)ops.of[ks][e];
            dl = quad_sum(dl);
            const float nl2 = -ops.l * LOG2E;
            if (lg == 0) {
                slse[q] = q < L ? nl2 : 0.f;
                sdel[q] = q < L ? dl : 0.f;
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dl), del_rsrc, (q < L && lg == 0) ? (unsigned)q * 4u : 0xFFFFFFF0u,
                                                  __builtin_amdgcn_readfirstlane((unsigned)head * (unsigned)L * 4u), 0);
            f32x4 dq[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int kend = CAUSAL ? min(Lp, ((row0 + 15) / 32 + 1) * 32) : Lp;
#pragma unroll 1
            for (int k0 = 0; k0 < kend; k0 += 32) {
                bf16x8 ka[KS], kb[KS], va[KS], vb[KS], ktr[DT];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    ka[ks] = frag_row<DH>(Kimg, k0, ks, li, lg);
                    kb[ks] = frag_row<DH>(Kimg, k0 + 16, ks, li, lg);
                    va[ks] = frag_row<DH>(Vimg, k0, ks, li, lg);
                    vb[ks] = frag_row<DH>(Vimg, k0 + 16, ks, li, lg);
                }
                f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s0 = sc_mfma16(ka[ks], ops.qf[ks], s0);
                    s1 = sc_mfma16(kb[ks], ops.qf[ks], s1);
                    p0 = sc_mfma16(va[ks], ops.dof[ks], p0);
                    p1 = sc_mfma16(vb[ks], ops.dof[ks], p1);
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) ktr[dt] = frag_tr<DH>(Kimg, k0, dt * 16, li, lg);
                const bool edge = (k0 + 32 > L) || CAUSAL;
                f32x4 e0 = exp2_affine(s0, c2, nl2), e1 = exp2_affine(s1, c2, nl2);
                if (edge) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k_a = k0 + 4 * lg + r, k_b = k_a + 16;
                        if (k_a >= L || (CAUSAL && k_a > q)) e0[r] = 0.f;
                        if (k_b >= L || (CAUSAL && k_b > q)) e1[r] = 0.f;
                    }
                }
                s0 = e0 * (p0 - dl);
                s1 = e1 * (p1 - dl);
                const bf16x8 dsf = pack8(s0, s1);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) dq[dt] = sc_mfma16(ktr[dt], dsf, dq[dt]);
            }
            if (wave == 0) TR(0, 1);
            store_tile(dq, scale, sdq);
            if (wave == 0) TR(0, 2);
        }
        wg_barrier();                                                       // B(i)
        if (wave == 0) TR(0, 3);
        // ---------------- pass B: dK, dV of my 16 keys
        const int lane = launder((int)(threadIdx.x & 63)), li = lane & 15, lg = lane >> 4, q = row0 + li;
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = frag_row<DH>(Kimg, row0, ks, li, lg);
            vf[ks] = frag_row<DH>(Vimg, row0, ks, li, lg);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_bump(khoist);
        f32x4 dk[DT], dv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dk[dt] = dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int qbeg = CAUSAL ? (row0 / 32) * 32 : 0;
#pragma unroll 1
        for (int q0 = qbeg; q0 < Lp; q0 += 32) {
            bf16x8 qa_[KS], qb_[KS], ga_[KS], gb_[KS], gtr[DT];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qa_[ks] = frag_row<DH>(Qimg, q0, ks, li, lg);
                qb_[ks] = frag_row<DH>(Qimg, q0 + 16, ks, li, lg);
                ga_[ks] = frag_row<DH>(Gimg, q0, ks, li, lg);
                gb_[ks] = frag_row<DH>(Gimg, q0 + 16, ks, li, lg);
            }
            f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s0 = sc_mfma16(qa_[ks], kf[ks], s0);
                s1 = sc_mfma16(qb_[ks], kf[ks], s1);
                p0 = sc_mfma16(ga_[ks], vf[ks], p0);
                p1 = sc_mfma16(gb_[ks], vf[ks], p1);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) gtr[dt] = frag_tr<DH>(Gimg, q0, dt * 16, li, lg);     // lands under the VALU section
            f32x4 pr0, pr1;
            const bool edge = (q0 + 32 > L) || (row0 + 16 > L) || CAUSAL;
            const f32x4 l2a = *reinterpret_cast<const f32x4*>(slse + q0 + 4 * lg), l2b = *reinterpret_cast<const f32x4*>(slse + q0 + 16 + 4 * lg);
            const f32x4 dla = *reinterpret_cast<const f32x4*>(sdel + q0 + 4 * lg), dlb = *reinterpret_cast<const f32x4*>(sdel + q0 + 16 + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qa = q0 + 4 * lg + r, qb = qa + 16;
                float pa = fast_exp2(fmaf(s0[r], c2, l2a[r])), pb = fast_exp2(fmaf(s1[r], c2, l2b[r]));
                float da = pa * (p0[r] - dla[r]), db = pb * (p1[r] - dlb[r]);
                if (edge) {
                    const bool ma = (qa >= L || q >= L || (CAUSAL && q > qa));
                    const bool mb = (qb >= L || q >= L || (CAUSAL && q > qb));
                    pa = ma ? 0.f : pa; da = ma ? 0.f : da;
                    pb = mb ? 0.f : pb; db = mb ? 0.f : db;
                }
                pr0[r] = pa;
                pr1[r] = pb;
                s0[r] = da;
                s1[r] = db;
            }
            const bf16x8 pf = pack8(pr0, pr1), dsf = pack8(s0, s1);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dv[dt] = sc_mfma16(gtr[dt], pf, dv[dt]);
                dk[dt] = sc_mfma16(frag_tr<DH>(Qimg, q0, dt * 16, li, lg), dsf, dk[dt]);
            }
        }
        if (wave == 0) TR(0, 4);
        // next head's pass-A operands first (their latency hides behind the stores and the barrier), then this head's rows
        const int next = head + gridDim.x;
        load_ops(min(next, nheads - 1), ops);              // unconditional: keeps the operands dead during pass B
        store_tile(dk, scale, sdq + 2u * (unsigned)d);
        store_tile(dv, 1.0f, sdq + 4u * (unsigned)d);
        // consume the prefetched operands HERE, in the same straight-line block as their loads and the four younger
        // stores: the compiler then waits with an exact vmcnt(4); left to the loop head it would wait with vmcnt(0),
        // i.e. for the stores of this head to retire
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(ops.qf[ks]), "+v"(ops.dof[ks]), "+v"(ops.of[ks]));
        asm volatile("" : "+v"(ops.l));
        if (wave == 0) TR(0, 5);
        wg_barrier();                                                       // A(i+1)
    }
}

template <typename K>
void set_lds_2(K kern, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

struct Bwd2Args {
    const bf16 *qkv, *out, *dout;
    const float* lse;
    float* delta;
    bf16* dqkv;
    int L, H, nheads;
    float scale;
    unsigned dqb;
};

template <int NB>
void launch_bwd2(bool causal, int grid, int threads, size_t lds, hipStream_t st, const Bwd2Args& a) {
    if (causal) {
        set_lds_2(attn_bwd2_kernel<NB, true>, lds);
        attn_bwd2_kernel<NB, true><<<grid, threads, lds, st>>>(a.qkv, a.out, a.dout, a.lse, a.delta, a.dqkv, a.L, a.H, a.nheads, a.scale, a.dqb);
    } else {
        set_lds_2(attn_bwd2_kernel<NB, false>, lds);
        attn_bwd2_kernel<NB, false><<<grid, threads, lds, st>>>(a.qkv, a.out, a.dout, a.lse, a.delta, a.dqkv, a.L, a.H, a.nheads, a.scale, a.dqb);
    }
}

}  // namespace

#ifdef SC_ATTN_TRACE
extern "C" int sc_debug_attn_trace(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_trace), sizeof(g_trace)) == hipSuccess ? 0 : -1;
}
#endif

// returns 1 if this kernel took the launch, 0 if the shape is outside its range (caller falls back)
int sc_attn_bwd_persistent(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                           int B, int L, int Lq, int H, int dh, int causal, hipStream_t st) {
    if (dh != PDH || L > 224 || Lq != L) return 0;
    const int nt = (L + 15) / 16, NB = (L + 31) / 32;
    if (nt + NLOADER > 16) return 0;
    const int Lp = NB * 32;
    const size_t lds = (size_t)4 * Lp * dh * 2 + (size_t)2 * NB * 2048 + (size_t)2 * Lp * 4 + 64;
    const long long dqb = (long long)B * L * 3 * H * dh * 2;
    if (dqb >= 0xFFFFFFF0ll || lds > 160 * 1024) return 0;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        ncu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    const int nheads = B * H;
    const int grid = nheads < ncu ? nheads : ncu;
    const int threads = (nt + NLOADER) * 64;
    const Bwd2Args a{(const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, delta, (bf16*)dqkv, L, H, nheads,
                     1.0f / sqrtf((float)dh), (unsigned)dqb};
    switch (NB) {
        case 1: launch_bwd2<1>(causal, grid, threads, lds, st, a); break;
        case 2: launch_bwd2<2>(causal, grid, threads, lds, st, a); break;
        case 3: launch_bwd2<3>(causal, grid, threads, lds, st, a); break;
        case 4: launch_bwd2<4>(causal, grid, threads, lds, st, a); break;
        case 5: launch_bwd2<5>(causal, grid, threads, lds, st, a); break;
        case 6: launch_bwd2<6>(causal, grid, threads, lds, st, a); break;
        case 7: launch_bwd2<7>(causal, grid, threads, lds, st, a); break;
        default: return 0;
    }
    return 1;
}
