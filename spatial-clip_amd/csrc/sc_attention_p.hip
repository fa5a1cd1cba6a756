// Persistent multi-head attention forward for short sequences (L <= 224, dh = 64: ViT-B/16's 197 tokens, CLIP text 77).
//
// Why a second forward kernel: the one-workgroup-per-head kernel (sc_attention.hip) pays a full memory round trip
// (K, V, Q of its head) before its first MFMA and hides it only behind a co-resident workgroup; at B*H = 3072 heads the
// chip spent as long loading as computing (76 us || 77 us, 116 us per ViT-B/16 layer).  Here ONE workgroup per CU walks
// a list of heads and the K / V images and Q tiles of head i+1 stream into the other half of a double buffer by LDS-DMA
// (global_load_lds_dwordx4: no VGPRs, no wave stalls) while head i computes, so the kernel runs at the pace of the
// slower of {HBM stream, compute} instead of their sum.
//
// Compute is restructured for instruction-level parallelism: a head's whole key range is LDS-resident and short
// (<= 7 blocks of 32 keys), so there is no online softmax -- all S^T = K.Q^T blocks are produced by back-to-back
// independent MFMAs, the row maximum is exact and taken once, then exp / sum / bf16 pack, then all P.V MFMAs (V by
// ds_read_b64_tr_b16 with immediate offsets, issued one block ahead of the MFMAs that consume them, counted lgkmcnt).
// No rescale branch, no dependent MFMA -> VALU -> MFMA chain per key block.
//
// Synchronisation per head (one s_barrier).  Loader waves: s_waitcnt vmcnt(0) (their DMA pieces of head i have landed)
// -> s_barrier -> issue K / V of head i+1 into the other buffer -> wait for the Q-slot arrival counter -> issue Q of
// head i+1.  Compute waves: s_barrier -> Q fragments to registers -> bump the arrival counter -> compute -> stores.
// RAW: data is read only behind the barrier that follows the issuing wave's vmcnt(0).  WAR: the buffer that head i+1
// lands in was last read by head i-1, which every compute wave finished before this head's barrier; a Q slot is
// refilled only after all compute waves reported their fragments in registers.  The bare s_barrier builtin is
// IntrNoMem to the compiler (ordinary LDS loads may be scheduled across it), hence wg_barrier() below.
// Measured (B = 256, L = 197, H = 12; one MI355X): 68-71 us per layer against 123 us for the one-workgroup-per-head
// kernel on the same box (SC_ATTN_PERSIST=0); DMA stream alone 34 us, compute alone 60 us.  Loader waves must run at
// raised priority: without s_setprio their DMA issue queues behind the compute waves' VALU (88 us).
//   reference: nn.MultiheadAttention via src/open_clip/transformer.py:253,272-287; causal mask :1080-1086.
#include "sc_attn_common.h"
#include <stdlib.h>

namespace {

constexpr int PDH = 64;                 // head dim of this kernel
constexpr int QSLOT = 16 * PDH * 2;     // one 16-query tile: 2 KiB
#ifndef SC_ATTN_NLOAD
#define SC_ATTN_NLOAD 3
#endif
constexpr int NLOAD = SC_ATTN_NLOAD;    // loader waves per workgroup
#ifndef SC_ATTN_QSYNC
#define SC_ATTN_QSYNC 1
#endif

SC_DEVICE float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// Wave roles.  Compute waves 0 .. nqt-1 own one 16-query tile each; the last NLOAD waves only move data: they issue
// every LDS-DMA piece of the NEXT head's K / V images (a piece costs 100-200 issue cycles with its address arithmetic,
// which would otherwise sit in the compute waves' streams) and are the only waves that wait on the DMA counter.
// With the cyclic wave -> SIMD placement (13 compute waves = 4,3,3,3) the three loaders land on the three lighter SIMDs.
// The Q tiles travel the same way into one 2-KiB slot per compute wave; a slot is single-buffered, so the loaders refill
// it only after an arrival counter in LDS says every compute wave has its Q fragments in registers (one barrier per
// head; the compute waves never wait for the loaders' issue loop).  The compute waves issue no
// global loads at all: nothing they do ever waits on the memory counter.
template <int NB, bool CAUSAL>
__global__ __launch_bounds__(1024) void attn_fwd_p_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                          float* __restrict__ lse, int L, int Lq, int H, int nheads,
                                                          float scale, unsigned out_bytes, unsigned lse_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DH = PDH, KS = DH / 32, DT = DH / 16;
    constexpr int Lp = NB * 32;
    constexpr int IMG = Lp * DH * 2;                 // bytes of one K or V image
    constexpr int PIECES = Lp / 8;                   // 1-KiB DMA pieces (8 rows x 128 B) per image
    const int t = threadIdx.x, lane = t & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwaves = blockDim.x >> 6;
    const int ncomp = nwaves - NLOAD;                // compute waves (>= number of query tiles)
    const int nqt = (Lq + 15) >> 4;
    const int d = H * DH;
    const long long rs = 3LL * d;

    // arrival counter of the compute waves ("my Q fragments are in registers"), after the Q slots
    const unsigned cnt_addr = (unsigned)(uintptr_t)(lptr_t)smem + 4 * IMG + nqt * QSLOT;
    if (t == 0) asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(cnt_addr), "v"(0u) : "memory");

    if (wave >= ncomp) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wave - ncomp;
        __builtin_amdgcn_s_setprio(3);               // the DMA issue must never queue behind compute waves' VALU
        const int prow = lane >> 3, pch = lane & 7;  // lane -> (row in piece, 16-byte chunk); swizzle on the SOURCE chunk
        auto issue_kv = [&](int head, int buf) {
            const int b = head / H, h = head % H;
            const bf16* base = qkv + (long long)b * L * rs + h * DH + d;
            for (int p = lw; p < 2 * PIECES; p += NLOAD) {
                const int img = p >= PIECES ? 1 : 0;
                const int pp = p - img * PIECES;
                const int row = pp * 8 + prow;
                const int rowc = min(row, L - 1);          // padding rows: finite copies of the last row (masked / p = 0)
                const int csrc = pch ^ Img<DH>::swz(row);
                dma16(base + img * d + (long long)rowc * rs + csrc * 8, smem + buf * 2 * IMG + img * IMG + pp * 1024);
            }
        };
        auto issue_q = [&](int head) {
            const int b = head / H, h = head % H;
            const bf16* base = qkv + (long long)b * L * rs + h * DH;
            for (int p = lw; p < 2 * nqt; p += NLOAD) {        // Q tiles: 2 pieces per 16-query tile, swizzled per tile row
                const int r = (p & 1) * 8 + prow;
                const int rowc = min((p >> 1) * 16 + r, L - 1);
                const int csrc = pch ^ Img<DH>::swz(r);
                dma16(base + (long long)rowc * rs + csrc * 8, smem + 4 * IMG + p * 1024);
            }
        };
        int head = blockIdx.x;
        if (head < nheads) { issue_kv(head, 0); issue_q(head); }
        for (int i = 0; head < nheads; ++i, head += gridDim.x) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this head's images and Q tiles have landed
            wg_barrier();                                       // ... and the other buffer's last reader is done
            const int next = head + gridDim.x;
#if SC_ATTN_QSYNC == 0
            wg_barrier();
            if (next < nheads) { issue_kv(next, (i & 1) ^ 1); issue_q(next); }
#else
            if (next < nheads) {
                issue_kv(next, (i & 1) ^ 1);                    // the other buffer is free: start streaming at once
                // the Q slots are single-buffered: refill them only after every compute wave has taken its fragments
                // (an arrival counter in LDS -- the compute waves never wait for the loaders here)
                const unsigned want = (unsigned)ncomp * (unsigned)(i + 1);
                unsigned seen;
                do {
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(cnt_addr) : "memory");
                    seen = __builtin_amdgcn_readfirstlane(seen);
                    if (seen < want) __builtin_amdgcn_s_sleep(2);
                } while (seen < want);
                issue_q(next);
            }
#endif
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const bool has_tile = wave < nqt;
    const float c2 = scale * 1.4426950408889634f;    // exp(x*scale) = exp2(x*c2)
    const __amdgpu_buffer_rsrc_t out_rsrc = sc_make_rsrc(out, out_bytes);
    const __amdgpu_buffer_rsrc_t lse_rsrc = sc_make_rsrc(lse, lse_bytes);
    const int q = wave * 16 + li;                          // this lane's query (MFMA column)
    const char* qslot = smem + 4 * IMG + wave * QSLOT;

    int head = blockIdx.x;
    for (int i = 0; head < nheads; ++i, head += gridDim.x) {
        const int cur = i & 1;
        wg_barrier();                                      // K / V / Q of this head are in LDS (loader waves waited)
        bf16x8 qf[KS];
        if (has_tile) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_row<DH>(qslot, 0, ks, li, lg);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if SC_ATTN_QSYNC == 0
        wg_barrier();
#else
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(cnt_addr), "v"(1u) : "memory");     // Q slot released
#endif
        if (!has_tile) continue;

        const char* Kimg = smem + cur * 2 * IMG;
        const int b = head / H, h = head % H;

        // ---- phase 1: every S^T block, independent MFMAs
        f32x4 s[NB][2];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            s[nb][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            s[nb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s[nb][0] = sc_mfma16(frag_row<DH>(Kimg, nb * 32, ks, li, lg), qf[ks], s[nb][0]);
                s[nb][1] = sc_mfma16(frag_row<DH>(Kimg, nb * 32 + 16, ks, li, lg), qf[ks], s[nb][1]);
            }
        }
        // ---- phase 2: masks (padding in the last block; causal diagonal) and the exact row maximum
        float mx = -1e30f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (nb == NB - 1 || CAUSAL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ka = nb * 32 + 4 * lg + r, kb = ka + 16;
                    if (ka >= L || (CAUSAL && ka > q)) s[nb][0][r] = -1e30f;
                    if (kb >= L || (CAUSAL && kb > q)) s[nb][1][r] = -1e30f;
                }
            }
            mx = max3(mx, s[nb][0][0], s[nb][0][1]);
            mx = max3(mx, s[nb][0][2], s[nb][0][3]);
            mx = max3(mx, s[nb][1][0], s[nb][1][1]);
            mx = max3(mx, s[nb][1][2], s[nb][1][3]);
        }
        mx = quad_max(mx);
        // ---- phase 3: probabilities, row sum, bf16 fragments (k-slot order = accumulator order, see sc_attention.hip)
        const float nbv = -mx * c2;
        f32x4 acc4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 pf[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 e0 = exp2_affine(s[nb][0], c2, nbv), e1 = exp2_affine(s[nb][1], c2, nbv);
            acc4 += e0;
            acc4 += e1;
            pf[nb] = pack8(e0, e1);
        }
        const float lsum = quad_sum((acc4[0] + acc4[1]) + (acc4[2] + acc4[3]));
        // ---- phase 4: O^T = V^T . P^T.  The compute waves never have an LDS-DMA in flight (the loaders own that
        // queue), so the transposed reads can be the compiler-visible builtin: it places the counted lgkmcnt waits and
        // keeps the fragments of block nb+1 in flight under the MFMAs of block nb.
        __builtin_amdgcn_sched_barrier(0);
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* Vimg = Kimg + IMG;
        bf16x8 vf[2][DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vf[0][dt] = frag_tr<DH>(Vimg, 0, dt * 16, li, lg);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (nb + 1 < NB) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) vf[(nb + 1) & 1][dt] = frag_tr<DH>(Vimg, (nb + 1) * 32, dt * 16, li, lg);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] = sc_mfma16(vf[nb & 1][dt], pf[nb], o[dt]);
        }
        // ---- epilogue: lane holds O[q][dt*16 + 4 lg .. +3]; rows >= Lq fall outside the descriptor and are dropped
        const float inv = 1.0f / lsum;
        const bool ok = q < Lq;
        // O rows leave through a 1-KiB wave-private LDS strip, eight query rows per pass, so that a store instruction writes
        // whole 128-byte rows (the accumulator layout would touch 16 rows x 32 B per instruction: see the backward kernels)
        char* ostrip = smem + 4 * IMG + nqt * QSLOT + 64 + wave * 1024;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((li >> 3) == half) {
                const int r8 = li & 7;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    *reinterpret_cast<u32x2*>(ostrip + r8 * 128 + (((dt * 2 + (lg >> 1)) ^ (r8 >> 1)) << 4) + (lg & 1) * 8) =
                        sc_pack4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            {
                const int r8 = lane >> 3, ch = lane & 7;
                const u32x4 v = *reinterpret_cast<const u32x4*>(ostrip + r8 * 128 + ((ch ^ (r8 >> 1)) << 4));
                const int qr = wave * 16 + half * 8 + r8;
                const unsigned off = qr < Lq ? (unsigned)((((long long)b * L + qr) * d + h * DH + ch * 8) * 2) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, off, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        const float l = mx * scale + __builtin_amdgcn_logf(lsum) * 0.6931471805599453f;     // lsum >= 1: raw v_log_f32
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, l), lse_rsrc,
                                              (ok && lg == 0) ? (unsigned)((((long long)b * H + h) * L + q) * 4) : 0xFFFFFFF0u, 0, 0);
    }
}

template <typename K>
void set_lds_p(K kern, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int NB>
void launch_fwd_p(bool causal, int grid, int nthreads, size_t lds, hipStream_t st, const bf16* qkv, bf16* out, float* lse,
                  int L, int Lq, int H, int nheads, float scale, unsigned ob, unsigned lb) {

    if (causal) {
        set_lds_p(attn_fwd_p_kernel<NB, true>, lds);
        attn_fwd_p_kernel<NB, true><<<grid, nthreads, lds, st>>>(qkv, out, lse, L, Lq, H, nheads, scale, ob, lb);
    } else {
        set_lds_p(attn_fwd_p_kernel<NB, false>, lds);
        attn_fwd_p_kernel<NB, false><<<grid, nthreads, lds, st>>>(qkv, out, lse, L, Lq, H, nheads, scale, ob, lb);
    }
}

}  // namespace

// returns 1 if the persistent kernel took the launch, 0 if the shape is outside its range (caller falls back)
int sc_attn_fwd_persistent(const void* qkv, void* out, float* lse, int B, int L, int Lq, int H, int dh, int causal,
                           hipStream_t st) {
    if (dh != PDH || L > 224) return 0;
    const int NB = (L + 31) / 32;
    const int nqt = (Lq + 15) / 16;
    const long long ob = (long long)B * L * H * dh * 2, lb = (long long)B * H * L * 4;
    if (ob >= 0xFFFFFFF0ll) return 0;
    const int nheads = B * H;
    if (nqt + NLOAD > 16) return 0;
    const int nwaves = nqt + NLOAD;                         // one compute wave per query tile + the loader waves
    const size_t lds = (size_t)4 * NB * 32 * dh * 2 + (size_t)nqt * QSLOT + 64 + (size_t)nqt * 1024;   // images, Q slots, counter, O strips
    if (lds > 160 * 1024) return 0;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        ncu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    const int grid = nheads < ncu ? nheads : ncu;
    const float scale = 1.0f / sqrtf((float)dh);
    const bf16* q = (const bf16*)qkv;
    bf16* o = (bf16*)out;
    const int nt = nwaves * 64;
    switch (NB) {
        case 1: launch_fwd_p<1>(causal, grid, nt, lds, st, q, o, lse, L, Lq, H, nheads, scale, (unsigned)ob, (unsigned)lb); break;
        case 2: launch_fwd_p<2>(causal, grid, nt, lds, st, q, o, lse, L, Lq, H, nheads, scale, (unsigned)ob, (unsigned)lb); break;
        case 3: launch_fwd_p<3>(causal, grid, nt, lds, st, q, o, lse, L, Lq, H, nheads, scale, (unsigned)ob, (unsigned)lb); break;
        case 4: launch_fwd_p<4>(causal, grid, nt, lds, st, q, o, lse, L, Lq, H, nheads, scale, (unsigned)ob, (unsigned)lb); break;
        case 5: launch_fwd_p<5>(causal, grid, nt, lds, st, q, o, lse, L, Lq, H, nheads, scale, (unsigned)ob, (unsigned)lb); break;
        case 6: launch_fwd_p<6>(causal, grid, nt, lds, st, q, o, lse, L, Lq, H, nheads, scale, (unsigned)ob, (unsigned)lb); break;
        case 7: launch_fwd_p<7>(causal, grid, nt, lds, st, q, o, lse, L, Lq, H, nheads, scale, (unsigned)ob, (unsigned)lb); break;
        default: return 0;
    }
    return 1;
}
