// Persistent multi-head attention forward for 224 < L <= 288, dh = 64 (round 5): ViT-L/14's 257 tokens
// (src/open_clip/model_configs/ViT-L-14.json; SDPA inside nn.MultiheadAttention, src/open_clip/transformer.py:253,272-287).
//
// sc_attention_p.hip stops at L = 224: seven 32-key blocks are what a double-buffered {K, V} image pair (4 x 28 KiB), one
// 2-KiB Q slot per 16-query tile and one compute wave per tile (13 + 3 loaders = 16 waves) fit into 160 KiB / 1024 threads.
// At 257 tokens there are nine key blocks (4 images of 36 KiB = 144 KiB before a single Q slot) and 17 query tiles.  The
// same design is kept -- one workgroup per CU walks a list of heads, loader waves stream the next head in by LDS-DMA while
// the compute waves work on the current one, no online softmax (the whole key range of a head is LDS-resident) -- with two
// changes that make it fit:
//   * a compute wave owns TWO query tiles (tile w and tile w + ncomp) and runs them one after the other: 9 compute + 3
//     loader waves = 12 waves (116-123 VGPRs, no scratch: nine score blocks of a tile stay in registers);
//   * only K is double-buffered.  V has ONE image: V of head i is requested right behind the barrier that ends head i - 1
//     (every P.V of that head is done) and is needed only after the scores, maxima and exponentials of the first tile, i.e.
//     a third of a head's time later; an LDS counter (`vready`) tells the compute waves it has landed.
// LDS at L = 257: K 2 x 36 KiB, V 36 KiB, 17 Q slots 34 KiB, counters, 9 output strips 9 KiB = 151.1 KiB.
// Synchronisation per head i: barrier B(i) [K(i), Q(i) landed; head i - 1 fully computed] -> loaders: V(i) [i > 0], bump
// `vready` when it has landed, K(i + 1) into the other K image, Q(i + 1) once the arrival counter says every compute wave
// holds its Q fragments -> compute: Q fragments, bump `arrive`, per tile {scores, maximum, exponentials, wait vready, P.V,
// store}.  Every wait is on a counter that is bumped by waves that do not wait for the waiter in between (loaders bump
// `vready` before they look at `arrive`; compute waves bump `arrive` before they look at `vready`).
#include "sc_attn_common.h"
#include <stdlib.h>

namespace {

constexpr int P2DH = 64;
constexpr int P2QSLOT = 16 * P2DH * 2;     // one 16-query tile: 2 KiB
constexpr int P2NLOAD = 3;                 // loader waves

SC_DEVICE float p2max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <int NB, bool CAUSAL>
__global__ __launch_bounds__(768) void attn_fwd_p2_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                          float* __restrict__ lse, int L, int Lq, int H, int nheads,
                                                          float scale, unsigned out_bytes, unsigned lse_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DH = P2DH, KS = DH / 32, DT = DH / 16;
    constexpr int Lp = NB * 32;
    constexpr int IMG = Lp * DH * 2;                 // bytes of one K or V image
    constexpr int PIECES = Lp / 8;                   // 1-KiB DMA pieces (8 rows x 128 B) per image
    const int t = threadIdx.x, lane = t & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwaves = blockDim.x >> 6;
    const int ncomp = nwaves - P2NLOAD;              // compute waves = ceil(query tiles / 2)
    const int nqt = (Lq + 15) >> 4;
    const int d = H * DH;
    const long long rs = 3LL * d;

    char* Vimg = smem + 2 * IMG;
    char* qbase = smem + 3 * IMG;
    const unsigned arrive = (unsigned)(uintptr_t)(lptr_t)smem + 3 * IMG + nqt * P2QSLOT;       // compute waves: Q fragments taken
    const unsigned vready = arrive + 4;                                                           // loader waves: V of this head landed
    if (t == 0) asm volatile("ds_write_b32 %0, %2\n\tds_write_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)" ::"v"(arrive), "v"(vready), "v"(0u) : "memory");
    wg_barrier();                                    // the counters are zero before any wave bumps or polls one

    if (wave >= ncomp) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wave - ncomp;
        __builtin_amdgcn_s_setprio(3);               // the DMA issue must never queue behind the compute waves' VALU
        const int prow = lane >> 3, pch = lane & 7;  // lane -> (row in piece, 16-byte chunk); swizzle on the SOURCE chunk
        auto issue_img = [&](int head, int which, char* dst) {        // which: 0 = K, 1 = V
            const int b = head / H, h = head % H;
            const bf16* base = qkv + (long long)b * L * rs + h * DH + (1 + which) * d;
            for (int p = lw; p < PIECES; p += P2NLOAD) {
                const int row = p * 8 + prow;
                const int rowc = min(row, L - 1);    // padding rows: finite copies of the last row (masked / p = 0)
                dma16(base + (long long)rowc * rs + (pch ^ Img<DH>::swz(row)) * 8, dst + p * 1024);
            }
        };
        auto issue_q = [&](int head) {
            const int b = head / H, h = head % H;
            const bf16* base = qkv + (long long)b * L * rs + h * DH;
            for (int p = lw; p < 2 * nqt; p += P2NLOAD) {              // 2 pieces per 16-query tile, swizzled per tile row
                const int r = (p & 1) * 8 + prow;
                const int rowc = min((p >> 1) * 16 + r, L - 1);
                dma16(base + (long long)rowc * rs + (pch ^ Img<DH>::swz(r)) * 8, qbase + p * 1024);
            }
        };
        int head = blockIdx.x;
        if (head < nheads) { issue_img(head, 0, smem); issue_q(head); issue_img(head, 1, Vimg); }
        for (int i = 0; head < nheads; ++i, head += gridDim.x) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // K, Q of this head (and V of the first) have landed
            wg_barrier();                                                // B(i): ... and head i - 1 is fully computed
            if (i > 0) {
                issue_img(head, 1, Vimg);                                // the single V image is free now
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (lane == 0) lds_bump(vready);                             // vready == NLOAD (i + 1): V of head i is in LDS
            const int next = head + gridDim.x;
            if (next < nheads) {
                issue_img(next, 0, smem + ((i + 1) & 1) * IMG);          // the other K image was last read in head i - 1
                lds_wait_ge(arrive, (unsigned)ncomp * (unsigned)(i + 1)); // every compute wave holds its Q fragments
                issue_q(next);
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const float c2 = scale * 1.4426950408889634f;    // exp(x*scale) = exp2(x*c2)
    const __amdgpu_buffer_rsrc_t out_rsrc = sc_make_rsrc(out, out_bytes);
    const __amdgpu_buffer_rsrc_t lse_rsrc = sc_make_rsrc(lse, lse_bytes);
    char* ostrip = qbase + nqt * P2QSLOT + 64 + wave * 1024;
    const bool two = wave + ncomp < nqt;             // does this wave own a second tile?

    int head = blockIdx.x;
    for (int i = 0; head < nheads; ++i, head += gridDim.x) {
        const char* Kimg = smem + (i & 1) * IMG;
        wg_barrier();                                // B(i): K / Q of this head are in LDS (the loader waves waited)
        const int b = head / H, h = head % H;

#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {         // two sequential tiles per wave (unrolled: one register allocation each)
            if (tt == 1 && !two) continue;
            const int tile = wave + tt * ncomp;
            // lane-dependent LDS addresses are re-derived from a laundered lane id per tile: hoisted out of the head loop they
            // cost ~30 registers and come back as scratch reloads, each of which waits for the previous tile's stores (vmcnt)
            int lane_l = lane;
            asm volatile("" : "+v"(lane_l));
            const int li = lane_l & 15, lg = lane_l >> 4;
            const int q = tile * 16 + li;            // this lane's query (MFMA column)
            // this tile's Q fragments; the wave's Q slots may be refilled once the LAST of them is in registers
            bf16x8 qf[KS];
            {
                const char* qslot = qbase + tile * P2QSLOT;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_row<DH>(qslot, 0, ks, li, lg);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0 && (tt == 1 || !two)) lds_bump(arrive);
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
                mx = p2max3(mx, s[nb][0][0], s[nb][0][1]);
                mx = p2max3(mx, s[nb][0][2], s[nb][0][3]);
                mx = p2max3(mx, s[nb][1][0], s[nb][1][1]);
                mx = p2max3(mx, s[nb][1][2], s[nb][1][3]);
            }
            mx = quad_max(mx);
            // ---- phase 3: probabilities, row sum, bf16 fragments (k-slot order = accumulator order)
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
            // ---- V of this head must have landed (single V image: requested behind B(i))
            lds_wait_ge(vready, (unsigned)P2NLOAD * (unsigned)(i + 1));
            // ---- phase 4: O^T = V^T . P^T, transposed V fragments one block ahead of the MFMAs that consume them
            __builtin_amdgcn_sched_barrier(0);
            f32x4 o[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
            // ---- epilogue: lane holds O[q][dt*16 + 4 lg .. +3]; whole 128-byte rows per store through the wave's strip
            const float inv = 1.0f / lsum;
            const bool ok = q < Lq;
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
                    const int r8 = lane_l >> 3, ch = lane_l & 7;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(ostrip + r8 * 128 + ((ch ^ (r8 >> 1)) << 4));
                    const int qr = tile * 16 + half * 8 + r8;
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
}

template <int NB>
void launch_fwd_p2(bool causal, int grid, int nthreads, size_t lds, hipStream_t st, const bf16* qkv, bf16* out, float* lse,
                   int L, int Lq, int H, int nheads, float scale, unsigned ob, unsigned lb) {
    if (causal) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_p2_kernel<NB, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attn_fwd_p2_kernel<NB, true><<<grid, nthreads, lds, st>>>(qkv, out, lse, L, Lq, H, nheads, scale, ob, lb);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_p2_kernel<NB, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attn_fwd_p2_kernel<NB, false><<<grid, nthreads, lds, st>>>(qkv, out, lse, L, Lq, H, nheads, scale, ob, lb);
    }
}

}  // namespace

// returns 1 if the kernel took the launch, 0 if the shape is outside its range (caller falls back)
int sc_attn_fwd_persistent2(const void* qkv, void* out, float* lse, int B, int L, int Lq, int H, int dh, int causal,
                            hipStream_t st) {
    if (dh != P2DH || L <= 224 || L > 288 || Lq < 1 || Lq > L) return 0;
    const int NB = (L + 31) / 32;                            // 8 or 9
    const int nqt = (Lq + 15) / 16;
    const int ncomp = (nqt + 1) / 2;
    if (ncomp + P2NLOAD > 12) return 0;
    const long long ob = (long long)B * L * H * dh * 2, lb = (long long)B * H * L * 4;
    if (ob >= 0xFFFFFFF0ll) return 0;
    const size_t lds = (size_t)3 * NB * 32 * dh * 2 + (size_t)nqt * P2QSLOT + 64 + (size_t)ncomp * 1024;
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
    const int nt = (ncomp + P2NLOAD) * 64;
    const bf16* q = (const bf16*)qkv;
    bf16* o = (bf16*)out;
    if (NB == 8) launch_fwd_p2<8>(causal, grid, nt, lds, st, q, o, lse, L, Lq, H, nheads, scale, (unsigned)ob, (unsigned)lb);
    else launch_fwd_p2<9>(causal, grid, nt, lds, st, q, o, lse, L, Lq, H, nheads, scale, (unsigned)ob, (unsigned)lb);
    return 1;
}
