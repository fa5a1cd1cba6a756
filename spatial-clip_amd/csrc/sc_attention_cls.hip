// Attention backward when only the FIRST query of every sequence is consumed (q_rows == 1): the last ViT block feeds only
// the class token on (pool_type='tok', src/open_clip/transformer.py:800-823).  The score matrix of a head is then one
// row: p[k] = softmax_k(q0 . K[k] / sqrt(dh)), and
//     dV[k] = p[k] dO0          dS[k] = p[k] (dO0 . V[k] - dO0 . O0)
//     dK[k] = dS[k] q0 / sqrt(dh)                 dQ[0] = sum_k dS[k] K[k] / sqrt(dh),   dQ[q > 0] = 0
// -- rank-one outputs, no matrix product.  The general kernels load four LDS images and run two MFMA passes for this
// (177 us per ViT-B/16 layer at B = 256, plus a 232 MB memset of dqkv in the caller); here one workgroup per (batch, head)
// streams K and V once (8 or 4 lanes per key row, so that every wave instruction moves whole lines), in fp32.  HBM-bound: reads K, V (155 MB) + writes dqkv (232 MB, the zero rows of dQ included, so no memset).
#include "sc_attn_common.h"

namespace {

template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_cls_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                           const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                           float* __restrict__ delta, bf16* __restrict__ dqkv, int L, int H,
                                                           float scale, int causal) {
    // CH lanes share a row (16 bytes each): one wave instruction covers 64 / CH whole rows, i.e. full 128- / 64-byte lines
    // (a lane-per-row layout made every 16-byte load touch 64 different lines: 195 us instead of ~80 at B = 256)
    constexpr int CH = DH / 8;                           // 16-byte chunks per row: 8 (dh 64) or 4 (dh 32)
    constexpr int RPW = 64 / CH;                         // rows per wave instruction
    __shared__ float q0[DH], g0[DH], red[4][DH], sdel;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c = lane % CH, r = lane / CH;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int d = H * DH;
    const long long rs = 3LL * d;
    const bf16* base = qkv + (long long)b * L * rs + h * DH;
    bf16* dbase = dqkv + (long long)b * L * rs + h * DH;
    if (t < 64) {
        float part = 0.f;
        if (t < DH) {
            const float qv = (float)base[t], gv = (float)dout[(long long)b * L * d + h * DH + t];
            q0[t] = qv;
            g0[t] = gv;
            part = gv * (float)out[(long long)b * L * d + h * DH + t];
        }
        part = sc_wave_sum(part);
        if (t == 0) {
            sdel = part;
            delta[((long long)b * H + h) * L] = part;
        }
    }
    __syncthreads();
    const float dl = sdel;
    const float nl2 = -lse[((long long)b * H + h) * L] * 1.4426950408889634f;
    const float c2 = scale * 1.4426950408889634f;
    float qc[8], gc[8], dq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { qc[e] = q0[c * 8 + e]; gc[e] = g0[c * 8 + e]; dq[e] = 0.f; }
    const u32x4 z = (u32x4){0u, 0u, 0u, 0u};
    for (int k0 = wave * RPW; k0 < L; k0 += 4 * RPW) {
        const int k = k0 + r;
        const bool live = k < L;
        const int kc = live ? k : L - 1;
        const bf16x8 kk = *reinterpret_cast<const bf16x8*>(base + (long long)kc * rs + d + c * 8);
        const bf16x8 vv = *reinterpret_cast<const bf16x8*>(base + (long long)kc * rs + 2 * d + c * 8);
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s = fmaf(qc[e], (float)kk[e], s);
            dp = fmaf(gc[e], (float)vv[e], dp);
        }
#pragma unroll
        for (int o = 1; o < CH; o <<= 1) {               // over the CH lanes of the row
            s += __shfl_xor(s, o, 64);
            dp += __shfl_xor(dp, o, 64);
        }
        float p = fast_exp2(fmaf(s, c2, nl2));
        if ((causal && k > 0) || !live) p = 0.f;         // query 0 sees key 0 only
        const float ds = p * (dp - dl);
        const float dsk = ds * scale;
        bf16x8 ok, ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ok[e] = (bf16)(dsk * qc[e]);
            ov[e] = (bf16)(p * gc[e]);
            dq[e] = fmaf(ds, (float)kk[e], dq[e]);       // dQ[0] = sum_k dS[k] K[k] (x scale at the end)
        }
        if (live) {
            *reinterpret_cast<bf16x8*>(dbase + (long long)k * rs + d + c * 8) = ok;
            *reinterpret_cast<bf16x8*>(dbase + (long long)k * rs + 2 * d + c * 8) = ov;
            if (k > 0) *reinterpret_cast<u32x4*>(dbase + (long long)k * rs + c * 8) = z;     // dQ of an unconsumed query
        }
    }
    // dQ[0]: sum the per-lane partials over the rows of the wave (lanes with equal c), then over the four waves
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = dq[e];
#pragma unroll
        for (int o = CH; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        dq[e] = v;
    }
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave][c * 8 + e] = dq[e];
    }
    __syncthreads();
    if (t < DH) dbase[t] = (bf16)((red[0][t] + red[1][t] + red[2][t] + red[3][t]) * scale);
}

}  // namespace

// returns 1 if this kernel took the launch, 0 if the shape is outside its range (caller falls back)
int sc_attn_bwd_cls(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
                    int L, int Lq, int H, int dh, int causal, hipStream_t st) {
    if (Lq != 1 || L < 2 || L > MAXL || (dh != 64 && dh != 32)) return 0;
    const float scale = 1.0f / sqrtf((float)dh);
    if (dh == 64)
        attn_bwd_cls_kernel<64><<<B * H, 256, 0, st>>>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, delta,
                                                       (bf16*)dqkv, L, H, scale, causal);
    else
        attn_bwd_cls_kernel<32><<<B * H, 256, 0, st>>>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, delta,
                                                       (bf16*)dqkv, L, H, scale, causal);
    return 1;
}
