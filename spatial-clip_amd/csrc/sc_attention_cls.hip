// Attention backward when only the FIRST query of every sequence is consumed (q_rows == 1): the last ViT block feeds only
// the class token on (pool_type='tok', src/open_clip/transformer.py:800-823).  The score matrix of a head is then one
// row: p[k] = softmax_k(q0 . K[k] / sqrt(dh)), and
//     dV[k] = p[k] dO0          dS[k] = p[k] (dO0 . V[k] - dO0 . O0)
//     dK[k] = dS[k] q0 / sqrt(dh)                 dQ[0] = sum_k dS[k] K[k] / sqrt(dh),   dQ[q > 0] = 0
// -- rank-one outputs, no matrix product.  The general kernels load four LDS images and run two MFMA passes for this
// (177 us per ViT-B/16 layer at B = 256, plus a 232 MB memset of dqkv in the caller); here one workgroup per (batch, head)
// streams K and V once, thread k owning key k (a whole 64- or 128-byte row per thread: full-line loads and stores), in
// fp32.  HBM-bound: reads K, V (155 MB) + writes dqkv (232 MB, the zero rows of dQ included, so no memset).
#include "sc_attn_common.h"

namespace {

template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_cls_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                           const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                           float* __restrict__ delta, bf16* __restrict__ dqkv, int L, int H,
                                                           float scale, int causal) {
    __shared__ float q0[DH], g0[DH], red[4][DH], sds[MAXL], sdel;
    constexpr int CH = DH / 8;                           // 16-byte chunks per row
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
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
    for (int k = t; k < L; k += 256) {
        const bf16* krow = base + (long long)k * rs + d;
        const bf16* vrow = krow + d;
        float s = 0.f, dp = 0.f;
        bf16x8 kk[CH], vv[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            kk[c] = *reinterpret_cast<const bf16x8*>(krow + c * 8);
            vv[c] = *reinterpret_cast<const bf16x8*>(vrow + c * 8);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s = fmaf(q0[c * 8 + e], (float)kk[c][e], s);
                dp = fmaf(g0[c * 8 + e], (float)vv[c][e], dp);
            }
        float p = fast_exp2(fmaf(s, c2, nl2));
        if (causal && k > 0) p = 0.f;                    // query 0 sees key 0 only
        const float ds = p * (dp - dl);
        sds[k] = ds;
        bf16* dkrow = dbase + (long long)k * rs + d;
        bf16* dvrow = dkrow + d;
        const float dsk = ds * scale;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            bf16x8 ok, ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ok[e] = (bf16)(dsk * q0[c * 8 + e]);
                ov[e] = (bf16)(p * g0[c * 8 + e]);
            }
            *reinterpret_cast<bf16x8*>(dkrow + c * 8) = ok;
            *reinterpret_cast<bf16x8*>(dvrow + c * 8) = ov;
        }
    }
    __syncthreads();
    // dQ[0][e] = scale * sum_k dS[k] K[k][e]: wave w takes keys w, w+4, ...; lanes = columns (K rows come from L2 now)
    {
        float acc = 0.f;
        if (lane < DH)
            for (int k = wave; k < L; k += 4) acc = fmaf(sds[k], (float)base[(long long)k * rs + d + lane], acc);
        if (lane < DH) red[wave][lane] = acc;
    }
    __syncthreads();
    if (t < DH) dbase[t] = (bf16)((red[0][t] + red[1][t] + red[2][t] + red[3][t]) * scale);
    // dQ of the queries nobody consumed: exact zeros (the qkv data-gradient GEMM reads every row of dqkv)
    const u32x4 z = (u32x4){0u, 0u, 0u, 0u};
    for (int i = t; i < (L - 1) * CH; i += 256) {
        const int row = 1 + i / CH, c = i % CH;
        *reinterpret_cast<u32x4*>(dbase + (long long)row * rs + c * 8) = z;
    }
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
