// Patch embedding glue of the ViT tower (reference: VisionTransformer._embeds, src/open_clip/transformer.py:783-798).
//   sc_im2col        : NCHW fp32 images -> [B*G*G, 3*P*P] bf16 patch matrix (inner order c,py,px = conv1.weight)
//   sc_embed_ln_fwd  : tokens = [class_embedding ; patch GEMM out] + positional_embedding, then ln_pre,
//                      written as the fp32 residual stream
//   sc_embed_ln_bwd  : LN backward of ln_pre; emits d(token) fp32 (in place), the packed bf16 d(patch out)
//                      for the conv wgrad GEMM, and dgamma/dbeta
//   sc_batch_sum     : d(positional_embedding)[t] = sum_b d(token)[b,t]  (row 0 is also d(class_embedding))
#include "sc_common.h"
#include "sc_kernels.h"

namespace {

constexpr int MAXV = 8;
SC_DEVICE f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
SC_DEVICE void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__global__ void im2col_kernel(const float* __restrict__ img, bf16* __restrict__ out, int B, int C, int H, int W, int P,
                              long long ld_out) {
    // one thread per (b, gy, gx, c, py): copies P pixels
    const int G_h = H / P, G_w = W / P;
    const long long total = (long long)B * G_h * G_w * C * P;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int gx = (int)(r % G_w); r /= G_w;
        const int py = (int)(r % P); r /= P;
        const int c = (int)(r % C); r /= C;
        const int gy = (int)(r % G_h); r /= G_h;
        const int b = (int)r;
        const float* src = img + (((long long)b * C + c) * H + gy * P + py) * W + gx * P;
        bf16* dst = out + ((long long)(b * G_h + gy) * G_w + gx) * ld_out + (c * P + py) * P;
        if ((P & 3) == 0 && (W & 3) == 0) {
            for (int x = 0; x < P; x += 4) {
                const f32x4 v = ld4(src + x);
                bf16x4 o;
                o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
                *reinterpret_cast<bf16x4*>(dst + x) = o;
            }
        } else {
            for (int x = 0; x < P; ++x) dst[x] = (bf16)src[x];
        }
    }
}

// im2col for patch sizes that are multiples of 8: one workgroup per (image, patch row) stages the C x P image rows of
// the strip in LDS as bf16 (coalesced 16-byte reads of whole image rows) and writes the strip's W / P patch rows as
// contiguous 16-byte chunks.  The thread-per-pixel-run kernel above writes 32-byte pieces to rows 1.5 KB apart
// (99 us for ViT-B/16 at B = 256: 2.3 TB/s).
__global__ __launch_bounds__(256) void im2col_strip_kernel(const float* __restrict__ img, bf16* __restrict__ out, int C, int H,
                                                           int W, int P, long long ld_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* tile = reinterpret_cast<bf16*>(smem);                  // [C * P rows][W]
    const int G_h = H / P, G_w = W / P;
    const int b = blockIdx.x / G_h, gy = blockIdx.x % G_h;
    const int w4 = W >> 2, rows = C * P;
    for (int i = threadIdx.x; i < rows * w4; i += 256) {
        const int row = i / w4, x4 = i - row * w4;
        const int c = row / P, py = row - c * P;
        const f32x4 v = ld4(img + (((long long)b * C + c) * H + gy * P + py) * W + x4 * 4);
        bf16x4 o;
        o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
        *reinterpret_cast<bf16x4*>(tile + row * W + x4 * 4) = o;
    }
    __syncthreads();
    const int cpr = rows * P / 8;                                 // 16-byte chunks per patch row
    const int p8 = P / 8;
    for (int i = threadIdx.x; i < G_w * cpr; i += 256) {
        const int gx = i / cpr, j = i - gx * cpr;
        const int row = j / p8, px0 = (j - row * p8) * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(tile + row * W + gx * P + px0);
        *reinterpret_cast<u32x4*>(out + ((long long)(b * G_h + gy) * G_w + gx) * ld_out + j * 8) = v;
    }
}

// one wave per token row; NV = 16-byte lane slots per row (ceil(d / 256)): a compile-time trip count keeps every load of a
// row in flight at once (the generic 8-slot loop with run-time predicates ran at 2.5 TB/s, the LayerNorm kernels at 5.7)
template <int NV, bool XB = false>      // XB: the residual stream starts in bf16 (x points at bf16 rows): no fp32 copy + cast pass
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ x,
                                                           float* __restrict__ mean, float* __restrict__ rstd, int B,
                                                           int L, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= B * L) return;
    const int b = row / L, tkn = row - b * L;
    const float* src = tkn == 0 ? cls : patch + ((long long)b * (L - 1) + (tkn - 1)) * d;
    const float* pr = pos + (long long)tkn * d;
    const int nv = d >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = i * 64 + lane;
        if (e < nv) {
            v[i] = ld4(src + e * 4) + ld4(pr + e * 4);
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mu = sc_wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = i * 64 + lane;
        if (e < nv) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float u = v[i][c] - mu; q += u * u; }
        }
    }
    const float rs = rsqrtf(sc_wave_sum(q) / (float)d + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = i * 64 + lane;
        if (e < nv) {
            const f32x4 g = ld4(gamma + e * 4), bb = ld4(beta + e * 4);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (v[i][c] - mu) * rs * g[c] + bb[c];
            if (XB) {
                bf16x4 ob;
#pragma unroll
                for (int c = 0; c < 4; ++c) ob[c] = (bf16)o[c];
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(x) + (long long)row * d + e * 4) = ob;
            } else {
                st4(x + (long long)row * d + e * 4, o);
            }
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(float* __restrict__ dres, const float* __restrict__ patch,
                                                           const float* __restrict__ cls, const float* __restrict__ pos,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, bf16* __restrict__ dpatch,
                                                           float* __restrict__ partial, int B, int L, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform: row pointers in SGPRs
    const int nv = d >> 2;
    const int rows = B * L;
    f32x4 ag[NV], ab[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = ab[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int e = i * 64 + lane;
        gm[i] = e < nv ? ld4(gamma + e * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int b = row / L, tkn = row - b * L;
        const float* src = tkn == 0 ? cls : patch + ((long long)b * (L - 1) + (tkn - 1)) * d;
        const float* pr = pos + (long long)tkn * d;
        const float mu = mean[row], rs = rstd[row];
        float* dr = dres + (long long)row * d;
        f32x4 g[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = i * 64 + lane;
            if (e < nv) {
                const f32x4 dyv = ld4(dr + e * 4);
                const f32x4 xv = ld4(src + e * 4) + ld4(pr + e * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    xh[i][c] = (xv[c] - mu) * rs;
                    g[i][c] = dyv[c] * gm[i][c];
                    s1 += g[i][c];
                    s2 += g[i][c] * xh[i][c];
                    ag[i][c] += dyv[c] * xh[i][c];
                    ab[i][c] += dyv[c];
                }
            }
        }
        s1 = sc_wave_sum(s1) / (float)d;
        s2 = sc_wave_sum(s2) / (float)d;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = i * 64 + lane;
            if (e < nv) {
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = rs * (g[i][c] - s1 - xh[i][c] * s2);
                st4(dr + e * 4, o);
                if (tkn > 0) {
                    bf16x4 ob;
                    ob[0] = (bf16)o[0]; ob[1] = (bf16)o[1]; ob[2] = (bf16)o[2]; ob[3] = (bf16)o[3];
                    *reinterpret_cast<bf16x4*>(dpatch + ((long long)b * (L - 1) + (tkn - 1)) * d + e * 4) = ob;
                }
            }
        }
    }
    float* sm = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = i * 64 + lane;
        if (e < nv) {
            st4(sm + (wave * 2 + 0) * d + e * 4, ag[i]);
            st4(sm + (wave * 2 + 1) * d + e * 4, ab[i]);
        }
    }
    __syncthreads();
    float* pout = partial + (long long)blockIdx.x * 2 * d;
    for (int e = threadIdx.x; e < 2 * d; e += 256) pout[e] = sm[e] + sm[2 * d + e] + sm[4 * d + e] + sm[6 * d + e];
}

// out_k[c] = sum_b partial[b][k][c], k = 0, 1: block = 64 columns x 16 row groups, eight loads in flight per thread, fixed
// summation order (deterministic)
__global__ __launch_bounds__(1024) void colvec2_finalize_kernel(const float* __restrict__ partial, int nblk, int d,
                                                                 float* __restrict__ o0, float* __restrict__ o1) {
    __shared__ float sm[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (e < 2 * d) {
        float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = ty;
        for (; b + 7 * 16 < nblk; b += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] += partial[(long long)(b + u * 16) * 2 * d + e];
        }
        for (int u = 0; b < nblk; b += 16, ++u) p[u & 7] += partial[(long long)b * 2 * d + e];
        s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    }
    sm[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && e < 2 * d) {
        s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += sm[k][tx];
        if (e < d) o0[e] = s; else o1[e - d] = s;
    }
}

// out[i] = sum_b x[b*n + i], i < n (n = L*d), float4 lanes
__global__ void batch_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 3 < B; b += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] += reinterpret_cast<const f32x4*>(x)[(long long)(b + u) * n4 + i];
    }
    for (int u = 0; b < B; ++b, ++u) p[u & 3] += reinterpret_cast<const f32x4*>(x)[(long long)b * n4 + i];
    reinterpret_cast<f32x4*>(out)[i] = (p[0] + p[1]) + (p[2] + p[3]);
}

}  // namespace

extern "C" int sc_im2col(const float* images, void* patches, int B, int C, int H, int W, int P, long long ld_out,
                         void* stream) {
    SC_CHECK(B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0, "sc_im2col: bad shape B=%d C=%d H=%d W=%d P=%d", B, C,
             H, W, P);
    SC_CHECK(ld_out >= (long long)C * P * P && (ld_out % 4) == 0, "sc_im2col: ld_out too small / unaligned");
    const size_t strip = (size_t)C * P * W * 2;
    if ((P % 8) == 0 && (W % 8) == 0 && (ld_out % 8) == 0 && strip <= 64 * 1024) {
        if (strip > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&im2col_strip_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)strip);
        im2col_strip_kernel<<<B * (H / P), 256, strip, (hipStream_t)stream>>>(images, (bf16*)patches, C, H, W, P, ld_out);
        SC_LAUNCH_CHECK();
        return 0;
    }
    const long long total = (long long)B * (H / P) * (W / P) * C * P;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    im2col_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(images, (bf16*)patches, B, C, H, W, P, ld_out);
    SC_LAUNCH_CHECK();
    return 0;
}

static int embed_ln_fwd_launch(const float* patch_out, const float* cls, const float* pos, const float* gamma,
                               const float* beta, float* x, bool xb, float* mean, float* rstd, int B, int L, int d, float eps,
                               void* stream);

extern "C" int sc_embed_ln_fwd(const float* patch_out, const float* cls, const float* pos, const float* gamma,
                               const float* beta, float* x, float* mean, float* rstd, int B, int L, int d, float eps,
                               void* stream) {
    SC_CHECK(B > 0 && L > 1 && d > 0 && (d % 4) == 0 && d <= MAXV * 256, "sc_embed_ln_fwd: bad shape B=%d L=%d d=%d", B, L, d);
    return embed_ln_fwd_launch(patch_out, cls, pos, gamma, beta, x, false, mean, rstd, B, L, d, eps, stream);
}

extern "C" int sc_embed_ln_fwd_x16(const float* patch_out, const float* cls, const float* pos, const float* gamma,
                                   const float* beta, void* x_bf16, float* mean, float* rstd, int B, int L, int d, float eps,
                                   void* stream) {
    SC_CHECK(B > 0 && L > 1 && d > 0 && (d % 4) == 0 && d <= MAXV * 256 && x_bf16 != nullptr, "sc_embed_ln_fwd_x16: bad shape B=%d L=%d d=%d", B, L, d);
    return embed_ln_fwd_launch(patch_out, cls, pos, gamma, beta, (float*)x_bf16, true, mean, rstd, B, L, d, eps, stream);
}

static int embed_ln_fwd_launch(const float* patch_out, const float* cls, const float* pos, const float* gamma,
                               const float* beta, float* x, bool xb, float* mean, float* rstd, int B, int L, int d, float eps,
                               void* stream) {
#define SC_EMBED_FWD(NV)                                                                                                          \
    do {                                                                                                                          \
        if (xb) embed_ln_fwd_kernel<NV, true><<<(B * L + 3) / 4, 256, 0, (hipStream_t)stream>>>(patch_out, cls, pos, gamma, beta, \
                                                                                                 x, mean, rstd, B, L, d, eps);     \
        else embed_ln_fwd_kernel<NV, false><<<(B * L + 3) / 4, 256, 0, (hipStream_t)stream>>>(patch_out, cls, pos, gamma, beta,   \
                                                                                               x, mean, rstd, B, L, d, eps);     \
    } while (0)
    switch ((d / 4 + 63) / 64) {
        case 1: SC_EMBED_FWD(1); break;
        case 2: SC_EMBED_FWD(2); break;
        case 3: SC_EMBED_FWD(3); break;
        case 4: SC_EMBED_FWD(4); break;
        case 5: SC_EMBED_FWD(5); break;
        case 6: SC_EMBED_FWD(6); break;
        case 7: SC_EMBED_FWD(7); break;
        default: SC_EMBED_FWD(8); break;
    }
#undef SC_EMBED_FWD
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" long long sc_embed_ln_bwd_ws_floats(int B, int L, int d) {
    int nblk = (B * L + 3) / 4;
    if (nblk > 1024) nblk = 1024;      // 4 workgroups of 4 row-waves per CU: one per CU left the HBM-bound row loop latency-bound
    return (long long)nblk * 2 * d;
}

extern "C" int sc_embed_ln_bwd(float* dres, const float* patch_out, const float* cls, const float* pos,
                               const float* mean, const float* rstd, const float* gamma, void* dpatch_bf16,
                               float* dgamma, float* dbeta, float* dpos, float* dcls, float* ws, int B, int L, int d,
                               void* stream) {
    SC_CHECK(B > 0 && L > 1 && d > 0 && (d % 4) == 0 && d <= MAXV * 256, "sc_embed_ln_bwd: bad shape B=%d L=%d d=%d", B, L, d);
    hipStream_t st = (hipStream_t)stream;
    int nblk = (B * L + 3) / 4;
    if (nblk > 1024) nblk = 1024;      // 4 workgroups of 4 row-waves per CU: one per CU left the HBM-bound row loop latency-bound
    const size_t lds = (size_t)4 * 2 * d * sizeof(float);
#define SC_EMBED_BWD(NV)                                                                                              \
    do {                                                                                                              \
        if (lds > 48 * 1024)                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&embed_ln_bwd_kernel<NV>),                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                          \
        embed_ln_bwd_kernel<NV><<<nblk, 256, lds, st>>>(dres, patch_out, cls, pos, mean, rstd, gamma,                  \
                                                        (bf16*)dpatch_bf16, ws, B, L, d);                             \
    } while (0)
    switch ((d / 4 + 63) / 64) {
        case 1: SC_EMBED_BWD(1); break;
        case 2: SC_EMBED_BWD(2); break;
        case 3: SC_EMBED_BWD(3); break;
        case 4: SC_EMBED_BWD(4); break;
        case 5: SC_EMBED_BWD(5); break;
        case 6: SC_EMBED_BWD(6); break;
        case 7: SC_EMBED_BWD(7); break;
        default: SC_EMBED_BWD(8); break;
    }
#undef SC_EMBED_BWD
    SC_LAUNCH_CHECK();
    colvec2_finalize_kernel<<<(2 * d + 63) / 64, 1024, 0, st>>>(ws, nblk, d, dgamma, dbeta);
    SC_LAUNCH_CHECK();
    const long long n4 = (long long)L * d / 4;
    batch_sum_kernel<<<(int)((n4 + 255) / 256), 256, 0, st>>>(dres, dpos, B, n4);
    SC_LAUNCH_CHECK();
    // d(class_embedding) = d(token row 0) summed over the batch = dpos[0]
    (void)hipMemcpyAsync(dcls, dpos, (size_t)d * sizeof(float), hipMemcpyDeviceToDevice, st);
    return 0;
}

// ============================================================================================ text tower glue
// CLIP.encode_text (src/open_clip/model.py:330-345): token_embedding gather + positional embedding; EOT pooling
// = row at text.argmax(-1) (src/open_clip/transformer.py:931-934).
namespace {

__global__ void token_embed_fwd_kernel(const long long* __restrict__ tokens, const float* __restrict__ table,
                                       const float* __restrict__ pos, float* __restrict__ x, int rows, int L, int d,
                                       int V) {
    const int nv = d >> 2;
    const long long total = (long long)rows * nv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / nv), e = (int)(i - (long long)r * nv) * 4;
        long long tk = tokens[r];
        tk = tk < 0 ? 0 : (tk >= V ? V - 1 : tk);
        const f32x4 v = ld4(table + tk * d + e) + ld4(pos + (long long)(r % L) * d + e);
        st4(x + (long long)r * d + e, v);
    }
}

__global__ void token_embed_bwd_kernel(const long long* __restrict__ tokens, const float* __restrict__ dres,
                                       float* __restrict__ dtable, int rows, int d, int V) {
    const long long total = (long long)rows * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / d), e = (int)(i - (long long)r * d);
        const float g = dres[i];
        if (g != 0.f) {
            long long tk = tokens[r];
            tk = tk < 0 ? 0 : (tk >= V ? V - 1 : tk);
            atomicAdd(dtable + tk * d + e, g);
        }
    }
}

// Deterministic form of the scatter-add (round 6).  The atomic kernel above sums the rows of a repeated token (<start_of_text>
// occurs in every caption) in whatever order the hardware serialises the float atomics: the last bit of dtable differs from run
// to run, and with it every weight after the first optimiser step.  Here table row t belongs to token class c = t % NC, and ONE
// wave per (class, 256-column slab) walks the token list IN ORDER: 64 positions per look-up (ballot), the rows of its class's
// tokens added in increasing position -- consecutive occurrences of one token in a register accumulator, a read-modify-write
// of the table row only when the token changes -- so every element of dtable is formed by one lane in one fixed order.
// Positions behind the pooled one (l > eot[b]: causal tower, pooled at the EOT token) carry an exactly zero gradient and are
// skipped.  The eight waves of a workgroup share ONE copy of the token list in LDS (int32, -1 = skipped; segments of 32 K
// positions): the walk reads LDS, not memory -- the first version walked global memory with a division and a dependent eot
// look-up per 64 positions and took 370-430 us at B = 256 against 34 us for the atomic kernel.  Up to four gradient rows are in
// flight per wave.
constexpr int TE_NC = 1024;
constexpr int TE_SEG = 32768;
__global__ __launch_bounds__(512) void token_embed_bwd_det_kernel(const long long* __restrict__ tokens, const int* __restrict__ eot,
                                                                  const float* __restrict__ dres, float* __restrict__ dtable,
                                                                  int rows, int L, int d, int V, int nslab, int npairs) {
    extern __shared__ int te_tok[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 8 + wave;                       // (class, slab) of this wave; waves past the last pair only help to load
    const bool live = pair < npairs;
    const int cls = live ? pair / nslab : 0, slab = live ? pair - cls * nslab : 0;
    const int col = slab * 256 + lane * 4;
    const bool active = live && col < d;
    int cur = -1;                                   // token whose partial sum sits in acc
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto flush = [&]() {
        if (cur >= 0 && active) {
            float* w = dtable + (long long)cur * d + col;
            st4(w, ld4(w) + acc);
        }
    };
    for (int seg0 = 0; seg0 < rows; seg0 += TE_SEG) {
        const int nseg = min(TE_SEG, rows - seg0);
        __syncthreads();                            // every wave is through the previous segment
        for (int i = threadIdx.x; i < nseg; i += 512) {
            const int p = seg0 + i;
            long long t = tokens[p];
            t = t < 0 ? 0 : (t >= V ? V - 1 : t);
            const int b = p / L, l = p - b * L;
            te_tok[i] = (eot == nullptr || l <= eot[b]) ? (int)t : -1;
        }
        __syncthreads();
        if (!live) continue;
        for (int base = 0; base < nseg; base += 64) {
            const int i = base + lane;
            const int tk = i < nseg ? te_tok[i] : -1;
            unsigned long long mask = __ballot(tk >= 0 && (tk % TE_NC) == cls);
            while (mask) {
                int bit[4], tok[4];
                f32x4 g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {       // up to four matches: their gradient rows requested together
                    bit[u] = mask ? __builtin_ctzll(mask) : -1;
                    if (mask) mask &= mask - 1;
                    tok[u] = bit[u] >= 0 ? __shfl(tk, bit[u], 64) : -1;
                    g[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (bit[u] >= 0 && active) g[u] = ld4(dres + (long long)(seg0 + base + bit[u]) * d + col);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (bit[u] < 0) break;
                    if (tok[u] != cur) {
                        flush();
                        cur = tok[u];
                        acc = g[u];
                    } else {
                        acc += g[u];
                    }
                }
            }
        }
    }
    flush();
}

// one wave per row: lanes take tokens lane, lane + 64, ...; first maximum (smallest index among equal values), like torch.argmax
__global__ __launch_bounds__(256) void argmax_rows_kernel(const long long* __restrict__ tokens, int* __restrict__ out, int B, int L) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    long long best = tokens[(long long)b * L + (lane < L ? lane : 0)];
    int bi = lane < L ? lane : 0;
    for (int t = lane + 64; t < L; t += 64) {
        const long long v = tokens[(long long)b * L + t];
        if (v > best) { best = v; bi = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) out[b] = bi;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, int L,
                                   float* __restrict__ dst, int B, int d) {
    const int nv = d >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nv) return;
    const int b = i / nv, e = (i - b * nv) * 4;
    st4(dst + (long long)b * d + e, ld4(src + ((long long)b * L + idx[b]) * d + e));
}

__global__ void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, int L,
                                    float* __restrict__ dst, bf16* __restrict__ dst_bf, int B, int d) {
    const int nv = d >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nv) return;
    const int b = i / nv, e = (i - b * nv) * 4;
    const f32x4 v = ld4(src + (long long)b * d + e);
    const long long row = (long long)b * L + idx[b];
    st4(dst + row * d + e, v);
    if (dst_bf) {
        bf16x4 o;
        o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
        *reinterpret_cast<bf16x4*>(dst_bf + row * d + e) = o;
    }
}

}  // namespace

extern "C" int sc_token_embed_fwd(const long long* tokens, const float* table, const float* pos, float* x, int B, int L,
                                  int d, int V, void* stream) {
    SC_CHECK(B > 0 && L > 0 && d > 0 && (d % 4) == 0 && V > 0, "sc_token_embed_fwd: bad shape");
    const long long total = (long long)B * L * (d / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    token_embed_fwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(tokens, table, pos, x, B * L, L, d, V);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_token_embed_bwd(const long long* tokens, const float* dres, float* dtable, float* dpos, int B, int L,
                                  int d, int V, void* stream) {
    SC_CHECK(B > 0 && L > 0 && d > 0 && (d % 4) == 0 && V > 0, "sc_token_embed_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(dtable, 0, (size_t)V * d * sizeof(float), st);
    const long long total = (long long)B * L * d;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    token_embed_bwd_kernel<<<blocks, 256, 0, st>>>(tokens, dres, dtable, B * L, d, V);
    SC_LAUNCH_CHECK();
    const long long n4 = (long long)L * d / 4;
    batch_sum_kernel<<<(int)((n4 + 255) / 256), 256, 0, st>>>(dres, dpos, B, n4);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_token_embed_bwd_det(const long long* tokens, const int* eot, const float* dres, float* dtable, float* dpos,
                                      int B, int L, int d, int V, void* stream) {
    SC_CHECK(B > 0 && L > 0 && d > 0 && (d % 4) == 0 && V > 0, "sc_token_embed_bwd_det: bad shape");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(dtable, 0, (size_t)V * d * sizeof(float), st);
    const int nslab = (d + 255) / 256;
    const int npairs = TE_NC * nslab;
    const int rows = B * L;
    const size_t lds = (size_t)(rows < TE_SEG ? rows : TE_SEG) * sizeof(int);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&token_embed_bwd_det_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              TE_SEG * (int)sizeof(int));
    token_embed_bwd_det_kernel<<<(npairs + 7) / 8, 512, lds, st>>>(tokens, eot, dres, dtable, rows, L, d, V, nslab, npairs);
    SC_LAUNCH_CHECK();
    const long long n4 = (long long)L * d / 4;
    batch_sum_kernel<<<(int)((n4 + 255) / 256), 256, 0, st>>>(dres, dpos, B, n4);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_argmax_rows_i64(const long long* tokens, int* out_idx, int B, int L, void* stream) {
    SC_CHECK(B > 0 && L > 0, "sc_argmax_rows_i64: bad shape");
    argmax_rows_kernel<<<(B + 3) / 4, 256, 0, (hipStream_t)stream>>>(tokens, out_idx, B, L);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_gather_rows_f32(const float* src, const int* idx, int L, float* dst, int B, int d, void* stream) {
    SC_CHECK(B > 0 && d > 0 && (d % 4) == 0, "sc_gather_rows_f32: bad shape");
    gather_rows_kernel<<<(B * (d / 4) + 255) / 256, 256, 0, (hipStream_t)stream>>>(src, idx, L, dst, B, d);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_scatter_rows_f32(const float* src, const int* idx, int L, float* dst, void* dst_bf16, int B, int d,
                                   void* stream) {
    SC_CHECK(B > 0 && d > 0 && (d % 4) == 0, "sc_scatter_rows_f32: bad shape");
    scatter_rows_kernel<<<(B * (d / 4) + 255) / 256, 256, 0, (hipStream_t)stream>>>(src, idx, L, dst, (bf16*)dst_bf16, B, d);
    SC_LAUNCH_CHECK();
    return 0;
}
