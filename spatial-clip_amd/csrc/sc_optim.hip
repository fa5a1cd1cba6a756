// Optimiser step on flat fp32 buffers: global grad-norm (for clip_grad_norm_ 1.0), fused AdamW with the clip
// coefficient and the 1/world_size gradient averaging applied on the fly, no host synchronisation.
//   reference: torch.optim.AdamW(lr, betas=(0.9,0.98), eps=1e-6, weight_decay=0.1) over ALL parameters
//   (src/models/spatial_clip_module.py:138-158, configs/optimizer/adamw.yaml) + Lightning gradient_clip_val 1.0
//   (configs/trainer/default.yaml:19 == torch.nn.utils.clip_grad_norm_).
#include "sc_common.h"
#include "sc_kernels.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n,
                                                            double* __restrict__ partial) {
    __shared__ double sm[4];
    double acc = 0.0;
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double a4[4] = {0.0, 0.0, 0.0, 0.0};                 // four loads in flight per thread, four independent fp64 chains
    for (; i + 3 * stride < n4; i += 4 * stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[i + u * stride];
            a4[u] += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
        }
    }
    for (; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        acc += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
    }
    acc += (a4[0] + a4[1]) + (a4[2] + a4[3]);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long long i = n4 << 2; i < n; ++i) acc += (double)x[i] * x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// out[0] = sqrt(sum)*gscale (the norm of the averaged gradient), out[1] = clip coefficient
__global__ void sumsq_final_kernel(const double* __restrict__ partial, int nblk, float gscale, float max_norm,
                                   float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(s) * gscale;
        out[0] = norm;
        out[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f;
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, float gscale, const float* __restrict__ normclip,
                                                    bf16* __restrict__ pbf, const float* __restrict__ hyper = nullptr) {
    if (hyper != nullptr) {     // step-dependent scalars from device memory: a captured hipGraph replays this launch every step
        lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2];
    }
    const float gs = gscale * (normclip ? normclip[1] : 1.0f);
    const long long n4 = n >> 2;
    const float decay = 1.0f - lr * wd;
    const float step = lr / bc1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float ge = gg[c] * gs;
            pp[c] *= decay;
            mm[c] = b1 * mm[c] + (1.0f - b1) * ge;
            vv[c] = b2 * vv[c] + (1.0f - b2) * ge * ge;
            const float denom = sqrtf(vv[c]) / bc2_sqrt + eps;
            pp[c] -= step * (mm[c] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        if (pbf) {      // bf16 mirror of the master buffer: the forward GEMM operands are views of it
            bf16x4 o;
            o[0] = (bf16)pp[0]; o[1] = (bf16)pp[1]; o[2] = (bf16)pp[2]; o[3] = (bf16)pp[3];
            reinterpret_cast<bf16x4*>(pbf)[i] = o;
        }
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
}

}  // namespace

extern "C" int sc_grad_norm(const float* grads, long long n, float grad_scale, float max_norm, double* ws,
                            float* norm_clip_out, void* stream) {
    SC_CHECK(n > 0 && ws && norm_clip_out, "sc_grad_norm: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = 1024;
    sumsq_partial_kernel<<<nblk, 256, 0, st>>>(grads, n, ws);
    SC_LAUNCH_CHECK();
    sumsq_final_kernel<<<1, 64, 0, st>>>(ws, nblk, grad_scale, max_norm, norm_clip_out);
    SC_LAUNCH_CHECK();
    return 0;
}

// The two halves of sc_grad_norm as separate entry points, for the sharded optimiser (comm.ShardedGradExchange): every rank
// sums the squares of ITS shard of the reduced gradient into 1024 fp64 partials per contiguous piece, the partial arrays are
// all-reduced (SUM, fp64: a few KiB), and every rank finishes with the same global norm / clip coefficient.
extern "C" int sc_grad_sumsq_partial(const float* grads, long long n, double* partial1024, void* stream) {
    SC_CHECK(n > 0 && partial1024, "sc_grad_sumsq_partial: bad args");
    sumsq_partial_kernel<<<1024, 256, 0, (hipStream_t)stream>>>(grads, n, partial1024);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_grad_norm_final(const double* partial, int n_partial, float grad_scale, float max_norm, float* norm_clip_out,
                                  void* stream) {
    SC_CHECK(n_partial > 0 && partial && norm_clip_out, "sc_grad_norm_final: bad args");
    sumsq_final_kernel<<<1, 64, 0, (hipStream_t)stream>>>(partial, n_partial, grad_scale, max_norm, norm_clip_out);
    SC_LAUNCH_CHECK();
    return 0;
}

// host-side: the two bias-correction scalars exactly as sc_adamw_step forms them (one definition for the eager and the
// graph-replayed update: bit-identical weights)
static inline void adamw_bias_corrections(float beta1, float beta2, int step, float* bc1, float* bc2s) {
    *bc1 = 1.0f - powf(beta1, (float)step);
    *bc2s = sqrtf(1.0f - powf(beta2, (float)step));
}

extern "C" int sc_adamw_hyper_host(float lr, float beta1, float beta2, int step, float* out3_host) {
    SC_CHECK(out3_host != nullptr && step >= 1, "sc_adamw_hyper_host: host output triple required, step >= 1");
    out3_host[0] = lr;
    adamw_bias_corrections(beta1, beta2, step, out3_host + 1, out3_host + 2);
    return 0;
}

extern "C" int sc_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                             const float* norm_clip, void* params_bf16, void* stream) {
    SC_CHECK(n > 0 && (n % 4) == 0 && step >= 1, "sc_adamw_step: n must be a positive multiple of 4, step >= 1");
    float bc1, bc2s;
    adamw_bias_corrections(beta1, beta2, step, &bc1, &bc2s);
    long long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    adamw_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                                                           weight_decay, bc1, bc2s, grad_scale, norm_clip,
                                                           (bf16*)params_bf16);
    SC_LAUNCH_CHECK();
    return 0;
}

// The same update with its step-dependent scalars read from DEVICE memory: hyper[3] = {lr, 1 - beta1^step,
// sqrt(1 - beta2^step)} (what sc_adamw_step derives from its host arguments).  A training step captured into a hipGraph
// (spatial_clip_amd/graph.py) replays identical launches; the host refreshes the three floats before each replay.
extern "C" int sc_adamw_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                                 const float* hyper, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                 const float* norm_clip, void* params_bf16, void* stream) {
    SC_CHECK(n > 0 && (n % 4) == 0 && hyper != nullptr, "sc_adamw_step_dev: n must be a positive multiple of 4, hyper required");
    long long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    adamw_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, n, 0.f, beta1, beta2, eps,
                                                           weight_decay, 1.f, 1.f, grad_scale, norm_clip, (bf16*)params_bf16,
                                                           hyper);
    SC_LAUNCH_CHECK();
    return 0;
}
