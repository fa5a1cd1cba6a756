// Contrastive head: global-batch ClipLoss and the multi-positive spatial-neighbour SpatialLoss, forward and
// backward, entirely on the device (the reference builds the soft labels with O(G + B*K) host syncs).
//   reference: src/open_clip/loss.py:91-155 (ClipLoss, local_loss layout)
//              src/models/components/losses.py:44-124 (SpatialLoss)        closed forms: SURVEY.md Appendix A
// Pipeline (all fp32, tiny next to the towers):
//   sc_sgemm_f32_grouped z = f . all_f^T  (cosine similarities, both directions; exact-fp32 MFMA, sc_head_gemm.hip)
//   sc_neighbor_join     tile-id join -> sparse soft labels (<= K+1 (col, weight) pairs per row)
//   sc_loss_rows_fwd     per row: logsumexp, E_p[z], sum q*logit, sum q*z
//   sc_loss_finalize     loss = 0.5*(CE_i + CE_t) + w*gap^2 ; gap
//   sc_loss_rows_bwd     dz (in place over z), d logit_scale, d logit_bias
//   sc_sgemm_f32_grouped d f_local = dz . all_f ,  d all_f = dz^T . f_local  (the latter is reduce-scattered)
#include "sc_common.h"
#include "sc_kernels.h"

namespace {

// ---------------------------------------------------------------- tile-id join -> sparse labels
// One block per local row i.  Entry 0 = the anchor column (rank*B + i, weight 1), entries 1..K = neighbours.
// Dict semantics of the reference (losses.py:92-93): for duplicate ids the LAST column wins; entries with
// alpha*scale <= 0 are skipped before lookup; labels are L1-normalised (all weights >= 0, eps 1e-12).
__global__ __launch_bounds__(256) void neighbor_join_kernel(const long long* __restrict__ all_img_ids,
                                                            const long long* __restrict__ all_txt_ids,
                                                            const long long* __restrict__ nbr_ids,
                                                            const float* __restrict__ nbr_alpha, int B, int G, int K,
                                                            int rank, float alpha_scale, int* __restrict__ lab_col,
                                                            float* __restrict__ lab_w) {
    __shared__ int s_col[2][64];
    __shared__ float s_w[64];
    const int i = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < K; k += 4) {
        const float a = fmaxf(nbr_alpha[(long long)i * K + k] * alpha_scale, 0.f);
        int ct = -1, ci = -1;
        if (a > 0.f) {
            const long long id = nbr_ids[(long long)i * K + k];
            for (int j = lane; j < G; j += 64) {
                if (all_txt_ids[j] == id) ct = j;   // ascending j -> keeps the last match per lane
                if (all_img_ids[j] == id) ci = j;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                ct = max(ct, __shfl_xor(ct, o, 64));
                ci = max(ci, __shfl_xor(ci, o, 64));
            }
        }
        if (lane == 0) { s_col[0][k] = ct; s_col[1][k] = ci; s_w[k] = a; }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int dir = threadIdx.x;   // 0: image->text labels (text ids), 1: text->image labels (image ids)
        float norm = 1.0f;
        for (int k = 0; k < K; ++k)
            if (s_col[dir][k] >= 0) norm += s_w[k];
        norm = fmaxf(norm, 1e-12f);
        int* lc = lab_col + ((long long)dir * B + i) * (K + 1);
        float* lw = lab_w + ((long long)dir * B + i) * (K + 1);
        lc[0] = rank * B + i;
        lw[0] = 1.0f / norm;
        for (int k = 0; k < K; ++k) {
            const int c = s_col[dir][k];
            lc[1 + k] = c;
            lw[1 + k] = c >= 0 ? s_w[k] / norm : 0.f;
        }
    }
}

__global__ void onehot_labels_kernel(int B, int rank, int* __restrict__ lab_col, float* __restrict__ lab_w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * B) return;
    lab_col[i] = rank * B + (i % B);
    lab_w[i] = 1.0f;
}

SC_DEVICE float block_sum(float v, float* sm) {
    v = sc_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}
SC_DEVICE float block_max(float v, float* sm) {
    v = sc_wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

SC_DEVICE float eff_scale(const float* scale, float cap) {
    const float s = *scale;
    return cap > 0.f ? fminf(s, cap) : s;
}

// rowstats[dir][i] = {lse, E_p[z], sum q*logit, sum q*z}
__global__ __launch_bounds__(256) void loss_rows_fwd_kernel(const float* __restrict__ z, int B, int G,
                                                            const float* __restrict__ scale, float cap,
                                                            const float* __restrict__ bias,
                                                            const int* __restrict__ lab_col,
                                                            const float* __restrict__ lab_w, int nlab,
                                                            float* __restrict__ rowstats) {
    __shared__ float sm[4];
    const int r = blockIdx.x;  // dir*B + i
    const float* zr = z + (long long)r * G;
    const float s = eff_scale(scale, cap);
    const float bz = bias ? *bias : 0.f;
    float mx = -3.0e38f;
    for (int j = threadIdx.x; j < G; j += 256) mx = fmaxf(mx, s * zr[j] + bz);
    mx = block_max(mx, sm);
    float se = 0.f, sz = 0.f;
    for (int j = threadIdx.x; j < G; j += 256) {
        const float zz = zr[j];
        const float e = __expf(s * zz + bz - mx);
        se += e;
        sz += e * zz;
    }
    se = block_sum(se, sm);
    sz = block_sum(sz, sm);
    if (threadIdx.x == 0) {
        float ql = 0.f, qz = 0.f;
        for (int k = 0; k < nlab; ++k) {
            const int c = lab_col[(long long)r * nlab + k];
            const float w = lab_w[(long long)r * nlab + k];
            if (c >= 0 && w > 0.f) {
                const float zz = zr[c];
                ql += w * (s * zz + bz);
                qz += w * zz;
            }
        }
        float* o = rowstats + (long long)r * 4;
        o[0] = mx + __logf(se);
        o[1] = sz / se;
        o[2] = ql;
        o[3] = qz;
    }
}

// out = {loss, gap, ce_image, ce_text}
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ rowstats, int B, float w,
                                                            float* __restrict__ out) {
    __shared__ float sm[4];
    float ce0 = 0.f, ce1 = 0.f, e0 = 0.f, e1 = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) {
        const float* a = rowstats + (long long)i * 4;
        const float* b = rowstats + (long long)(B + i) * 4;
        ce0 += a[0] - a[2];
        ce1 += b[0] - b[2];
        e0 += a[1] - a[3];
        e1 += b[1] - b[3];
    }
    ce0 = block_sum(ce0, sm) / B;
    ce1 = block_sum(ce1, sm) / B;
    e0 = block_sum(e0, sm) / B;
    e1 = block_sum(e1, sm) / B;
    if (threadIdx.x == 0) {
        const float gap = 0.5f * (e0 + e1);
        out[0] = 0.5f * (ce0 + ce1) + (w > 0.f ? w * gap * gap : 0.f);
        out[1] = gap;
        out[2] = ce0;
        out[3] = ce1;
    }
}

// dz in place; rowgrad[r] = {sum_j dl_ij z_ij, sum_j dl_ij}
__global__ __launch_bounds__(256) void loss_rows_bwd_kernel(float* __restrict__ z, int B, int G,
                                                            const float* __restrict__ scale, float cap,
                                                            const float* __restrict__ bias,
                                                            const int* __restrict__ lab_col,
                                                            const float* __restrict__ lab_w, int nlab,
                                                            const float* __restrict__ rowstats,
                                                            const float* __restrict__ lossout, float w,
                                                            const float* __restrict__ gout, float* __restrict__ rowgrad) {
    __shared__ float sm[4];
    const int r = blockIdx.x;
    float* zr = z + (long long)r * G;
    const float s = eff_scale(scale, cap);
    const float bz = bias ? *bias : 0.f;
    const float go = gout ? *gout : 1.0f;
    const float lse = rowstats[(long long)r * 4 + 0], mi = rowstats[(long long)r * 4 + 1];
    const float c1 = go * 0.5f / B;                               // CE weight
    const float c2 = w > 0.f ? go * w * lossout[1] / B : 0.f;     // w*gap/B
    float sdz = 0.f, sdl = 0.f;
    // dense part: dl = c1*p + c2*p*(z - m) ; dz = s*dl + c2*p
    for (int j = threadIdx.x; j < G; j += 256) {
        const float zz = zr[j];
        const float p = __expf(s * zz + bz - lse);
        const float dl = c1 * p + c2 * p * (zz - mi);
        sdz += dl * zz;
        sdl += dl;
        zr[j] = s * dl + c2 * p;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // sparse label part: dl -= c1*q  =>  dz -= (s*c1 + c2)*q ; repeated columns accumulate
        for (int k = 0; k < nlab; ++k) {
            const int c = lab_col[(long long)r * nlab + k];
            const float q = lab_w[(long long)r * nlab + k];
            if (c >= 0 && q > 0.f) {
                sdl -= c1 * q;
                zr[c] -= (s * c1 + c2) * q;
            }
        }
    }
    sdz = block_sum(sdz, sm);
    sdl = block_sum(sdl, sm);
    if (threadIdx.x == 0) {
        // sum_j dl_ij z_ij  needs  - c1 * sum_j q_ij z_ij  = - c1 * rowstats[3]
        rowgrad[(long long)r * 2 + 0] = sdz - c1 * rowstats[(long long)r * 4 + 3];
        rowgrad[(long long)r * 2 + 1] = sdl;
    }
}

__global__ __launch_bounds__(256) void loss_scalar_grads_kernel(const float* __restrict__ rowgrad, int nrows,
                                                                float* __restrict__ dscale, float* __restrict__ dbias) {
    __shared__ float sm[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nrows; i += 256) {
        a += rowgrad[(long long)i * 2];
        b += rowgrad[(long long)i * 2 + 1];
    }
    a = block_sum(a, sm);
    b = block_sum(b, sm);
    if (threadIdx.x == 0) {
        if (dscale) *dscale = a;
        if (dbias) *dbias = b;
    }
}

// recall@{1,5,10} hit counts of the local [B,B] block of z (columns col0..col0+B): rank of the diagonal
__global__ __launch_bounds__(256) void recall_kernel(const float* __restrict__ z, int G, int B, int col0,
                                                     int* __restrict__ hits) {
    __shared__ float sm[4];
    const int i = blockIdx.x;
    const float* zr = z + (long long)i * G + col0;
    const float dgl = zr[i];
    float cnt = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) cnt += (zr[j] > dgl || (zr[j] == dgl && j < i)) ? 1.f : 0.f;
    cnt = block_sum(cnt, sm);
    if (threadIdx.x == 0) {
        const int c = (int)cnt;
        if (c < 1) atomicAdd(&hits[0], 1);
        if (c < 5) atomicAdd(&hits[1], 1);
        if (c < 10) atomicAdd(&hits[2], 1);
    }
}

__global__ void exp_scalar_kernel(const float* __restrict__ x, float* __restrict__ y) { *y = __expf(*x); }
__global__ void exp_scalar_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                      float mult) {
    *dx = (*dy) * (*y) * mult;
}

// ---------------------------------------------------------------------------------------------- zero-shot PCC rows
// Sample-wise Pearson correlation between predicted gene logits and rank-weighted targets
// (ref:src/metrics/zero_shot.py:72-88).  One workgroup per row, two passes over the row like the reference (means
// first, then centred sums; the row comes from L2 the second time); rows with a denominator <= 1e-6 score 0.
// sum_count[0] += sum of the rows' pcc, sum_count[1] += number of rows (float atomics: 2 per row).
__global__ __launch_bounds__(256) void pcc_rows_kernel(const float* __restrict__ pred, long long ldp,
                                                       const float* __restrict__ tgt, long long ldt, int cols,
                                                       float* __restrict__ pcc, float* __restrict__ sum_count) {
    __shared__ float red[3][4];
    const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float* p = pred + (long long)row * ldp;
    const float* q = tgt + (long long)row * ldt;
    float sp = 0.f, sq = 0.f;
    for (int c = t; c < cols; c += 256) { sp += p[c]; sq += q[c]; }
    sp = sc_wave_sum(sp); sq = sc_wave_sum(sq);
    if (lane == 0) { red[0][w] = sp; red[1][w] = sq; }
    __syncthreads();
    const float mp = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)cols;
    const float mq = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)cols;
    __syncthreads();
    float num = 0.f, pp = 0.f, qq = 0.f;
    for (int c = t; c < cols; c += 256) {
        const float a = p[c] - mp, b = q[c] - mq;
        num += a * b; pp += a * a; qq += b * b;
    }
    num = sc_wave_sum(num); pp = sc_wave_sum(pp); qq = sc_wave_sum(qq);
    if (lane == 0) { red[0][w] = num; red[1][w] = pp; red[2][w] = qq; }
    __syncthreads();
    if (t == 0) {
        const float n = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const float a = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const float b = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        const float den = sqrtf(a) * sqrtf(b);
        const float r = den > 1e-6f ? n / den : 0.f;
        if (pcc) pcc[row] = r;
        if (sum_count) { atomicAdd(sum_count, r); atomicAdd(sum_count + 1, 1.0f); }
    }
}

}  // namespace

extern "C" int sc_neighbor_join(const long long* all_image_tile_ids, const long long* all_text_tile_ids,
                                const long long* neighbor_tile_ids, const float* neighbor_alphas, int B, int G, int K,
                                int rank, float neighbor_alpha_scale, int* lab_col, float* lab_w, void* stream) {
    SC_CHECK(B > 0 && G >= B && K >= 0 && K <= 63, "sc_neighbor_join: bad shape B=%d G=%d K=%d", B, G, K);
    SC_CHECK(rank >= 0 && (long long)(rank + 1) * B <= G, "sc_neighbor_join: rank %d out of range", rank);
    neighbor_join_kernel<<<B, 256, 0, (hipStream_t)stream>>>(all_image_tile_ids, all_text_tile_ids, neighbor_tile_ids,
                                                            neighbor_alphas, B, G, K, rank, neighbor_alpha_scale,
                                                            lab_col, lab_w);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_onehot_labels(int B, int rank, int* lab_col, float* lab_w, void* stream) {
    SC_CHECK(B > 0 && rank >= 0, "sc_onehot_labels: bad args");
    onehot_labels_kernel<<<(2 * B + 255) / 256, 256, 0, (hipStream_t)stream>>>(B, rank, lab_col, lab_w);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_contrastive_loss_fwd(const float* z, int B, int G, const float* logit_scale, float cap_logit_scale,
                                       const float* logit_bias, const int* lab_col, const float* lab_w, int nlab,
                                       float temp_reg_weight, float* rowstats, float* loss_out, void* stream) {
    SC_CHECK(B > 0 && G >= B && nlab >= 1, "sc_contrastive_loss_fwd: bad shape B=%d G=%d nlab=%d", B, G, nlab);
    hipStream_t st = (hipStream_t)stream;
    loss_rows_fwd_kernel<<<2 * B, 256, 0, st>>>(z, B, G, logit_scale, cap_logit_scale, logit_bias, lab_col, lab_w, nlab,
                                                rowstats);
    SC_LAUNCH_CHECK();
    loss_finalize_kernel<<<1, 256, 0, st>>>(rowstats, B, temp_reg_weight, loss_out);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_contrastive_loss_bwd(float* z_inout, int B, int G, const float* logit_scale, float cap_logit_scale,
                                       const float* logit_bias, const int* lab_col, const float* lab_w, int nlab,
                                       float temp_reg_weight, const float* rowstats, const float* loss_out,
                                       const float* grad_out, float* rowgrad, float* dscale, float* dbias,
                                       void* stream) {
    SC_CHECK(B > 0 && G >= B && nlab >= 1, "sc_contrastive_loss_bwd: bad shape B=%d G=%d nlab=%d", B, G, nlab);
    hipStream_t st = (hipStream_t)stream;
    loss_rows_bwd_kernel<<<2 * B, 256, 0, st>>>(z_inout, B, G, logit_scale, cap_logit_scale, logit_bias, lab_col, lab_w,
                                                nlab, rowstats, loss_out, temp_reg_weight, grad_out, rowgrad);
    SC_LAUNCH_CHECK();
    loss_scalar_grads_kernel<<<1, 256, 0, st>>>(rowgrad, 2 * B, dscale, dbias);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_pcc_rows(const float* pred, long long ldp, const float* target, long long ldt, int rows, int cols,
                           float* pcc, float* sum_count, void* stream) {
    SC_CHECK(rows > 0 && cols > 0 && ldp >= cols && ldt >= cols, "sc_pcc_rows: bad shape rows=%d cols=%d", rows, cols);
    pcc_rows_kernel<<<rows, 256, 0, (hipStream_t)stream>>>(pred, ldp, target, ldt, cols, pcc, sum_count);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_recall_hits(const float* z_image_rows, int G, int B, int col0, int* hits3, void* stream) {
    SC_CHECK(B > 0 && G >= B && col0 >= 0 && col0 + B <= G, "sc_recall_hits: bad shape");
    recall_kernel<<<B, 256, 0, (hipStream_t)stream>>>(z_image_rows, G, B, col0, hits3);
    SC_LAUNCH_CHECK();
    return 0;
}

// out[i] = x[i] * *s : the upstream gradient of the loss (a device scalar) applied to the feature / scale gradients the
// fused head produced for an upstream gradient of 1 (autograd's loss.backward() passes 1.0, but any scalar is honoured)
__global__ void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ out,
                                       long long n) {
    const float f = *s;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = x[i] * f;
}

extern "C" int sc_scale_by_scalar(const float* x, const float* s, float* out, long long n, void* stream) {
    SC_CHECK(n > 0 && x != nullptr && s != nullptr && out != nullptr, "sc_scale_by_scalar: bad arguments (n=%lld)", n);
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    scale_by_scalar_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(x, s, out, n);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_exp_scalar(const float* x, float* y, void* stream) {
    exp_scalar_kernel<<<1, 1, 0, (hipStream_t)stream>>>(x, y);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_exp_scalar_bwd(const float* y, const float* dy, float* dx, float mult, void* stream) {
    exp_scalar_bwd_kernel<<<1, 1, 0, (hipStream_t)stream>>>(y, dy, dx, mult);
    SC_LAUNCH_CHECK();
    return 0;
}
