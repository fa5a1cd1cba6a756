// Device side of the input pipeline: the two data-dependent steps the reference runs on CPU
// dataloader workers, moved next to the consumer so that a ~10 k pairs/s/GPU trainer is not fed at PIL speed.
//   sc_knn_alpha      spatial neighbours of every tile of one slide from its (x, y) centroid and their loss weights
//                     (docs/spatial_clip_data_pipeline.html "Step 1": KNN inside the same tissue sample,
//                     weight = 1 / (distance + 1e-6), alpha = weight / sum(weights); the Gaussian variant of
//                     notebooks/d1_dataset_construct_cw.ipynb is selectable)
//   sc_augment_tiles  RandomResizedCrop(scale, ratio) -> bilinear resize -> ColorJitter(brightness, contrast,
//                     saturation in a per-sample order) -> Normalize(mean, std) on decoded uint8 tiles
//                     (configs/model/spatial_clip.yaml:12-17 aug_cfg; src/open_clip/constants.py:1-2 mean / std)
// Random draws stay on the host (one small parameter row per sample), so a run is reproducible from its seed and the
// kernels are pure functions of their inputs.  Both are HBM-bound byte movers: coalesced reads of the source rows,
// one pass for the statistics the contrast step needs, one pass that writes the normalised fp32 NCHW tile.
#include "sc_common.h"
#include "sc_kernels.h"

namespace {

// One wave per query tile.  K rounds; round r finds the candidate with the smallest (distance^2, index) key that is
// larger than the key selected in round r-1 -- no per-lane candidate lists, ties broken by index (deterministic).
__global__ __launch_bounds__(256) void knn_alpha_kernel(const float* __restrict__ xy, int N, int K, int mode, float sigma,
                                                        int* __restrict__ nbr, float* __restrict__ alpha) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= N) return;
    const float qx = xy[2 * wave], qy = xy[2 * wave + 1];
    float last_d = -1.f;
    int last_i = -1;
    float wsum = 0.f;
    for (int r = 0; r < K; ++r) {
        float best_d = 3.0e38f;
        int best_i = 0x7fffffff;
        for (int j = lane; j < N; j += 64) {
            if (j == wave) continue;
            const float dx = xy[2 * j] - qx, dy = xy[2 * j + 1] - qy;
            const float d = dx * dx + dy * dy;
            const bool after = d > last_d || (d == last_d && j > last_i);
            if (after && (d < best_d || (d == best_d && j < best_i))) { best_d = d; best_i = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(best_d, o, 64);
            const int oi = __shfl_xor(best_i, o, 64);
            if (od < best_d || (od == best_d && oi < best_i)) { best_d = od; best_i = oi; }
        }
        const bool found = best_i != 0x7fffffff;
        float w = 0.f;
        if (found) {
            const float dist = sqrtf(best_d);
            w = mode == 0 ? 1.0f / (dist + 1e-6f) : __expf(-best_d / (2.f * sigma * sigma));
            last_d = best_d;
            last_i = best_i;
        }
        wsum += w;
        if (lane == 0) {
            nbr[(long long)wave * K + r] = found ? best_i : -1;
            alpha[(long long)wave * K + r] = w;
        }
        if (!found) {                       // fewer than K other tiles: pad the rest
            for (int rr = r + 1; rr < K && lane == 0; ++rr) { nbr[(long long)wave * K + rr] = -1; alpha[(long long)wave * K + rr] = 0.f; }
            break;
        }
    }
    if (lane == 0 && wsum > 0.f) {
        const float inv = 1.0f / wsum;
        for (int r = 0; r < K; ++r) alpha[(long long)wave * K + r] *= inv;
    }
}

struct AugParams {          // one row of 12 floats per sample
    float x0, y0, cw, ch;   // crop box in source pixels
    float b, c, s;          // brightness / contrast / saturation factors (1 = identity)
    float order;            // permutation code 0..5 of (brightness, contrast, saturation)
    float flip, pad0, pad1, pad2;
};

SC_DEVICE void fetch_rgb(const unsigned char* __restrict__ img, int H, int W, float sx, float sy, float (&rgb)[3]) {
    // bilinear sample at source coordinate (sx, sy) in pixel-centre convention (align_corners = False), edge clamped
    sx = fminf(fmaxf(sx, 0.f), (float)(W - 1));
    sy = fminf(fmaxf(sy, 0.f), (float)(H - 1));
    const int x0 = (int)sx, y0 = (int)sy;
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float fx = sx - x0, fy = sy - y0;
    const unsigned char* p00 = img + ((long long)y0 * W + x0) * 3;
    const unsigned char* p01 = img + ((long long)y0 * W + x1) * 3;
    const unsigned char* p10 = img + ((long long)y1 * W + x0) * 3;
    const unsigned char* p11 = img + ((long long)y1 * W + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = p00[c] + fx * ((float)p01[c] - p00[c]);
        const float bot = p10[c] + fx * ((float)p11[c] - p10[c]);
        rgb[c] = (top + fy * (bot - top)) * (1.0f / 255.0f);
    }
}
SC_DEVICE float gray(const float (&v)[3]) { return 0.299f * v[0] + 0.587f * v[1] + 0.114f * v[2]; }
SC_DEVICE float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// ops 0 = brightness, 1 = contrast (needs the image-wide mean of the grayscale at that point), 2 = saturation
SC_DEVICE void apply_op(int op, const AugParams& P, float mean_gray, float (&v)[3]) {
    if (op == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = clamp01(v[c] * P.b);
    } else if (op == 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = clamp01(P.c * v[c] + (1.f - P.c) * mean_gray);
    } else {
        const float g = gray(v);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = clamp01(P.s * v[c] + (1.f - P.s) * g);
    }
}
__constant__ int kPerm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};

// one workgroup per sample: pass 1 = mean grayscale of the image as it is when the contrast step meets it,
// pass 2 = everything, normalised, written as fp32 [3][S][S]
__global__ __launch_bounds__(1024) void augment_kernel(const unsigned char* __restrict__ src, int H, int W,
                                                       const float* __restrict__ params, float* __restrict__ out, int S,
                                                       float m0, float m1, float m2, float is0, float is1, float is2) {
    __shared__ float red[16];
    const int b = blockIdx.x, t = threadIdx.x;
    AugParams P;
    {
        const float* p = params + (long long)b * 12;
        P.x0 = p[0]; P.y0 = p[1]; P.cw = p[2]; P.ch = p[3]; P.b = p[4]; P.c = p[5]; P.s = p[6]; P.order = p[7]; P.flip = p[8];
    }
    const unsigned char* img = src + (long long)b * H * W * 3;
    const int* perm = kPerm[min(max((int)P.order, 0), 5)];
    const int contrast_pos = perm[0] == 1 ? 0 : (perm[1] == 1 ? 1 : 2);
    const float scx = P.cw / S, scy = P.ch / S;
    float part = 0.f;
    for (int i = t; i < S * S; i += blockDim.x) {
        const int oy = i / S, ox0 = i % S;
        const int ox = P.flip > 0.5f ? S - 1 - ox0 : ox0;
        float v[3];
        fetch_rgb(img, H, W, P.x0 + (ox + 0.5f) * scx - 0.5f, P.y0 + (oy + 0.5f) * scy - 0.5f, v);
        for (int k = 0; k < contrast_pos; ++k) apply_op(perm[k], P, 0.f, v);
        part += gray(v);
    }
    part = sc_wave_sum(part);
    if ((t & 63) == 0) red[t >> 6] = part;
    __syncthreads();
    float mean_gray = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) mean_gray += red[w];
    mean_gray /= (float)(S * S);
    float* o = out + (long long)b * 3 * S * S;
    for (int i = t; i < S * S; i += blockDim.x) {
        const int oy = i / S, ox0 = i % S;
        const int ox = P.flip > 0.5f ? S - 1 - ox0 : ox0;
        float v[3];
        fetch_rgb(img, H, W, P.x0 + (ox + 0.5f) * scx - 0.5f, P.y0 + (oy + 0.5f) * scy - 0.5f, v);
#pragma unroll
        for (int k = 0; k < 3; ++k) apply_op(perm[k], P, mean_gray, v);
        o[i] = (v[0] - m0) * is0;
        o[S * S + i] = (v[1] - m1) * is1;
        o[2 * S * S + i] = (v[2] - m2) * is2;
    }
}

}  // namespace

extern "C" int sc_knn_alpha(const float* xy, int N, int K, int mode, float sigma, int* nbr_index, float* alpha,
                            void* stream) {
    SC_CHECK(N >= 1 && K >= 1 && K <= 64, "sc_knn_alpha: bad shape N=%d K=%d", N, K);
    SC_CHECK(mode == 0 || (mode == 1 && sigma > 0.f), "sc_knn_alpha: mode 0 (inverse distance) or 1 (gaussian, sigma > 0)");
    const int blocks = (N + 3) / 4;          // 4 waves (= 4 query tiles) per workgroup
    knn_alpha_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(xy, N, K, mode, sigma, nbr_index, alpha);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_augment_tiles(const void* src_u8_hwc, int B, int H, int W, const float* params12, float* out_nchw,
                                int S, const float* mean3_host, const float* std3_host, void* stream) {
    SC_CHECK(B >= 1 && H >= 1 && W >= 1 && S >= 1, "sc_augment_tiles: bad shape B=%d H=%d W=%d S=%d", B, H, W, S);
    SC_CHECK(mean3_host && std3_host && std3_host[0] > 0 && std3_host[1] > 0 && std3_host[2] > 0,
             "sc_augment_tiles: mean / std (host pointers to 3 floats) required");
    augment_kernel<<<B, 1024, 0, (hipStream_t)stream>>>((const unsigned char*)src_u8_hwc, H, W, params12, out_nchw, S,
                                                        mean3_host[0], mean3_host[1], mean3_host[2], 1.f / std3_host[0],
                                                        1.f / std3_host[1], 1.f / std3_host[2]);
    SC_LAUNCH_CHECK();
    return 0;
}
