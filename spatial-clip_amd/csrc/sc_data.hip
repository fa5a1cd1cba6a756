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

// ---- What the reference's train transform does to a tile (src/open_clip/transform.py:186-204 with `use_timm: true`,
// configs/model/spatial_clip.yaml:12-17): timm's create_transform on a PIL image = RandomResizedCropAndInterpolation
// (torchvision F.resized_crop: img.crop(box).resize(size, BICUBIC)) -> RandomHorizontalFlip -> ColorJitter (PIL
// ImageEnhance.Brightness / Contrast / Color in a random order) -> ToTensor -> Normalize.  Everything up to ToTensor is
// 8-bit PIL arithmetic, restated here operation for operation so that the device result equals PIL's byte for byte:
//   * Image.resize = two separable passes (horizontal, then vertical) with an 8-bit intermediate image; per output pixel the
//     cubic-convolution filter (a = -0.5) with its support stretched by max(scale, 1) (that stretch IS the antialiasing),
//     taps clipped to the cropped image and renormalised; coefficients rounded to 22-bit fixed point, accumulator started
//     at 1 << 21, result >> 22 clipped to [0, 255] (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
//     ImagingResampleHorizontal_8bpc / Vertical_8bpc);
//   * ImageEnhance.X(img).enhance(f) = Image.blend(degenerate, img, f): out = (uint8)(d + f * (v - d)) in float32 with
//     truncation (clipped when f is outside [0, 1]; libImaging/Blend.c); degenerate = black (Brightness), the rounded mean
//     of the L image (Contrast), the L image (Color); L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16 (Convert.c).
constexpr int kResampleBits = 32 - 8 - 2;           // PRECISION_BITS of Resample.c

SC_DEVICE double pil_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// coefficients of output position `xx` for an axis of `in_size` source pixels resized to `out_size` (precompute_coeffs)
SC_DEVICE void pil_coeffs(int in_size, int out_size, int xx, int T, int* kk, int& xmin_out, int& cnt_out) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > T) xmax = T;                          // cannot happen: T is sized from the worst scale by the launcher
    double w[64];
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        w[x] = pil_bicubic((x + xmin - center + 0.5) * ss);
        ww += w[x];
    }
    for (int x = 0; x < xmax; ++x) {
        double k = ww != 0.0 ? w[x] / ww : w[x];
        kk[x] = k < 0 ? (int)(-0.5 + k * (1 << kResampleBits)) : (int)(0.5 + k * (1 << kResampleBits));
    }
    xmin_out = xmin;
    cnt_out = xmax;
}
SC_DEVICE int clip8(int ss) {
    ss >>= kResampleBits;
    return ss < 0 ? 0 : (ss > 255 ? 255 : ss);
}
SC_DEVICE int pil_luma(int r, int g, int b) { return (19595 * r + 38470 * g + 7471 * b + 0x8000) >> 16; }
// Image.blend(degenerate d, image v, alpha) on one 8-bit value (float32 arithmetic without contraction, truncation)
SC_DEVICE int pil_blend(int d, int v, float alpha) {
    const float t = __fadd_rn((float)d, __fmul_rn(alpha, (float)(v - d)));
    if (alpha >= 0.f && alpha <= 1.0f) return (int)t & 255;
    return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}
// ops 0 = brightness, 1 = contrast (degenerate = mean L of the whole image at that point), 2 = saturation
SC_DEVICE void apply_op(int op, const AugParams& P, int mean_l, int (&v)[3]) {
    if (op == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = pil_blend(0, v[c], P.b);
    } else if (op == 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = pil_blend(mean_l, v[c], P.c);
    } else {
        const int g = pil_luma(v[0], v[1], v[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = pil_blend(g, v[c], P.s);
    }
}
__constant__ int kPerm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};

// One workgroup per sample.  LDS: coefficient tables of both axes (int32 [S][T] + xmin / count per output position) and the
// 8-bit intermediate image of the horizontal pass for one channel ([crop_h][S]).  The resized 8-bit image is parked in the
// output tensor (as floats 0..255, flip applied on the way in) between the passes; the last pass overwrites it in place.
__global__ __launch_bounds__(1024) void augment_kernel(const unsigned char* __restrict__ src, int H, int W,
                                                       const float* __restrict__ params, float* __restrict__ out, int S, int T,
                                                       float m0, float m1, float m2, float s0, float s1, float s2) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ long long red[16];
    const int b = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
    AugParams P;
    {
        const float* p = params + (long long)b * 12;
        P.x0 = p[0]; P.y0 = p[1]; P.cw = p[2]; P.ch = p[3]; P.b = p[4]; P.c = p[5]; P.s = p[6]; P.order = p[7]; P.flip = p[8];
    }
    // integer crop box inside the tile (RandomResizedCrop draws integers; clamp defensively)
    int cw = min(max((int)P.cw, 1), W), ch = min(max((int)P.ch, 1), H);
    int x0 = min(max((int)P.x0, 0), W - cw), y0 = min(max((int)P.y0, 0), H - ch);
    int* kx = reinterpret_cast<int*>(lds);               // [S][T]
    int* ky = kx + S * T;                                // [S][T]
    int* xmin_x = ky + S * T;                            // [S] each
    int* cnt_x = xmin_x + S;
    int* ymin_y = cnt_x + S;
    int* cnt_y = ymin_y + S;
    unsigned char* tmp = reinterpret_cast<unsigned char*>(cnt_y + S);      // [ch][S]
    for (int i = t; i < 2 * S; i += nt) {
        if (i < S) pil_coeffs(cw, S, i, T, kx + i * T, xmin_x[i], cnt_x[i]);
        else pil_coeffs(ch, S, i - S, T, ky + (i - S) * T, ymin_y[i - S], cnt_y[i - S]);
    }
    __syncthreads();
    const unsigned char* img = src + (long long)b * H * W * 3;
    float* o = out + (long long)b * 3 * S * S;
    const bool flip = P.flip > 0.5f;
    for (int c = 0; c < 3; ++c) {
        for (int i = t; i < ch * S; i += nt) {           // horizontal pass over the rows of the crop
            const int y = i / S, ox = i - y * S;
            const unsigned char* row = img + ((long long)(y0 + y) * W + x0 + xmin_x[ox]) * 3 + c;
            const int* k = kx + ox * T;
            int ss = 1 << (kResampleBits - 1);
            for (int x = 0; x < cnt_x[ox]; ++x) ss += (int)row[x * 3] * k[x];
            tmp[i] = (unsigned char)clip8(ss);
        }
        __syncthreads();
        for (int i = t; i < S * S; i += nt) {            // vertical pass
            const int oy = i / S, ox = i - oy * S;
            const int* k = ky + oy * T;
            const unsigned char* col = tmp + ymin_y[oy] * S + ox;
            int ss = 1 << (kResampleBits - 1);
            for (int y = 0; y < cnt_y[oy]; ++y) ss += (int)col[y * S] * k[y];
            o[c * S * S + oy * S + (flip ? S - 1 - ox : ox)] = (float)clip8(ss);
        }
        __syncthreads();
    }
    const int* perm = kPerm[min(max((int)P.order, 0), 5)];
    const int contrast_pos = perm[0] == 1 ? 0 : (perm[1] == 1 ? 1 : 2);
    // mean of the L image as the contrast step meets it: int(ImageStat.Stat(L).mean[0] + 0.5)
    long long part = 0;
    for (int i = t; i < S * S; i += nt) {
        int v[3] = {(int)o[i], (int)o[S * S + i], (int)o[2 * S * S + i]};
        for (int k = 0; k < contrast_pos; ++k) apply_op(perm[k], P, 0, v);
        part += pil_luma(v[0], v[1], v[2]);
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((t & 63) == 0) red[t >> 6] = part;
    __syncthreads();
    long long tot = 0;
    for (int w = 0; w < (nt >> 6); ++w) tot += red[w];
    const int mean_l = (int)((double)tot / (double)(S * S) + 0.5);
    for (int i = t; i < S * S; i += nt) {
        int v[3] = {(int)o[i], (int)o[S * S + i], (int)o[2 * S * S + i]};
#pragma unroll
        for (int k = 0; k < 3; ++k) apply_op(perm[k], P, mean_l, v);
        // ToTensor (uint8 / 255) then Normalize ((x - mean) / std), float32 like torchvision
        o[i] = ((float)v[0] / 255.0f - m0) / s0;
        o[S * S + i] = ((float)v[1] / 255.0f - m1) / s1;
        o[2 * S * S + i] = ((float)v[2] / 255.0f - m2) / s2;
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
    // taps per output position: support 2 * max(scale, 1) either side of the centre, worst case = the whole tile
    const double fs = fmax(1.0, fmax((double)W, (double)H) / (double)S);
    const int T = (int)(4.0 * fs + 0.5) + 2;
    SC_CHECK(T <= 64, "sc_augment_tiles: downsampling %dx%d tiles to %d needs %d filter taps (> 64)", W, H, S, T);
    const size_t lds = (size_t)(2 * S * T + 4 * S) * sizeof(int) + (size_t)H * S;
    SC_CHECK(lds <= 160 * 1024 - 256, "sc_augment_tiles: %d x %d tiles at output size %d need %zu bytes of LDS", H, W, S, lds);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&augment_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024 - 256);
        attr_done = true;
    }
    augment_kernel<<<B, 1024, lds, (hipStream_t)stream>>>((const unsigned char*)src_u8_hwc, H, W, params12, out_nchw, S, T,
                                                          mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0],
                                                          std3_host[1], std3_host[2]);
    SC_LAUNCH_CHECK();
    return 0;
}
