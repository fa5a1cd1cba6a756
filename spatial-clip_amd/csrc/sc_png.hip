// PNG tiles decoded on the device (SURVEY.md 8f rank 3 "decode/augment"): the shards_v1 backend stores every tile as a
// PNG member of a tar (the reference's tests/test_spatial_datasets.py:57-75 writes them; its dataset classes open them with
// PIL on CPU dataloader workers).  At ~7.5 k pairs/s per GPU the host cannot inflate ~1.1 GB/s of zlib streams, so the
// compressed files are copied to HBM as they are and decoded next to their consumer (sc_augment_tiles).
//
// One wave per tile.  A DEFLATE stream is sequential, so the parallelism is ACROSS tiles (a batch is 256 of them); within a
// tile the 64 lanes run the decoder's control flow with identical values (sc_png_core.h, the same source the CPU test
// harness compiles) and help where the format allows it:
//   * input   the IDAT payloads are pulled through a 1-KiB LDS window, refilled by all lanes (chunk boundaries handled there);
//   * output  literals collect in a 256-byte LDS stage and leave as one coalesced store burst; an LZ77 match is ONE
//             wave-wide gather / scatter whatever its length (an overlapping match repeats with period `dist`, so lane i reads
//             byte i mod dist of the source run) -- stores and the later loads of other lanes are ordered by workgroup-scope
//             release / acquire fences (one wave = one workgroup: a wait, no cache maintenance);
//   * filters PNG's Sub / Up / Average / Paeth predictors depend on the left, upper and upper-left neighbours: 64 rows are
//             reconstructed together, row r one pixel behind row r - 1, and the neighbours of the row above arrive by
//             lane shifts (no memory round trip inside a band).
// Footprint: 5 KiB of LDS, 138 VGPRs, one wave per workgroup: it runs wherever a SIMD has room (beside LayerNorm / loss /
// optimiser kernels; the 256x256 GEMM workgroups fill the register file on their own).
// Limits (anything else sets status != 0 and the host decodes that tile with PIL): 8-bit RGB / RGBA, no interlace, tile
// size = the requested H x W, at most 32 IDAT chunks.
#include "sc_common.h"
#include "sc_kernels.h"
#include "sc_png_core.h"

namespace {

struct Lds {
    sc_png::Tables T;
    sc_png::Header h;
    unsigned in[256];                  // 1-KiB input window, read as dwords
    unsigned char stage[256];
};

struct DevIO {
    const unsigned char* file;
    Lds* L;
    unsigned char* out;
    long long cap, n;        // n: bytes committed to `out` (the stage holds `pending` more)
    int lane;
    int seg;
    unsigned pos;            // cursor inside IDAT segment `seg`
    int in_pos, in_fill;
    int pending;

    __device__ void refill() {
        __syncthreads();                                   // everybody is done with the old window
        int fill = 0;
        while (fill < 1024 && seg < L->h.nseg) {
            const unsigned left = L->h.seg_len[seg] - pos;
            if (left == 0) { ++seg; pos = 0; continue; }
            const int take = (int)min((unsigned)(1024 - fill), left);
            const unsigned char* src = file + L->h.seg_off[seg] + pos;
            for (int i = lane; i < take; i += 64) reinterpret_cast<unsigned char*>(L->in)[fill + i] = src[i];
            fill += take;
            pos += take;
        }
        in_fill = fill;
        in_pos = 0;
        __syncthreads();
    }
    __device__ int get_byte() {
        if (in_pos >= in_fill) {
            refill();
            if (in_fill == 0) return -1;
        }
        return reinterpret_cast<unsigned char*>(L->in)[in_pos++];
    }
    __device__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
    // the window is refilled in whole KiB (except the stream's tail), so a dword read is aligned whenever in_pos is
    __device__ int get_word(unsigned& w) {
        if (in_pos >= in_fill) {
            refill();
            if (in_fill == 0) { w = 0; return 0; }
        }
        if ((in_pos & 3) == 0 && in_pos + 4 <= in_fill) {
            w = (unsigned)__builtin_amdgcn_readfirstlane((int)L->in[in_pos >> 2]);
            in_pos += 4;
            return 4;
        }
        w = 0;
        int nb = 0;
        for (; nb < 4 && in_pos < in_fill; ++nb) w |= (unsigned)reinterpret_cast<unsigned char*>(L->in)[in_pos++] << (8 * nb);
        return nb;
    }
    __device__ void flush() {
        if (pending == 0) return;
        __syncthreads();                                   // the stage is complete
        for (int i = lane; i < pending; i += 64) out[n + i] = L->stage[i];
        __syncthreads();                                   // ... and may be overwritten
        n += pending;
        pending = 0;
    }
    __device__ bool put_literal(int b) {
        if (n + pending >= cap) return false;
        if (lane == 0) L->stage[pending] = (unsigned char)b;
        if (++pending == 256) flush();
        return true;
    }
    __device__ bool copy_match(int dist, int len) {
        flush();
        if (dist > n || n + len > cap) return false;
        // the source run was written by other lanes (and possibly a moment ago): make those stores visible to this wave's loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const unsigned char* src = out + n - dist;
        for (int i = lane; i < len; i += 64) out[n + i] = src[i % dist];
        n += len;
        return true;
    }
    __device__ bool copy_stored(int k) {
        for (int i = 0; i < k; ++i) {
            const int b = get_byte();
            if (b < 0 || !put_literal(b)) return false;
        }
        return true;
    }
};

__global__ __launch_bounds__(64) void png_decode_kernel(const unsigned char* __restrict__ files, const long long* __restrict__ offsets,
                                                        unsigned char* __restrict__ out_rgb, int H, int W,
                                                        unsigned char* __restrict__ scratch, long long scratch_per_image,
                                                        int* __restrict__ status) {
    __shared__ Lds L;
    const int b = blockIdx.x, lane = threadIdx.x;
    const unsigned char* file = files + offsets[b];
    const long long nbytes = offsets[b + 1] - offsets[b];
    int rc = sc_png::OK;
    if (lane == 0) {
        rc = nbytes > 0 ? sc_png::parse(file, nbytes, L.h) : sc_png::ERR_TRUNCATED;
        if (rc == sc_png::OK && (L.h.width != W || L.h.height != H)) rc = sc_png::ERR_SIZE;
    }
    rc = __shfl(rc, 0, 64);
    __syncthreads();
    if (rc != sc_png::OK) {
        if (lane == 0) status[b] = rc;
        return;
    }
    const int bpp = L.h.channels, rowb = W * bpp, stride = rowb + 1;
    const long long raw = (long long)H * stride;
    unsigned char* buf = scratch + (long long)b * scratch_per_image;
    DevIO io{file, &L, buf, raw, 0, lane, 0, 0u, 0, 0, 0};
    rc = sc_png::inflate(io, L.T);
    io.flush();
    if (rc == sc_png::OK && io.n != raw) rc = sc_png::ERR_TRUNCATED;
    if (rc != sc_png::OK) {
        if (lane == 0) status[b] = rc;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ---- reverse the scanline filters, 64 rows at a time, row r one pixel behind row r - 1; rows are rebuilt IN PLACE
    // (the alpha channel of an RGBA file is needed by its neighbours although it is not part of the output)
    unsigned char* o = out_rgb + (long long)b * H * W * 3;
    bool bad_filter = false;
    for (int y0 = 0; y0 < H; y0 += 64) {
        const int y = y0 + lane;
        const bool active = y < H;
        unsigned char* row = buf + (long long)(active ? y : 0) * stride;
        const unsigned char* above = buf + (long long)(y > 0 ? y - 1 : 0) * stride;      // used by lane 0 only
        const int ft = active ? row[0] : 0;
        if (ft > 4) bad_filter = true;
        int left[4] = {0, 0, 0, 0}, last1[4] = {0, 0, 0, 0}, last2[4] = {0, 0, 0, 0};
        for (int t = 0; t < W + 63; ++t) {
            const int x = t - lane;
            const bool on = active && x >= 0 && x < W;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c >= bpp) break;
                int up = __shfl_up(last1[c], 1, 64), ul = __shfl_up(last2[c], 1, 64);
                if (lane == 0) {
                    up = (on && y > 0) ? above[1 + x * bpp + c] : 0;
                    ul = (on && y > 0 && x > 0) ? above[1 + (x - 1) * bpp + c] : 0;
                }
                if (on) {
                    const int a = x > 0 ? left[c] : 0;
                    if (x == 0) ul = 0;
                    int v = row[1 + x * bpp + c];
                    if (ft == 1) v += a;
                    else if (ft == 2) v += up;
                    else if (ft == 3) v += (a + up) >> 1;
                    else if (ft == 4) v += sc_png::paeth(a, up, ul);
                    v &= 255;
                    row[1 + x * bpp + c] = (unsigned char)v;
                    if (c < 3) o[((long long)y * W + x) * 3 + c] = (unsigned char)v;
                    left[c] = v;
                    last2[c] = last1[c];
                    last1[c] = v;
                }
            }
        }
        // the next band's first row reads this band's last row from memory
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    bad_filter = __any(bad_filter);
    if (lane == 0) status[b] = bad_filter ? sc_png::ERR_FORMAT : sc_png::OK;
}

}  // namespace

extern "C" long long sc_png_decode_scratch_bytes(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return 0;
    const long long per = (((long long)H * (4LL * W + 1)) + 255) / 256 * 256;
    return per * B;
}

extern "C" int sc_png_decode(const void* files, const long long* offsets, int B, void* out_rgb, int H, int W, void* scratch,
                             int* status, void* stream) {
    SC_CHECK(B >= 1 && H >= 1 && W >= 1 && files && offsets && out_rgb && scratch && status,
             "sc_png_decode: bad arguments B=%d H=%d W=%d", B, H, W);
    const long long per = sc_png_decode_scratch_bytes(1, H, W);
    png_decode_kernel<<<B, 64, 0, (hipStream_t)stream>>>((const unsigned char*)files, offsets, (unsigned char*)out_rgb, H, W,
                                                         (unsigned char*)scratch, per, status);
    SC_LAUNCH_CHECK();
    return 0;
}
