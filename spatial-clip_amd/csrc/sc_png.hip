// PNG tiles decoded on the device (SURVEY.md 8f rank 3 "decode/augment"): the shards_v1 backend stores every tile as a
// PNG member of a tar (the reference's tests/test_spatial_datasets.py:57-75 writes them; its dataset classes open them with
// PIL on CPU dataloader workers).  At ~7.5 k pairs/s per GPU the host cannot inflate ~1.1 GB/s of zlib streams, so the
// compressed files are copied to HBM as they are and decoded next to their consumer (sc_augment_tiles).
//
// One wave per tile; the parallelism is across tiles (the data module inflates several batches per launch) and, inside a
// tile, wherever the format allows it.  The decoder's control state is wave-uniform and kept in SGPRs (sc_png_core.h, the same
// source the CPU test harness compiles, with the device policy below):
//   * input    the IDAT payloads are pulled through a 2-KiB LDS window, refilled or slid by all lanes (chunk boundaries there);
//   * literals 64 stream bits per round: every lane looks up the code that would start at ITS bit, the wave walks the chain of
//              code starts with v_readlane and the lanes on the chain store their literals (DevIO::literal_run);
//   * matches  an LZ77 match is ONE wave-wide gather / scatter whatever its length (an overlapping match repeats with period
//              `dist`, so lane i reads byte i mod dist of the source run) -- stores and the later loads of other lanes are
//              ordered by workgroup-scope release / acquire fences (one wave = one workgroup: a wait, no cache maintenance);
//   * filters  PNG's Sub / Up / Average / Paeth predictors depend on the left, upper and upper-left neighbours: 64 rows are
//              reconstructed together, row r one pixel behind row r - 1, and the neighbours of the row above arrive by
//              lane shifts (no memory round trip inside a band).
// Footprint: 5.9 KiB of LDS, 82 VGPRs, one wave per workgroup: five waves per SIMD, and it runs wherever a SIMD has room.
// Limits (anything else sets status != 0 and the host decodes that tile with PIL): 8-bit RGB / RGBA, no interlace, tile
// size = the requested H x W, at most 32 IDAT chunks.
#include "sc_common.h"
#include "sc_kernels.h"
#include "sc_png_core.h"

namespace {

constexpr int kWin = 2048;             // bytes of compressed input held in LDS

struct Lds {
    sc_png::Tables T;
    sc_png::Header h;
    unsigned in[kWin / 4 + 2];         // input window, read as dwords (+2: the literal probe of the last lanes reads past its bits)
};

struct DevIO {
    const unsigned char* file;
    Lds* L;
    unsigned char* out;
    int cap, n;              // n: bytes committed to `out` (`pending` more wait in `acc`); H (4 W + 1) < 2^31 is checked at launch
    int lane;
    int seg;
    unsigned pos;            // cursor inside IDAT segment `seg`
    int in_pos, in_fill;
    int pending;
    int acc;                 // up to 64 decoded literals, lane i holds literal i, stored as one 64-byte burst
    int skip;                // literal_run back-off after a round that found (almost) no literals

    // Every value that steers the decoder is wave-uniform, but the compiler only believes it for values it can prove so:
    // whatever comes out of LDS is passed through v_readfirstlane (u()) and every function is force-inlined (a call passes
    // `this` through memory), otherwise the whole control flow is compiled as divergent: exec-mask bookkeeping around every `if`.
    static __device__ __forceinline__ int u(int v) { return __builtin_amdgcn_readfirstlane(v); }
    __device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

    // fill the window from byte `start` on with the next IDAT payload bytes (chunk boundaries are crossed here)
    __device__ __forceinline__ void refill(int start) {
        __syncthreads();                                   // everybody is done with the old window
        int fill = start;
        const int nseg = u(L->h.nseg);
        while (fill < kWin && seg < nseg) {
            const unsigned left = (unsigned)u((int)L->h.seg_len[seg]) - pos;
            if (left == 0) { ++seg; pos = 0; continue; }
            const int take = (int)min((unsigned)(kWin - fill), left);
            const unsigned char* src = file + (unsigned)u((int)L->h.seg_off[seg]) + pos;
            for (int i = lane; i < take; i += 64) reinterpret_cast<unsigned char*>(L->in)[fill + i] = src[i];
            fill += take;
            pos += take;
        }
        in_fill = fill;
        __syncthreads();
    }
    // drop the bytes before `keep_from` (a multiple of 4; at most 256 bytes stay) and fill up behind what stays
    __device__ __forceinline__ void slide(int keep_from) {
        const int nd = (in_fill - keep_from + 3) >> 2;
        __syncthreads();
        unsigned v = 0;
        if (lane < nd) v = L->in[(keep_from >> 2) + lane];
        __syncthreads();
        if (lane < nd) L->in[lane] = v;
        in_pos -= keep_from;
        refill(in_fill - keep_from);
    }
    __device__ __forceinline__ int get_byte() {
        if (in_pos >= in_fill) {
            refill(0);
            in_pos = 0;
            if (in_fill == 0) return -1;
        }
        return u(reinterpret_cast<unsigned char*>(L->in)[in_pos++]);
    }
    // up to four stream bytes: a whole dword when the cursor is aligned, otherwise the bytes up to the next dword boundary
    __device__ __forceinline__ int get_word(unsigned& w) {
        if (in_pos >= in_fill) {
            refill(0);
            in_pos = 0;
            if (in_fill == 0) { w = 0; return 0; }
        }
        if ((in_pos & 3) == 0 && in_pos + 4 <= in_fill) {
            w = (unsigned)u((int)L->in[in_pos >> 2]);
            in_pos += 4;
            return 4;
        }
        w = 0;
        int nb = 0;
        const int lim = min(in_fill, (in_pos | 3) + 1);
        for (; in_pos < lim; ++nb) w |= (unsigned)u(reinterpret_cast<unsigned char*>(L->in)[in_pos++]) << (8 * nb);
        return nb;
    }
    __device__ __forceinline__ void flush() {
        if (pending == 0) return;
        // (the wave barriers keep this lane-dependent branch from being merged with the uniform ones around it, which would
        // turn the decoder's whole control flow divergent)
        __builtin_amdgcn_wave_barrier();
        if (lane < pending) out[n + lane] = (unsigned char)acc;
        __builtin_amdgcn_wave_barrier();
        n += pending;
        pending = 0;
    }
    __device__ __forceinline__ bool put_literal(int b) {
        if (n + pending >= cap) return false;
        acc = lane == pending ? b : acc;
        if (++pending == 64) flush();
        return true;
    }
    __device__ __forceinline__ bool copy_match(int dist, int len) {
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
    __device__ __forceinline__ bool copy_stored(int k) {
        for (int i = 0; i < k; ++i) {
            const int b = get_byte();
            if (b < 0 || !put_literal(b)) return false;
        }
        return true;
    }

    // Runs of literals, 64 stream bits per round (photographs of tissue compress to literals only: 150,752 of them and not
    // one match in the bench tile).  A Huffman stream is sequential because a code's position depends on the lengths of all
    // codes before it -- but WHICH code starts at a given bit does not.  Lane k looks up the code that would start at bit
    // cursor + k (one LDS look-up for 64 candidate positions); the wave then follows the chain 0 -> 0 + len -> ... through the
    // lanes with v_readlane (a few scalar instructions per symbol instead of a dependent LDS round trip) and the lanes
    // on the chain store their literals.  The chain stops at the first symbol that is not a short-coded literal (length / end-of-block
    // symbols, codes longer than the look-up table): the scalar decoder takes that one and the next round starts behind it.
    template <class BR>
    __device__ __forceinline__ void literal_run(BR& br, const sc_png::Tables& T) {
        if (skip > 0) { --skip; return; }
        int c = in_pos * 8 - br.cnt;                       // stream cursor as a bit offset into the window
        if (c < 0) return;                                 // the bit buffer still holds bytes of the previous window
        bool moved = false;
        flush();                                           // the rounds store behind what the scalar path collected
        for (;;) {
            if (c + 80 > in_fill * 8) {                    // lane 63 needs bits up to c + 63 + kFastBitsL
                if (in_fill < kWin) break;                 // the stream's tail: scalar
                const int keep = (c >> 3) & ~3;
                if (keep == 0) break;
                slide(keep);
                c -= keep * 8;
                if (c + 80 > in_fill * 8) break;
            }
            if (n + 64 > cap) break;                       // a round emits at most 64 literals
            const int bit = c + lane;
            const unsigned lo = L->in[bit >> 5], hi = L->in[(bit >> 5) + 1];
            const unsigned w = __builtin_amdgcn_alignbit(hi, lo, bit & 31);
            const unsigned e = T.fast_l[w & ((1u << sc_png::kFastBitsL) - 1)];
            const bool lit = e != 0 && e < (256u << 4);
            // next position of the chain from this lane; bit 6 ends the walk (a literal that ends at or past bit 64 -- or, with
            // bit 7, a symbol the walk must stop IN FRONT of)
            const int next = lit ? lane + (int)(e & 15) : 0xC0 | lane;
            unsigned long long mask;                       // the lanes on the chain
            int p;
            // four scalar issue slots per symbol (+2 wait states: a v_readlane result may not select the next v_readlane's
            // lane earlier than 4 states later); the compiler's version of this loop took 17 with three branches
            asm volatile("s_mov_b64 %0, 0\n\t"
                         "s_mov_b32 %1, 0\n"
                         "sc_png_chase_%=:\n\t"
                         "s_bitset1_b64 %0, %1\n\t"
                         "s_nop 1\n\t"
                         "v_readlane_b32 %1, %2, %1\n\t"
                         "s_bitcmp0_b32 %1, 6\n\t"
                         "s_cbranch_scc1 sc_png_chase_%=\n\t"
                         : "=&s"(mask), "=&s"(p)
                         : "v"(next)
                         : "scc");
            if (p & 0x80) {                                // stopped in front of lane p & 63, which the walk marked
                p &= 63;
                mask &= ~(1ull << p);
            }
            // the chain's literals leave in stream order: lane -> rank among the chain's lanes (the bytes of one round are
            // adjacent, the L2 merges the partial lines of successive rounds)
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            __builtin_amdgcn_wave_barrier();
            if ((mask >> lane) & 1) out[n + rank] = (unsigned char)(e >> 4);
            __builtin_amdgcn_wave_barrier();
            const int got = __builtin_popcountll(mask);
            n += got;
            c += p;
            moved = true;
            if (p < 64) {                                  // stopped in front of a symbol for the scalar decoder
                if (got < 2) skip = 8;
                break;
            }
        }
        if (moved) {                                       // re-seat the bit reader at the new cursor
            in_pos = c >> 3;
            br.buf = 0;
            br.cnt = 0;
            if (c & 7) br.bits(c & 7);
        }
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
    rc = __builtin_amdgcn_readfirstlane(rc);
    __syncthreads();
    if (rc != sc_png::OK) {
        if (lane == 0) status[b] = rc;
        return;
    }
    const int bpp = __builtin_amdgcn_readfirstlane(L.h.channels), rowb = W * bpp, stride = rowb + 1;
    const int raw = H * stride;
    unsigned char* buf = scratch + (long long)b * scratch_per_image;
    DevIO io{file, &L, buf, raw, 0, lane, 0, 0u, 0, 0, 0, 0, 0};
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
        // the bytes of pixel x + 1 (and, for lane 0, of the pixel above it) are requested before pixel x is rebuilt and stored:
        // one exposed memory latency per pixel step would otherwise be the whole cost of this stage
        int cur[4] = {0, 0, 0, 0}, upc[4] = {0, 0, 0, 0}, upprev[4] = {0, 0, 0, 0};
        auto load_px = [&](int x, int (&dst)[4], int (&updst)[4]) {
            const bool on = active && x >= 0 && x < W;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c >= bpp) break;
                dst[c] = on ? row[1 + x * bpp + c] : 0;
                if (lane == 0) updst[c] = (on && y > 0) ? above[1 + x * bpp + c] : 0;
            }
        };
        load_px(-lane, cur, upc);
        for (int t = 0; t < W + 63; ++t) {
            const int x = t - lane;
            const bool on = active && x >= 0 && x < W;
            int nxt[4] = {0, 0, 0, 0}, upn[4] = {0, 0, 0, 0};
            load_px(x + 1, nxt, upn);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c >= bpp) break;
                int up = __shfl_up(last1[c], 1, 64), ul = __shfl_up(last2[c], 1, 64);
                if (lane == 0) {
                    up = upc[c];
                    ul = upprev[c];                        // the pixel above x - 1 (zeros in front of the row)
                }
                if (on) {
                    const int a = x > 0 ? left[c] : 0;
                    if (x == 0) ul = 0;
                    int v = cur[c];
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
                    upprev[c] = upc[c];
                }
                cur[c] = nxt[c];
                upc[c] = upn[c];
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
    SC_CHECK((long long)H * (4LL * W + 1) < (1LL << 31), "sc_png_decode: tile %d x %d too large", H, W);
    const long long per = sc_png_decode_scratch_bytes(1, H, W);
    png_decode_kernel<<<B, 64, 0, (hipStream_t)stream>>>((const unsigned char*)files, offsets, (unsigned char*)out_rgb, H, W,
                                                         (unsigned char*)scratch, per, status);
    SC_LAUNCH_CHECK();
    return 0;
}
