// TEST INFRASTRUCTURE: the DEFLATE / PNG core of the device decoder (spatial-clip_amd/csrc/sc_png_core.h) compiled for the
// CPU with plain-array IO, so that tests/test_cpu_png.py can check the bit-stream logic against PIL / zlib on the build
// machine.  The product path never links this file (the device kernel instantiates the same header in sc_png.hip).
//   g++ -O2 -shared -fPIC -o oracle/_host/libpngcore.so oracle/png_core_host.cpp
#include "../spatial-clip_amd/csrc/sc_png_core.h"
#include <stdlib.h>
#include <string.h>

namespace {
struct HostIO {
    const uint8_t* file;
    const sc_png::Header* h;
    int seg = 0;
    uint32_t pos = 0;
    uint8_t* out;
    long long cap, n = 0;
    int get_byte() {
        while (seg < h->nseg && pos >= h->seg_len[seg]) { ++seg; pos = 0; }
        if (seg >= h->nseg) return -1;
        return file[h->seg_off[seg] + pos++];
    }
    int uniform(int v) { return v; }
    template <class BR> void literal_run(BR&, const sc_png::Tables&) {}   // the device's wave-parallel fast path
    int get_word(uint32_t& w) {
        w = 0;
        int nb = 0;
        for (; nb < 4; ++nb) {
            const int b = get_byte();
            if (b < 0) break;
            w |= (uint32_t)b << (8 * nb);
        }
        return nb;
    }
    long long n_lit = 0, n_match = 0, n_match_bytes = 0, n_far = 0;
    bool put_literal(int b) {
        if (n >= cap) return false;
        out[n++] = (uint8_t)b;
        ++n_lit;
        return true;
    }
    bool copy_match(int dist, int len) {
        if (dist > n || n + len > cap) return false;
        ++n_match; n_match_bytes += len; n_far += dist > 7900;
        for (int i = 0; i < len; ++i, ++n) out[n] = out[n - dist];
        return true;
    }
    bool copy_stored(int k) {
        for (int i = 0; i < k; ++i) {
            const int b = get_byte();
            if (b < 0 || !put_literal(b)) return false;
        }
        return true;
    }
};
}  // namespace

// -> 0 and out_rgb[H][W][3], or an sc_png error code
extern "C" int sc_png_host_decode(const uint8_t* file, long long n, uint8_t* out_rgb, int H, int W) {
    sc_png::Header h;
    int rc = sc_png::parse(file, n, h);
    if (rc) return rc;
    if (h.width != W || h.height != H) return sc_png::ERR_SIZE;
    const int bpp = h.channels, rowb = W * bpp;
    const long long raw = (long long)H * (rowb + 1);
    uint8_t* buf = (uint8_t*)malloc(raw);
    sc_png::Tables* T = (sc_png::Tables*)malloc(sizeof(sc_png::Tables));
    HostIO io{file, &h, 0, 0, buf, raw, 0};
    rc = sc_png::inflate(io, *T);
    if (rc == 0 && io.n != raw) rc = sc_png::ERR_TRUNCATED;
    if (rc == 0) {
        uint8_t* prev = (uint8_t*)calloc(rowb, 1);
        uint8_t* cur = (uint8_t*)malloc(rowb);
        for (int y = 0; y < H && rc == 0; ++y) {
            const uint8_t* src = buf + (long long)y * (rowb + 1);
            const int ft = src[0];
            if (ft > 4) { rc = sc_png::ERR_FORMAT; break; }
            for (int x = 0; x < rowb; ++x) {
                const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
                int v = src[1 + x];
                if (ft == 1) v += a; else if (ft == 2) v += b; else if (ft == 3) v += (a + b) >> 1; else if (ft == 4) v += sc_png::paeth(a, b, c);
                cur[x] = (uint8_t)v;
            }
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < 3; ++c) out_rgb[((long long)y * W + x) * 3 + c] = cur[x * bpp + c];
            memcpy(prev, cur, rowb);
        }
        free(prev); free(cur);
    }
    free(buf); free(T);
    return rc;
}

// symbol statistics of a file's DEFLATE stream (tools/bench_input_pipeline.py): literals, matches, matched bytes, matches
// further back than an 8-KiB window would hold
extern "C" int sc_png_host_stats(const uint8_t* file, long long n, long long* out4) {
    sc_png::Header h;
    int rc = sc_png::parse(file, n, h);
    if (rc) return rc;
    const long long raw = (long long)h.height * ((long long)h.width * h.channels + 1);
    uint8_t* buf = (uint8_t*)malloc(raw);
    sc_png::Tables* T = (sc_png::Tables*)malloc(sizeof(sc_png::Tables));
    HostIO io{file, &h, 0, 0, buf, raw, 0};
    rc = sc_png::inflate(io, *T);
    out4[0] = io.n_lit; out4[1] = io.n_match; out4[2] = io.n_match_bytes; out4[3] = io.n_far;
    free(buf); free(T);
    return rc;
}
