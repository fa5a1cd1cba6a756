// DEFLATE (RFC 1951) decoder core of the device PNG reader (sc_png.hip), written once for both sides: the HIP kernel
// instantiates it with a wave-cooperative IO policy, tests/png_core_host.cpp (g++, CPU) with plain arrays, so that the
// bit-stream logic is checked against zlib on the build machine before it ever runs on a GPU.
//
// The control flow is strictly sequential (a DEFLATE stream cannot be entered in the middle); on the device every lane of
// the wave executes it with identical values, and the IO policy turns "emit a byte" into a store by one lane and "copy a
// match" into a copy by all 64.
//
// IO policy:   int  get_byte()                 next byte of the zlib stream, -1 past its end
//              int  get_word(uint32_t& w)      next up-to-four bytes (little endian) -> how many are real (0 = end of stream)
//              int  uniform(int v)             identity; the device policy returns lane 0's copy (v_readfirstlane), which tells
//                                              the compiler that a table entry read from LDS is wave-uniform -> scalar ALU
//              bool put_literal(int b)         append one byte to the output; false when the output is full
//              bool copy_match(int dist, int len)   append len bytes starting dist bytes back (may overlap); false on a
//                                              bad distance / overflow
//              bool copy_stored(int n)         append the next n input bytes (byte-aligned "stored" block)
// Reference behaviour being replaced: PIL.Image.open(...).convert("RGB") on the dataloader workers of the shards_v1
// backend (tests/test_spatial_datasets.py:57-75 writes the tiles as PNG members of a tar).
#pragma once
#include <stdint.h>

#ifndef SC_HD
#ifdef __HIPCC__
#define SC_HD __host__ __device__ __forceinline__
#else
#define SC_HD inline
#endif
#endif

namespace sc_png {

constexpr int kFastBitsL = 10;          // literal/length codes up to this many bits resolve in one table look-up
constexpr int kFastBitsD = 8;
enum { OK = 0, ERR_TRUNCATED = 1, ERR_BLOCK_TYPE = 2, ERR_STORED_LEN = 3, ERR_CODE_LENGTHS = 4, ERR_SYMBOL = 5,
       ERR_DISTANCE = 6, ERR_OUTPUT_FULL = 7, ERR_FORMAT = 8, ERR_UNSUPPORTED = 9, ERR_SIZE = 10 };

struct Tables {
    uint16_t fast_l[1 << kFastBitsL];   // (symbol << 4) | length, 0 = longer than kFastBits: canonical walk
    uint16_t fast_d[1 << kFastBitsD];
    uint16_t cnt_l[16], cnt_d[16];      // codes per length
    uint16_t sym_l[288], sym_d[32];     // symbols sorted by (length, symbol)
    uint8_t len[352];                   // scratch: 19 code-length-code lengths, then up to 286 + 30 code lengths
};

template <class IO>
struct BitReader {
    IO& io;
    uint64_t buf = 0;
    int cnt = 0;
    bool eof = false;
    SC_HD explicit BitReader(IO& i) : io(i) {}
    // Refill in 32-bit gulps (io.get_word: up to four stream bytes, little endian, and how many of them are real): on the
    // device every look-up is a dependent LDS round trip of ~100 cycles, so the input side must not add one per byte.
    SC_HD void fill(int need) {
        while (cnt < need && !eof) {
            if (cnt > 32) {                          // no room for a whole word: single bytes (only near `need` > 32, unused)
                const int b = io.get_byte();
                if (b < 0) { eof = true; break; }
                buf |= (uint64_t)(unsigned)b << cnt;
                cnt += 8;
                continue;
            }
            uint32_t w = 0;
            const int nb = io.get_word(w);
            if (nb <= 0) { eof = true; break; }
            buf |= (uint64_t)w << cnt;
            cnt += 8 * nb;                           // nb < 4: a window / stream boundary; the next call tells which
        }
    }
    SC_HD int bits(int n) {             // n <= 16; -1 when the stream ends first
        if (n == 0) return 0;
        fill(n);
        if (cnt < n) return -1;
        const int v = (int)(buf & ((1u << n) - 1));
        buf >>= n;
        cnt -= n;
        return v;
    }
    SC_HD void align() { const int drop = cnt & 7; buf >>= drop; cnt -= drop; }
};

// canonical-code tables from code lengths len[0..n): counts, sorted symbols, and the fast look-up table
template <class IO>
SC_HD int build(IO& io, const uint8_t* len, int n, uint16_t* cnt, uint16_t* sym, uint16_t* fast, int fast_bits) {
    int c[16], offs[16];                             // counts in registers: the tables themselves live in (device: LDS) memory
    for (int l = 0; l < 16; ++l) c[l] = 0;
    for (int s = 0; s < n; ++s) {
        const int l = io.uniform(len[s]);
        for (int k = 0; k < 16; ++k) c[k] += (k == l);
    }
    for (int l = 0; l < 16; ++l) cnt[l] = (uint16_t)c[l];
    if (c[0] == n) return 0;                         // no codes at all: legal for an unused distance alphabet
    int left = 1;
    for (int l = 1; l < 16; ++l) {
        left <<= 1;
        left -= c[l];
        if (left < 0) return -1;                     // over-subscribed
    }
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + c[l];
    for (int s = 0; s < n; ++s) {
        const int l = io.uniform(len[s]);
        if (l) {
            int o = 0;
            for (int k = 1; k < 16; ++k) { o += (k == l) ? offs[k] : 0; offs[k] += (k == l); }
            sym[o] = (uint16_t)s;
        }
    }
    for (int i = 0; i < (1 << fast_bits); ++i) fast[i] = 0;
    int code = 0, idx = 0;
    for (int l = 1; l <= fast_bits; ++l) {
        const int cl = io.uniform(cnt[l]);
        for (int k = 0; k < cl; ++k, ++code, ++idx) {
            int rev = 0;                             // DEFLATE packs Huffman codes MSB first into an LSB-first stream
            for (int b = 0; b < l; ++b) rev |= ((code >> b) & 1) << (l - 1 - b);
            const uint16_t e = (uint16_t)((io.uniform(sym[idx]) << 4) | l);
            for (int j = rev; j < (1 << fast_bits); j += 1 << l) fast[j] = e;
        }
        code <<= 1;
    }
    return left;                                     // > 0: incomplete code (allowed only in special cases by the caller)
}

template <class IO>
SC_HD int decode_sym(BitReader<IO>& br, const uint16_t* cnt, const uint16_t* sym, const uint16_t* fast, int fast_bits) {
    br.fill(15);
    const uint16_t e = (uint16_t)br.io.uniform(fast[br.buf & ((1u << fast_bits) - 1)]);
    if (e) {
        const int l = e & 15;
        if (l > br.cnt) return -1;
        br.buf >>= l;
        br.cnt -= l;
        return e >> 4;
    }
    int code = 0, first = 0, index = 0;              // canonical walk, one bit at a time (codes longer than fast_bits)
#ifdef SC_PNG_STATS
    ++sc_png_stats_long_codes;
#endif
    for (int l = 1; l <= 15; ++l) {
        if (br.cnt < 1) return -1;
        code |= (int)(br.buf & 1);
        br.buf >>= 1;
        br.cnt -= 1;
        const int count = br.io.uniform(cnt[l]);
        if (code - count < first) return br.io.uniform(sym[index + (code - first)]);
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

SC_HD void fixed_lengths(uint8_t* len) {
    int s = 0;
    for (; s < 144; ++s) len[s] = 8;
    for (; s < 256; ++s) len[s] = 9;
    for (; s < 280; ++s) len[s] = 7;
    for (; s < 288; ++s) len[s] = 8;
}

// zlib stream (2-byte header, DEFLATE blocks; the Adler-32 trailer is not verified) -> output through io
template <class IO>
SC_HD int inflate(IO& io, Tables& T) {
    const int base_l[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const int extra_l[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const int base_d[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    const int extra_d[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    BitReader<IO> br(io);
    const int cmf = br.bits(8), flg = br.bits(8);
    if (cmf < 0 || flg < 0) return ERR_TRUNCATED;
    if ((cmf & 15) != 8 || ((cmf << 8) + flg) % 31 != 0 || (flg & 32)) return ERR_FORMAT;
    for (;;) {
        const int last = br.bits(1), type = br.bits(2);
        if (last < 0 || type < 0) return ERR_TRUNCATED;
        if (type == 0) {
            br.align();
            // whole bytes still in the bit buffer belong to LEN / NLEN and the data: hand them back through bits()
            const int lo = br.bits(8), hi = br.bits(8), nlo = br.bits(8), nhi = br.bits(8);
            if (lo < 0 || hi < 0 || nlo < 0 || nhi < 0) return ERR_TRUNCATED;
            int n = lo | (hi << 8);
            if ((n ^ (nlo | (nhi << 8))) != 0xFFFF) return ERR_STORED_LEN;
            while (n > 0 && br.cnt >= 8) {           // bytes already pulled into the bit buffer
                if (!io.put_literal(br.bits(8))) return ERR_OUTPUT_FULL;
                --n;
            }
            if (n > 0 && !io.copy_stored(n)) return ERR_TRUNCATED;
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                fixed_lengths(T.len);
                build(io, T.len, 288, T.cnt_l, T.sym_l, T.fast_l, kFastBitsL);
                for (int s = 0; s < 30; ++s) T.len[s] = 5;
                build(io, T.len, 30, T.cnt_d, T.sym_d, T.fast_d, kFastBitsD);
            } else {
                const int hlit = br.bits(5), hdist = br.bits(5), hclen = br.bits(4);
                if (hlit < 0 || hdist < 0 || hclen < 0) return ERR_TRUNCATED;
                const int nl = hlit + 257, nd = hdist + 1, nc = hclen + 4;
                if (nl > 286 || nd > 30) return ERR_CODE_LENGTHS;
                for (int i = 0; i < 19; ++i) T.len[i] = 0;
                for (int i = 0; i < nc; ++i) {
                    const int v = br.bits(3);
                    if (v < 0) return ERR_TRUNCATED;
                    T.len[order[i]] = (uint8_t)v;
                }
                // the code-length code borrows the distance tables (rebuilt below)
                if (build(io, T.len, 19, T.cnt_d, T.sym_d, T.fast_d, 7) != 0) return ERR_CODE_LENGTHS;
                int i = 0;
                while (i < nl + nd) {
                    const int s = decode_sym(br, T.cnt_d, T.sym_d, T.fast_d, 7);
                    if (s < 0) return ERR_CODE_LENGTHS;
                    if (s < 16) {
                        T.len[19 + i++] = (uint8_t)s;
                    } else {
                        int prev = 0, rep;
                        if (s == 16) {
                            if (i == 0) return ERR_CODE_LENGTHS;
                            prev = io.uniform(T.len[19 + i - 1]);
                            rep = 3 + br.bits(2);
                            if (rep < 3) return ERR_TRUNCATED;
                        } else if (s == 17) {
                            rep = 3 + br.bits(3);
                            if (rep < 3) return ERR_TRUNCATED;
                        } else {
                            rep = 11 + br.bits(7);
                            if (rep < 11) return ERR_TRUNCATED;
                        }
                        if (i + rep > nl + nd) return ERR_CODE_LENGTHS;
                        while (rep--) T.len[19 + i++] = (uint8_t)prev;
                    }
                }
                if (io.uniform(T.len[19 + 256]) == 0) return ERR_CODE_LENGTHS;            // no end-of-block code
                // the two builds read T.len[19..]; build() does not write T.len
                const int rl = build(io, T.len + 19, nl, T.cnt_l, T.sym_l, T.fast_l, kFastBitsL);
                if (rl < 0 || (rl > 0 && nl - T.cnt_l[0] != 1)) return ERR_CODE_LENGTHS;
                const int rd = build(io, T.len + 19 + nl, nd, T.cnt_d, T.sym_d, T.fast_d, kFastBitsD);
                if (rd < 0 || (rd > 0 && nd - T.cnt_d[0] != 1)) return ERR_CODE_LENGTHS;
            }
            for (;;) {
                io.literal_run(br, T);                   // device: a wave-parallel run of literals; elsewhere a no-op
                int s = decode_sym(br, T.cnt_l, T.sym_l, T.fast_l, kFastBitsL);
                if (s < 0) return br.eof ? ERR_TRUNCATED : ERR_SYMBOL;
                if (s < 256) {
                    if (!io.put_literal(s)) return ERR_OUTPUT_FULL;
                } else if (s == 256) {
                    break;
                } else {
                    s -= 257;
                    if (s >= 29) return ERR_SYMBOL;
                    const int xl = br.bits(extra_l[s]);
                    if (xl < 0) return ERR_TRUNCATED;
                    const int len = base_l[s] + xl;
                    const int ds = decode_sym(br, T.cnt_d, T.sym_d, T.fast_d, kFastBitsD);
                    if (ds < 0 || ds >= 30) return ERR_DISTANCE;
                    const int xd = br.bits(extra_d[ds]);
                    if (xd < 0) return ERR_TRUNCATED;
                    if (!io.copy_match(base_d[ds] + xd, len)) return ERR_DISTANCE;
                }
            }
        } else {
            return ERR_BLOCK_TYPE;
        }
        if (last) return OK;
    }
}

// ---- PNG container: signature, IHDR, IDAT segments.  Returns OK and fills the header / segment list.
struct Header {
    int width, height, channels;        // channels of the stored image: 3 (RGB) or 4 (RGBA), 8 bits each, not interlaced
    int nseg;
    uint32_t seg_off[32];               // IDAT payloads inside the file
    uint32_t seg_len[32];
};
SC_HD uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

SC_HD int parse(const uint8_t* f, long long n, Header& h) {
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 + 25) return ERR_TRUNCATED;
    for (int i = 0; i < 8; ++i)
        if (f[i] != sig[i]) return ERR_FORMAT;
    long long pos = 8;
    h.nseg = 0;
    h.width = h.height = h.channels = 0;
    while (pos + 12 <= n) {
        const uint32_t len = be32(f + pos);
        const uint8_t* t = f + pos + 4;
        if (pos + 12 + (long long)len > n) return ERR_TRUNCATED;
        if (t[0] == 'I' && t[1] == 'H' && t[2] == 'D' && t[3] == 'R') {
            if (len != 13) return ERR_FORMAT;
            const uint8_t* d = f + pos + 8;
            h.width = (int)be32(d);
            h.height = (int)be32(d + 4);
            if (d[8] != 8 || d[12] != 0 || d[10] != 0 || d[11] != 0) return ERR_UNSUPPORTED;      // depth, interlace
            if (d[9] == 2) h.channels = 3;
            else if (d[9] == 6) h.channels = 4;
            else return ERR_UNSUPPORTED;                                                          // gray / palette
        } else if (t[0] == 'I' && t[1] == 'D' && t[2] == 'A' && t[3] == 'T') {
            if (h.nseg == 32) return ERR_UNSUPPORTED;
            h.seg_off[h.nseg] = (uint32_t)(pos + 8);
            h.seg_len[h.nseg] = len;
            h.nseg++;
        } else if (t[0] == 'I' && t[1] == 'E' && t[2] == 'N' && t[3] == 'D') {
            break;
        }
        pos += 12 + (long long)len;
    }
    if (h.channels == 0 || h.nseg == 0) return ERR_FORMAT;
    return OK;
}

SC_HD int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace sc_png
