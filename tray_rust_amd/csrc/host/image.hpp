// Image files for the `image` / `animated_image` / `movie` textures of the scene format (scene.rs:317-394 calls image::open of
// the `image` crate 0.18, not vendored by the reference): decoded to RGBA8 in the layout DynamicImage::get_pixel presents --
// row 0 on top, grey expanded to (l, l, l, 255), RGB given alpha 255 (texture/image.rs:18-32 reads px.data[0..4]).
// Formats: PNG (non-interlaced; grey / grey+alpha / RGB / RGBA / palette with tRNS; 1-16 bits, 16-bit samples keep their high
// byte), binary PPM / PGM (P6 / P5, maxval <= 255), BMP (uncompressed 24 / 32 bit), TGA (uncompressed true colour 24 / 32 bit and
// grey 8 bit), baseline and progressive JPEG (see decode_jpeg), GIF (first frame, see decode_gif), TIFF (baseline strips: uncompressed / LZW / PackBits,
// see decode_tiff), Radiance HDR (tone-mapped to 8 bit as the crate does, see decode_hdr), ICO (the best entry, PNG or BMP payload), WebP (simple lossy
// files, as the GREY image of their luma plane that image 0.18 makes of them: webp.hpp). No third-party code: the inflate below is the textbook
// RFC 1951 decoder.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "webp.hpp"

namespace trayh {

struct ImageRGBA8 {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> px;   // width * height * 4
};

namespace img_detail {

// ---- RFC 1951 inflate (zlib stream without preset dictionary) ----
struct BitReader {
    const uint8_t* p; size_t n, pos = 0; uint32_t bitbuf = 0; int bitcnt = 0; bool bad = false;
    BitReader(const uint8_t* d, size_t len) : p(d), n(len) {}
    uint32_t bits(int need) {
        uint32_t v = bitbuf;
        while (bitcnt < need) {
            if (pos >= n) { bad = true; return 0; }
            v |= (uint32_t)p[pos++] << bitcnt;
            bitcnt += 8;
        }
        bitbuf = need < 32 ? v >> need : 0;
        bitcnt -= need;
        return need < 32 ? v & ((1u << need) - 1u) : v;
    }
};
struct Huffman { uint16_t count[16]; uint16_t symbol[288]; };
inline bool build_huffman(Huffman& h, const uint8_t* length, int n) {
    std::memset(h.count, 0, sizeof h.count);
    for (int s = 0; s < n; ++s) h.count[length[s]]++;
    if (h.count[0] == n) return true;   // no codes: legal for the distance table of a literal-only block
    int left = 1;
    for (int len = 1; len < 16; ++len) { left <<= 1; left -= h.count[len]; if (left < 0) return false; }
    uint16_t offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = offs[len] + h.count[len];
    for (int s = 0; s < n; ++s) if (length[s]) h.symbol[offs[length[s]]++] = (uint16_t)s;
    return true;
}
inline int decode_symbol(BitReader& br, const Huffman& h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; ++len) {
        code |= (int)br.bits(1);
        if (br.bad) return -1;
        const int count = h.count[len];
        if (code - count < first) return h.symbol[index + (code - first)];
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;
}
inline bool inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (n < 6 || (src[0] & 0x0f) != 8 || ((src[0] << 8 | src[1]) % 31) != 0 || (src[1] & 0x20)) return false;
    BitReader br(src + 2, n - 2);
    out.clear();
    out.reserve(expect);
    for (;;) {
        const uint32_t last = br.bits(1), type = br.bits(2);
        if (br.bad) return false;
        if (type == 0) {
            br.bitbuf = 0; br.bitcnt = 0;
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | br.p[br.pos + 1] << 8, nlen = br.p[br.pos + 2] | br.p[br.pos + 3] << 8;
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > br.n) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lencode, distcode;
            uint8_t lengths[320];
            if (type == 1) {
                int s = 0;
                for (; s < 144; ++s) lengths[s] = 8;
                for (; s < 256; ++s) lengths[s] = 9;
                for (; s < 280; ++s) lengths[s] = 7;
                for (; s < 288; ++s) lengths[s] = 8;
                build_huffman(lencode, lengths, 288);
                for (s = 0; s < 30; ++s) lengths[s] = 5;
                build_huffman(distcode, lengths, 30);
            } else {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (br.bad || nlen > 286 || ndist > 30) return false;
                std::memset(lengths, 0, sizeof lengths);
                for (int i = 0; i < ncode; ++i) lengths[order[i]] = (uint8_t)br.bits(3);
                Huffman cl;
                if (!build_huffman(cl, lengths, 19)) return false;
                int idx = 0;
                uint8_t ll[320];
                std::memset(ll, 0, sizeof ll);
                while (idx < nlen + ndist) {
                    const int sym = decode_symbol(br, cl);
                    if (sym < 0) return false;
                    if (sym < 16) ll[idx++] = (uint8_t)sym;
                    else {
                        int prev = 0, rep;
                        if (sym == 16) { if (idx == 0) return false; prev = ll[idx - 1]; rep = 3 + (int)br.bits(2); }
                        else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (br.bad || idx + rep > nlen + ndist) return false;
                        while (rep--) ll[idx++] = (uint8_t)prev;
                    }
                }
                if (ll[256] == 0) return false;
                if (!build_huffman(lencode, ll, nlen) || !build_huffman(distcode, ll + nlen, ndist)) return false;
            }
            for (;;) {
                const int sym = decode_symbol(br, lencode);
                if (sym < 0) return false;
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    const int ls = sym - 257;
                    if (ls >= 29) return false;
                    const int len = lbase[ls] + (int)br.bits(lext[ls]);
                    const int ds = decode_symbol(br, distcode);
                    if (ds < 0 || ds >= 30) return false;
                    const size_t dist = dbase[ds] + br.bits(dext[ds]);
                    if (br.bad || dist > out.size()) return false;
                    for (int k = 0; k < len; ++k) out.push_back(out[out.size() - dist]);
                }
                if (out.size() > expect + 65536) return false;   // a decompression bomb is not an image
            }
        } else return false;
        if (last) break;
    }
    return true;
}

inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

inline bool decode_png(const std::vector<uint8_t>& f, ImageRGBA8& out, std::string& err) {
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool end = false;
    while (!end && pos + 12 <= f.size()) {
        const uint32_t len = be32(&f[pos]);
        const char* tag = reinterpret_cast<const char*>(&f[pos + 4]);
        if (pos + 12 + (size_t)len > f.size()) { err = "truncated PNG chunk"; return false; }
        const uint8_t* d = &f[pos + 8];
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len < 13) { err = "bad IHDR"; return false; }
            w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
            if (d[10] != 0 || d[11] != 0) { err = "unknown PNG compression / filter method"; return false; }
        } else if (!std::memcmp(tag, "PLTE", 4)) plte.assign(d, d + len);
        else if (!std::memcmp(tag, "tRNS", 4)) trns.assign(d, d + len);
        else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!std::memcmp(tag, "IEND", 4)) end = true;
        pos += 12 + (size_t)len;
    }
    if (ctype < 0 || w == 0 || h == 0 || w > 32768 || h > 32768) { err = "PNG without a usable IHDR"; return false; }
    if (interlace != 0) { err = "interlaced PNG files are not supported"; return false; }
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!channels || !depth_ok) { err = "unsupported PNG colour type / bit depth"; return false; }
    const size_t bpp_bits = (size_t)channels * depth, stride = (w * bpp_bits + 7) / 8, bpp = (bpp_bits + 7) / 8;
    std::vector<uint8_t> raw;
    if (!inflate(idat.data(), idat.size(), raw, (stride + 1) * h) || raw.size() < (stride + 1) * h) { err = "corrupt PNG data stream"; return false; }
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    out.width = w; out.height = h;
    out.px.assign((size_t)w * h * 4, 255);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* line = &raw[(stride + 1) * y];
        const int ft = line[0];
        if (ft > 4) { err = "corrupt PNG filter byte"; return false; }
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int v = line[1 + i];
            if (ft == 1) v += a;
            else if (ft == 2) v += b;
            else if (ft == 3) v += (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            cur[i] = (uint8_t)v;
        }
        uint8_t* o = &out.px[(size_t)y * w * 4];
        for (uint32_t x = 0; x < w; ++x) {
            uint32_t s[4] = {0, 0, 0, 255};
            for (int ch = 0; ch < channels; ++ch) {
                if (depth == 8) s[ch] = cur[(size_t)x * channels + ch];
                else if (depth == 16) s[ch] = cur[((size_t)x * channels + ch) * 2];   // high byte
                else { const size_t bit = (size_t)x * depth; s[ch] = (cur[bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1u); }
            }
            if (ctype == 3) {
                const uint32_t idx = s[0];
                if ((size_t)idx * 3 + 2 >= plte.size()) { err = "PNG palette index out of range"; return false; }
                o[4 * x] = plte[idx * 3]; o[4 * x + 1] = plte[idx * 3 + 1]; o[4 * x + 2] = plte[idx * 3 + 2];
                o[4 * x + 3] = idx < trns.size() ? trns[idx] : 255;
            } else if (ctype == 0 || ctype == 4) {
                uint32_t l = s[0];
                if (ctype == 0 && depth < 8) l = l * 255u / ((1u << depth) - 1u);
                o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = (uint8_t)l;
                o[4 * x + 3] = ctype == 4 ? (uint8_t)s[1] : 255;
            } else {
                o[4 * x] = (uint8_t)s[0]; o[4 * x + 1] = (uint8_t)s[1]; o[4 * x + 2] = (uint8_t)s[2];
                o[4 * x + 3] = ctype == 6 ? (uint8_t)s[3] : 255;
            }
        }
        prev.swap(cur);
    }
    return true;
}

inline bool decode_pnm(const std::vector<uint8_t>& f, ImageRGBA8& out, std::string& err) {
    size_t pos = 2;
    auto next_int = [&](uint32_t& v) {
        for (;;) {
            while (pos < f.size() && (f[pos] == ' ' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\t')) ++pos;
            if (pos < f.size() && f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') ++pos; continue; }
            break;
        }
        if (pos >= f.size() || f[pos] < '0' || f[pos] > '9') return false;
        v = 0;
        while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') { v = v * 10 + (uint32_t)(f[pos] - '0'); if (v > 100000) return false; ++pos; }
        return true;
    };
    const int ch = f[1] == '6' ? 3 : 1;
    uint32_t w, h, maxv;
    if (!next_int(w) || !next_int(h) || !next_int(maxv) || w == 0 || h == 0 || maxv == 0 || maxv > 255) { err = "unsupported PNM header"; return false; }
    ++pos;   // one whitespace byte after maxval
    if (pos + (size_t)w * h * ch > f.size()) { err = "truncated PNM file"; return false; }
    out.width = w; out.height = h;
    out.px.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        for (int c = 0; c < 3; ++c) out.px[4 * i + c] = (uint8_t)(f[pos + i * ch + (ch == 3 ? c : 0)] * 255u / maxv);
        out.px[4 * i + 3] = 255;
    }
    return true;
}

inline bool decode_bmp(const std::vector<uint8_t>& f, ImageRGBA8& out, std::string& err) {
    if (f.size() < 54) { err = "truncated BMP file"; return false; }
    auto le32 = [&](size_t o) { return (uint32_t)f[o] | (uint32_t)f[o + 1] << 8 | (uint32_t)f[o + 2] << 16 | (uint32_t)f[o + 3] << 24; };
    const uint32_t off = le32(10), w = le32(18);
    const int32_t hs = (int32_t)le32(22);
    const uint32_t bits = f[28] | f[29] << 8, comp = le32(30);
    if ((bits != 24 && bits != 32) || (comp != 0 && comp != 3) || w == 0 || hs == 0 || w > 32768) { err = "only uncompressed 24 / 32-bit BMP files are supported"; return false; }
    const uint32_t h = (uint32_t)(hs < 0 ? -hs : hs);
    const size_t stride = ((size_t)w * bits / 8 + 3) & ~(size_t)3;
    if (off + stride * h > f.size()) { err = "truncated BMP file"; return false; }
    out.width = w; out.height = h;
    out.px.resize((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = &f[off + stride * (hs < 0 ? y : h - 1 - y)];
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t* p = row + (size_t)x * bits / 8;
            uint8_t* o = &out.px[((size_t)y * w + x) * 4];
            o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = bits == 32 ? p[3] : 255;
        }
    }
    return true;
}

inline bool decode_tga(const std::vector<uint8_t>& f, ImageRGBA8& out, std::string& err) {
    if (f.size() < 18) { err = "truncated TGA file"; return false; }
    const int idlen = f[0], cmap = f[1], type = f[2], bits = f[16], desc = f[17];
    const uint32_t w = f[12] | f[13] << 8, h = f[14] | f[15] << 8;
    if (cmap != 0 || !((type == 2 && (bits == 24 || bits == 32)) || (type == 3 && bits == 8)) || w == 0 || h == 0) { err = "only uncompressed true-colour / grey TGA files are supported"; return false; }
    const size_t bpp = (size_t)bits / 8, off = 18 + (size_t)idlen;
    if (off + (size_t)w * h * bpp > f.size()) { err = "truncated TGA file"; return false; }
    out.width = w; out.height = h;
    out.px.resize((size_t)w * h * 4);
    const bool top = (desc & 0x20) != 0;
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t* p = &f[off + ((size_t)(top ? y : h - 1 - y) * w + x) * bpp];
            uint8_t* o = &out.px[((size_t)y * w + x) * 4];
            if (bpp == 1) { o[0] = o[1] = o[2] = p[0]; o[3] = 255; }
            else { o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = bpp == 4 ? p[3] : 255; }
        }
    return true;
}

// ---- baseline JPEG (ITU T.81 sequential DCT, Huffman, 8 bit; 1 or 3 components; sampling 1x1 / 2x1 / 1x2 / 2x2; restart intervals).
// image 0.18 decodes JPEG with jpeg-decoder 0.1.13 (Cargo.lock; not vendored by the reference: PARITY UNPINNED). Restated from that
// crate's published design: the integer IDCT it ports from stb_image (stbi__idct_block: 12-bit fixed-point constants, column pass
// >> 10, row pass >> 17 with the +128 level shift folded in), its linear ("fancy") chroma upsampling -- h2v1 (3 a + b + 2) >> 2,
// h2v2 the 9-3-3-1 triangle (3 (3 near + far) + neighbour + 8) >> 4 with replicated edges -- and its f32 YCbCr -> RGB
// (1.402, 0.34414, 0.71414, 1.772; + 0.5, truncate, clamp). Lossy decoders differ by an LSB or two between implementations; the test
// (tests/test_textures.py) holds this one within 2 LSB of libjpeg's output. Progressive files (SOF2, T.81 Annex G: spectral selection and
// successive approximation; DC first / refinement scans interleaved or not, AC first / refinement scans per component with end-of-band
// runs) are collected coefficient by coefficient and transformed after the last scan, as jpeg-decoder does. Lossless, hierarchical and
// arithmetic-coded files are refused.
struct JpegComp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, dc_pred = 0, bw = 0, bh = 0; std::vector<uint8_t> plane;
                  std::vector<int16_t> coef; };   // progressive frames: 64 coefficients per block (natural order), filled scan by scan
struct JpegHuff { uint8_t bits[17] = {0}; uint8_t vals[256] = {0}; int mincode[17], maxcode[18], valptr[17]; bool present = false; };
inline void jpeg_build(JpegHuff& h) {   // T.81 F.2.2.3
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        h.valptr[l] = k; h.mincode[l] = code;
        code += h.bits[l]; k += h.bits[l];
        h.maxcode[l] = h.bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    h.present = true;
}
struct JpegBits {
    const uint8_t* p; size_t n, pos; uint32_t buf = 0; int cnt = 0; bool bad = false; int marker = 0;
    JpegBits(const uint8_t* d, size_t len, size_t start) : p(d), n(len), pos(start) {}
    int bit() {
        if (cnt == 0) {
            if (marker || pos >= n) { buf = 0; cnt = 8; if (pos >= n && !marker) bad = true; }   // past a marker: zero bits (T.81 F.2.2.5 pads with ones, decoders feed zeros)
            else {
                uint8_t b = p[pos++];
                if (b == 0xff) {
                    uint8_t b2 = pos < n ? p[pos] : 0;
                    if (b2 == 0) ++pos;                     // stuffed byte
                    else { marker = b2; --pos; b = 0; }     // a marker ends the entropy-coded segment
                }
                buf = b; cnt = 8;
            }
        }
        --cnt;
        return (int)((buf >> cnt) & 1u);
    }
    int receive(int s) { int v = 0; for (int i = 0; i < s; ++i) v = (v << 1) | bit(); return v; }
    void reset() { cnt = 0; buf = 0; }
};
inline int jpeg_decode(JpegBits& br, const JpegHuff& h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    br.bad = true;
    return 0;
}
inline int jpeg_extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }
inline uint8_t jpeg_clamp(long long x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
// stb_image's stbi__idct_block as jpeg-decoder ports it. The arithmetic is 64 bits wide: a valid file's values stay far inside 32 bits
// (same results), a corrupt file's coefficients must not overflow a signed int on the way to being clamped
inline void jpeg_idct(const int* d, uint8_t* out, int stride) {
    auto f2f = [](double x) { return (long long)(x * 4096.0 + 0.5); };
    long long val[64];
#define TR_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                          \
    long long t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                          \
    p2 = s2; p3 = s6; p1 = (p2 + p3) * f2f(0.5411961); t2 = p1 + p3 * f2f(-1.847759065); t3 = p1 + p2 * f2f(0.765366865); \
    p2 = s0; p3 = s4; t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                        \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                                \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                                    \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2; p5 = (p3 + p4) * f2f(1.175875602);             \
    t0 = t0 * f2f(0.298631336); t1 = t1 * f2f(2.053119869); t2 = t2 * f2f(3.072711026); t3 = t3 * f2f(1.501321110); \
    p1 = p5 + p1 * f2f(-0.899976223); p2 = p5 + p2 * f2f(-2.562915447); p3 = p3 * f2f(-1.961570560); p4 = p4 * f2f(-0.390180644); \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
    for (int i = 0; i < 8; ++i) {   // columns
        const int* c = d + i;
        long long* v = val + i;
        if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
            const long long dc = (long long)c[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
        } else {
            TR_IDCT_1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10; v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10; v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    for (int i = 0; i < 8; ++i) {   // rows
        const long long* v = val + i * 8;
        uint8_t* o = out + i * stride;
        TR_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128ll << 17); x1 += 65536 + (128ll << 17); x2 += 65536 + (128ll << 17); x3 += 65536 + (128ll << 17);
        o[0] = jpeg_clamp((x0 + t3) >> 17); o[7] = jpeg_clamp((x0 - t3) >> 17); o[1] = jpeg_clamp((x1 + t2) >> 17); o[6] = jpeg_clamp((x1 - t2) >> 17);
        o[2] = jpeg_clamp((x2 + t1) >> 17); o[5] = jpeg_clamp((x2 - t1) >> 17); o[3] = jpeg_clamp((x3 + t0) >> 17); o[4] = jpeg_clamp((x3 - t0) >> 17);
    }
#undef TR_IDCT_1D
}
inline void jpeg_up_h2(const uint8_t* in, int w, uint8_t* out) {   // one row, twice as wide
    if (w == 1) { out[0] = out[1] = in[0]; return; }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    for (int i = 1; i < w - 1; ++i) {
        const int s = 3 * in[i] + 2;
        out[2 * i] = (uint8_t)((s + in[i - 1]) >> 2);
        out[2 * i + 1] = (uint8_t)((s + in[i + 1]) >> 2);
    }
    out[2 * (w - 1)] = (uint8_t)((in[w - 1] * 3 + in[w - 2] + 2) >> 2);
    out[2 * (w - 1) + 1] = in[w - 1];
}
inline void jpeg_up_h2v2(const uint8_t* near, const uint8_t* far, int w, uint8_t* out) {   // one output row from its two source rows
    if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); return; }
    int t0 = 3 * near[0] + far[0], t1 = 3 * near[1] + far[1];
    out[0] = (uint8_t)((t0 + 2) >> 2);
    out[1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
    for (int i = 2; i < w; ++i) {
        const int tp = t0;
        t0 = t1; t1 = 3 * near[i] + far[i];
        out[2 * i - 2] = (uint8_t)((3 * t0 + tp + 8) >> 4);
        out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
    }
    out[2 * w - 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
    out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
}
inline bool decode_jpeg(const std::vector<uint8_t>& f, ImageRGBA8& out, std::string& err) {
    static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    uint16_t qt[4][64] = {};
    bool qt_present[4] = {false, false, false, false};
    JpegHuff dc[4], ac[4];
    std::vector<JpegComp> comps;
    int width = 0, height = 0, hmax = 1, vmax = 1, restart = 0, adobe_transform = -1;
    bool have_frame = false, progressive = false;
    size_t pos = 2;
    const size_t n = f.size();
    auto be16 = [&](size_t at) { return (int)f[at] << 8 | (int)f[at + 1]; };
    while (pos + 4 <= n) {
        if (f[pos] != 0xff) { err = "JPEG: marker expected"; return false; }
        while (pos < n && f[pos] == 0xff) ++pos;   // fill bytes
        if (pos >= n) break;
        const int m = f[pos++];
        if (m == 0xd9) break;                       // EOI
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
        if (pos + 2 > n) break;
        const int len = be16(pos);
        if (len < 2 || pos + (size_t)len > n) { err = "JPEG: truncated segment"; return false; }
        const size_t seg = pos + 2, end = pos + (size_t)len;
        if (m == 0xdb) {   // DQT
            size_t q = seg;
            while (q < end) {
                const int pq = f[q] >> 4, tq = f[q] & 15; ++q;
                if (tq > 3 || pq > 1 || q + (size_t)(pq ? 128 : 64) > end) { err = "JPEG: bad quantisation table"; return false; }
                for (int k = 0; k < 64; ++k) { qt[tq][k] = pq ? (uint16_t)be16(q + 2 * (size_t)k) : f[q + (size_t)k]; }
                q += pq ? 128 : 64;
                qt_present[tq] = true;
            }
        } else if (m == 0xc4) {   // DHT
            size_t q = seg;
            while (q < end) {
                const int tc = f[q] >> 4, th = f[q] & 15; ++q;
                if (tc > 1 || th > 3 || q + 16 > end) { err = "JPEG: bad Huffman table"; return false; }
                JpegHuff& h = tc ? ac[th] : dc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = f[q + (size_t)l - 1]; total += h.bits[l]; }
                q += 16;
                if (total > 256 || q + (size_t)total > end) { err = "JPEG: bad Huffman table"; return false; }
                for (int k = 0; k < total; ++k) h.vals[k] = f[q + (size_t)k];
                q += (size_t)total;
                jpeg_build(h);
            }
        } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {   // SOF0 / SOF1: sequential Huffman; SOF2: progressive Huffman
            progressive = m == 0xc2;
            if (len < 8 || f[seg] != 8) { err = "JPEG: only 8-bit samples are supported"; return false; }
            if (have_frame) { err = "JPEG: more than one frame header"; return false; }   // (a second SOF would meet the first one's sampling factors and planes)
            height = be16(seg + 1); width = be16(seg + 3);
            if ((size_t)width * (size_t)height > ((size_t)1 << 28)) { err = "JPEG: image larger than 2^28 pixels"; return false; }   // (planes are allocated from the header alone)
            // ... and a file of a few hundred bytes must not make them gigabytes: a coded 8 x 8 block takes at least a few
            // bits per component, i.e. well under 1024 pixels per byte of file even for a flat image
            if ((size_t)width * (size_t)height > std::max<size_t>((size_t)1 << 20, f.size() * 1024u)) { err = "JPEG: the frame header announces more pixels than a file of this size can hold"; return false; }
            const int nc = f[seg + 5];
            if ((nc != 1 && nc != 3) || width <= 0 || height <= 0 || seg + 6 + 3 * (size_t)nc > end) { err = "JPEG: unsupported frame (1 or 3 components)"; return false; }
            comps.assign((size_t)nc, JpegComp());
            for (int c = 0; c < nc; ++c) {
                JpegComp& k = comps[(size_t)c];
                k.id = f[seg + 6 + 3 * (size_t)c]; k.h = f[seg + 7 + 3 * (size_t)c] >> 4; k.v = f[seg + 7 + 3 * (size_t)c] & 15; k.tq = f[seg + 8 + 3 * (size_t)c];
                if (k.h < 1 || k.h > 2 || k.v < 1 || k.v > 2 || k.tq > 3) { err = "JPEG: unsupported sampling factors (1x1, 2x1, 1x2, 2x2)"; return false; }
                hmax = k.h > hmax ? k.h : hmax; vmax = k.v > vmax ? k.v : vmax;
            }
            if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }   // a single component is never subsampled
            const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            for (JpegComp& k : comps) { k.bw = mcux * k.h; k.bh = mcuy * k.v; k.plane.assign((size_t)k.bw * 8 * (size_t)k.bh * 8, 0); }
            if (progressive) for (JpegComp& k : comps) k.coef.assign((size_t)k.bw * (size_t)k.bh * 64, 0);
            have_frame = true;
        } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {
            err = "JPEG: lossless / hierarchical / arithmetic-coded files are not supported (baseline and progressive Huffman only)"; return false;
        } else if (m == 0xdd) {
            if (len >= 4) restart = be16(seg);
        } else if (m == 0xee && len >= 14 && !std::memcmp(&f[seg], "Adobe", 5)) {
            adobe_transform = f[seg + 11];
        } else if (m == 0xda) {   // SOS + entropy-coded data
            if (!have_frame) { err = "JPEG: scan before frame header"; return false; }
            if (len < 3 || seg >= end) { err = "JPEG: bad scan header"; return false; }   // (the component count is the first byte of the segment's body)
            const int ns = f[seg];
            if (ns < 1 || ns > (int)comps.size() || seg + 1 + 2 * (size_t)ns + 3 > end) { err = "JPEG: bad scan header"; return false; }
            // spectral selection Ss..Se and successive approximation Ah / Al (T.81 B.2.3; 0, 63, 0, 0 in a sequential scan)
            const int ss = f[seg + 1 + 2 * (size_t)ns], se = f[seg + 2 + 2 * (size_t)ns], ah = f[seg + 3 + 2 * (size_t)ns] >> 4, al = f[seg + 3 + 2 * (size_t)ns] & 15;
            if (progressive && (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13 || (ah != 0 && ah != al + 1))) { err = "JPEG: bad progressive scan parameters"; return false; }
            std::vector<JpegComp*> sc;
            for (int i = 0; i < ns; ++i) {
                const int cid = f[seg + 1 + 2 * (size_t)i], t = f[seg + 2 + 2 * (size_t)i];
                JpegComp* k = nullptr;
                for (JpegComp& c : comps) if (c.id == cid) k = &c;
                const bool need_dc = !progressive || ss == 0, need_ac = !progressive || ss > 0;
                if (!k || (t >> 4) > 3 || (t & 15) > 3 || (need_dc && ah == 0 && !dc[t >> 4].present) || (need_ac && !ac[t & 15].present) || !qt_present[k->tq]) { err = "JPEG: scan refers to a missing component / table"; return false; }
                k->td = t >> 4; k->ta = t & 15; k->dc_pred = 0;
                sc.push_back(k);
            }
            JpegBits br(f.data(), n, end);
            // a scan of one component covers ceil(w_c / 8) x ceil(h_c / 8) blocks (T.81 A.2.2), an interleaved one whole MCUs
            const bool single = ns == 1;
            int nx, ny;
            if (single) {
                const int cw = (width * sc[0]->h + hmax - 1) / hmax, ch = (height * sc[0]->v + vmax - 1) / vmax;
                nx = (cw + 7) / 8; ny = (ch + 7) / 8;
            } else { nx = comps[0].bw / comps[0].h; ny = comps[0].bh / comps[0].v; }
            int until_restart = restart, rst_expect = 0, eobrun = 0;
            for (int my = 0; my < ny; ++my)
                for (int mx = 0; mx < nx; ++mx) {
                    if (restart && until_restart == 0) {   // RSTn: byte-align, check the marker, reset the predictors
                        br.reset();
                        if (!br.marker) { while (br.pos + 1 < n && !(f[br.pos] == 0xff && f[br.pos + 1] >= 0xd0 && f[br.pos + 1] <= 0xd7)) ++br.pos; br.marker = br.pos + 1 < n ? f[br.pos + 1] : 0; }
                        if (br.marker != 0xd0 + rst_expect) { err = "JPEG: restart marker out of sequence"; return false; }
                        br.pos += 2; br.marker = 0; rst_expect = (rst_expect + 1) & 7; until_restart = restart;
                        for (JpegComp* k : sc) k->dc_pred = 0;
                        eobrun = 0;
                    }
                    for (JpegComp* k : sc) {
                        const int bh_ = single ? 1 : k->h, bv_ = single ? 1 : k->v;
                        for (int by = 0; by < bv_; ++by)
                            for (int bx = 0; bx < bh_; ++bx) {
                                const int px = (mx * bh_ + bx) * 8, py = (my * bv_ + by) * 8;
                                if (progressive) {
                                    if (px + 8 > k->bw * 8 || py + 8 > k->bh * 8) { err = "JPEG: block outside the frame"; return false; }
                                    int16_t* c = &k->coef[((size_t)(py / 8) * (size_t)k->bw + (size_t)(px / 8)) * 64];
                                    if (ss == 0) {   // DC scan (G.1.2.1): first pass = the difference, shifted; refinement = one more bit
                                        if (ah == 0) {
                                            const int s = jpeg_decode(br, dc[k->td]);
                                            if (s > 11) { err = "JPEG: bad DC category"; return false; }
                                            k->dc_pred += jpeg_extend(br.receive(s), s);
                                            if (k->dc_pred > 32767 || k->dc_pred < -32768) { err = "JPEG: DC coefficient out of range"; return false; }   // (12 bits in a valid file; keeps the products below in range)
                                            c[0] = (int16_t)((unsigned)k->dc_pred << al);
                                        } else if (br.bit()) c[0] = (int16_t)(c[0] | (1 << al));
                                    } else if (ah == 0) {   // AC first pass (G.1.2.2) with end-of-band runs
                                        if (eobrun > 0) --eobrun;
                                        else for (int i = ss; i <= se;) {
                                            const int rs = jpeg_decode(br, ac[k->ta]), r = rs >> 4, sz = rs & 15;
                                            if (sz == 0) {
                                                if (r == 15) { i += 16; continue; }
                                                eobrun = (1 << r) - 1;
                                                if (r) eobrun += br.receive(r);
                                                break;
                                            }
                                            i += r;
                                            if (i > se) { err = "JPEG: AC coefficient index out of range"; return false; }
                                            c[zigzag[i]] = (int16_t)(jpeg_extend(br.receive(sz), sz) * (1 << al));
                                            ++i;
                                        }
                                    } else {   // AC refinement (G.1.2.3): a correction bit for every coefficient that is already nonzero, new +-1 << al ones in between
                                        const int p1 = 1 << al;
                                        auto refine = [&](int16_t& v) { if (br.bit() && (v & p1) == 0) v = (int16_t)(v >= 0 ? v + p1 : v - p1); };
                                        int i = ss;
                                        if (eobrun == 0) {
                                            while (i <= se) {
                                                const int rs = jpeg_decode(br, ac[k->ta]), sz = rs & 15;
                                                int r = rs >> 4, value = 0;
                                                if (sz == 0) {
                                                    if (r < 15) { eobrun = 1 << r; if (r) eobrun += br.receive(r); break; }
                                                } else {
                                                    if (sz != 1) { err = "JPEG: bad refinement scan"; return false; }
                                                    value = br.bit() ? p1 : -p1;
                                                }
                                                while (i <= se) {   // r zero-valued coefficients are skipped, the nonzero ones in between refined
                                                    int16_t& v = c[zigzag[i++]];
                                                    if (v != 0) refine(v);
                                                    else { if (r == 0) { if (value) v = (int16_t)value; break; } --r; }
                                                }
                                                if (br.bad) break;
                                            }
                                        }
                                        if (eobrun > 0) {   // the rest of the band belongs to an end-of-band run: refinement bits only
                                            for (; i <= se; ++i) { int16_t& v = c[zigzag[i]]; if (v != 0) refine(v); }
                                            --eobrun;
                                        }
                                    }
                                    if (br.bad) { err = "JPEG: corrupt entropy-coded data"; return false; }
                                    continue;
                                }
                                int blk[64] = {0};
                                const int s = jpeg_decode(br, dc[k->td]);
                                if (s > 11) { err = "JPEG: bad DC category"; return false; }
                                k->dc_pred += jpeg_extend(br.receive(s), s);
                                if (k->dc_pred > 32767 || k->dc_pred < -32768) { err = "JPEG: DC coefficient out of range"; return false; }   // (12 bits in a valid file; keeps the product in range)
                                blk[0] = k->dc_pred * qt[k->tq][0];
                                for (int i = 1; i < 64;) {
                                    const int rs = jpeg_decode(br, ac[k->ta]), r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) { if (r == 15) { i += 16; continue; } break; }
                                    i += r;
                                    if (i > 63) { err = "JPEG: AC coefficient index out of range"; return false; }
                                    blk[zigzag[i]] = jpeg_extend(br.receive(sz), sz) * qt[k->tq][i];
                                    ++i;
                                }
                                if (br.bad) { err = "JPEG: corrupt entropy-coded data"; return false; }
                                if (px + 8 <= k->bw * 8 && py + 8 <= k->bh * 8) jpeg_idct(blk, &k->plane[(size_t)py * (size_t)k->bw * 8 + (size_t)px], k->bw * 8);
                            }
                    }
                    if (restart) --until_restart;
                }
            // continue behind the entropy-coded segment
            pos = br.pos;
            while (pos + 1 < n && !(f[pos] == 0xff && f[pos + 1] != 0 && !(f[pos + 1] >= 0xd0 && f[pos + 1] <= 0xd7))) ++pos;
            continue;
        }
        pos = end;
    }
    if (!have_frame) { err = "JPEG: no frame header"; return false; }
    if (progressive)   // every scan has contributed: dequantise (the table is indexed in zigzag order) and transform
        for (JpegComp& k : comps) {
            if (!qt_present[k.tq]) { err = "JPEG: missing quantisation table"; return false; }
            int qnat[64];
            for (int i = 0; i < 64; ++i) qnat[zigzag[i]] = qt[k.tq][i];
            for (int by = 0; by < k.bh; ++by)
                for (int bx = 0; bx < k.bw; ++bx) {
                    const int16_t* c = &k.coef[((size_t)by * (size_t)k.bw + (size_t)bx) * 64];
                    int blk[64];
                    for (int i = 0; i < 64; ++i) blk[i] = c[i] * qnat[i];
                    jpeg_idct(blk, &k.plane[(size_t)by * 8 * (size_t)k.bw * 8 + (size_t)bx * 8], k.bw * 8);
                }
        }
    // upsample every component to the frame and convert
    std::vector<std::vector<uint8_t>> full(comps.size());
    for (size_t c = 0; c < comps.size(); ++c) {
        const JpegComp& k = comps[c];
        const int cw = (width * k.h + hmax - 1) / hmax, ch = (height * k.v + vmax - 1) / vmax, stride = k.bw * 8;
        const int fx = hmax / k.h, fy = vmax / k.v;
        std::vector<uint8_t>& o = full[c];
        o.assign((size_t)width * (size_t)height, 0);
        std::vector<uint8_t> row((size_t)cw * 2 + 2);
        for (int y = 0; y < height; ++y) {
            const uint8_t* src;
            if (fx == 1 && fy == 1) src = &k.plane[(size_t)y * (size_t)stride];
            else if (fx == 2 && fy == 1) { jpeg_up_h2(&k.plane[(size_t)y * (size_t)stride], cw, row.data()); src = row.data(); }
            else {
                const int near = y / 2;
                int far = (y & 1) ? near + 1 : near - 1;
                far = far < 0 ? 0 : (far > ch - 1 ? ch - 1 : far);
                if (fx == 2) { jpeg_up_h2v2(&k.plane[(size_t)near * (size_t)stride], &k.plane[(size_t)far * (size_t)stride], cw, row.data()); src = row.data(); }
                else {   // 1 x 2: vertical triangle only
                    for (int x = 0; x < cw; ++x) row[(size_t)x] = (uint8_t)((3 * k.plane[(size_t)near * (size_t)stride + (size_t)x] + k.plane[(size_t)far * (size_t)stride + (size_t)x] + 2) >> 2);
                    src = row.data();
                }
            }
            std::memcpy(&o[(size_t)y * (size_t)width], src, (size_t)width);
        }
    }
    out.width = (uint32_t)width; out.height = (uint32_t)height;
    out.px.assign((size_t)width * (size_t)height * 4, 255);
    for (size_t i = 0; i < (size_t)width * (size_t)height; ++i) {
        uint8_t* o = &out.px[i * 4];
        if (comps.size() == 1) { o[0] = o[1] = o[2] = full[0][i]; }
        else if (adobe_transform == 0) { o[0] = full[0][i]; o[1] = full[1][i]; o[2] = full[2][i]; }   // Adobe APP14 transform 0: RGB stored as is
        else {
            const float y = (float)full[0][i], cb = (float)full[1][i] - 128.0f, cr = (float)full[2][i] - 128.0f;
            o[0] = jpeg_clamp((int)(y + 1.40200f * cr + 0.5f));
            o[1] = jpeg_clamp((int)(y - 0.34414f * cb - 0.71414f * cr + 0.5f));
            o[2] = jpeg_clamp((int)(y + 1.77200f * cb + 0.5f));
        }
    }
    return true;
}

// ---- GIF (87a / 89a): what image::open returns for a .gif -- the FIRST frame as RGBA (image 0.18 decodes GIF with the gif crate 0.9,
// Cargo.lock; not vendored by the reference: PARITY UNPINNED): palette lookup through the frame's local colour table or the global one,
// the graphic control extension's transparent index -> (r, g, b, 0), interlaced frames put back in row order, variable-width LZW
// (clear / end codes, 12-bit table, deferred clear). The image crate builds the picture from the screen's size and the frame's buffer,
// so a first frame that does not cover the logical screen is refused here (it is an error there).
inline bool decode_gif(const std::vector<uint8_t>& f, ImageRGBA8& out, std::string& err) {
    const size_t n = f.size();
    if (n < 13 || std::memcmp(f.data(), "GIF8", 4) || (f[4] != '7' && f[4] != '9') || f[5] != 'a') { err = "GIF: bad signature"; return false; }
    auto le16 = [&](size_t at) { return (int)f[at] | (int)f[at + 1] << 8; };
    const int sw = le16(6), sh = le16(8);
    size_t pos = 13;
    std::vector<uint8_t> global;
    if (f[10] & 0x80) { const size_t sz = (size_t)3 << ((f[10] & 7) + 1); if (pos + sz > n) { err = "GIF: truncated colour table"; return false; } global.assign(f.begin() + (long)pos, f.begin() + (long)(pos + sz)); pos += sz; }
    int transparent = -1;
    while (pos < n) {
        const int b = f[pos++];
        if (b == 0x3b) break;                         // trailer
        if (b == 0x21) {                              // extension: label, sub-blocks
            if (pos >= n) break;
            const int label = f[pos++];
            if (label == 0xf9 && pos + 5 < n && f[pos] == 4 && (f[pos + 1] & 1)) transparent = f[pos + 4];   // graphic control: transparent colour flag + index
            while (pos < n && f[pos] != 0) pos += (size_t)f[pos] + 1;
            ++pos;
            continue;
        }
        if (b != 0x2c) { err = "GIF: unexpected block"; return false; }
        if (pos + 9 > n) { err = "GIF: truncated image descriptor"; return false; }
        const int left = le16(pos), top = le16(pos + 2), w = le16(pos + 4), h = le16(pos + 6), flags = f[pos + 8];
        pos += 9;
        if (w <= 0 || h <= 0 || (size_t)w * (size_t)h > ((size_t)1 << 28)) { err = "GIF: bad frame size"; return false; }
        if (left != 0 || top != 0 || w != sw || h != sh) { err = "GIF: the first frame does not cover the logical screen"; return false; }
        std::vector<uint8_t> local;
        if (flags & 0x80) { const size_t sz = (size_t)3 << ((flags & 7) + 1); if (pos + sz > n) { err = "GIF: truncated colour table"; return false; } local.assign(f.begin() + (long)pos, f.begin() + (long)(pos + sz)); pos += sz; }
        const std::vector<uint8_t>& pal = local.empty() ? global : local;
        if (pal.empty()) { err = "GIF: no colour table"; return false; }
        if (pos >= n) { err = "GIF: truncated image data"; return false; }
        const int min_code = f[pos++];
        if (min_code < 2 || min_code > 8) { err = "GIF: bad LZW minimum code size"; return false; }
        std::vector<uint8_t> data;                    // the sub-blocks, concatenated
        while (pos < n && f[pos] != 0) { const size_t len = f[pos]; if (pos + 1 + len > n) { err = "GIF: truncated image data"; return false; } data.insert(data.end(), f.begin() + (long)pos + 1, f.begin() + (long)(pos + 1 + len)); pos += len + 1; }
        // LZW, least significant bit first
        std::vector<uint8_t> idx;
        idx.reserve((size_t)w * (size_t)h);
        const int clear = 1 << min_code, end_code = clear + 1;
        std::vector<int16_t> prefix(4096, -1);
        std::vector<uint8_t> suffix(4096, 0), stack(4097);
        for (int i = 0; i < clear; ++i) suffix[(size_t)i] = (uint8_t)i;
        int width = min_code + 1, next = end_code + 1, prev = -1;
        uint32_t acc = 0; int nbits = 0; size_t dp = 0;
        while (idx.size() < (size_t)w * (size_t)h) {
            while (nbits < width && dp < data.size()) { acc |= (uint32_t)data[dp++] << nbits; nbits += 8; }
            if (nbits < width) break;                 // data ran out: the rest of the frame stays at index 0
            const int code = (int)(acc & ((1u << width) - 1u)); acc >>= width; nbits -= width;
            if (code == clear) { width = min_code + 1; next = end_code + 1; prev = -1; continue; }
            if (code == end_code) break;
            if (prev < 0) { if (code >= clear) { err = "GIF: corrupt LZW stream"; return false; } idx.push_back((uint8_t)code); prev = code; continue; }
            int cur = code, sp = 0;
            if (code > next || (code == next && next >= 4096)) { err = "GIF: corrupt LZW stream"; return false; }
            if (code == next) { cur = prev; }         // the string being defined: prev + first(prev)
            while (cur >= clear) { if (cur == end_code || sp >= 4096) { err = "GIF: corrupt LZW stream"; return false; } stack[(size_t)sp++] = suffix[(size_t)cur]; cur = prefix[(size_t)cur]; }
            const uint8_t first = (uint8_t)cur;
            stack[(size_t)sp++] = first;
            while (sp > 0) idx.push_back(stack[(size_t)--sp]);
            if (code == next) idx.push_back(first);
            if (next < 4096) {
                prefix[(size_t)next] = (int16_t)prev; suffix[(size_t)next] = first; ++next;
                if (next == (1 << width) && width < 12) ++width;
            }
            prev = code;
        }
        if (idx.size() < (size_t)w * (size_t)h) { err = "GIF: image data ends before the frame is complete"; return false; }   // (image::open fails on a truncated frame; round 4 padded with index 0)
        idx.resize((size_t)w * (size_t)h, 0);
        out.width = (uint32_t)w; out.height = (uint32_t)h;
        out.px.assign((size_t)w * (size_t)h * 4, 0);
        // interlaced frames store rows 0, 8, 16, ... then 4, 12, ... then 2, 6, ... then 1, 3, ...
        std::vector<int> row_of((size_t)h);
        if (flags & 0x40) { int r = 0; const int start[4] = {0, 4, 2, 1}, step[4] = {8, 8, 4, 2}; for (int p = 0; p < 4; ++p) for (int y = start[p]; y < h; y += step[p]) row_of[(size_t)r++] = y; }
        else for (int y = 0; y < h; ++y) row_of[(size_t)y] = y;
        for (int r = 0; r < h; ++r)
            for (int x = 0; x < w; ++x) {
                const int i = idx[(size_t)r * (size_t)w + (size_t)x];
                uint8_t* o = &out.px[((size_t)row_of[(size_t)r] * (size_t)w + (size_t)x) * 4];
                if ((size_t)i * 3 + 2 < pal.size()) { o[0] = pal[(size_t)i * 3]; o[1] = pal[(size_t)i * 3 + 1]; o[2] = pal[(size_t)i * 3 + 2]; }
                o[3] = i == transparent ? 0 : 255;
            }
        return true;
    }
    err = "GIF: no image";
    return false;
}


// ---- TIFF (baseline, strips). image 0.18 carries its own tiff decoder (not vendored by the reference: PARITY UNPINNED): grey / RGB(A) strips of 8 or 16 bits,
// uncompressed, LZW or PackBits, horizontal predictor. Restated from the TIFF 6.0 specification; accepted beyond that crate: palette images, grey with
// 1 / 4 bits. Tiles, planar-separate samples, JPEG / deflate compression are refused with a message. 16-bit samples keep their high byte (as PNG above).
inline bool tiff_lzw(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {   // TIFF 6.0 section 13: MSB-first codes, 9 - 12 bits, "early change"
    std::vector<uint16_t> prefix(4096, 0);
    std::vector<uint8_t> suffix(4096, 0), tmp;
    for (int i = 0; i < 256; ++i) suffix[i] = (uint8_t)i;
    uint32_t acc = 0; int have = 0, width = 9; uint32_t next = 258; int old = -1;
    size_t pos = 0;
    auto first_of = [&](int c) { while (c >= 258) c = prefix[c]; return suffix[c]; };
    auto emit = [&](int c) {
        tmp.clear();
        while (c >= 258) { tmp.push_back(suffix[c]); c = prefix[c]; if (tmp.size() > 4096) return false; }
        tmp.push_back(suffix[c]);
        out.insert(out.end(), tmp.rbegin(), tmp.rend());
        return true;
    };
    while (out.size() < expect) {
        while (have < width) { if (pos >= n) return out.size() >= expect; acc = (acc << 8) | src[pos++]; have += 8; }
        const int code = (int)((acc >> (have - width)) & ((1u << width) - 1u));
        have -= width;
        if (code == 257) break;
        if (code == 256) { width = 9; next = 258; old = -1; continue; }
        if (old < 0) { if (code >= 256) return false; out.push_back((uint8_t)code); old = code; continue; }
        if ((uint32_t)code < next) {
            if (code >= 256 && code < 258) return false;
            if (!emit(code)) return false;
            if (next < 4096) { prefix[next] = (uint16_t)old; suffix[next] = first_of(code); ++next; }
        } else if ((uint32_t)code == next && next < 4096) {
            prefix[next] = (uint16_t)old; suffix[next] = first_of(old); ++next;
            if (!emit(code)) return false;
        } else return false;
        old = code;
        if (next + 1 >= (1u << width) && width < 12) ++width;
    }
    return true;
}

inline bool decode_tiff(const std::vector<uint8_t>& f, ImageRGBA8& out, std::string& err) {
    if (f.size() < 8) { err = "truncated TIFF file"; return false; }
    const bool le = f[0] == 'I';
    bool bad = false;
    auto u16 = [&](size_t o) -> uint32_t { if (o + 2 > f.size()) { bad = true; return 0; } return le ? (uint32_t)(f[o] | f[o + 1] << 8) : (uint32_t)(f[o] << 8 | f[o + 1]); };
    auto u32 = [&](size_t o) -> uint32_t { if (o + 4 > f.size()) { bad = true; return 0; }
        return le ? (uint32_t)f[o] | (uint32_t)f[o + 1] << 8 | (uint32_t)f[o + 2] << 16 | (uint32_t)f[o + 3] << 24
                  : (uint32_t)f[o] << 24 | (uint32_t)f[o + 1] << 16 | (uint32_t)f[o + 2] << 8 | (uint32_t)f[o + 3]; };
    if (u16(2) != 42) { err = "not a TIFF file (BigTIFF is not supported)"; return false; }
    const size_t ifd = u32(4);
    const uint32_t n_entries = u16(ifd);
    if (bad || ifd + 2 + (size_t)n_entries * 12 > f.size()) { err = "truncated TIFF directory"; return false; }
    struct Entry { uint32_t type = 0, count = 0; size_t at = 0; bool present = false; };
    auto find = [&](uint32_t tag) {
        Entry e;
        for (uint32_t i = 0; i < n_entries; ++i) {
            const size_t o = ifd + 2 + (size_t)i * 12;
            if (u16(o) != tag) continue;
            e.type = u16(o + 2); e.count = u32(o + 4);
            const size_t size = e.type == 3 ? 2 : (e.type == 4 ? 4 : 1);
            e.at = (size * e.count <= 4) ? o + 8 : (size_t)u32(o + 8);
            e.present = e.type == 1 || e.type == 3 || e.type == 4;
            break;
        }
        return e;
    };
    auto value = [&](const Entry& e, uint32_t k) -> uint32_t { return e.type == 3 ? u16(e.at + 2 * (size_t)k) : (e.type == 4 ? u32(e.at + 4 * (size_t)k) : (e.at + k < f.size() ? f[e.at + k] : (bad = true, 0u))); };
    auto scalar = [&](uint32_t tag, uint32_t dflt) { const Entry e = find(tag); return e.present && e.count ? value(e, 0) : dflt; };
    const uint32_t w = scalar(256, 0), h = scalar(257, 0), comp = scalar(259, 1), photo = scalar(262, 0xffffu), spp = scalar(277, 1);
    const uint32_t rows_per_strip = std::max(1u, std::min(scalar(278, h), h)), planar = scalar(284, 1), predictor = scalar(317, 1);
    const Entry bps = find(258), offs = find(273), counts = find(279), cmap = find(320);
    if (find(322).present || find(324).present) { err = "tiled TIFF files are not supported"; return false; }
    uint32_t bits = 1;
    if (bps.present) { bits = value(bps, 0); for (uint32_t k = 1; k < bps.count && k < spp; ++k) if (value(bps, k) != bits) { err = "TIFF: samples of different widths are not supported"; return false; } }
    if (bad || w == 0 || h == 0 || w > 32768 || h > 32768 || !offs.present) { err = "TIFF without usable dimensions / strips"; return false; }
    if ((uint64_t)w * h > std::max<uint64_t>(1u << 20, (uint64_t)f.size() * 1024u)) { err = "TIFF dimensions out of proportion to the file size"; return false; }   // (no allocation from a forged header)
    if (comp != 1 && comp != 5 && comp != 32773) { err = "TIFF compression " + std::to_string(comp) + " is not supported (uncompressed, LZW, PackBits)"; return false; }
    if (planar != 1 && spp > 1) { err = "TIFF with separate sample planes is not supported"; return false; }
    const bool grey = photo == 0 || photo == 1, rgb = photo == 2, pal = photo == 3;
    const bool shape_ok = (grey && (spp == 1 || spp == 2) && (bits == 8 || bits == 16 || (spp == 1 && (bits == 1 || bits == 4)))) ||
                          (rgb && (spp == 3 || spp == 4) && (bits == 8 || bits == 16)) ||
                          (pal && spp == 1 && (bits == 4 || bits == 8) && cmap.present && cmap.type == 3 && cmap.count >= 3u << bits);
    if (!shape_ok) { err = "unsupported TIFF photometric interpretation / sample layout"; return false; }
    if (predictor != 1 && (predictor != 2 || bits < 8)) { err = "unsupported TIFF predictor"; return false; }
    const size_t row_bytes = ((size_t)w * spp * bits + 7) / 8;
    const uint32_t n_strips = (h + rows_per_strip - 1) / rows_per_strip;
    if (offs.count < n_strips || (counts.present && counts.count < n_strips)) { err = "TIFF strip tables are shorter than the image"; return false; }
    out.width = w; out.height = h;
    out.px.assign((size_t)w * h * 4, 255);
    std::vector<uint8_t> raw;
    for (uint32_t s = 0; s < n_strips; ++s) {
        const uint32_t y0 = s * rows_per_strip, rows = std::min(rows_per_strip, h - y0);
        const size_t expect = row_bytes * rows, at = value(offs, s);
        size_t len = counts.present ? (size_t)value(counts, s) : (comp == 1 ? expect : f.size() - std::min(f.size(), at));
        if (bad || at > f.size() || len > f.size() - at) { err = "TIFF strip outside the file"; return false; }
        raw.clear();
        if (comp == 1) { if (len < expect) { err = "truncated TIFF strip"; return false; } raw.assign(f.begin() + (long)at, f.begin() + (long)(at + expect)); }
        else if (comp == 5) { raw.reserve(expect); if (!tiff_lzw(&f[at], len, raw, expect) || raw.size() < expect) { err = "corrupt LZW data in a TIFF strip"; return false; } }
        else {   // PackBits
            size_t p = at; const size_t end = at + len;
            while (raw.size() < expect && p < end) {
                const int nb = (int8_t)f[p++];
                if (nb >= 0) { if (p + (size_t)nb + 1 > end) break; raw.insert(raw.end(), f.begin() + (long)p, f.begin() + (long)(p + (size_t)nb + 1)); p += (size_t)nb + 1; }
                else if (nb != -128) { if (p >= end) break; raw.insert(raw.end(), (size_t)(1 - nb), f[p++]); }
            }
            if (raw.size() < expect) { err = "truncated PackBits data in a TIFF strip"; return false; }
        }
        for (uint32_t r = 0; r < rows; ++r) {
            uint8_t* line = &raw[row_bytes * r];
            if (predictor == 2) {
                if (bits == 8) for (size_t i = spp; i < (size_t)w * spp; ++i) line[i] = (uint8_t)(line[i] + line[i - spp]);
                else for (size_t i = spp; i < (size_t)w * spp; ++i) {
                    const size_t a = 2 * i, b = 2 * (i - spp);
                    const uint32_t cur = le ? (uint32_t)(line[a] | line[a + 1] << 8) : (uint32_t)(line[a] << 8 | line[a + 1]);
                    const uint32_t prv = le ? (uint32_t)(line[b] | line[b + 1] << 8) : (uint32_t)(line[b] << 8 | line[b + 1]);
                    const uint32_t v = (cur + prv) & 0xffffu;
                    if (le) { line[a] = (uint8_t)v; line[a + 1] = (uint8_t)(v >> 8); } else { line[a] = (uint8_t)(v >> 8); line[a + 1] = (uint8_t)v; }
                }
            }
            uint8_t* o = &out.px[(size_t)(y0 + r) * w * 4];
            for (uint32_t x = 0; x < w; ++x) {
                uint32_t sm[4] = {0, 0, 0, 255};
                for (uint32_t ch = 0; ch < spp; ++ch) {
                    if (bits == 8) sm[ch] = line[(size_t)x * spp + ch];
                    else if (bits == 16) sm[ch] = line[((size_t)x * spp + ch) * 2 + (le ? 1 : 0)];   // high byte
                    else { const size_t bit = (size_t)x * bits; sm[ch] = (line[bit / 8] >> (8 - bits - bit % 8)) & ((1u << bits) - 1u); }
                }
                if (pal) {
                    const uint32_t n = 1u << bits;
                    o[4 * x] = (uint8_t)(value(cmap, sm[0]) >> 8); o[4 * x + 1] = (uint8_t)(value(cmap, n + sm[0]) >> 8); o[4 * x + 2] = (uint8_t)(value(cmap, 2 * n + sm[0]) >> 8);
                } else if (grey) {
                    uint32_t l = sm[0];
                    if (bits < 8) l = l * 255u / ((1u << bits) - 1u);
                    if (photo == 0) l = 255u - l;   // WhiteIsZero
                    o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = (uint8_t)l;
                    if (spp == 2) o[4 * x + 3] = (uint8_t)sm[1];
                } else {
                    o[4 * x] = (uint8_t)sm[0]; o[4 * x + 1] = (uint8_t)sm[1]; o[4 * x + 2] = (uint8_t)sm[2];
                    if (spp == 4) o[4 * x + 3] = (uint8_t)sm[3];
                }
            }
        }
    }
    if (bad) { err = "TIFF tables outside the file"; return false; }
    return true;
}

// ---- Radiance HDR (.hdr / .pic). image 0.18 opens these through its HDRAdapter, which hands out 8-bit RGB: RGBE -> f32 (mantissa * 2^(e - 136), e = 0 is
// black), then `powf(v, 2.2) * 255 + 0.5`, clamped and truncated (RGBE8Pixel::to_ldr = to_ldr_scale_gamma(1.0, 2.2), as that crate wrote it; restated from
// the crate's published source, PARITY UNPINNED). Scanlines: the new run-length form (2, 2, width; four channel planes) or flat pixels; "-Y h +X w" only.
inline bool decode_hdr(const std::vector<uint8_t>& f, ImageRGBA8& out, std::string& err) {
    size_t pos = 0;
    auto line = [&](std::string& l) { l.clear(); while (pos < f.size() && f[pos] != '\n') l.push_back((char)f[pos++]); if (pos >= f.size()) return false; ++pos; return true; };
    std::string l;
    if (!line(l) || (l != "#?RADIANCE" && l != "#?RGBE")) { err = "not a Radiance HDR file"; return false; }
    for (;;) {
        if (!line(l)) { err = "truncated HDR header"; return false; }
        if (l.empty()) break;
        if (l.rfind("FORMAT=", 0) == 0 && l != "FORMAT=32-bit_rle_rgbe") { err = "the HDR file's pixel format is not supported (32-bit_rle_rgbe only)"; return false; }
    }
    if (!line(l)) { err = "HDR file without a resolution line"; return false; }
    unsigned long hh = 0, ww = 0;
    if (std::sscanf(l.c_str(), "-Y %lu +X %lu", &hh, &ww) != 2 || ww == 0 || hh == 0 || ww > 32768 || hh > 32768) { err = "unsupported HDR resolution line (only -Y h +X w)"; return false; }
    const uint32_t w = (uint32_t)ww, h = (uint32_t)hh;
    if ((uint64_t)w * h > std::max<uint64_t>(1u << 20, (uint64_t)f.size() * 1024u)) { err = "HDR dimensions out of proportion to the file size"; return false; }
    out.width = w; out.height = h;
    out.px.assign((size_t)w * h * 4, 255);
    std::vector<uint8_t> scan((size_t)w * 4);
    auto ldr = [](float v) { const float fv = std::pow(v, 2.2f) * 255.0f + 0.5f; return (uint8_t)(fv < 0.0f ? 0.0f : (fv > 255.0f ? 255.0f : fv)); };
    for (uint32_t y = 0; y < h; ++y) {
        if (pos + 4 > f.size()) { err = "truncated HDR scanline"; return false; }
        if (w >= 8 && w <= 32767 && f[pos] == 2 && f[pos + 1] == 2 && (f[pos + 2] & 0x80) == 0 && ((uint32_t)f[pos + 2] << 8 | f[pos + 3]) == w) {
            pos += 4;
            for (int ch = 0; ch < 4; ++ch) {
                uint32_t x = 0;
                while (x < w) {
                    if (pos >= f.size()) { err = "truncated HDR scanline"; return false; }
                    uint32_t c = f[pos++];
                    if (c > 128) {
                        c -= 128;
                        if (pos >= f.size() || x + c > w) { err = "corrupt HDR run"; return false; }
                        const uint8_t v = f[pos++];
                        for (uint32_t k = 0; k < c; ++k) scan[(size_t)(x + k) * 4 + ch] = v;
                    } else {
                        if (c == 0 || pos + c > f.size() || x + c > w) { err = "corrupt HDR run"; return false; }
                        for (uint32_t k = 0; k < c; ++k) scan[(size_t)(x + k) * 4 + ch] = f[pos++];
                    }
                    x += c;
                }
            }
        } else {
            if (pos + (size_t)w * 4 > f.size()) { err = "truncated HDR scanline"; return false; }
            std::memcpy(scan.data(), &f[pos], (size_t)w * 4);
            pos += (size_t)w * 4;
        }
        uint8_t* o = &out.px[(size_t)y * w * 4];
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t* p = &scan[(size_t)x * 4];
            const float scale = p[3] == 0 ? 0.0f : std::ldexp(1.0f, (int)p[3] - 136);
            o[4 * x] = ldr(scale * (float)p[0]); o[4 * x + 1] = ldr(scale * (float)p[1]); o[4 * x + 2] = ldr(scale * (float)p[2]);
        }
    }
    return true;
}

// ---- ICO: the directory's best entry as image 0.18 picks it (ico/decoder.rs best_entry: the highest (bits per pixel, width x height), the LAST entry
// among equals only if nothing before it is strictly better), its payload a PNG file or a BMP without the file header (height doubled: colour rows, then
// the 1-bit AND mask; a set mask bit makes the pixel transparent). PARITY UNPINNED.
inline bool decode_ico(const std::vector<uint8_t>& f, ImageRGBA8& out, std::string& err) {
    auto le16 = [&](size_t o) { return (uint32_t)(f[o] | f[o + 1] << 8); };
    auto le32 = [&](size_t o) { return (uint32_t)f[o] | (uint32_t)f[o + 1] << 8 | (uint32_t)f[o + 2] << 16 | (uint32_t)f[o + 3] << 24; };
    if (f.size() < 6 || le16(0) != 0 || le16(2) != 1) { err = "not an ICO file"; return false; }
    const uint32_t n = le16(4);
    if (n == 0 || 6 + (size_t)n * 16 > f.size()) { err = "ICO file without a usable directory"; return false; }
    auto score = [&](uint32_t i) { const size_t o = 6 + (size_t)i * 16; const uint64_t ew = f[o] ? f[o] : 256u, eh = f[o + 1] ? f[o + 1] : 256u; return ((uint64_t)le16(o + 6) << 32) | (ew * eh); };
    uint32_t best = n - 1;
    for (uint32_t i = 0; i + 1 < n; ++i) if (score(i) > score(best)) best = i;
    const size_t eo = 6 + (size_t)best * 16, size = le32(eo + 8), at = le32(eo + 12);
    if (at > f.size() || size > f.size() - at || size < 40) { err = "ICO entry outside the file"; return false; }
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (!std::memcmp(&f[at], png_sig, 8)) { const std::vector<uint8_t> sub(f.begin() + (long)at, f.begin() + (long)(at + size)); return decode_png(sub, out, err); }
    const uint32_t hdr = le32(at), w = le32(at + 4), h2 = le32(at + 8), bpp = le16(at + 14), comp = le32(at + 16), used = le32(at + 32);
    const uint32_t h = h2 / 2;
    if (hdr < 40 || comp != 0 || w == 0 || h == 0 || w > 256 || h > 256 || (bpp != 1 && bpp != 4 && bpp != 8 && bpp != 24 && bpp != 32)) { err = "unsupported ICO bitmap"; return false; }
    const uint32_t n_pal = bpp <= 8 ? (used ? used : 1u << bpp) : 0u;
    const size_t pal_at = at + hdr, xor_at = pal_at + (size_t)n_pal * 4, stride = ((size_t)w * bpp + 31) / 32 * 4, and_stride = ((size_t)w + 31) / 32 * 4, and_at = xor_at + stride * h;
    if (n_pal > 256 || xor_at + stride * h > at + size) { err = "truncated ICO bitmap"; return false; }
    const bool has_mask = and_at + and_stride * h <= at + size;
    out.width = w; out.height = h;
    out.px.assign((size_t)w * h * 4, 255);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = &f[xor_at + stride * (h - 1 - y)];
        const uint8_t* mrow = has_mask ? &f[and_at + and_stride * (h - 1 - y)] : nullptr;
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t* o = &out.px[((size_t)y * w + x) * 4];
            if (bpp >= 24) { const uint8_t* p = row + (size_t)x * bpp / 8; o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = bpp == 32 ? p[3] : 255; }
            else {
                const size_t bit = (size_t)x * bpp;
                const uint32_t idx = (row[bit / 8] >> (8 - bpp - bit % 8)) & ((1u << bpp) - 1u);
                if (idx >= n_pal) { err = "ICO palette index out of range"; return false; }
                const uint8_t* p = &f[pal_at + (size_t)idx * 4];
                o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = 255;
            }
            if (mrow && (mrow[x / 8] >> (7 - x % 8) & 1)) o[3] = 0;
        }
    }
    return true;
}

}  // namespace img_detail

// image::open: the format is taken from the file's content
inline bool load_image(const std::string& path, ImageRGBA8& out, std::string& err) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { err = "cannot open " + path; return false; }
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    bool ok;
    const size_t dot = path.rfind('.');
    if (f.size() >= 8 && !std::memcmp(f.data(), png_sig, 8)) ok = img_detail::decode_png(f, out, err);
    else if (f.size() >= 3 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) ok = img_detail::decode_pnm(f, out, err);
    else if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M') ok = img_detail::decode_bmp(f, out, err);
    else if (f.size() >= 3 && f[0] == 0xff && f[1] == 0xd8) ok = img_detail::decode_jpeg(f, out, err);
    else if (f.size() >= 6 && !std::memcmp(f.data(), "GIF8", 4)) ok = img_detail::decode_gif(f, out, err);
    else if (f.size() >= 4 && ((f[0] == 'I' && f[1] == 'I' && f[2] == 42 && f[3] == 0) || (f[0] == 'M' && f[1] == 'M' && f[2] == 0 && f[3] == 42))) ok = img_detail::decode_tiff(f, out, err);
    else if (f.size() >= 10 && (!std::memcmp(f.data(), "#?RADIANCE", 10) || !std::memcmp(f.data(), "#?RGBE", 6))) ok = img_detail::decode_hdr(f, out, err);
    else if (f.size() >= 12 && !std::memcmp(f.data(), "RIFF", 4) && !std::memcmp(f.data() + 8, "WEBP", 4)) {   // image 0.18: ColorType::Gray(8) of the luma plane
        std::vector<uint8_t> luma;
        ok = decode_webp_luma(f, out.width, out.height, luma, err);
        if (ok) {
            out.px.resize(luma.size() * 4);
            for (size_t i = 0; i < luma.size(); ++i) { uint8_t* o = &out.px[4 * i]; o[0] = o[1] = o[2] = luma[i]; o[3] = 255; }
        }
    }
    else if (dot != std::string::npos && (path.substr(dot) == ".ico" || path.substr(dot) == ".ICO")) ok = img_detail::decode_ico(f, out, err);
    else if (dot != std::string::npos && (path.substr(dot) == ".tga" || path.substr(dot) == ".TGA")) ok = img_detail::decode_tga(f, out, err);
    else { ok = false; err = "unrecognised image format (PNG, JPEG, GIF, TIFF, Radiance HDR, ICO, WebP, binary PPM / PGM, BMP and TGA are supported)"; }
    if (!ok) err = path + ": " + err;   // (which file: a scene names dozens of textures)
    return ok;
}

}  // namespace trayh
