// Image files for the `image` / `animated_image` / `movie` textures of the scene format (scene.rs:317-394 calls image::open of
// the `image` crate 0.18, not vendored by the reference): decoded to RGBA8 in the layout DynamicImage::get_pixel presents --
// row 0 on top, grey expanded to (l, l, l, 255), RGB given alpha 255 (texture/image.rs:18-32 reads px.data[0..4]).
// Formats: PNG (non-interlaced; grey / grey+alpha / RGB / RGBA / palette with tRNS; 1-16 bits, 16-bit samples keep their high
// byte), binary PPM / PGM (P6 / P5, maxval <= 255), BMP (uncompressed 24 / 32 bit), TGA (uncompressed true colour 24 / 32 bit and
// grey 8 bit). No third-party code: the inflate below is the textbook RFC 1951 decoder.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

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

}  // namespace img_detail

// image::open: the format is taken from the file's content
inline bool load_image(const std::string& path, ImageRGBA8& out, std::string& err) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { err = "cannot open " + path; return false; }
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() >= 8 && !std::memcmp(f.data(), png_sig, 8)) return img_detail::decode_png(f, out, err);
    if (f.size() >= 3 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) return img_detail::decode_pnm(f, out, err);
    if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M') return img_detail::decode_bmp(f, out, err);
    if (f.size() >= 3 && f[0] == 0xff && f[1] == 0xd8) { err = "JPEG files are not supported by this loader (convert to PNG)"; return false; }
    const size_t dot = path.rfind('.');
    if (dot != std::string::npos && (path.substr(dot) == ".tga" || path.substr(dot) == ".TGA")) return img_detail::decode_tga(f, out, err);
    err = "unrecognised image format (PNG, binary PPM / PGM, BMP and TGA are supported)";
    return false;
}

}  // namespace trayh
