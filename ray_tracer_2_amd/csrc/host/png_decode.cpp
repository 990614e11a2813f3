// png_decode.cpp -- minimal PNG -> RGBA8 decoder on zlib's inflate, standing
// in for `image::load_from_memory(..)` + the RGBA8 view that
// `imageops::flip_horizontal` takes of it (src/core/asset.rs:77; image 0.25.8
// / png 0.18.0 are not vendored).  PNG decoding is lossless, so any correct
// decoder yields the same texels.  Supports bit depths 1-16, colour types
// 0/2/3/4/6, tRNS, non-interlaced and Adam7 images.  16-bit samples are
// reduced like image's u16 -> u8 conversion: (v + 128) / 257 ... which equals
// rounding v / 257.
#include <zlib.h>

#include <cstring>
#include <fstream>
#include <sstream>

#include "scene.h"

namespace rt2 {
namespace {

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

struct Hdr {
    uint32_t w, h;
    int depth, ctype, interlace;
    int channels() const { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }
    int bpp_bits() const { return channels() * depth; }
};

// Unfilter one pass of width pw, height ph starting at src; returns bytes used.
bool unfilter(const uint8_t* src, size_t avail, uint32_t pw, uint32_t ph, int bits,
              std::vector<uint8_t>& out, size_t& used) {
    size_t stride = ((size_t)pw * bits + 7) / 8;
    size_t bpp = (size_t)(bits + 7) / 8;
    if (bpp < 1) bpp = 1;
    out.assign(stride * ph, 0);
    used = 0;
    for (uint32_t y = 0; y < ph; ++y) {
        if (used + 1 + stride > avail) return false;
        int ft = src[used];
        const uint8_t* in = src + used + 1;
        uint8_t* cur = out.data() + (size_t)y * stride;
        const uint8_t* prev = y ? cur - stride : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= bpp ? cur[i - bpp] : 0;
            int b = prev ? prev[i] : 0;
            int c = (prev && i >= bpp) ? prev[i - bpp] : 0;
            int v = in[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: v += paeth(a, b, c); break;
                default: return false;
            }
            cur[i] = (uint8_t)v;
        }
        used += 1 + stride;
    }
    return true;
}

uint8_t to8(uint32_t v, int depth) {
    switch (depth) {
        case 1: return v ? 255 : 0;
        case 2: return (uint8_t)(v * 85);
        case 4: return (uint8_t)(v * 17);
        case 8: return (uint8_t)v;
        default: return (uint8_t)((v + 128) / 257);
    }
}

}  // namespace

static bool decode_png_file_impl(const std::string& path, Image& out);

// (never throws: a file that cannot be decoded -- whatever the reason, an allocation that fails included -- is `false`)
bool decode_png_file(const std::string& path, Image& out) {
    try {
        return decode_png_file_impl(path, out);
    } catch (const std::exception&) {
        out = Image();
        return false;
    }
}

static bool decode_png_file_impl(const std::string& path, Image& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    std::string buf = ss.str();
    const uint8_t* p = (const uint8_t*)buf.data();
    size_t n = buf.size();
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (n < 8 || memcmp(p, sig, 8)) return false;
    Hdr h{};
    bool have_hdr = false;
    std::vector<uint8_t> idat, plte, trns;
    size_t off = 8;
    while (off + 12 <= n) {
        uint32_t len = be32(p + off);
        const uint8_t* type = p + off + 4;
        const uint8_t* data = p + off + 8;
        if (off + 12 + (size_t)len > n) return false;
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) return false;
            h.w = be32(data);
            h.h = be32(data + 4);
            h.depth = data[8];
            h.ctype = data[9];
            h.interlace = data[12];
            have_hdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!memcmp(type, "tRNS", 4)) {
            trns.assign(data, data + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        off += 12 + (size_t)len;
    }
    if (!have_hdr || h.w == 0 || h.h == 0 || idat.empty()) return false;
    if (h.ctype != 0 && h.ctype != 2 && h.ctype != 3 && h.ctype != 4 && h.ctype != 6) return false;
    // the bit depths PNG allows per colour type (anything else would index samples with undefined shifts below)
    {
        const int d = h.depth;
        const bool ok = h.ctype == 0 ? (d == 1 || d == 2 || d == 4 || d == 8 || d == 16)
                      : h.ctype == 3 ? (d == 1 || d == 2 || d == 4 || d == 8)
                                     : (d == 8 || d == 16);
        if (!ok || (h.interlace != 0 && h.interlace != 1)) return false;
    }
    // A header may claim any size: refuse what would decode to more than 512 MiB of RGBA8 (the default allocation limit
    // of the `image` crate the reference decodes with, image 0.25: Limits::default) before anything is allocated.
    if ((uint64_t)h.w * h.h > (512ull << 20) / 4 || idat.size() > 0x7fffffffull) return false;
    int bits = h.bpp_bits();
    // inflate
    size_t raw_cap = 0;
    if (!h.interlace) {
        raw_cap = (((size_t)h.w * bits + 7) / 8 + 1) * h.h;
    } else {
        static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1};
        static const int dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int k = 0; k < 7; ++k) {
            uint32_t pw = (h.w + dx[k] - 1 - xs[k]) / dx[k], ph = (h.h + dy[k] - 1 - ys[k]) / dy[k];
            if (pw && ph) raw_cap += (((size_t)pw * bits + 7) / 8 + 1) * ph;
        }
    }
    std::vector<uint8_t> raw(raw_cap);
    z_stream zs{};
    if (inflateInit(&zs) != Z_OK) return false;
    zs.next_in = idat.data();
    zs.avail_in = (uInt)idat.size();
    zs.next_out = raw.data();
    zs.avail_out = (uInt)raw.size();
    int zr = inflate(&zs, Z_FINISH);
    size_t got = zs.total_out;
    inflateEnd(&zs);
    if (zr != Z_STREAM_END && zr != Z_OK && zr != Z_BUF_ERROR) return false;
    if (got < raw_cap) return false;

    out.width = h.w;
    out.height = h.h;
    out.rgba.assign((size_t)h.w * h.h * 4, 0);

    auto sample = [&](const uint8_t* row, uint32_t x, int ch) -> uint32_t {
        int nch = h.channels();
        size_t idx = (size_t)x * nch + ch;
        if (h.depth == 8) return row[idx];
        if (h.depth == 16) return (uint32_t)row[idx * 2] << 8 | row[idx * 2 + 1];
        size_t bit = idx * h.depth;
        return (row[bit / 8] >> (8 - h.depth - (bit % 8))) & ((1u << h.depth) - 1);
    };
    auto put = [&](const uint8_t* row, uint32_t sx, uint32_t x, uint32_t y) {
        uint8_t* o = out.rgba.data() + ((size_t)y * h.w + x) * 4;
        switch (h.ctype) {
            case 0: {
                uint32_t v = sample(row, sx, 0);
                o[0] = o[1] = o[2] = to8(v, h.depth);
                o[3] = 255;
                if (trns.size() >= 2 && v == ((uint32_t)trns[0] << 8 | trns[1])) o[3] = 0;
                break;
            }
            case 2: {
                uint32_t r = sample(row, sx, 0), g = sample(row, sx, 1), b = sample(row, sx, 2);
                o[0] = to8(r, h.depth); o[1] = to8(g, h.depth); o[2] = to8(b, h.depth);
                o[3] = 255;
                if (trns.size() >= 6 && r == ((uint32_t)trns[0] << 8 | trns[1]) &&
                    g == ((uint32_t)trns[2] << 8 | trns[3]) && b == ((uint32_t)trns[4] << 8 | trns[5]))
                    o[3] = 0;
                break;
            }
            case 3: {
                uint32_t i = sample(row, sx, 0);
                if (i * 3 + 2 < plte.size()) {
                    o[0] = plte[i * 3]; o[1] = plte[i * 3 + 1]; o[2] = plte[i * 3 + 2];
                }
                o[3] = i < trns.size() ? trns[i] : 255;
                break;
            }
            case 4: {
                o[0] = o[1] = o[2] = to8(sample(row, sx, 0), h.depth);
                o[3] = to8(sample(row, sx, 1), h.depth);
                break;
            }
            default: {
                for (int c = 0; c < 4; ++c) o[c] = to8(sample(row, sx, c), h.depth);
            }
        }
    };

    std::vector<uint8_t> px;
    size_t used = 0;
    if (!h.interlace) {
        if (!unfilter(raw.data(), raw.size(), h.w, h.h, bits, px, used)) return false;
        size_t stride = ((size_t)h.w * bits + 7) / 8;
        for (uint32_t y = 0; y < h.h; ++y)
            for (uint32_t x = 0; x < h.w; ++x) put(px.data() + y * stride, x, x, y);
    } else {
        static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1};
        static const int dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
        size_t pos = 0;
        for (int k = 0; k < 7; ++k) {
            uint32_t pw = (h.w + dx[k] - 1 - xs[k]) / dx[k], ph = (h.h + dy[k] - 1 - ys[k]) / dy[k];
            if (!pw || !ph) continue;
            if (!unfilter(raw.data() + pos, raw.size() - pos, pw, ph, bits, px, used)) return false;
            pos += used;
            size_t stride = ((size_t)pw * bits + 7) / 8;
            for (uint32_t y = 0; y < ph; ++y)
                for (uint32_t x = 0; x < pw; ++x)
                    put(px.data() + y * stride, x, xs[k] + x * dx[k], ys[k] + y * dy[k]);
        }
    }
    return true;
}

}  // namespace rt2
