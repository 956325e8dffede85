// image_io.hpp -- host-side image helpers of the headless driver:
//   * Radiance .hdr (RGBE) reader standing in for the `radiant` crate (src/io.rs:108);
//   * PNG writer (8-bit RGB, zlib level 9, adaptive row filters) standing in for io::save_colors_as_image
//     (src/io.rs:310-336);
// None of this is on the GPU path.
#ifndef RT_HOST_IMAGE_IO_HPP
#define RT_HOST_IMAGE_IO_HPP

#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace imageio {

struct HdrImage {
    uint32_t width = 0, height = 0;
    std::vector<float> rgb;  // row-major, file scanline order (first scanline = top = "up" for the renderer)
};

// Radiance RGBE, flat or new-style run-length encoded scanlines, -Y H +X W orientation.
inline HdrImage load_hdr(const std::string &path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::string line;
    bool magic = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.rfind("#?", 0) == 0) magic = true;
        if (line.empty()) break;
    }
    if (!magic) throw std::runtime_error(path + ": not a Radiance file");
    if (!std::getline(in, line)) throw std::runtime_error(path + ": missing resolution line");
    int h = 0, w = 0;
    if (std::sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || h <= 0 || w <= 0)
        throw std::runtime_error(path + ": unsupported orientation '" + line + "'");
    HdrImage img;
    img.width = (uint32_t)w;
    img.height = (uint32_t)h;
    img.rgb.resize((size_t)w * h * 3);
    std::vector<uint8_t> scan((size_t)w * 4);
    auto decode = [&](int y) {
        for (int x = 0; x < w; x++) {
            const uint8_t *p = &scan[(size_t)x * 4];
            float *o = &img.rgb[((size_t)y * w + x) * 3];
            if (p[3] == 0) {
                o[0] = o[1] = o[2] = 0.0f;
            } else {
                float f = std::ldexp(1.0f, (int)p[3] - (128 + 8));
                o[0] = p[0] * f, o[1] = p[1] * f, o[2] = p[2] * f;
            }
        }
    };
    for (int y = 0; y < h; y++) {
        uint8_t head[4];
        in.read((char *)head, 4);
        if (!in) throw std::runtime_error(path + ": truncated");
        if (w >= 8 && w < 32768 && head[0] == 2 && head[1] == 2 && ((head[2] << 8) | head[3]) == w) {
            for (int c = 0; c < 4; c++) {
                int x = 0;
                while (x < w) {
                    int n = in.get();
                    if (n < 0) throw std::runtime_error(path + ": truncated");
                    if (n > 128) {
                        n -= 128;
                        int v = in.get();
                        if (x + n > w) throw std::runtime_error(path + ": bad run");
                        for (int k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = (uint8_t)v;
                    } else {
                        if (x + n > w) throw std::runtime_error(path + ": bad run");
                        for (int k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = (uint8_t)in.get();
                    }
                }
            }
        } else {
            std::memcpy(scan.data(), head, 4);
            in.read((char *)scan.data() + 4, (std::streamsize)((size_t)w * 4 - 4));
            if (!in) throw std::runtime_error(path + ": truncated");
        }
        decode(y);
    }
    return img;
}

// float RGB -> RGBE (Ward's float2rgbe: mantissas truncated, shared exponent of the largest component)
inline void float_to_rgbe(const float *c, uint8_t px[4]) {
    float m = std::fmax(c[0], std::fmax(c[1], c[2]));
    px[0] = px[1] = px[2] = px[3] = 0;
    if (m > 1e-32f) {
        int e;
        float f = std::frexp(m, &e) * 256.0f / m;
        px[0] = (uint8_t)(c[0] * f), px[1] = (uint8_t)(c[1] * f), px[2] = (uint8_t)(c[2] * f), px[3] = (uint8_t)(e + 128);
    }
}

// Writes a Radiance file, -Y H +X W: flat RGBE pixels, or (rle) new-style run-length encoded scanlines -- each
// scanline "2 2 hi(W) lo(W)" followed by its four channels one after another, as runs (128 + n, value) of 4..127
// equal bytes and literal stretches (n, bytes...) of at most 128.  Used to ship procedural environments as files.
inline void save_hdr(const std::string &path, uint32_t w, uint32_t h, const float *rgb, bool rle = false) {
    std::ofstream out(path, std::ios::binary);
    if (!out) throw std::runtime_error("cannot write " + path);
    out << "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y " << h << " +X " << w << "\n";
    rle = rle && w >= 8 && w < 32768;
    std::vector<uint8_t> scan((size_t)w * 4), enc;
    for (uint32_t y = 0; y < h; y++) {
        for (uint32_t x = 0; x < w; x++) float_to_rgbe(rgb + ((size_t)y * w + x) * 3, &scan[(size_t)x * 4]);
        if (!rle) {
            out.write((const char *)scan.data(), (std::streamsize)scan.size());
            continue;
        }
        enc.clear();
        enc.push_back(2), enc.push_back(2), enc.push_back((uint8_t)(w >> 8)), enc.push_back((uint8_t)(w & 0xFF));
        for (int c = 0; c < 4; c++) {
            uint32_t x = 0;
            while (x < w) {
                // length of the run of equal bytes starting at x
                uint32_t run = 1;
                while (x + run < w && run < 127 && scan[(size_t)(x + run) * 4 + c] == scan[(size_t)x * 4 + c]) run++;
                if (run >= 4) {
                    enc.push_back((uint8_t)(128 + run)), enc.push_back(scan[(size_t)x * 4 + c]);
                    x += run;
                    continue;
                }
                // a literal stretch: up to the next run of at least 4, at most 128 bytes
                uint32_t n = 0;
                while (x + n < w && n < 128) {
                    uint32_t r = 1;
                    while (x + n + r < w && r < 4 && scan[(size_t)(x + n + r) * 4 + c] == scan[(size_t)(x + n) * 4 + c]) r++;
                    if (r >= 4) break;
                    n++;
                }
                enc.push_back((uint8_t)n);
                for (uint32_t k = 0; k < n; k++) enc.push_back(scan[(size_t)(x + k) * 4 + c]);
                x += n;
            }
        }
        out.write((const char *)enc.data(), (std::streamsize)enc.size());
    }
}

namespace detail {
inline uint32_t crc32(const uint8_t *d, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ d[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
inline void put32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back(x >> 24), v.push_back(x >> 16), v.push_back(x >> 8), v.push_back(x);
}
inline void chunk(std::ofstream &out, const char *type, const std::vector<uint8_t> &data) {
    std::vector<uint8_t> buf;
    put32(buf, (uint32_t)data.size());
    buf.insert(buf.end(), type, type + 4);
    buf.insert(buf.end(), data.begin(), data.end());
    uint32_t c = crc32(buf.data() + 4, buf.size() - 4);
    put32(buf, c);
    out.write((const char *)buf.data(), (std::streamsize)buf.size());
}
}  // namespace detail

// 8-bit RGB PNG from packed 0xFFRRGGBB pixels (alpha dropped as int_to_rgb does, src/io.rs:301-307), written as the
// reference asks its encoder to (src/io.rs:319-325): Compression::Best = zlib's level 9, and an adaptive filter: for
// every row the one of the five PNG filters whose output has the smallest sum of absolute values (bytes read as
// signed), the usual minimum-sum-of-absolute-differences heuristic.  The pixels a decoder gets back are what matters;
// the compressed bytes themselves need not equal another encoder's.
inline void save_png(const std::string &path, const uint32_t *argb, uint32_t w, uint32_t h) {
    std::ofstream out(path, std::ios::binary);
    if (!out) throw std::runtime_error("cannot write " + path);
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    out.write((const char *)sig, 8);
    std::vector<uint8_t> ihdr;
    detail::put32(ihdr, w);
    detail::put32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
    detail::chunk(out, "IHDR", ihdr);
    const size_t stride = 3 * (size_t)w;
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (1 + stride));
    std::vector<uint8_t> cur(stride), prev(stride, 0), cand[5];
    for (auto &c : cand) c.resize(stride);
    for (uint32_t y = 0; y < h; y++) {
        for (uint32_t x = 0; x < w; x++) {
            uint32_t c = argb[(size_t)y * w + x];
            cur[3 * (size_t)x] = (c >> 16) & 0xFF, cur[3 * (size_t)x + 1] = (c >> 8) & 0xFF, cur[3 * (size_t)x + 2] = c & 0xFF;
        }
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= 3 ? cur[i - 3] : 0, b = prev[i], c = i >= 3 ? prev[i - 3] : 0;
            const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
            const int paeth = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            cand[0][i] = cur[i];
            cand[1][i] = (uint8_t)(cur[i] - a);
            cand[2][i] = (uint8_t)(cur[i] - b);
            cand[3][i] = (uint8_t)(cur[i] - ((a + b) >> 1));
            cand[4][i] = (uint8_t)(cur[i] - paeth);
        }
        int best = 0;
        uint64_t best_sum = ~0ull;
        for (int f = 0; f < 5; f++) {
            uint64_t sum = 0;
            for (size_t i = 0; i < stride; i++) sum += (uint64_t)std::abs((int)(int8_t)cand[f][i]);
            if (sum < best_sum) best_sum = sum, best = f;
        }
        raw.push_back((uint8_t)best);
        raw.insert(raw.end(), cand[best].begin(), cand[best].end());
        prev = cur;
    }
    uLongf bound = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(bound);
    if (compress2(z.data(), &bound, raw.data(), (uLong)raw.size(), 9) != Z_OK) throw std::runtime_error("zlib: compress2 failed");
    z.resize(bound);
    detail::chunk(out, "IDAT", z);
    detail::chunk(out, "IEND", {});
}

}  // namespace imageio
#endif
