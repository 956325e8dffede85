// image_io.hpp -- host-side image helpers of the headless driver:
//   * Radiance .hdr (RGBE) reader standing in for the `radiant` crate (src/io.rs:108);
//   * PNG writer (8-bit RGB) standing in for io::save_colors_as_image (src/io.rs:310-336);
// None of this is on the GPU path.
#ifndef RT_HOST_IMAGE_IO_HPP
#define RT_HOST_IMAGE_IO_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
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

// Writes a flat RGBE file (used to ship a procedural environment as a fixture).
inline void save_hdr(const std::string &path, uint32_t w, uint32_t h, const float *rgb) {
    std::ofstream out(path, std::ios::binary);
    out << "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y " << h << " +X " << w << "\n";
    for (size_t i = 0; i < (size_t)w * h; i++) {
        float r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        float m = std::fmax(r, std::fmax(g, b));
        uint8_t px[4] = {0, 0, 0, 0};
        if (m > 1e-32f) {
            int e;
            float f = std::frexp(m, &e) * 256.0f / m;
            px[0] = (uint8_t)(r * f), px[1] = (uint8_t)(g * f), px[2] = (uint8_t)(b * f), px[3] = (uint8_t)(e + 128);
        }
        out.write((const char *)px, 4);
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

// 8-bit RGB PNG from packed 0xFFRRGGBB pixels (alpha dropped as int_to_rgb does, src/io.rs:301-307).
// The zlib stream uses stored blocks: valid everywhere, no compression library needed.
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
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (1 + 3 * (size_t)w));
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);
        for (uint32_t x = 0; x < w; x++) {
            uint32_t c = argb[(size_t)y * w + x];
            raw.push_back((c >> 16) & 0xFF), raw.push_back((c >> 8) & 0xFF), raw.push_back(c & 0xFF);
        }
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        bool last = off + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(n & 0xFF), z.push_back(n >> 8), z.push_back(~n & 0xFF), z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + n));
        if (last) break;
    }
    for (uint8_t v : raw) {
        a = (a + v) % 65521;
        b = (b + a) % 65521;
    }
    detail::put32(z, (b << 16) | a);
    detail::chunk(out, "IDAT", z);
    detail::chunk(out, "IEND", {});
}


}  // namespace imageio
#endif
