// 8-bit RGB image files: PNG through zlib (the reference saves with the `image` crate, format by
// extension, src/main.rs:49; PNG is what its report harness writes) and binary PPM.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace host {

inline void png_chunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t n) {
    auto be32 = [&out](uint32_t v) {
        for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(v >> s));
    };
    be32((uint32_t)n);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data, data + n);
    uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start));
    be32(crc);
}

inline std::vector<uint8_t> encode_png(const uint8_t* rgb, uint32_t width, uint32_t height) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (1 + (size_t)width * 3));
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);  // filter type None
        raw.insert(raw.end(), rgb + (size_t)y * width * 3, rgb + (size_t)(y + 1) * width * 3);
    }
    uLongf bound = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(bound);
    if (compress2(z.data(), &bound, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("zlib compress failed");
    z.resize(bound);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    uint8_t ihdr[13];
    for (int k = 0; k < 4; ++k) {
        ihdr[k] = (uint8_t)(width >> (24 - 8 * k));
        ihdr[4 + k] = (uint8_t)(height >> (24 - 8 * k));
    }
    ihdr[8] = 8;   // bit depth
    ihdr[9] = 2;   // colour type: truecolour
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    png_chunk(out, "IHDR", ihdr, 13);
    png_chunk(out, "IDAT", z.data(), z.size());
    png_chunk(out, "IEND", nullptr, 0);
    return out;
}

inline void save_image(const std::string& path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    size_t dot = path.rfind('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    for (char& c : ext) c = (char)tolower((unsigned char)c);
    std::vector<uint8_t> bytes;
    if (ext == "png") {
        bytes = encode_png(rgb, width, height);
    } else if (ext == "ppm") {
        std::string head = "P6\n" + std::to_string(width) + " " + std::to_string(height) + "\n255\n";
        bytes.assign(head.begin(), head.end());
        bytes.insert(bytes.end(), rgb, rgb + (size_t)width * height * 3);
    } else {
        throw std::runtime_error("unsupported image format \"" + ext + "\" (png and ppm are written)");
    }
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    size_t n = std::fwrite(bytes.data(), 1, bytes.size(), f);
    if (std::fclose(f) != 0 || n != bytes.size()) throw std::runtime_error("short write to " + path);
}

}  // namespace host
