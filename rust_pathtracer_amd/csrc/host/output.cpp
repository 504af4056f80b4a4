// output.cpp -- PNG output (main.rs:66-91).  The container is the only thing here: signature, IHDR, one IDAT holding
// the zlib stream of the filtered scanlines, IEND; CRC-32 and DEFLATE come from zlib.
#include "output.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace pt {
namespace {

void put_u32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}

void put_chunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t n) {
    put_u32(out, (uint32_t)n);
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    const uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4));
    put_u32(out, crc);
}

}  // namespace

void write_file(const std::string& path, size_t width, size_t height, const uint8_t* pixels, size_t n_bytes) {
    if (width == 0 || height == 0 || width > 0x7FFFFFFFull || height > 0x7FFFFFFFull) throw std::runtime_error("write_file: bad image size");
    if (!pixels || n_bytes != width * height * 3) throw std::runtime_error("write_file: pixel buffer is not width*height*3 bytes");
    // scanlines, each preceded by its filter type (0 = None)
    std::vector<uint8_t> raw((width * 3 + 1) * height);
    for (size_t y = 0; y < height; ++y) {
        raw[y * (width * 3 + 1)] = 0;
        std::memcpy(&raw[y * (width * 3 + 1) + 1], pixels + y * width * 3, width * 3);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("write_file: zlib compress2 failed");
    std::vector<uint8_t> out;
    out.reserve(zn + 64);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    out.insert(out.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put_u32(ihdr, (uint32_t)width);
    put_u32(ihdr, (uint32_t)height);
    ihdr.push_back(8);  // bit depth
    ihdr.push_back(2);  // colour type: RGB (png::ColorType::Rgb, main.rs:87)
    ihdr.push_back(0);  // compression
    ihdr.push_back(0);  // filter method
    ihdr.push_back(0);  // no interlace
    put_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(out, "IDAT", z.data(), zn);
    put_chunk(out, "IEND", nullptr, 0);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("could not create image file: " + path);  // main.rs:69
    const size_t wrote = std::fwrite(out.data(), 1, out.size(), f);
    const int rc = std::fclose(f);
    if (wrote != out.size() || rc != 0) throw std::runtime_error("could not write image data: " + path);  // main.rs:71
}

void write_file(const std::string& path, size_t width, size_t height, const std::vector<uint8_t>& pixels) {
    write_file(path, width, height, pixels.data(), pixels.size());
}

std::string frame_path(const std::string& dir, size_t frame_index) {
    char name[64];
    std::snprintf(name, sizeof(name), "image_%04zu.png", frame_index);
    return dir.empty() ? std::string(name) : dir + "/" + name;
}

}  // namespace pt
