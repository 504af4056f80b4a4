// output.hpp -- the stage after the render call: main.rs:66-91 (one RGB8 PNG per frame, ./output/image_%04d.png).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace pt {

// write_file (main.rs:78-91): an 8-bit RGB PNG of `pixels` (width*height*3 bytes, top row first), as the png crate
// writes it with its defaults (colour type 2, bit depth 8, no interlace; here: filter 0 on every row, zlib level 6).
// Throws std::runtime_error on I/O failure or size mismatch.
void write_file(const std::string& path, size_t width, size_t height, const std::vector<uint8_t>& pixels);
void write_file(const std::string& path, size_t width, size_t height, const uint8_t* pixels, size_t n_bytes);

// "{dir}/image_{index:04}.png" (main.rs:67)
std::string frame_path(const std::string& dir, size_t frame_index);

}  // namespace pt
