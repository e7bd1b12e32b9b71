// Host-side image file helpers (texture decode, output encode).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rth {

// 8-bit RGB PNG.  flip_vertically writes the last buffer row first (the reference sets
// stbi_flip_vertically_on_write(1) because its framebuffer is bottom-row-first, src/main.cpp:419).
bool write_png_rgb8(const std::string &path, int w, int h, const uint8_t *rgb, bool flip_vertically);
// 3-channel float EXR, rows in buffer order.
bool write_exr_rgb32f(const std::string &path, int w, int h, const float *rgb);
bool read_exr_rgba(const std::string &path, int &w, int &h, std::vector<float> &rgba, std::string &err);
// PNG, else JPEG -> 8-bit RGB.
bool decode_image_rgb8(const std::vector<uint8_t> &file, int &w, int &h, std::vector<uint8_t> &rgb, std::string &err);
// Baseline / progressive JPEG -> 8-bit RGB.
bool decode_jpeg(const std::vector<uint8_t> &file, int &w, int &h, std::vector<uint8_t> &rgb, std::string &err);

}  // namespace rth
