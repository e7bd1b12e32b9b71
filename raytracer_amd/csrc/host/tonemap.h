#pragma once
#include <cstdint>
namespace rth {
// hdr: width*height*3 floats; out: width*height*3 bytes, same pixel order.
void apply_tonemap(int tmo, float opt1, float opt2, float saturation, float gamma, int width, int height, const float *hdr,
                   uint8_t *out);
}
