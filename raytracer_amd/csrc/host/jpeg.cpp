// JPEG texture decoding.
#include "image_io.h"

namespace rth {

bool decode_jpeg(const std::vector<uint8_t> &file, int &w, int &h, std::vector<uint8_t> &rgb, std::string &err) {
    (void)file; (void)w; (void)h; (void)rgb;
    err = "JPEG decoding not available";
    return false;
}

}  // namespace rth
