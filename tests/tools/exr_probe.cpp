// Decodes every file named on the command line with the host library's EXR reader and prints one line per file.
// Built with -fsanitize=address,undefined by tests/test_exr_hostile.py: a malformed file may be rejected or decoded, but
// must not make the reader touch memory it does not own.
#include <cstdio>
#include <string>
#include <vector>

#include "image_io.h"

int main(int argc, char **argv) {
    for (int i = 1; i < argc; ++i) {
        int w = 0, h = 0;
        std::vector<float> rgba;
        std::string err;
        const bool ok = rth::read_exr_rgba(argv[i], w, h, rgba, err);
        printf("%s %s %d %d %zu %s\n", argv[i], ok ? "ok" : "rejected", w, h, rgba.size(), err.c_str());
    }
    return 0;
}
