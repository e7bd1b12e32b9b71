// TEST INFRASTRUCTURE.  Decodes image files with the reference's own 8-bit texture decoder, called the way the
// reference calls it (stbi_load(path, &w, &h, &n, 3), reference src/texture.cpp:46), and writes the raw result:
//   stb_dump in.jpg out.bin      out.bin = int32 width, int32 height, width*height*3 bytes (all zero sizes on failure)
// Compiled against the header where it lies in the reference checkout (oracle/Makefile); used by
// tests/golden/make_jpeg_golden.py to produce the expected texels of the JPEG fixtures.
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"

#include <cstdint>
#include <cstdio>

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    int w = 0, h = 0, n = 0;
    unsigned char *data = stbi_load(argv[1], &w, &h, &n, 3);
    FILE *f = fopen(argv[2], "wb");
    if (!f) return 2;
    int32_t dims[2] = {data ? w : 0, data ? h : 0};
    fwrite(dims, sizeof dims, 1, f);
    if (data) fwrite(data, 1, (size_t)w * h * 3, f);
    fclose(f);
    if (!data) fprintf(stderr, "%s: %s\n", argv[1], stbi_failure_reason());
    return data ? 0 : 1;
}
