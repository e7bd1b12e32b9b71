"""Differential fuzzing of the JPEG texture decoder against the reference's decoder (oracle/_ref/stb_dump).

Mutates the fixture files of tests/golden/jpeg.npz (bit flips, random bytes, 0xff bytes, truncation) and compares, file
by file, whether both decoders accept it and whether the decoded texels are identical.  Needs the reference checkout (to
have built oracle/_ref/stb_dump), so it runs in the development container only:   python tools/fuzz_jpeg.py [rounds]
"""
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = r'''
#include "image_io.h"
#include <cstdio>
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    std::vector<uint8_t> d(1 << 22);
    d.resize(fread(d.data(), 1, d.size(), f));
    fclose(f);
    int w = 0, h = 0;
    std::vector<uint8_t> rgb;
    std::string e;
    bool ok = rth::decode_jpeg(d, w, h, rgb, e);
    FILE *o = fopen(argv[2], "wb");
    int dims[2] = {ok ? w : 0, ok ? h : 0};
    fwrite(dims, 8, 1, o);
    if (ok) fwrite(rgb.data(), 1, rgb.size(), o);
    fclose(o);
    if (!ok) fprintf(stderr, "%s\n", e.c_str());
    return ok ? 0 : 1;
}
'''


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    stb = os.path.join(ROOT, "oracle", "_ref", "stb_dump")
    golden = np.load(os.path.join(ROOT, "tests", "golden", "jpeg.npz"))
    rng = random.Random(5)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "h.cpp")
        open(src, "w").write(HARNESS)
        exe = os.path.join(tmp, "jdump")
        host = os.path.join(ROOT, "raytracer_amd", "csrc", "host")
        subprocess.run(["g++", "-std=c++17", "-O2", "-I" + host, src, os.path.join(host, "jpeg.cpp"), "-o", exe], check=True)
        both = same = only_ref = only_ours = rejected = 0
        notes = []
        for name in golden["names"]:
            data = bytes(golden["file_" + str(name)])
            for it in range(rounds):
                m = bytearray(data)
                for _ in range(rng.randint(1, 3)):
                    p, mode = rng.randrange(len(m)), rng.randrange(3)
                    m[p] = m[p] ^ (1 << rng.randrange(8)) if mode == 0 else rng.randrange(256) if mode == 1 else 0xff
                if rng.randrange(6) == 0:
                    m = m[: rng.randrange(len(m))]
                f = os.path.join(tmp, "m.jpg")
                open(f, "wb").write(m)
                a, b = os.path.join(tmp, "a.bin"), os.path.join(tmp, "b.bin")
                r1 = subprocess.run([stb, f, a], capture_output=True, text=True)
                r2 = subprocess.run([exe, f, b], capture_output=True, text=True)
                if r1.returncode and r2.returncode:
                    rejected += 1
                elif r1.returncode != r2.returncode:
                    only_ref += r1.returncode == 0
                    only_ours += r2.returncode == 0
                    keep = os.path.join("/tmp", f"jpegdiff_{name}_{it}.jpg")
                    open(keep, "wb").write(m)
                    notes.append((str(name), it, "ref: " + (r1.stderr.strip()[-40:] or "ok"), "ours: " + (r2.stderr.strip() or "ok"), keep))
                else:
                    both += 1
                    if open(a, "rb").read() == open(b, "rb").read():
                        same += 1
                    else:
                        keep = os.path.join("/tmp", f"jpegdiff_{name}_{it}.jpg")
                        open(keep, "wb").write(m)
                        notes.append((str(name), it, "texels differ", keep))
        print(f"both accept: {both} (identical texels: {same}); both reject: {rejected}; only reference accepts: {only_ref}; "
              f"only ours accepts: {only_ours}")
        for n in notes[:60]:
            print("  ", n)


if __name__ == "__main__":
    main()
