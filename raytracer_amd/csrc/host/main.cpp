// Command-line front end: `./raytracer <scene.json|dir> [threading_type 0..3] [thread_count]`.
//
// Same argv contract, validation, exit codes, log lines and output files as the reference's main()
// and process_file() (reference src/main.cpp:347-737).  The std::thread row/tile/pixel scheduler of
// the reference (src/main.cpp:437-555) is replaced by one rt_render() call per camera; the threading
// arguments are still parsed and validated so existing invocations keep working, but they no longer
// select anything — the GPU schedules the pixels.  RT_GPU=<index> picks the device (default 0).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <map>
#include <regex>
#include <string>
#include <thread>
#include <vector>

#include "rt_host.h"
#include "image_io.h"
#include "scene.h"
#include "tonemap.h"

namespace fs = std::filesystem;

static FILE *g_log = fopen("output.txt", "a");

// tee to stdout and ./output.txt (reference src/helper.cpp:3-19)
static void log_printf(const char *fmt, ...) {
    va_list args;
    va_start(args, fmt);
    vprintf(fmt, args);
    va_end(args);
    if (g_log) {
        va_start(args, fmt);
        vfprintf(g_log, fmt, args);
        va_end(args);
        fflush(g_log);
    }
    fflush(stdout);
}

// PNG sequences `<prefix>_<digits>.png` in `folder` -> `<prefix>.mp4` through the bundled ffmpeg
// (reference src/main.cpp:29-158); frames are moved to `used/` after a successful encode.
static void create_videos_from_png_sequences(const std::string &folder) {
    std::regex pattern(R"(^(.+)_(\d+)\.png$)", std::regex::ECMAScript | std::regex::icase);
    std::map<std::string, std::vector<std::pair<int, fs::path>>> sequences;
    std::error_code ec;
    for (const auto &entry : fs::directory_iterator(folder, ec)) {
        if (!entry.is_regular_file()) continue;
        std::string name = entry.path().filename().string();
        std::smatch m;
        if (std::regex_match(name, m, pattern)) sequences[m[1].str()].push_back({std::stoi(m[2].str()), entry.path()});
    }
    const std::string ffmpeg = "./linux_ffmpeg/ffmpeg";
    if (!fs::exists(ffmpeg)) {
        fprintf(stderr, "ERROR: FFmpeg executable not found. Looked for it at: %s\n", ffmpeg.c_str());
        return;
    }
    for (auto &[prefix, frames] : sequences) {
        if (frames.empty()) continue;
        std::sort(frames.begin(), frames.end(), [](auto &a, auto &b) { return a.first < b.first; });
        std::smatch m;
        std::string first = frames[0].second.filename().string();
        if (!std::regex_match(first, m, pattern)) continue;
        int pad = (int)m[2].str().length();
        std::string input = (fs::path(folder) / (prefix + "_%0" + std::to_string(pad) + "d.png")).string();
        std::string output = (fs::path(folder) / (prefix + ".mp4")).string();
        std::string cmd = ffmpeg + " -y -framerate 30 -i \"" + input +
                          "\" -c:v libopenh264 -b:v 8M -maxrate 8M -bufsize 16M -preset slow -profile:v high -pix_fmt yuv420p \"" + output + "\"";
        printf("Running: %s\n", cmd.c_str());
        if (system(cmd.c_str()) != 0) {
            fprintf(stderr, "FFmpeg execution failed\n");
            continue;
        }
        fs::path used = fs::path(folder) / "used";
        fs::create_directory(used, ec);
        for (auto &f : frames) fs::rename(f.second, used / f.second.filename(), ec);
    }
}

static int process_file(const char *filename, int thread_count, int threading_type) {
    static const char *names[] = {"row_dynamic", "row_static", "square_dynamic", "pixel_dynamic"};
    log_printf("SIMD On / Thread count: %d / Threading type: %s\n", thread_count, names[threading_type]);
    auto start = std::chrono::high_resolution_clock::now();
    log_printf("JSON parsing started\n");
    rth::HostScene scene;
    std::string err;
    if (!rth::load_scene(filename, scene, err)) {
        log_printf("%s\n", err.c_str());
        exit(-1);   // the reference exits on unreadable meshes / missing base meshes (src/parser.cpp:855, src/ply_loader.cpp:74)
    }
    for (auto &w : scene.warnings) log_printf("%s\n", w.c_str());
    // the reference makes this check per camera, before it spawns its threads (src/main.cpp:542-545)
    if (!scene.cameras.empty() && (thread_count > 4096 || thread_count < 1)) exit(-2);
    rt_scene_desc desc = scene.desc();
    rt_scene *gpu = nullptr;
    if (rt_scene_create(&desc, &gpu) != 0) {
        log_printf("rt_scene_create failed: %s\n", rt_last_error());
        exit(-1);
    }
    auto end = std::chrono::high_resolution_clock::now();
    log_printf("JSON parsing and grid preparation is done: %d ms\n", (int)std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count());

    for (auto &cam : scene.cameras) {
        log_printf("%s started\n", cam.image_name.c_str());
        start = std::chrono::high_resolution_clock::now();
        const int W = cam.desc.resx, H = cam.desc.resy;
        const std::string &name = cam.image_name;
        const bool exr = cam.desc.is_hdr || (name.size() >= 4 && name.compare(name.size() - 4, 4, ".exr") == 0);
        // Only an EXR needs the float frame on the host; for PNG outputs the frame stays in HBM and is tone-mapped
        // or quantised there (rt_tonemap / rt_quantize_ldr), so 8-bit pixels are all that crosses PCIe.
        std::vector<float> hdr;
        float *frame = nullptr, *device_frame = nullptr;
        if (exr) {
            hdr.assign((size_t)W * H * 3, 0.0f);
            frame = hdr.data();
        } else {
            if (hipMalloc((void **)&device_frame, (size_t)W * H * 3 * sizeof(float)) != hipSuccess) {
                log_printf("cannot allocate the device framebuffer\n");
                exit(-1);
            }
            frame = device_frame;
        }
        rt_stats stats;
        if (rt_render(gpu, &cam.desc, frame, &stats) != 0) {
            log_printf("rt_render failed: %s\n", rt_last_error());
            exit(-1);
        }
        end = std::chrono::high_resolution_clock::now();
        log_printf("%s rendering done: %d ms\n", cam.image_name.c_str(), (int)std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count());
        const unsigned long long rays = stats.primary_rays + stats.shadow_rays + stats.secondary_rays;
        log_printf("%s rays: %llu primary, %llu shadow, %llu secondary; device %.3f ms; %.1f Mrays/s\n", cam.image_name.c_str(),
                   (unsigned long long)stats.primary_rays, (unsigned long long)stats.shadow_rays, (unsigned long long)stats.secondary_rays,
                   stats.kernel_ms, stats.kernel_ms > 0 ? rays / stats.kernel_ms / 1e3 : 0.0);

        const size_t n_values = (size_t)W * H * 3;
        if (exr) {
            log_printf("%s write EXR started\n", name.c_str());
            if (!rth::write_exr_rgb32f("outputs/" + name, W, H, hdr.data())) fprintf(stderr, "Save EXR Error: cannot write outputs/%s\n", name.c_str());
        } else if (cam.tonemaps.empty()) {
            std::vector<uint8_t> ldr(n_values);
            if (rt_quantize_ldr(frame, (int64_t)n_values, ldr.data()) != 0) {
                log_printf("rt_quantize_ldr failed: %s\n", rt_last_error());
                exit(-1);
            }
            rth::write_png_rgb8("outputs/" + name, W, H, ldr.data(), true);
        }
        if (!cam.tonemaps.empty()) {
            std::vector<uint8_t> tm_out(n_values);
            for (auto &tm : cam.tonemaps) {
                const rt_tonemap_desc td{tm.type, tm.opt1, tm.opt2, tm.saturation, tm.gamma};
                if (rt_tonemap(frame, W, H, &td, tm_out.data()) != 0) {
                    log_printf("rt_tonemap failed: %s\n", rt_last_error());
                    exit(-1);
                }
                std::string base = name;
                size_t dot = base.find_last_of('.');
                if (dot != std::string::npos) base = base.substr(0, dot);
                std::string tm_name = base + tm.extension;
                log_printf("Saving tone mapped image: %s\n", tm_name.c_str());
                rth::write_png_rgb8("outputs/" + tm_name, W, H, tm_out.data(), true);
            }
        }
        if (device_frame) (void)hipFree(device_frame);
    }
    rt_scene_destroy(gpu);
    return 0;
}

int main(int argc, char **argv) {
    log_printf("\n>>> NEW RUN <<<\n");
    fs::create_directories("outputs");
    int thread_count = 1, tt = 0;
    if (argc == 2) {
        thread_count = (int)std::thread::hardware_concurrency();
    } else if (argc == 3 || argc == 4) {
        thread_count = argc == 4 ? atoi(argv[3]) : (int)std::thread::hardware_concurrency();
        int tmp = atoi(argv[2]);
        if (tmp >= 4 || tmp < 0) {
            log_printf("Invalid threading type\n");
            return -3;
        }
        tt = tmp;
    } else {
        // the reference renders three hard-coded benchmark scenes here (src/main.cpp:669-698); there are none to ship
        log_printf("usage: %s <scene.json|directory> [threading_type 0..3] [thread_count]\n", argv[0]);
        return 0;
    }
    if (thread_count <= 0) {
        log_printf("Invalid thread count\n");
        return -1;
    }
    if (const char *dev = getenv("RT_GPU")) {
        if (hipSetDevice(atoi(dev)) != hipSuccess) {
            log_printf("Error: cannot select GPU %s\n", dev);
            return 1;
        }
    }
    std::vector<std::string> files;
    std::string input = argv[1];
    if (input.size() >= 5 && input.compare(input.size() - 5, 5, ".json") == 0) {
        files.push_back(input);
    } else if (fs::is_directory(input)) {
        for (const auto &entry : fs::recursive_directory_iterator(input))
            if (entry.is_regular_file() && entry.path().extension() == ".json") files.push_back(entry.path().string());
    } else {
        log_printf("Error: %s is neither a .json file nor a directory.\n", input.c_str());
        return 1;
    }
    for (auto &f : files) process_file(f.c_str(), thread_count, tt);
    create_videos_from_png_sequences("outputs");
    return 0;
}
