// Command-line front end: `./raytracer <scene.json|dir> [threading_type 0..3] [thread_count]`.
//
// Same argv contract, validation, exit codes, log lines and output files as the reference's main()
// and process_file() (reference src/main.cpp:347-737).  The std::thread row/tile/pixel scheduler of
// the reference (src/main.cpp:437-555) is replaced by one rt_render() call per camera; the threading
// arguments are still parsed and validated so existing invocations keep working, but they no longer
// select anything — the GPU schedules the pixels.  RT_GPU=<index> picks the device; without it the files of a
// directory are spread over all visible GPUs (frame -> GPU).  RT_GPUS=<k> makes teams of k GPUs that render a frame
// together: each GPU its share of the 32x32 tiles, gathered on the team's first over RCCL (rt_render_gather); the teams
// take frames in turn beside each other, and the frames left over when fewer remain than there are teams are rendered
// by all GPUs regrouped into that many teams (frame-and-tile parallel: no GPU idles while a frame is left).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <condition_variable>
#include <map>
#include <mutex>
#include <regex>
#include <string>
#include <thread>
#include <vector>

#include <cerrno>
#include <sys/wait.h>
#include <unistd.h>

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

// Runs args[0] with the given arguments and waits for it; true when it exits with status 0.
static bool run_program(const std::vector<std::string> &args) {
    std::vector<char *> argv;
    for (const auto &a : args) argv.push_back(const_cast<char *>(a.c_str()));
    argv.push_back(nullptr);
    const pid_t pid = fork();
    if (pid < 0) return false;
    if (pid == 0) {
        execv(argv[0], argv.data());
        _exit(127);
    }
    int status = 0;
    while (waitpid(pid, &status, 0) < 0)
        if (errno != EINTR) return false;
    return WIFEXITED(status) && WEXITSTATUS(status) == 0;
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
        // argv, not a shell line: the names come from the scene files' ImageName
        const std::vector<std::string> args = {ffmpeg, "-y", "-framerate", "30", "-i", input, "-c:v", "libopenh264", "-b:v", "8M", "-maxrate", "8M",
                                               "-bufsize", "16M", "-preset", "slow", "-profile:v", "high", "-pix_fmt", "yuv420p", output};
        printf("Running:");
        for (const auto &a : args) printf(" %s", a.c_str());
        printf("\n");
        fflush(stdout);
        if (!run_program(args)) {
            fprintf(stderr, "FFmpeg execution failed\n");
            continue;
        }
        fs::path used = fs::path(folder) / "used";
        fs::create_directory(used, ec);
        for (auto &f : frames) fs::rename(f.second, used / f.second.filename(), ec);
    }
}

// ---- a directory of scene files as a pipeline --------------------------------------------------------------------
// The reference handles files one after the other on the main thread: parse, build, render, encode, write
// (src/main.cpp:705-733).  With the render down to milliseconds the host stages dominate a sequence of frames, so
// they run beside it: loader threads parse and build the next scenes, one render thread per GPU takes scenes in
// file order (frame -> GPU, no communication between GPUs), writer threads encode and write the images.  Every
// file's log lines are collected and printed in file order, so the log reads as the reference's does.
struct FileJob {
    std::string path, log;
    rth::HostScene scene;
    std::string error;
    bool loaded = false, load_ok = false, done = false;
    std::chrono::high_resolution_clock::time_point start;
};

// 8-bit frames on their way from the device to a writer thread live in pinned host memory that goes round: the copy from
// the device runs at the link's rate, and no frame pays for fresh pages (a 1080p frame is 6 MB: 1500 page faults and a
// staged copy otherwise, half a millisecond of the render thread per file).  Blocks are kept until the process ends.
class PinnedPool {
public:
    uint8_t *take(size_t bytes, size_t &room) {
        {
            std::lock_guard<std::mutex> lk(m_);
            for (size_t k = 0; k < idle_.size(); ++k)
                if (idle_[k].second >= bytes) {
                    uint8_t *p = idle_[k].first;
                    room = idle_[k].second;
                    idle_.erase(idle_.begin() + (long)k);
                    return p;
                }
        }
        void *p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        room = bytes;
        return (uint8_t *)p;
    }
    void give(uint8_t *p, size_t room) {
        std::lock_guard<std::mutex> lk(m_);
        if (idle_.size() < 64) {
            idle_.emplace_back(p, room);
            return;
        }
        (void)hipHostFree(p);
    }

private:
    std::mutex m_;
    std::vector<std::pair<uint8_t *, size_t>> idle_;
};
static PinnedPool g_pinned_pixels;

class HostPixels {   // n bytes, pinned when the pool can give them, ordinary memory otherwise
public:
    HostPixels() = default;
    explicit HostPixels(size_t n) {
        pinned_ = g_pinned_pixels.take(n, room_);
        if (!pinned_) heap_.resize(n);
    }
    HostPixels(HostPixels &&o) noexcept : pinned_(o.pinned_), room_(o.room_), heap_(std::move(o.heap_)) { o.pinned_ = nullptr; }
    HostPixels &operator=(HostPixels &&o) noexcept {
        if (this != &o) {
            release();
            pinned_ = o.pinned_, room_ = o.room_, heap_ = std::move(o.heap_);
            o.pinned_ = nullptr;
        }
        return *this;
    }
    HostPixels(const HostPixels &) = delete;
    HostPixels &operator=(const HostPixels &) = delete;
    ~HostPixels() { release(); }
    uint8_t *data() { return pinned_ ? pinned_ : heap_.data(); }

private:
    void release() {
        if (pinned_) g_pinned_pixels.give(pinned_, room_);
        pinned_ = nullptr;
    }
    uint8_t *pinned_ = nullptr;
    size_t room_ = 0;
    std::vector<uint8_t> heap_;
};

struct WriteTask {
    enum Kind { PNG_RGB8, EXR_RGB32F } kind;
    std::string path;
    int width, height;
    HostPixels rgb8;
    std::vector<float> rgb32f;
};

static void job_printf(FileJob &job, const char *fmt, ...) {
    char buf[1024];
    va_list args;
    va_start(args, fmt);
    vsnprintf(buf, sizeof buf, fmt, args);
    va_end(args);
    job.log += buf;
}

// GPUs that render a frame together: the first device leads (it receives the frame), the others run beside it.
struct Team {
    std::vector<int> devices;
    std::vector<rt_comm *> comms;   // one per device when the team has more than one
    // on devices[0], kept from frame to frame (grown when a frame needs more): the float frame a PNG is made from, and its
    // 8-bit pixels before they cross to the host
    float *frame = nullptr;
    size_t frame_values = 0;
    uint8_t *ldr = nullptr;
    size_t ldr_bytes = 0;
    bool room_for(size_t n_values) {   // both buffers hold a frame of n_values floats; false when the device is out of memory
        if (frame_values < n_values) {
            if (frame) (void)hipFree(frame);
            frame = nullptr, frame_values = 0;
            if (hipMalloc((void **)&frame, n_values * sizeof(float)) != hipSuccess) return false;
            frame_values = n_values;
        }
        if (ldr_bytes < n_values) {
            if (ldr) (void)hipFree(ldr);
            ldr = nullptr, ldr_bytes = 0;
            if (hipMalloc((void **)&ldr, n_values) != hipSuccess) return false;
            ldr_bytes = n_values;
        }
        return true;
    }
    void release_buffers() {
        if (frame) (void)hipFree(frame);
        if (ldr) (void)hipFree(ldr);
        frame = nullptr, ldr = nullptr, frame_values = ldr_bytes = 0;
    }
};

// `gpus` cut into teams of `size` (the last one takes what is left).
static std::vector<Team> make_teams(const std::vector<int> &gpus, size_t size) {
    std::vector<Team> teams;
    for (size_t at = 0; at < gpus.size(); at += size) {
        Team t;
        t.devices.assign(gpus.begin() + at, gpus.begin() + std::min(gpus.size(), at + size));
        teams.push_back(std::move(t));
    }
    return teams;
}

// `gpus` spread over `n` teams as evenly as they go (n <= gpus.size()).
static std::vector<Team> spread_teams(const std::vector<int> &gpus, size_t n) {
    std::vector<Team> teams(n);
    size_t at = 0;
    for (size_t k = 0; k < n; ++k) {
        const size_t take = gpus.size() / n + (k < gpus.size() % n ? 1 : 0);
        teams[k].devices.assign(gpus.begin() + at, gpus.begin() + at + take);
        at += take;
    }
    return teams;
}

class Pipeline {
public:
    // `gpus`: every device that renders (a device named twice gets two ranks: the many-GPU paths rehearsed on one GPU);
    // `team_size`: devices that render a frame together.  The frames go to the teams in turn (the reference's dynamic
    // hand-out of work, src/main.cpp:486-539, with a team in the place of a thread); when fewer frames are left than there
    // are teams, all devices are regrouped into as many teams as frames are left (`tail_tiles`), so that a lone last
    // frame, or a single file, is split by tiles over every GPU instead of leaving all but one idle.
    Pipeline(std::vector<std::string> files, int thread_count, int threading_type, std::vector<int> gpus, size_t team_size, bool tail_tiles)
        : thread_count_(thread_count), threading_type_(threading_type), gpus_(std::move(gpus)) {
        jobs_.resize(files.size());
        for (size_t i = 0; i < files.size(); ++i) jobs_[i].path = files[i];
        teams_ = make_teams(gpus_, std::max<size_t>(1, team_size));
        const size_t left_over = jobs_.size() % teams_.size();
        bulk_frames_ = jobs_.size();
        if (tail_tiles && left_over && gpus_.size() > left_over && spread_teams(gpus_, left_over)[0].devices.size() > teams_[0].devices.size()) {
            bulk_frames_ = jobs_.size() - left_over;
            tail_teams_ = spread_teams(gpus_, left_over);
        }
    }
    ~Pipeline() {
        for (auto *teams : {&teams_, &tail_teams_})
            for (Team &t : *teams)
                for (rt_comm *c : t.comms) rt_comm_destroy(c);
    }

    // false (with a message) when a team's communicators cannot be made
    static bool connect(std::vector<Team> &teams) {
        for (Team &t : teams) {
            if (t.devices.size() < 2 || !t.comms.empty()) continue;
            t.comms.assign(t.devices.size(), nullptr);
            std::vector<int32_t> devices(t.devices.begin(), t.devices.end());
            if (rt_comm_init_all((int32_t)devices.size(), devices.data(), t.comms.data()) != 0) {
                log_printf("Error: cannot set up %zu GPUs for tile-sharded frames: %s\n", t.devices.size(), rt_last_error());
                t.comms.clear();
                return false;
            }
        }
        return true;
    }
    bool prepare() { return bulk_frames_ == 0 || connect(teams_); }

    void run() {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        // per render thread (one per team) up to eight loaders and six writers: a 1080p frame of a light scene keeps its
        // render thread for 3 ms (scene to the device, render, tone map), takes 20 ms to parse and build its tree and 15 ms to
        // encode as PNG, so the helpers of one GPU fill fifteen cores and eight GPUs need eight times as many
        // (RT_LOADERS / RT_WRITERS: the counts themselves)
        const unsigned n_teams = (unsigned)std::max<size_t>(1, teams_.size());
        auto wanted = [&](const char *name, unsigned per_team) {
            const char *e = getenv(name);
            const unsigned n = e && atoi(e) > 0 ? (unsigned)atoi(e) : std::min(per_team * n_teams, std::max(1u, hw / 2));
            return std::max(1u, std::min(n, 256u));
        };
        const unsigned n_load = (unsigned)std::min<size_t>(jobs_.size(), wanted("RT_LOADERS", 8));
        const unsigned n_write = wanted("RT_WRITERS", 6);
        load_ahead_ = std::max<size_t>(2 * teams_.size() + 2, n_load + teams_.size());
        std::vector<std::thread> loaders, renderers, writers;
        for (unsigned i = 0; i < n_load; ++i) loaders.emplace_back([this] { load_loop(); });
        for (unsigned i = 0; i < n_write; ++i) writers.emplace_back([this] { write_loop(); });
        // the teams beside each other while there is a frame for each, then everybody on what is left
        render_limit_ = bulk_frames_;
        if (bulk_frames_)
            for (Team &team : teams_) renderers.emplace_back([this, &team] { render_loop(team); });
        for (auto &t : renderers) t.join();
        renderers.clear();
        if (!tail_teams_.empty()) {
            if (!connect(tail_teams_)) {   // (said above) the frames that are left go to the teams as they were
                if (!connect(teams_)) fail(flushed_, 1);
                tail_teams_.clear();
            }
            {
                std::lock_guard<std::mutex> lk(m_);
                render_limit_ = jobs_.size();
            }
            for (Team &team : tail_teams_.empty() ? teams_ : tail_teams_) renderers.emplace_back([this, &team] { render_loop(team); });
            for (auto &t : renderers) t.join();
        }
        for (auto &t : loaders) t.join();
        {
            std::lock_guard<std::mutex> lk(m_);
            writers_may_stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : writers) t.join();
    }

private:
    // ---- stage 1: parse + BVH build, at most load_ahead_ files beyond the last one reported -------------------
    void load_loop() {
        for (;;) {
            size_t i;
            {
                std::vector<rth::HostScene> spent;   // scenes the render threads are done with: released here, outside the lock
                std::unique_lock<std::mutex> lk(m_);
                spent.swap(spent_scenes_);
                lk.unlock();
                spent.clear();
                lk.lock();
                i = next_load_++;
                if (i >= jobs_.size()) return;
                cv_.wait(lk, [&] { return i < flushed_ + load_ahead_; });   // bounds the scenes held in host memory
            }
            FileJob &job = jobs_[i];
            static const char *names[] = {"row_dynamic", "row_static", "square_dynamic", "pixel_dynamic"};
            job_printf(job, "SIMD On / Thread count: %d / Threading type: %s\n", thread_count_, names[threading_type_]);
            job.start = std::chrono::high_resolution_clock::now();
            job_printf(job, "JSON parsing started\n");
            const bool ok = rth::load_scene(job.path, job.scene, job.error);
            if (getenv("RT_CLI_TIMING"))
                job_printf(job, "[timing] load %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - job.start).count());
            {
                std::lock_guard<std::mutex> lk(m_);
                job.load_ok = ok;
                job.loaded = true;
            }
            cv_.notify_all();
        }
    }

    // the reference exits the process where it fails; earlier files' log lines come out first
    // (and their images are on disk: the reference has written them by the time a later file fails)
    [[noreturn]] void fail(size_t index, int code) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return flushed_ == index; });
        log_printf("%s", jobs_[index].log.c_str());
        fflush(stdout);
        cv_.wait(lk, [&] { return write_queue_.empty() && writes_in_flight_ == 0; });
        _exit(code & 0xff);
    }

    // ---- stage 2: one thread per team of GPUs, scenes in file order ---------------------------------------------
    void render_loop(Team &team) {
        struct Release {   // the team's frame buffers go when its thread ends
            Team &t;
            ~Release() {
                if (!t.devices.empty() && hipSetDevice(t.devices[0]) == hipSuccess) t.release_buffers();
            }
        } release{team};
        // the first HIP call on a device takes 0.15 s (the driver makes the process's context): made now, beside the loaders'
        // first files, not when the first scene arrives
        if (!team.devices.empty() && hipSetDevice(team.devices[0]) == hipSuccess) (void)hipFree(nullptr);
        for (;;) {
            size_t i;
            const auto t_wait = std::chrono::high_resolution_clock::now();
            {
                std::unique_lock<std::mutex> lk(m_);
                if (next_render_ >= render_limit_) return;
                i = next_render_++;
                cv_.wait(lk, [&] { return jobs_[i].loaded; });
            }
            FileJob &job = jobs_[i];
            if (getenv("RT_CLI_TIMING"))
                job_printf(job, "[timing] render thread waited %.1f ms for this file\n",
                           std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_wait).count());
            if (!job.load_ok) {
                job_printf(job, "%s\n", job.error.c_str());
                fail(i, -1);   // unreadable meshes / missing base meshes (src/parser.cpp:855, src/ply_loader.cpp:74)
            }
            for (auto &w : job.scene.warnings) job_printf(job, "%s\n", w.c_str());
            // the reference makes this check per camera, before it spawns its threads (src/main.cpp:542-545)
            if (!job.scene.cameras.empty() && (thread_count_ > 4096 || thread_count_ < 1)) fail(i, -2);
            render_scene(i, job, team);
            const auto t_tail = std::chrono::high_resolution_clock::now();
            rth::HostScene spent;           // the host arrays are released by a loader thread (half a millisecond this thread has no time for)
            std::swap(spent, job.scene);
            const double free_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_tail).count();
            {
                std::lock_guard<std::mutex> lk(m_);
                spent_scenes_.push_back(std::move(spent));
                job.done = true;
                while (flushed_ < jobs_.size() && jobs_[flushed_].done) {
                    log_printf("%s", jobs_[flushed_].log.c_str());
                    jobs_[flushed_].log.clear();
                    ++flushed_;
                }
            }
            cv_.notify_all();
            if (getenv("RT_CLI_TIMING"))
                fprintf(stderr, "[timing] file %zu: release %.1f ms, release + log flush %.1f ms (at %.0f ms)\n", i, free_ms,
                        std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_tail).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now().time_since_epoch()).count() - 1.7e12);
        }
    }

    void render_scene(size_t index, FileJob &job, Team &crew) {
        const int gpu_index = crew.devices[0];
        const std::vector<int> &team_ = crew.devices;
        const std::vector<rt_comm *> &comms_ = crew.comms;
        static const bool timing = getenv("RT_CLI_TIMING") != nullptr;   // diagnostic: stage times per file
        auto t_begin = std::chrono::high_resolution_clock::now();
        auto since = [&](std::chrono::high_resolution_clock::time_point t) {
            return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t).count();
        };
        if (hipSetDevice(gpu_index) != hipSuccess) {   // the device is a per-thread setting
            job_printf(job, "Error: cannot select GPU %d\n", gpu_index);
            fail(index, 1);
        }
        rt_scene_desc desc = job.scene.desc();
        rt_scene *gpu = nullptr;
        // A team of GPUs: the other ranks run beside this thread, each with its own replica of the scene, and meet it in
        // rt_render_gather camera by camera.  They start rendering only when every replica exists.
        const size_t n_ranks = comms_.empty() ? 1 : comms_.size();
        struct Sync {
            std::mutex m;
            std::condition_variable cv;
            size_t created = 0, failed = 0;
            std::vector<std::vector<rt_stats>> stats;   // [camera][rank], filled as the ranks finish a camera
            std::vector<size_t> reported;               // [camera]: ranks that have
            std::string error;
            bool broken = false;                        // a rank has left its camera loop with an error
        } team;
        team.stats.assign(job.scene.cameras.size(), std::vector<rt_stats>(n_ranks));
        team.reported.assign(job.scene.cameras.size(), 0);
        std::vector<std::thread> helpers;
        for (size_t r = 1; r < n_ranks; ++r)
            helpers.emplace_back([&, r] {
                rt_scene *mine = nullptr;
                bool ok = hipSetDevice(team_[r]) == hipSuccess && rt_scene_create(&desc, &mine) == 0;
                {
                    std::unique_lock<std::mutex> lk(team.m);
                    if (!ok) {
                        ++team.failed;
                        team.error = rt_last_error();
                    }
                    ++team.created;
                    team.cv.notify_all();
                    team.cv.wait(lk, [&] { return team.created == n_ranks; });
                    ok = team.failed == 0;
                }
                for (size_t k = 0; ok && k < job.scene.cameras.size(); ++k) {
                    rt_stats st{};
                    // (a frame fails for every rank of the team, each with an error of its own: this rank's, or "a peer rank
                    // failed"; all of them leave the camera loop here and the leader reports)
                    const int rc = rt_render_gather(mine, comms_[r], &job.scene.cameras[k].desc, nullptr, &st);
                    std::lock_guard<std::mutex> lk(team.m);
                    if (rc != 0) {
                        if (rc != 12) team.error = "GPU " + std::to_string(team_[r]) + ": " + rt_last_error();
                        team.broken = true;
                        ok = false;
                    } else {
                        team.stats[k][r] = st;
                        ++team.reported[k];
                    }
                    team.cv.notify_all();
                }
                if (mine) rt_scene_destroy(mine);
            });
        bool created = rt_scene_create(&desc, &gpu) == 0;
        if (!created && n_ranks == 1 && desc.n_bvh_nodes == 0 && desc.n_bvh_prims > 0) {
            // RT_GPU_BVH=1 and the device builder declined the scene (a tree deeper than the walk's stack): the host's tree
            rth::ensure_host_bvh(job.scene);
            desc = job.scene.desc();
            created = rt_scene_create(&desc, &gpu) == 0;
        }
        {
            std::unique_lock<std::mutex> lk(team.m);
            if (!created) {
                ++team.failed;
                team.error = rt_last_error();
            }
            ++team.created;
            team.cv.notify_all();
            team.cv.wait(lk, [&] { return team.created == n_ranks; });
        }
        if (team.failed) {
            for (auto &t : helpers) t.join();
            job_printf(job, "rt_scene_create failed: %s\n", team.error.c_str());
            fail(index, -1);
        }
        size_t camera_index = 0;
        auto end = std::chrono::high_resolution_clock::now();
        job_printf(job, "JSON parsing and grid preparation is done: %d ms\n",
                   (int)std::chrono::duration_cast<std::chrono::milliseconds>(end - job.start).count());
        if (timing) job_printf(job, "[timing] waited for the render thread since load start, scene upload %.1f ms\n", since(t_begin));
        for (auto &cam : job.scene.cameras) {
            job_printf(job, "%s started\n", cam.image_name.c_str());
            auto start = std::chrono::high_resolution_clock::now();
            const int W = cam.desc.resx, H = cam.desc.resy;
            const std::string &name = cam.image_name;
            const bool exr = cam.desc.is_hdr || (name.size() >= 4 && name.compare(name.size() - 4, 4, ".exr") == 0);
            const size_t n_values = (size_t)W * H * 3;
            // Only an EXR needs the float frame on the host; for PNG outputs the frame stays in HBM and is tone-mapped
            // or quantised there (rt_tonemap / rt_quantize_ldr), so 8-bit pixels are all that crosses PCIe.
            WriteTask exr_task{WriteTask::EXR_RGB32F, "outputs/" + name, W, H, {}, {}};
            float *frame = nullptr;
            if (exr) {
                exr_task.rgb32f.assign(n_values, 0.0f);
                frame = exr_task.rgb32f.data();
            } else {
                if (!crew.room_for(n_values)) {
                    job_printf(job, "cannot allocate the device framebuffer\n");
                    fail(index, -1);
                }
                frame = crew.frame;
            }
            // 8-bit pixels are made in the team's device buffer and copied out (an EXR frame is on the host: the library stages it)
            auto to_host = [&](HostPixels &pixels) { return exr || hipMemcpy(pixels.data(), crew.ldr, n_values, hipMemcpyDeviceToHost) == hipSuccess; };
            uint8_t *const ldr_out_dev = exr ? nullptr : crew.ldr;
            rt_stats stats;
            if (n_ranks == 1) {
                if (rt_render(gpu, &cam.desc, frame, &stats) != 0) {
                    job_printf(job, "rt_render failed: %s\n", rt_last_error());
                    fail(index, -1);
                }
            } else {
                if (const int rc = rt_render_gather(gpu, comms_[0], &cam.desc, frame, &stats)) {
                    const std::string own = rt_last_error();
                    for (auto &t : helpers) t.join();   // every rank has left the frame with an error
                    job_printf(job, "rt_render_gather failed: %s\n", rc == 12 && !team.error.empty() ? team.error.c_str() : own.c_str());
                    fail(index, -1);
                }
                std::unique_lock<std::mutex> lk(team.m);   // the frame is complete; the other ranks' ray counts follow
                team.cv.wait(lk, [&] { return team.reported[camera_index] == n_ranks - 1 || team.broken; });
                if (team.broken) {   // (an error behind the agreed status: the exchange itself)
                    lk.unlock();
                    for (auto &t : helpers) t.join();
                    job_printf(job, "rt_render_gather failed: %s\n", team.error.c_str());
                    fail(index, -1);
                }
                for (size_t r = 1; r < n_ranks; ++r) {
                    stats.primary_rays += team.stats[camera_index][r].primary_rays;
                    stats.shadow_rays += team.stats[camera_index][r].shadow_rays;
                    stats.secondary_rays += team.stats[camera_index][r].secondary_rays;
                    stats.kernel_ms = std::max(stats.kernel_ms, team.stats[camera_index][r].kernel_ms);
                }
            }
            ++camera_index;
            end = std::chrono::high_resolution_clock::now();
            job_printf(job, "%s rendering done: %d ms\n", name.c_str(), (int)std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count());
            const unsigned long long rays = stats.primary_rays + stats.shadow_rays + stats.secondary_rays;
            job_printf(job, "%s rays: %llu primary, %llu light samples, %llu secondary; device %.3f ms; %.1f Mrays/s\n", name.c_str(),
                       (unsigned long long)stats.primary_rays, (unsigned long long)stats.shadow_rays, (unsigned long long)stats.secondary_rays,
                       stats.kernel_ms, stats.kernel_ms > 0 ? rays / stats.kernel_ms / 1e3 : 0.0);

            std::vector<WriteTask> tasks;
            if (!exr && cam.tonemaps.empty()) {
                WriteTask t{WriteTask::PNG_RGB8, "outputs/" + name, W, H, HostPixels(n_values), {}};
                if (rt_quantize_ldr(frame, (int64_t)n_values, ldr_out_dev ? ldr_out_dev : t.rgb8.data()) != 0 || !to_host(t.rgb8)) {
                    job_printf(job, "rt_quantize_ldr failed: %s\n", rt_last_error());
                    fail(index, -1);
                }
                tasks.push_back(std::move(t));
            }
            for (auto &tm : cam.tonemaps) {
                std::string base = name;
                size_t dot = base.find_last_of('.');
                if (dot != std::string::npos) base = base.substr(0, dot);
                WriteTask t{WriteTask::PNG_RGB8, "outputs/" + base + tm.extension, W, H, HostPixels(n_values), {}};
                const rt_tonemap_desc td{tm.type, tm.opt1, tm.opt2, tm.saturation, tm.gamma};
                if (rt_tonemap(frame, W, H, &td, ldr_out_dev ? ldr_out_dev : t.rgb8.data()) != 0 || !to_host(t.rgb8)) {
                    job_printf(job, "rt_tonemap failed: %s\n", rt_last_error());
                    fail(index, -1);
                }
                tasks.push_back(std::move(t));
            }
            if (exr) {
                job_printf(job, "%s write EXR started\n", name.c_str());
                submit(std::move(exr_task));
            }
            for (auto &t : tasks) {
                if (!cam.tonemaps.empty()) job_printf(job, "Saving tone mapped image: %s\n", t.path.substr(strlen("outputs/")).c_str());
                submit(std::move(t));
            }
            if (timing) job_printf(job, "[timing] camera total %.1f ms (render + tone map + hand-off)\n", since(start));
        }
        auto t_destroy = std::chrono::high_resolution_clock::now();
        for (auto &t : helpers) t.join();
        rt_scene_destroy(gpu);
        if (timing) job_printf(job, "[timing] destroy %.1f ms, scene total on the render thread %.1f ms\n", since(t_destroy), since(t_begin));
    }

    // ---- stage 3: encode and write ---------------------------------------------------------------------------
    void submit(WriteTask &&t) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return write_queue_.size() < 16; });   // bounds the frames held in host memory
        write_queue_.push_back(std::move(t));
        lk.unlock();
        cv_.notify_all();
    }

    void write_loop() {
        for (;;) {
            WriteTask t;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !write_queue_.empty() || writers_may_stop_; });
                if (write_queue_.empty()) return;
                t = std::move(write_queue_.front());
                write_queue_.erase(write_queue_.begin());
                ++writes_in_flight_;
            }
            cv_.notify_all();
            struct Done {   // runs after the file is written
                Pipeline *p;
                ~Done() {
                    {
                        std::lock_guard<std::mutex> lk(p->m_);
                        --p->writes_in_flight_;
                    }
                    p->cv_.notify_all();
                }
            } done{this};
            const auto t_write = std::chrono::high_resolution_clock::now();
            struct Report {
                std::chrono::high_resolution_clock::time_point t0;
                const std::string &path;
                ~Report() {
                    if (getenv("RT_CLI_TIMING"))
                        fprintf(stderr, "[timing] wrote %s in %.1f ms (at %.0f ms)\n", path.c_str(),
                                std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count(),
                                std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now().time_since_epoch()).count() - 1.7e12);
                }
            } report{t_write, t.path};
            if (t.kind == WriteTask::EXR_RGB32F) {
                if (!rth::write_exr_rgb32f(t.path, t.width, t.height, t.rgb32f.data())) fprintf(stderr, "Save EXR Error: cannot write %s\n", t.path.c_str());
            } else {
                rth::write_png_rgb8(t.path, t.width, t.height, t.rgb8.data(), true);
            }
        }
    }

    int thread_count_, threading_type_;
    std::vector<int> gpus_;
    std::vector<Team> teams_, tail_teams_;   // who renders the frames in turn, and who renders the frames left over
    size_t load_ahead_ = 4;                  // files that may be loaded beyond the last one whose log has been printed
    size_t bulk_frames_ = 0;                 // frames handed to teams_; the rest goes to tail_teams_
    size_t render_limit_ = 0;                // frames that may be taken in the current phase
    std::vector<FileJob> jobs_;
    std::vector<WriteTask> write_queue_;
    std::vector<rth::HostScene> spent_scenes_;
    std::mutex m_;
    std::condition_variable cv_;
    size_t next_load_ = 0, next_render_ = 0, flushed_ = 0, writes_in_flight_ = 0;
    bool writers_may_stop_ = false;
};

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
    // RT_GPU=<index> renders on that device, RT_GPU=<a>,<b>,... with one render thread per entry (a device named twice
    // gets two threads: the dispatcher of several GPUs rehearsed on one); otherwise the files of a directory are spread
    // over every visible GPU
    std::vector<int> gpus;
    if (const char *dev = getenv("RT_GPU")) {
        for (const char *p = dev; *p;) {
            gpus.push_back(atoi(p));
            const char *comma = strchr(p, ',');
            if (!comma) break;
            p = comma + 1;
        }
        if (gpus.empty()) gpus.push_back(0);
    } else {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count < 1) count = 1;   // the first HIP call of the render thread reports the error
        for (int g = 0; g < count; ++g) gpus.push_back(g);
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
    // RT_GPUS=<k>: teams of k GPUs render a frame together, split by 32x32 tiles, the teams beside each other (k = all the
    // GPUs: every frame on the whole node, the case of a single heavy frame; a directory of light frames is served better
    // by the default, one frame per GPU).  RT_GPUS_DEVICES="0,0,0,0" names the devices instead of RT_GPU (a repeated device
    // rehearses the paths on one GPU); given alone it makes them one team.  Frames left over when fewer remain than there
    // are teams go to all GPUs regrouped (RT_TAIL_TILES=0: they go to the teams as they are, the others idle).
    size_t team_size = 1;
    if (const char *names = getenv("RT_GPUS_DEVICES")) {
        gpus.clear();
        for (const char *p = names; *p;) {
            gpus.push_back(atoi(p));
            const char *comma = strchr(p, ',');
            if (!comma) break;
            p = comma + 1;
        }
        if (gpus.empty()) gpus.push_back(0);
        team_size = gpus.size();
    }
    if (const char *n = getenv("RT_GPUS")) {
        const int want = atoi(n);
        if (want < 1 || (size_t)want > gpus.size()) {
            log_printf("Error: RT_GPUS=%s but %zu GPU(s) are there to render\n", n, gpus.size());
            return 1;
        }
        team_size = (size_t)want;
    }
    const char *tail = getenv("RT_TAIL_TILES");
    const auto t_run = std::chrono::high_resolution_clock::now();
    {
        Pipeline pipeline(files, thread_count, tt, gpus, team_size, !(tail && atoi(tail) == 0));
        if (!pipeline.prepare()) return 1;
        pipeline.run();
    }
    if (getenv("RT_CLI_TIMING"))
        log_printf("[timing] %zu file(s) in %.1f ms\n", files.size(),
                   std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_run).count());
    create_videos_from_png_sequences("outputs");
    return 0;
}
