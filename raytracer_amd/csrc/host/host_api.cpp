// C entry points of the host library (see include/rt_host.h).
#include "rt_host.h"

#include <algorithm>
#include <cstring>
#include <string>

#include "image_io.h"
#include "scene.h"
#include "tonemap.h"

struct rth_scene {
    rth::HostScene scene;
    rt_scene_desc desc;
};

static thread_local std::string g_error;

extern "C" {

int rth_scene_load(const char *json_path, rth_scene **out) {
    if (!json_path || !out) {
        g_error = "null argument";
        return 1;
    }
    auto *s = new rth_scene();
    std::string err;
    if (!rth::load_scene(json_path, s->scene, err)) {
        g_error = err;
        delete s;
        return 2;
    }
    s->desc = s->scene.desc();
    *out = s;
    return 0;
}

const rt_scene_desc *rth_scene_desc(const rth_scene *s) { return &s->desc; }
void rth_scene_build_bvh(rth_scene *s) {
    rth::ensure_host_bvh(s->scene);
    s->desc = s->scene.desc();
}
int32_t rth_scene_camera_count(const rth_scene *s) { return (int32_t)s->scene.cameras.size(); }
const rt_camera_desc *rth_scene_camera(const rth_scene *s, int32_t i) {
    if (i < 0 || i >= (int32_t)s->scene.cameras.size()) return nullptr;
    return &s->scene.cameras[i].desc;
}
const char *rth_scene_image_name(const rth_scene *s, int32_t i) {
    if (i < 0 || i >= (int32_t)s->scene.cameras.size()) return "";
    return s->scene.cameras[i].image_name.c_str();
}
int32_t rth_scene_tonemap_count(const rth_scene *s, int32_t camera) {
    if (camera < 0 || camera >= (int32_t)s->scene.cameras.size()) return 0;
    return (int32_t)s->scene.cameras[camera].tonemaps.size();
}
const char *rth_scene_tonemap_extension(const rth_scene *s, int32_t camera, int32_t index) {
    if (camera < 0 || camera >= (int32_t)s->scene.cameras.size()) return "";
    const auto &tms = s->scene.cameras[camera].tonemaps;
    if (index < 0 || index >= (int32_t)tms.size()) return "";
    return tms[index].extension.c_str();
}
void rth_scene_timings(const rth_scene *s, double *ingest_ms, double *bvh_ms) {
    if (ingest_ms) *ingest_ms = s->scene.ingest_ms;
    if (bvh_ms) *bvh_ms = s->scene.bvh_ms;
}
void rth_scene_free(rth_scene *s) { delete s; }

int rth_tonemap(int32_t tmo, float opt1, float opt2, float saturation, float gamma, int32_t width, int32_t height, const float *hdr,
                uint8_t *out_rgb8) {
    rth::apply_tonemap(tmo, opt1, opt2, saturation, gamma, width, height, hdr, out_rgb8);
    return 0;
}
int rth_scene_tonemap_apply(const rth_scene *s, int32_t camera, int32_t index, const float *hdr, uint8_t *out_rgb8) {
    if (camera < 0 || camera >= (int32_t)s->scene.cameras.size()) return 1;
    const auto &cam = s->scene.cameras[camera];
    if (index < 0 || index >= (int32_t)cam.tonemaps.size()) return 1;
    const auto &tm = cam.tonemaps[index];
    rth::apply_tonemap(tm.type, tm.opt1, tm.opt2, tm.saturation, tm.gamma, cam.desc.resx, cam.desc.resy, hdr, out_rgb8);
    return 0;
}
void rth_quantize_ldr(int64_t n, const float *hdr, uint8_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)std::clamp(hdr[i] * 255.0f, 0.0f, 255.0f);
}
int rth_write_png(const char *path, int32_t w, int32_t h, const uint8_t *rgb8, int32_t flip) {
    return rth::write_png_rgb8(path, w, h, rgb8, flip != 0) ? 0 : 1;
}
int rth_write_exr(const char *path, int32_t w, int32_t h, const float *rgb) { return rth::write_exr_rgb32f(path, w, h, rgb) ? 0 : 1; }
int rth_read_exr(const char *path, int32_t *w, int32_t *h, float *rgba) {
    std::vector<float> data;
    std::string err;
    int ww = 0, hh = 0;
    if (!rth::read_exr_rgba(path, ww, hh, data, err)) {
        g_error = err;
        return 1;
    }
    *w = ww;
    *h = hh;
    if (rgba) memcpy(rgba, data.data(), data.size() * sizeof(float));
    return 0;
}
int rth_decode_image_rgb8(const uint8_t *file, int64_t n_bytes, int32_t *w, int32_t *h, uint8_t *rgb8) {
    std::vector<uint8_t> bytes(file, file + n_bytes), rgb;
    std::string err;
    int ww = 0, hh = 0;
    if (!rth::decode_image_rgb8(bytes, ww, hh, rgb, err)) {
        g_error = err;
        return 1;
    }
    *w = ww;
    *h = hh;
    if (rgb8) memcpy(rgb8, rgb.data(), rgb.size());
    return 0;
}
const char *rth_last_error(void) { return g_error.c_str(); }

}  // extern "C"
