/*
 * rt_host.h — C entry points of the host library (scene ingest, BVH build, tone mapping, image
 * files).  These wrap the parts of the reference that stay on the host: parser (src/parser.cpp),
 * ply_loader (src/ply_loader.cpp), bvh::bvh (src/bvh.cpp:41-344), apply_tonemap
 * (src/main.cpp:267-345) and the PNG/EXR writers (src/main.cpp:561-607).
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "gpu_render.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;

/* Parses a scene JSON file (and the PLY / image files it names) and builds the BVH. */
int rth_scene_load(const char *json_path, rth_scene **out);
/* Flat view of the scene; valid until rth_scene_free. */
const rt_scene_desc *rth_scene_desc(const rth_scene *s);
int32_t rth_scene_camera_count(const rth_scene *s);
const rt_camera_desc *rth_scene_camera(const rth_scene *s, int32_t index);
const char *rth_scene_image_name(const rth_scene *s, int32_t index);
int32_t rth_scene_tonemap_count(const rth_scene *s, int32_t camera);
/* The file-name extension of tone map `index` of `camera` (the scene file's Tonemap.Extension, e.g. "_phot.png"). */
const char *rth_scene_tonemap_extension(const rth_scene *s, int32_t camera, int32_t index);
/* Builds the SAH tree of a scene that was loaded without one (RT_GPU_BVH=1 leaves the tree to the rendering core, which
 * builds it on the device); no-op when the scene has its tree.  rth_scene_desc changes with it. */
void rth_scene_build_bvh(rth_scene *s);
/* ingest and BVH-build wall time in milliseconds */
void rth_scene_timings(const rth_scene *s, double *ingest_ms, double *bvh_ms);
void rth_scene_free(rth_scene *s);

/* Tone-maps a bottom-row-first float RGB buffer into 8-bit RGB (same row order). */
int rth_tonemap(int32_t tmo, float opt1, float opt2, float saturation, float gamma, int32_t width, int32_t height,
                const float *hdr, uint8_t *out_rgb8);
/* Applies tone-map operator `index` of `camera`. */
int rth_scene_tonemap_apply(const rth_scene *s, int32_t camera, int32_t index, const float *hdr, uint8_t *out_rgb8);
/* Plain LDR conversion of the reference: (unsigned char)clamp(v*255, 0, 255). */
void rth_quantize_ldr(int64_t n_values, const float *hdr, uint8_t *out);
int rth_write_png(const char *path, int32_t width, int32_t height, const uint8_t *rgb8, int32_t flip_vertically);
int rth_write_exr(const char *path, int32_t width, int32_t height, const float *rgb);
/* Reads an EXR into caller memory of width*height*4 floats; pass rgba=NULL to query the size. */
int rth_read_exr(const char *path, int32_t *width, int32_t *height, float *rgba);
/* Decodes an 8-bit texture file held in memory (PNG, else JPEG) to RGB, as the reference's
 * stbi_load(path, &w, &h, &n, 3) does (reference src/texture.cpp:46); pass rgb8=NULL to query the size. */
int rth_decode_image_rgb8(const uint8_t *file, int64_t n_bytes, int32_t *width, int32_t *height, uint8_t *rgb8);

const char *rth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
