// Host-side scene container: the flat arrays handed to the rendering core through rt_scene_desc,
// plus the per-camera output description (image name, tone-map operators) that stays on the host.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "gpu_render.h"

namespace rth {

struct Tonemap {          // reference src/camera.h:13-21
    int type = RT_TMO_PHOTOGRAPHIC;
    float opt1 = 0.f, opt2 = 0.f;
    float saturation = 1.f;
    float gamma = 2.2f;
    std::string extension;
};

struct HostCamera {
    rt_camera_desc desc{};
    std::string image_name;
    std::vector<Tonemap> tonemaps;
};

struct PrimBounds {       // world-space padded box of one (shape, placement) pair
    float bmin[3], bmax[3];
};

struct SceneMapping;   // a scene cache file mapped into memory (scene_cache.cpp)

struct HostScene {
    std::vector<rt_triangle> triangles;
    std::vector<rt_instance> instances;
    std::vector<rt_sphere> spheres;
    std::vector<rt_plane> planes;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<int32_t> texture_refs;
    std::vector<rt_image> images;
    std::vector<uint8_t> texels_u8;
    std::vector<float> texels_f32;
    std::vector<rt_light> lights;
    std::vector<rt_bvh_node> bvh_nodes;
    std::vector<rt_bvh_prim> bvh_prims;
    std::vector<HostCamera> cameras;

    // BVH build input in the reference's enumeration order (src/bvh.cpp:55-73)
    std::vector<rt_bvh_prim> build_prims;
    std::vector<PrimBounds> build_bounds;

    float intersection_epsilon = 1e-6f;
    float shadow_ray_epsilon = 1e-3f;
    float max_depth = 6.f;
    float ambient_light[3] = {0, 0, 0};
    float background_color[3] = {0, 0, 0};
    int32_t background_texture = -1;
    int32_t is_probe = 0;

    double ingest_ms = 0, bvh_ms = 0;
    std::vector<std::string> warnings;

    // files the scene pulled in (PLY meshes, images), for the scene cache
    std::vector<std::string> dependencies;
    // A scene that came out of the scene cache keeps its arrays in the mapped file instead of the vectors above: `mapping`
    // holds the file, `mapped_desc` points into it.
    std::shared_ptr<SceneMapping> mapping;
    rt_scene_desc mapped_desc{};

    rt_scene_desc desc() const;
};

// Parses a scene file of the reference's JSON dialect (src/parser.cpp) including PLY meshes
// (src/ply_loader.cpp) and builds the BVH.  Returns false and fills `err` on failure.
bool load_scene(const std::string &json_path, HostScene &out, std::string &err);

// The scene cache (RT_SCENE_CACHE=<directory>): load_scene tries scene_cache_load first and stores what it built.
bool scene_cache_load(const std::string &json_path, HostScene &out);
void scene_cache_store(const std::string &json_path, const HostScene &scene);

// SAH BVH build over build_prims/build_bounds (reference src/bvh.cpp:41-344).
void build_bvh(HostScene &scene);
// Builds the tree of a scene that was loaded without one (RT_GPU_BVH=1); no-op otherwise.
void ensure_host_bvh(HostScene &scene);

// PLY reader (reference src/ply_loader.cpp:63-433).  Appends vertices to `vertices` and
// triangles to scene.triangles.
struct PlyResult {
    size_t vertex_count = 0, tri_count = 0;
    bool has_normals = false, has_uvs = false;
};
bool load_ply(const std::string &path, std::vector<float> &vertices, HostScene &scene, bool smooth_shading, PlyResult &res,
              std::string &err);

// Image decoding for textures: PNG / JPEG (8-bit RGB) and EXR (float RGBA).
bool load_image_file(const std::string &path, HostScene &scene, rt_image &out, std::string &err);

}  // namespace rth
