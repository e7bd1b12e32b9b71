/*
 * gpu_render.h — C ABI of the MI355X rendering core.
 *
 * The reference renderer has no plugin/FFI layer; the seam this header defines is the body of the
 * per-camera loop of process_file() (reference src/main.cpp:428-559): everything between
 * "hdr_output allocated" (:435) and "rendering done" (:557-559).  The host program ingests the
 * JSON/PLY scene, builds the SAH BVH, flattens everything into the plain arrays below and calls:
 *
 *     rt_scene_create   replaces  `new ray_tracer(...)`          (src/main.cpp:421, src/ray_trace.h:53-65)
 *     rt_render         replaces  the std::thread scheduler block (src/main.cpp:437-555) and all it calls
 *     rt_scene_destroy  replaces  `delete rt`                     (src/main.cpp:625)
 *
 * All structs are plain C, pointers + counts, 4-byte scalars only.  Matrices are the first three
 * rows of a row-major 4x4 (12 floats); the 4th row of an affine matrix is never read on the path.
 */
#ifndef GPU_RENDER_H
#define GPU_RENDER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3

/* ---- enums (values follow the reference's enum order) ---------------------------------------- */
enum { RT_MAT_REGULAR = 0, RT_MAT_MIRROR = 1, RT_MAT_CONDUCTOR = 2, RT_MAT_DIELECTRIC = 3 }; /* src/shapes.h:20-26 */
enum { RT_BRDF_ORIGINAL_BLINN_PHONG = 0, RT_BRDF_ORIGINAL_PHONG = 1, RT_BRDF_MODIFIED_BLINN_PHONG = 2,
       RT_BRDF_MODIFIED_PHONG = 3, RT_BRDF_TORRANCE_SPARROW = 4 };                           /* src/shapes.h:37-44 */
enum { RT_DECAL_REPLACE_KD = 0, RT_DECAL_BLEND_KD = 1, RT_DECAL_REPLACE_KS = 2, RT_DECAL_REPLACE_BACKGROUND = 3,
       RT_DECAL_REPLACE_NORMAL = 4, RT_DECAL_BUMP_NORMAL = 5, RT_DECAL_REPLACE_ALL = 6 };    /* src/texture.h:19-28 */
enum { RT_INTERP_NEAREST = 0, RT_INTERP_BILINEAR = 1 };                                      /* src/texture.h:30-34 */
enum { RT_NOISE_ABSVAL = 0, RT_NOISE_LINEAR = 1 };                                           /* src/texture.h:36-40 */
enum { RT_TEX_IMAGE = 0, RT_TEX_CHECKER = 1, RT_TEX_PERLIN = 2 };
enum { RT_LIGHT_POINT = 0, RT_LIGHT_AREA = 1, RT_LIGHT_DIRECTIONAL = 2, RT_LIGHT_SPOT = 3,
       RT_LIGHT_ENV = 4, RT_LIGHT_TRIANGLE = 5, RT_LIGHT_SPHERE = 6 };                       /* src/lights.h */
enum { RT_TMO_PHOTOGRAPHIC = 0, RT_TMO_FILMIC = 1, RT_TMO_ACES = 2 };                        /* src/camera.h:6-11 */

/* ---- scene records ---------------------------------------------------------------------------- */

/* material (src/shapes.h:46-61) */
typedef struct rt_material {
    int32_t type;        /* RT_MAT_* */
    int32_t brdf;        /* RT_BRDF_* */
    int32_t normalized;
    int32_t kdfresnel;
    float ambient[3];
    float diffuse[3];
    float specular[3];
    float phong_exponent;
    float mirror[3];
    float absorption_coefficient[3];
    float refraction_index;
    float absorption_index;
    float roughness;
} rt_material;

/* one mesh placement = one entry of all_mesh_infos::mesh_infos (src/shapes.h:70-84) */
typedef struct rt_instance {
    float model[12];
    float inv_model[12];
    float normal[12];      /* transpose(inverse(model)), rows 0..2 */
    float motionblur[3];
    float emission[3];
    int32_t material;      /* index into materials */
    int32_t tex_offset;    /* range in texture_refs */
    int32_t tex_count;
    int32_t ref_id;        /* the reference's instance id: mesh _id, MeshInstance _id, or 123456 */
    int32_t identity;      /* 1 when model is exactly the identity.  inv_model is then k I, with k what the caller's inverse makes
                            * of 1 / 1 (the reference's: 1 - 2^-24, src/algebra.cpp:345-347), and a triangle test multiplies the
                            * ray's origin and direction by k in place of the matrix product (the same values, src/shapes.cpp:224-230).
                            * One k per scene: the first such placement's inv_model[0]; placements whose inverse is another
                            * matrix take the general way. */
} rt_instance;

/* triangle in OBJECT space (src/shapes.h:110-149).  The reference intersects in object space with
 * vertices scaled by 1000 (src/shapes.cpp:212-279); the scaled values are derived by the core. */
typedef struct rt_triangle {
    float c1[3], c2[3], c3[3];
    float n1[3], n2[3], n3[3];   /* per-vertex normals when smooth */
    float tex_u[3], tex_v[3];    /* per-vertex texture coordinates */
    int32_t smooth;
} rt_triangle;

/* sphere (src/shapes.h:151-184) */
typedef struct rt_sphere {
    float center[3];
    float radius;
    float model[12];
    float inv_model[12];
    float normal[12];
    float emission[3];
    int32_t material;
    int32_t tex_offset;
    int32_t tex_count;
} rt_sphere;

/* unbounded plane, already in world space (src/shapes.h:186-204, src/parser.cpp:1020-1067) */
typedef struct rt_plane {
    float point[3];
    float normal[3];       /* transformed and normalised */
    float raw_normal_x;    /* x of the normal as written in the scene file: the reference's helper-axis choices for
                              plane texture coordinates, tangents and hemisphere bases read this value, because its
                              vec3 keeps the parsed scalars next to the transformed SIMD value (src/shapes.cpp:533,554,
                              src/ray_trace.cpp:45) */
    int32_t material;
    int32_t tex_offset;
    int32_t tex_count;
} rt_plane;

/* decoded image (src/texture.h:4-17): 8-bit RGB (3 bytes/texel) or float RGBA (4 floats/texel) */
typedef struct rt_image {
    int32_t width, height;
    int32_t is_hdr;
    int32_t offset;        /* texel offset: bytes into texels_u8 or floats into texels_f32 */
} rt_image;

/* texture map (src/texture.h:42-62) */
typedef struct rt_texture {
    int32_t kind;          /* RT_TEX_* */
    int32_t image;         /* index into images or -1 */
    int32_t decal_mode;    /* RT_DECAL_* */
    int32_t interp;        /* RT_INTERP_* */
    float bump_factor;
    float noise_scale;
    int32_t noise_conversion;
    float num_octaves;
    float black[3];
    float white[3];
    float scale;
    float offset;
} rt_texture;

/* light (src/lights.h).  Field use by type:
 *   POINT       position, color=intensity
 *   AREA        position, normal (as parsed, not normalised), u, v (basis), size, color=radiance, samples
 *   DIRECTIONAL direction (normalised), color=radiance
 *   SPOT        position, direction (normalised), color=intensity, cos_coverage, cos_falloff
 *   ENV         image, cosine_sampling, is_probe
 *   TRIANGLE    v0,v1,v2 = world vertices, normal, area, color=radiance            (samples 16)
 *   SPHERE      position=centre (object space), size=radius, sphere = index of the rt_sphere,
 *               area = 4*pi*|model*(r,0,0)|^2, color=radiance                       (samples 16)
 */
typedef struct rt_light {
    int32_t type;
    int32_t samples;
    float position[3];
    float direction[3];
    float normal[3];
    float u[3], v[3];
    float v0[3], v1[3], v2[3];
    float color[3];
    float size;
    float area;
    float cos_coverage, cos_falloff;
    int32_t image;
    int32_t cosine_sampling;
    int32_t is_probe;
    int32_t sphere;
} rt_light;

/* flattened BVH in the reference's depth-first layout (src/bvh.h:12-28, src/bvh.cpp:316-343):
 * the left child of interior node i is i+1, the right child is `offset`. */
typedef struct rt_bvh_node {
    float bmin[3];
    float bmax[3];
    int32_t offset;        /* leaf: first entry in prims; interior: index of the second child */
    int32_t n_prims;       /* 0 for interior nodes */
    int32_t axis;          /* split axis of interior nodes */
} rt_bvh_node;

/* leaf entry = (shape, placement) pair (src/bvh.h:24-28).  Shapes are numbered in the order of
 * parser::get_shapes(): all triangles, then spheres, then planes. */
typedef struct rt_bvh_prim {
    int32_t shape;         /* < n_triangles: triangle; else sphere (shape - n_triangles) */
    int32_t instance;      /* index into instances for triangles, -1 for spheres */
} rt_bvh_prim;

typedef struct rt_scene_desc {
    int32_t abi_version;

    int32_t n_triangles;   const rt_triangle *triangles;
    int32_t n_instances;   const rt_instance *instances;
    int32_t n_spheres;     const rt_sphere *spheres;
    int32_t n_planes;      const rt_plane *planes;
    int32_t n_materials;   const rt_material *materials;
    int32_t n_textures;    const rt_texture *textures;
    int32_t n_texture_refs; const int32_t *texture_refs;   /* per-object texture index lists */
    int32_t n_images;      const rt_image *images;
    int64_t n_texels_u8;   const uint8_t *texels_u8;
    int64_t n_texels_f32;  const float *texels_f32;
    int32_t n_lights;      const rt_light *lights;

    int32_t n_bvh_nodes;   const rt_bvh_node *bvh_nodes;
    int32_t n_bvh_prims;   const rt_bvh_prim *bvh_prims;
    /* A scene may come without a tree: n_bvh_nodes = 0, bvh_prims = every (shape, placement) pair in any order and
     * bvh_prim_bounds = their world-space boxes (6 floats each: min xyz, max xyz; the padded boxes the reference's builder
     * starts from, src/bvh.cpp:55-73).  The library then builds a tree on the device (Morton order, one leaf per pair).
     * Such a tree is not the reference's: the closest hit is the same, hits at exactly equal distance may resolve to
     * another primitive.  NULL when the tree is given. */
    const float *bvh_prim_bounds;

    float intersection_epsilon;   /* src/parser.cpp:1239-1244 */
    float shadow_ray_epsilon;     /* src/parser.cpp:1246-1251 */
    float max_depth;              /* src/parser.cpp:1253-1258 (kept as float, compared with int depth) */
    float ambient_light[3];
    float background_color[3];
    int32_t background_texture;   /* index into textures, or -1 (src/main.cpp:380-417) */
    int32_t is_probe;
} rt_scene_desc;

/* post-constructor camera state (src/camera.h:35-56) + RenderSettings (src/camera.h:23-33) */
typedef struct rt_camera_desc {
    float position[3];
    float u[3], v[3], w[3];
    float center[3];
    float near_left, near_right, near_bottom, near_top;
    int32_t resx, resy;
    int32_t num_samples;
    float focus_distance;
    float aperture_size;
    int32_t is_hdr;
    int32_t path_tracing;
    int32_t importance_sampling;
    int32_t next_event_estimation;
    int32_t russian_roulette;
    int32_t splitting_factor;
    float sample_max_val;
    int32_t min_recursion_depth;
    uint32_t seed;           /* the reference's RNG is unseedable; the core's is counter based */
} rt_camera_desc;

typedef struct rt_stats {
    uint64_t primary_rays;
    uint64_t shadow_rays;        /* light samples with non-zero incident radiance = shadow rays of the reference's loop.  Exact
                                    with the work counters on (rt_set_counters); with them off, the samples of a sphere emitter
                                    are evaluated only at shading points its shadow ray reaches, and only those are counted */
    uint64_t secondary_rays;
    uint64_t nodes_visited;      /* BVH nodes whose box was tested */
    uint64_t triangle_tests;
    uint64_t sphere_tests;
    uint64_t plane_tests;
    uint64_t shaded_hits;
    uint64_t texel_fetches;
    float kernel_ms;             /* device time of the render kernels of the last call */
    float total_ms;              /* wall time of the call including copies */
    uint64_t shadow_rays_traced; /* any-hit traversals run: light samples of a shading point whose rays are identical bit for
                                    bit share one traversal (with the work counters on none do: then it equals shadow_rays) */
    uint64_t any_hit_nodes_visited;   /* the any-hit (shadow) share of nodes_visited ... plane_tests */
    uint64_t any_hit_triangle_tests;
    uint64_t any_hit_sphere_tests;
    uint64_t any_hit_plane_tests;
} rt_stats;

/* kernels of the render path, in launch order within a batch (indices of rt_kernel_times' output) */
enum { RT_KERNEL_GENERATE = 0, RT_KERNEL_TRACE_CLOSEST = 1, RT_KERNEL_SHADE = 2, RT_KERNEL_LIGHT_SETUP = 3,
       RT_KERNEL_TRACE_ANY = 4, RT_KERNEL_LIGHT_GATHER = 5, RT_KERNEL_RESOLVE = 6, RT_KERNEL_COUNT = 7 };
typedef struct rt_kernel_time {
    uint64_t launches;
    double total_ms;     /* sum over the launches of (end event - begin event) on the library's stream */
} rt_kernel_time;

typedef struct rt_scene rt_scene;

/* Copies every array of `desc` to the current HIP device; the caller may free them on return.
 * Returns 0 on success. */
int rt_scene_create(const rt_scene_desc *desc, rt_scene **out);

/* Renders one camera synchronously.  hdr_out: caller-owned host or device memory, resx*resy*3
 * floats, interleaved RGB, row 0 = bottom image row (index = j*resx + i, src/main.cpp:453).
 * stats may be NULL.  Returns 0 on success. */
int rt_render(rt_scene *scene, const rt_camera_desc *cam, float *hdr_out, rt_stats *stats);

/* Image-space shard of one frame (32x32 tiles, the reference's square_dynamic tile size,
 * src/main.cpp:441): renders only the tiles owned by `rank` of `world` (rt_shard_owner_map) into hdr_out at their
 * full-frame positions; other pixels are left untouched. */
int rt_render_shard(rt_scene *scene, const rt_camera_desc *cam, int32_t rank, int32_t world, float *hdr_out,
                    rt_stats *stats);

/* Host-only helper for the gather step: owner_out[ty*tiles_x + tx] = rank that renders tile (tx, ty) when a
 * resx x resy frame is split over `world` ranks ((tx + skew*ty) mod world with a skew coprime to world, so that the ranks
 * alternate along rows and columns of tiles and every rank's share is spread over the whole image); tiles_x = ceil(resx/32), tiles_y = ceil(resy/32), tile row 0 is the bottom of the
 * image.  Touches no GPU.  (The reference hands tiles to threads dynamically, src/main.cpp:486-512; a static map is
 * what lets every rank know the layout without communication.) */
int rt_shard_owner_map(int32_t resx, int32_t resy, int32_t world, int32_t *owner_out);

/* Host-only test hook (no reference counterpart: the reference has no queues).  Inside a device a batch of camera
 * samples is dealt to 64 queue shards, one workgroup = one sample of a quarter tile at a time: workgroup `q` of sample `s`
 * goes to shard (q + 37 s) mod 64.  rt_queue_shard_of returns that shard; rt_queue_fill_bound the most workgroups one
 * shard receives for `groups_per_sample` workgroups per sample and `samples` samples, which is what the batch planner
 * checks against the room of a shard.  Touch no GPU. */
int32_t rt_queue_shard_of(uint32_t group_of_sample, uint32_t sample);
uint64_t rt_queue_fill_bound(uint32_t groups_per_sample, uint32_t samples);

/* Test hook: 1 when the scene's shadow rays are walked in any order.  A shadow ray (the any-hit case of bvh::intersect,
 * src/bvh.cpp:383-480, as calculate_color uses it, src/ray_trace.cpp:232-241) gets the same answer in every order
 * unless a surface can lie within the shadow-ray epsilon of a light sample; rt_scene_create decides that from the lights
 * and the tree, and keeps the reference's order where it cannot rule it out (RT_ANY_ORDERED=1: always). */
int32_t rt_scene_shadow_order_free(const rt_scene *scene);
/* The same decision from a scene description alone (host only, no GPU needed; RT_ANY_ORDERED is not consulted). */
int32_t rt_shadow_order_free(const rt_scene_desc *desc);
/* Test hook: 1 when the scene is shaded by the build of the shading kernel that holds only what a scene of regular
 * materials without textures runs of calculate_color, trace_rec and path_trace (src/ray_trace.cpp:154-447, 646-869, 871-1139):
 * no texture lookups, no mirror, conductor or dielectric branch.  RT_PLAIN_SHADING=0 at rt_scene_create: never. */
int32_t rt_scene_plain_shading(const rt_scene *scene);
/* Test hook: 1 when the scene holds the four-box records that uncounted frames walk instead of bvh::intersect's node-by-node
 * records (src/bvh.cpp:383-480) -- made on the device once the scene has traced enough rays to repay them (RT_WIDE_NODES=1: at
 * rt_scene_create), or with the tree when that is built on the device; RT_WIDE_NODES=0: never. */
int32_t rt_scene_wide_records(const rt_scene *scene);
/* Test hook: runs the builder of the four-box records on a bare node array (n_nodes >= 3, nodes[0] interior; the layout of
 * bvh::nodes, src/bvh.h:9-22) with a walk's stack of `stack_entries`.  Returns the number of records the tree gets, -1 when
 * the builder declines it (a walk could need more stack, or a box is not finite: such a scene renders from the node-by-node
 * records), -2 on a malformed array or an error. */
int64_t rt_selftest_wide_records(const rt_bvh_node *nodes, int32_t n_nodes, int32_t stack_entries);

/* ---- one frame on several GPUs (src/main.cpp:437-555 with a GPU in the place of a thread) -----------------------------
 * Every rank holds a replica of the scene (rt_scene_create on its device) and a communicator.  rt_render_gather renders
 * the rank's tiles into a packed tile-major buffer, moves the buffers to rank 0 in one grouped RCCL send / receive and,
 * on rank 0, writes the frame to hdr_out (host or device memory, resx*resy*3 floats; ignored on the other ranks, may be
 * NULL there).  All ranks of a communicator must make the call for the same camera; it returns when the rank's part is
 * done.  The frame equals rt_render's (random numbers are keyed by absolute pixel).
 *   rt_comm_init_all    ranks are threads of this process, rank r on devices[r] (ncclCommInitAll); call
 *                       rt_render_gather from one thread per rank.  Ranks that share a device (a rehearsal on one GPU)
 *                       exchange by device-to-device copies instead of RCCL, which allows one rank per device.
 *   rt_comm_unique_id + rt_comm_init_rank
 *                       ranks are processes: rank 0 makes the id (RT_COMM_ID_BYTES bytes), the launcher hands it to
 *                       every rank, each calls rt_comm_init_rank with its device current (ncclCommInitRank).
 * A frame fails as a whole.  Before the tiles travel the ranks agree on one status word (an all-reduce of one integer), so
 * a rank that could not render its tiles still takes part in every step the others wait in: it returns its own error,
 * every other rank returns 12 ("a peer rank failed"), and no rank is left waiting.  The communicator stays usable.
 * RCCL is opened at run time (dlopen) when the first communicator is made; RT_RCCL_LIB=<path> names the library to open
 * in place of the one beside the HIP runtime. */
/* Host-only: the layout of the gather.  tiles_of_rank_out[r] = tiles rank r renders; tile_slot_out[ty*tiles_x + tx] = the
 * tile's slot (units of 32*32*3 floats) in the root's receive buffer, where the tiles of rank r follow those of ranks
 * 0 .. r-1 and a rank's tiles stand in row-major order, the order in which it packs them.  Touches no GPU. */
int rt_shard_layout(int32_t resx, int32_t resy, int32_t world, int32_t *tiles_of_rank_out, int32_t *tile_slot_out);

#define RT_COMM_ID_BYTES 128
typedef struct rt_comm rt_comm;
/* 1 when the RCCL entry points the communicators need could be opened in this process (no GPU work, no communication):
 * ranks can agree on this before any of them enters rt_comm_init_rank, which -- like ncclCommInitRank -- returns only when
 * every rank has made the call. */
int32_t rt_comm_available(void);
int rt_comm_unique_id(void *id_out, int32_t id_bytes);
int rt_comm_init_rank(const void *id, int32_t id_bytes, int32_t rank, int32_t world, rt_comm **out);
int rt_comm_init_all(int32_t n, const int32_t *devices, rt_comm **comms_out);
int rt_render_gather(rt_scene *scene, rt_comm *comm, const rt_camera_desc *cam, float *hdr_out, rt_stats *stats);
void rt_comm_destroy(rt_comm *comm);

/* Primary visibility only (bvh::intersect for each pixel's first sample, src/bvh.cpp:383-480):
 * shape index, reference instance id (-1/-1 on a miss) and hit distance (1e30 on a miss); moving objects
 * are taken at time 0. */
int rt_trace_primary(rt_scene *scene, const rt_camera_desc *cam, int32_t *shape_out, int32_t *id_out, float *t_out,
                     rt_stats *stats);

/* Per-kernel timing of subsequent renders: every kernel launch is bracketed by two HIP events on the stream the
 * library launches on.  Calling it (with either value) clears the totals.  rt_kernel_times fills
 * out[RT_KERNEL_COUNT].  No reference counterpart (the reference times whole frames, src/main.cpp:557-559). */
int rt_set_profiling(rt_scene *scene, int32_t enabled);
int rt_kernel_times(const rt_scene *scene, rt_kernel_time *out);

/* Enables per-ray traversal counters (nodes visited, primitive tests) in subsequent renders.  Counted renders walk the
 * BVH node by node as bvh::intersect does (src/bvh.cpp:383-480), so the counts are the reference's; renders without
 * counters may walk a wide form of the same tree (four quantised boxes per record, leaves filtered by their exact box)
 * that tests the same primitives in the same order and returns the same hits.  No reference counterpart (the reference
 * does not count). */
int rt_set_counters(rt_scene *scene, int32_t enabled);

void rt_scene_destroy(rt_scene *scene);

/* Self-test hook: evaluates arithmetic definition `fn` of the core on the device for n element
 * pairs (0 sin, 1 cos, 2 asin, 3 acos, 4 atan2, 5 exp, 6 log, 7 pow, 8 divide, 9 sqrt, 10 random
 * number from (key, dimension) bit patterns, 11 x*y+x unfused).  Used to check that device
 * arithmetic is bit-identical to a strict IEEE-754 evaluation on the host. */
int rt_selftest_math(int32_t fn, int32_t n, const float *a, const float *b, float *out);

/* Tone mapping of a finished frame on the device: apply_tonemap (src/main.cpp:267-345) -- log-average luminance,
 * exposure opt1 / average, burn-percent (opt2) white point as an exact order statistic, photographic / filmic /
 * ACES curve, saturation (+0.2 bias) and gamma (0 = 2.2) -- into 8-bit RGB in the same pixel order.  hdr and
 * rgb8_out may each be host or device memory; with both on the device the frame never crosses PCIe as floats. */
typedef struct rt_tonemap_desc {
    int32_t tmo;         /* RT_TMO_* */
    float opt1, opt2;    /* TMOOptions: key value, burn percent */
    float saturation;
    float gamma;
} rt_tonemap_desc;
int rt_tonemap(const float *hdr, int32_t width, int32_t height, const rt_tonemap_desc *tm, uint8_t *rgb8_out);

/* rt_tonemap for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc):
 * no counterpart in the reference.  The same kernels and the same result, byte for byte, as rt_tonemap; what rt_tonemap
 * allocates per call lives in a fixed array of device memory inside the library instead (9.3 KB per GPU: the partial
 * sums of the log-average, the state and histogram of the white-point selection), and the luminance plane is not kept
 * but re-evaluated.  hdr and rgb8_out must both be device memory on the current device; at most
 * RT_TONEMAP_NOALLOC_MAX_PIXELS pixels (error 5 beyond).  The call allocates and frees nothing, synchronises the device
 * before it returns, and every call on a GPU uses that GPU's one array: calls that overlap on the same GPU race. */
#define RT_TONEMAP_NOALLOC_MAX_PIXELS (1 << 28)
int rt_tonemap_noalloc(const float *hdr, int32_t width, int32_t height, const rt_tonemap_desc *tm, uint8_t *rgb8_out);

/* The reference's plain LDR conversion, (unsigned char)clamp(v * 255, 0, 255) (src/main.cpp:579-583), on the device. */
int rt_quantize_ldr(const float *hdr, int64_t n_values, uint8_t *out);

/* Message of the last failing call on this thread ("" if none).  The reference's convention is
 * print + exit(code) (src/main.cpp:544,654; src/parser.cpp:855); the ABI returns nonzero instead. */
const char *rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
