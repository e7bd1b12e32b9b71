"""ctypes mirror of include/gpu_render.h and include/rt_host.h (field order and types must match)."""
import ctypes as C

f32, i32, u32, i64, u64 = C.c_float, C.c_int32, C.c_uint32, C.c_int64, C.c_uint64
F3 = f32 * 3
F12 = f32 * 12


class Material(C.Structure):
    _fields_ = [("type", i32), ("brdf", i32), ("normalized", i32), ("kdfresnel", i32), ("ambient", F3), ("diffuse", F3),
                ("specular", F3), ("phong_exponent", f32), ("mirror", F3), ("absorption_coefficient", F3),
                ("refraction_index", f32), ("absorption_index", f32), ("roughness", f32)]


class Instance(C.Structure):
    _fields_ = [("model", F12), ("inv_model", F12), ("normal", F12), ("motionblur", F3), ("emission", F3), ("material", i32),
                ("tex_offset", i32), ("tex_count", i32), ("ref_id", i32), ("identity", i32)]


class Triangle(C.Structure):
    _fields_ = [("c1", F3), ("c2", F3), ("c3", F3), ("n1", F3), ("n2", F3), ("n3", F3), ("tex_u", F3), ("tex_v", F3),
                ("smooth", i32)]


class Sphere(C.Structure):
    _fields_ = [("center", F3), ("radius", f32), ("model", F12), ("inv_model", F12), ("normal", F12), ("emission", F3),
                ("material", i32), ("tex_offset", i32), ("tex_count", i32)]


class Plane(C.Structure):
    _fields_ = [("point", F3), ("normal", F3), ("raw_normal_x", f32), ("material", i32), ("tex_offset", i32), ("tex_count", i32)]


class Image(C.Structure):
    _fields_ = [("width", i32), ("height", i32), ("is_hdr", i32), ("offset", i32)]


class Texture(C.Structure):
    _fields_ = [("kind", i32), ("image", i32), ("decal_mode", i32), ("interp", i32), ("bump_factor", f32),
                ("noise_scale", f32), ("noise_conversion", i32), ("num_octaves", f32), ("black", F3), ("white", F3),
                ("scale", f32), ("offset", f32)]


class Light(C.Structure):
    _fields_ = [("type", i32), ("samples", i32), ("position", F3), ("direction", F3), ("normal", F3), ("u", F3), ("v", F3),
                ("v0", F3), ("v1", F3), ("v2", F3), ("color", F3), ("size", f32), ("area", f32), ("cos_coverage", f32),
                ("cos_falloff", f32), ("image", i32), ("cosine_sampling", i32), ("is_probe", i32), ("sphere", i32)]


class BvhNode(C.Structure):
    _fields_ = [("bmin", F3), ("bmax", F3), ("offset", i32), ("n_prims", i32), ("axis", i32)]


class BvhPrim(C.Structure):
    _fields_ = [("shape", i32), ("instance", i32)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", i32),
        ("n_triangles", i32), ("triangles", C.POINTER(Triangle)),
        ("n_instances", i32), ("instances", C.POINTER(Instance)),
        ("n_spheres", i32), ("spheres", C.POINTER(Sphere)),
        ("n_planes", i32), ("planes", C.POINTER(Plane)),
        ("n_materials", i32), ("materials", C.POINTER(Material)),
        ("n_textures", i32), ("textures", C.POINTER(Texture)),
        ("n_texture_refs", i32), ("texture_refs", C.POINTER(i32)),
        ("n_images", i32), ("images", C.POINTER(Image)),
        ("n_texels_u8", i64), ("texels_u8", C.POINTER(C.c_uint8)),
        ("n_texels_f32", i64), ("texels_f32", C.POINTER(f32)),
        ("n_lights", i32), ("lights", C.POINTER(Light)),
        ("n_bvh_nodes", i32), ("bvh_nodes", C.POINTER(BvhNode)),
        ("n_bvh_prims", i32), ("bvh_prims", C.POINTER(BvhPrim)),
        ("bvh_prim_bounds", C.POINTER(f32)),
        ("intersection_epsilon", f32), ("shadow_ray_epsilon", f32), ("max_depth", f32),
        ("ambient_light", F3), ("background_color", F3), ("background_texture", i32), ("is_probe", i32),
    ]


class CameraDesc(C.Structure):
    _fields_ = [("position", F3), ("u", F3), ("v", F3), ("w", F3), ("center", F3), ("near_left", f32), ("near_right", f32),
                ("near_bottom", f32), ("near_top", f32), ("resx", i32), ("resy", i32), ("num_samples", i32),
                ("focus_distance", f32), ("aperture_size", f32), ("is_hdr", i32), ("path_tracing", i32),
                ("importance_sampling", i32), ("next_event_estimation", i32), ("russian_roulette", i32),
                ("splitting_factor", i32), ("sample_max_val", f32), ("min_recursion_depth", i32), ("seed", u32)]


class Stats(C.Structure):
    _fields_ = [("primary_rays", u64), ("shadow_rays", u64), ("secondary_rays", u64), ("nodes_visited", u64),
                ("triangle_tests", u64), ("sphere_tests", u64), ("plane_tests", u64), ("shaded_hits", u64),
                ("texel_fetches", u64), ("kernel_ms", f32), ("total_ms", f32), ("shadow_rays_traced", u64),
                ("any_hit_nodes_visited", u64), ("any_hit_triangle_tests", u64), ("any_hit_sphere_tests", u64),
                ("any_hit_plane_tests", u64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

    @property
    def total_rays(self):
        return self.primary_rays + self.shadow_rays + self.secondary_rays

    def traversal_bytes(self, any_hit):
        """Algorithmic bytes of the closest-hit (any_hit=False) or any-hit traversals under the same accounting."""
        n, t, s, p = (self.any_hit_nodes_visited, self.any_hit_triangle_tests, self.any_hit_sphere_tests,
                      self.any_hit_plane_tests)
        rays = self.shadow_rays_traced
        if not any_hit:
            n, t, s, p = self.nodes_visited - n, self.triangle_tests - t, self.sphere_tests - s, self.plane_tests - p
            rays = self.primary_rays + self.secondary_rays
        return 32 * n + 36 * t + 64 * s + 24 * p + 48 * rays

    @property
    def algorithmic_bytes(self):
        """Traversal bytes per the fixed accounting: 32 B/node visit, 36 B/triangle test, 64 B/sphere test,
        24 B/plane test, 32 B ray read + 16 B hit write per ray; shading 64 B material + 12 B framebuffer per shaded
        hit + 16 B per texel fetch."""
        return (32 * self.nodes_visited + 36 * self.triangle_tests + 64 * self.sphere_tests + 24 * self.plane_tests
                + 48 * self.total_rays + 76 * self.shaded_hits + 16 * self.texel_fetches)


KERNEL_NAMES = ["generate", "trace_closest", "shade", "light_setup", "trace_any", "light_gather", "resolve"]


class KernelTime(C.Structure):
    _fields_ = [("launches", u64), ("total_ms", C.c_double)]


class TonemapDesc(C.Structure):
    _fields_ = [("tmo", i32), ("opt1", f32), ("opt2", f32), ("saturation", f32), ("gamma", f32)]
