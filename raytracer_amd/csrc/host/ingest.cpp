// Scene ingest: the reference's JSON dialect (reference src/parser.cpp) -> flat arrays.
//
// Everything here runs once per scene file on the host.  Lookups follow the reference's rules:
// transformations, materials and vertex indices are positional and 1-based; textures and images are
// looked up by `_id` (src/parser.cpp:161-167, 704-711) except the environment light, which indexes
// images by position (src/parser.cpp:1450-1456).
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "hmath.h"
#include "json.h"
#include "scene.h"

namespace rth {

using rtjson::Value;

namespace {

struct TransformTable {   // src/shapes.h:28-35, indexed by position
    std::vector<M4> translations, rotations, scales, composites;
};

// whitespace-separated floats, as read by `istringstream >> float`
std::vector<float> floats_of(const std::string &s) {
    std::vector<float> out;
    const char *p = s.c_str();
    while (*p) {
        while (*p && isspace((unsigned char)*p)) ++p;
        if (!*p) break;
        char *end = nullptr;
        float v = strtof(p, &end);
        if (end == p) break;
        out.push_back(v);
        p = end;
    }
    return out;
}
std::vector<long long> ints_of(const std::string &s) {
    std::vector<long long> out;
    const char *p = s.c_str();
    while (*p) {
        while (*p && isspace((unsigned char)*p)) ++p;
        if (!*p) break;
        char *end = nullptr;
        long long v = strtoll(p, &end, 10);
        if (end == p) break;
        out.push_back(v);
        p = end;
    }
    return out;
}
V3 vec3_of(const Value &v) {
    auto f = floats_of(v.str);
    f.resize(3, 0.f);
    return V3(f[0], f[1], f[2]);
}
float float_of(const Value &v) { return strtof(v.c_str(), nullptr); }
int int_of(const Value &v) { return (int)strtol(v.c_str(), nullptr, 10); }
void put3(float *d, V3 v) {
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
}
void put34(float *d, const M4 &m) { memcpy(d, m.m, 12 * sizeof(float)); }

// src/parser.cpp:54-91 — tokens t#/r#/s#/c# applied left to right as M = T_i * M
M4 compose_transformations(const Value &obj, const TransformTable &t) {
    M4 model;
    const Value *tr = obj.find("Transformations");
    if (!tr) return model;
    std::istringstream iss(tr->str);
    std::string word;
    while (iss >> word) {
        if (word.size() < 2) continue;
        int index = atoi(word.c_str() + 1);
        const std::vector<M4> *list = nullptr;
        switch (word[0]) {
            case 't': list = &t.translations; break;
            case 'r': list = &t.rotations; break;
            case 's': list = &t.scales; break;
            case 'c': list = &t.composites; break;
            default: break;
        }
        if (!list || index < 1 || index > (int)list->size()) continue;
        model = mul((*list)[index - 1], model);
    }
    return model;
}

// src/parser.cpp:1496-1575
TransformTable parse_transformations(const Value &scene) {
    TransformTable res;
    const Value &tr = scene["Transformations"];
    if (!tr.is_object()) return res;
    auto each = [&](const char *name, auto &&make) {
        tr[name].each([&](const Value &node) {
            if (!node.is_object() || !node.has("_id") || !node.has("_data")) return;
            make(node["_data"].str);
        });
    };
    each("Translation", [&](const std::string &d) {
        auto f = floats_of(d);
        f.resize(3, 0.f);
        res.translations.push_back(translate(V3(f[0], f[1], f[2])));
    });
    each("Rotation", [&](const std::string &d) {
        auto f = floats_of(d);
        f.resize(4, 0.f);
        // degrees * (pi / 180.0f) evaluated in double, returned as float (src/parser.cpp:1491-1494)
        float angle = (float)((double)f[0] * (3.1415926535897932384626433832 / 180.0));
        res.rotations.push_back(rotate(V3(f[1], f[2], f[3]), angle));
    });
    each("Scaling", [&](const std::string &d) {
        auto f = floats_of(d);
        f.resize(3, 0.f);
        res.scales.push_back(scale(V3(f[0], f[1], f[2])));
    });
    each("Composite", [&](const std::string &d) {
        auto f = floats_of(d);
        f.resize(16, 0.f);
        M4 m;
        memcpy(m.m, f.data(), sizeof m.m);
        res.composites.push_back(m);
    });
    return res;
}

bool ends_with(const std::string &s, const char *suffix) {
    size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// src/parser.cpp:276-412 and the two camera constructors src/camera.cpp:6-115
void parse_cameras(const Value &scene, const TransformTable &tt, HostScene &out) {
    if (!scene.has("Cameras")) return;
    scene["Cameras"]["Camera"].each([&](const Value &cam) {
        HostCamera hc;
        rt_camera_desc &c = hc.desc;
        std::string type = cam.has("_type") ? cam["_type"].str : "pinhole";
        M4 model = compose_transformations(cam, tt);
        V3 pos = vec3_of(cam["Position"]);
        V3 up = vec3_of(cam["Up"]);
        float near_dist = float_of(cam["NearDistance"]);
        auto res = ints_of(cam["ImageResolution"].str);
        res.resize(2, 0);
        c.resx = (int)res[0];
        c.resy = (int)res[1];
        c.num_samples = cam.has("NumSamples") ? int_of(cam["NumSamples"]) : 1;
        c.focus_distance = cam.has("FocusDistance") ? float_of(cam["FocusDistance"]) : 0.f;
        c.aperture_size = cam.has("ApertureSize") ? float_of(cam["ApertureSize"]) : 0.f;
        hc.image_name = cam["ImageName"].str;

        V3 w_local, u_local, v_local;
        if (type == "lookAt") {
            V3 gaze_point = vec3_of(cam["GazePoint"]);
            float fovy = float_of(cam["FovY"]);
            V3 gaze_dir = normalize(gaze_point - pos);
            w_local = gaze_dir * -1.0f;
            u_local = normalize(cross(up, w_local));
            v_local = cross(w_local, u_local);
            float aspect = (float)c.resx / (float)c.resy;
            float near_height = 2.0f * near_dist * tanf((float)((double)(fovy * 0.5f) * M_PI / 180.0));
            float near_width = near_height * aspect;
            c.near_left = -near_width / 2.0f;
            c.near_right = near_width / 2.0f;
            c.near_bottom = -near_height / 2.0f;
            c.near_top = near_height / 2.0f;
        } else {
            V3 gaze = vec3_of(cam["Gaze"]);
            auto np = floats_of(cam["NearPlane"].str);
            np.resize(4, 0.f);
            w_local = normalize(gaze) * -1.0f;
            u_local = normalize(cross(up, w_local));
            v_local = cross(w_local, u_local);
            c.near_left = np[0];
            c.near_right = np[1];
            c.near_bottom = np[2];
            c.near_top = np[3];
        }
        V3 w = normalize(xform_dir(model, w_local));
        V3 u = normalize(xform_dir(model, u_local));
        V3 v = normalize(xform_dir(model, v_local));
        V3 position = xform_point(model, pos);
        V3 center = position - w * near_dist;
        put3(c.position, position);
        put3(c.u, u);
        put3(c.v, v);
        put3(c.w, w);
        put3(c.center, center);

        c.is_hdr = ends_with(hc.image_name, ".exr") ? 1 : 0;
        c.splitting_factor = 1;
        c.sample_max_val = -1.0f;
        c.min_recursion_depth = 0;
        if (cam.has("Renderer") && cam["Renderer"].str == "PathTracing") c.path_tracing = 1;
        if (cam.has("RendererParams")) {
            const std::string &p = cam["RendererParams"].str;
            if (p.find("ImportanceSampling") != std::string::npos) c.importance_sampling = 1;
            if (p.find("NextEventEstimation") != std::string::npos) c.next_event_estimation = 1;
            if (p.find("RussianRoulette") != std::string::npos) c.russian_roulette = 1;
        }
        if (cam.has("SplittingFactor")) c.splitting_factor = int_of(cam["SplittingFactor"]);
        if (cam.has("SampleMaxVal")) c.sample_max_val = float_of(cam["SampleMaxVal"]);
        if (cam.has("MinRecursionDepth")) c.min_recursion_depth = int_of(cam["MinRecursionDepth"]);
        if (cam.has("Tonemap")) {
            cam["Tonemap"].each([&](const Value &val) {
                Tonemap tm;
                if (val.has("TMO")) {
                    if (val["TMO"].str == "Filmic")
                        tm.type = RT_TMO_FILMIC;
                    else if (val["TMO"].str == "ACES")
                        tm.type = RT_TMO_ACES;
                }
                if (val.has("TMOOptions")) {
                    auto f = floats_of(val["TMOOptions"].str);
                    if (f.size() > 0) tm.opt1 = f[0];
                    if (f.size() > 1) tm.opt2 = f[1];
                }
                if (val.has("Saturation")) tm.saturation = float_of(val["Saturation"]);
                if (val.has("Gamma")) tm.gamma = float_of(val["Gamma"]);
                if (val.has("Extension")) tm.extension = val["Extension"].str;
                hc.tonemaps.push_back(tm);
            });
        }
        out.cameras.push_back(hc);
    });
}

// src/parser.cpp:454-656
void parse_materials(const Value &scene, HostScene &out) {
    struct BrdfInfo {
        int type;
        bool normalized = false, has_exponent = false, kdfresnel = false;
        float exponent = 1.0f;
    };
    std::unordered_map<int, BrdfInfo> brdfs;
    if (scene.has("BRDFs")) {
        const Value &b = scene["BRDFs"];
        const char *names[5] = {"OriginalBlinnPhong", "OriginalPhong", "ModifiedBlinnPhong", "ModifiedPhong", "TorranceSparrow"};
        for (int k = 0; k < 5; ++k) {
            b[names[k]].each([&](const Value &node) {
                if (!node.has("_id")) return;
                BrdfInfo info;
                info.type = k;
                if (node.has("_normalized")) info.normalized = node["_normalized"].str == "true";
                if (node.has("Exponent")) {
                    info.exponent = float_of(node["Exponent"]);
                    info.has_exponent = true;
                }
                if (node.has("_kdfresnel")) info.kdfresnel = node["_kdfresnel"].str == "true";
                brdfs[int_of(node["_id"])] = info;
            });
        }
    }
    if (!scene.has("Materials") || !scene["Materials"].has("Material")) return;
    scene["Materials"]["Material"].each([&](const Value &mj) {
        rt_material m{};
        m.type = RT_MAT_REGULAR;
        m.brdf = RT_BRDF_ORIGINAL_BLINN_PHONG;
        m.phong_exponent = 1.0f;
        m.refraction_index = 0.0f;
        m.absorption_index = 1.0f;
        if (mj.has("_type")) {
            const std::string &t = mj["_type"].str;
            if (t == "mirror")
                m.type = RT_MAT_MIRROR;
            else if (t == "dielectric")
                m.type = RT_MAT_DIELECTRIC;
            else if (t == "conductor")
                m.type = RT_MAT_CONDUCTOR;
        }
        if (mj.has("_BRDF")) {
            auto it = brdfs.find(int_of(mj["_BRDF"]));
            if (it != brdfs.end()) {
                m.brdf = it->second.type;
                m.normalized = it->second.normalized;
                m.kdfresnel = it->second.kdfresnel;
                if (it->second.has_exponent) m.phong_exponent = it->second.exponent;
            }
        }
        bool degamma = mj.has("_degamma") && mj["_degamma"].str == "true";
        auto color = [&](const char *key, float *dst) {
            if (!mj.has(key)) return;
            V3 c = vec3_of(mj[key]);
            if (degamma) c = V3(powf(c.x, 2.2f), powf(c.y, 2.2f), powf(c.z, 2.2f));
            put3(dst, c);
        };
        color("AmbientReflectance", m.ambient);
        color("DiffuseReflectance", m.diffuse);
        color("SpecularReflectance", m.specular);
        if (mj.has("PhongExponent")) m.phong_exponent = float_of(mj["PhongExponent"]);
        if (mj.has("MirrorReflectance")) put3(m.mirror, vec3_of(mj["MirrorReflectance"]));
        if (mj.has("AbsorptionCoefficient")) put3(m.absorption_coefficient, vec3_of(mj["AbsorptionCoefficient"]));
        if (mj.has("RefractionIndex")) m.refraction_index = float_of(mj["RefractionIndex"]);
        if (mj.has("AbsorptionIndex")) m.absorption_index = float_of(mj["AbsorptionIndex"]);
        if (mj.has("Roughness")) m.roughness = float_of(mj["Roughness"]);
        out.materials.push_back(m);
    });
}

struct IdIndex {   // declaration-ordered (id -> index) list: "first match by _id"
    std::vector<std::pair<int, int>> items;
    void add(int id, int index) { items.emplace_back(id, index); }
};

// src/parser.cpp:93-133
void parse_images(const Value &scene, const std::string &base_dir, HostScene &out, IdIndex &image_ids) {
    if (!scene.has("Textures")) return;
    const Value &t = scene["Textures"];
    if (!t.has("Images") || !t["Images"].has("Image")) return;
    t["Images"]["Image"].each([&](const Value &img) {
        if (!img.has("_id") || !img.has("_data")) return;
        int id = int_of(img["_id"]);
        std::string full = (std::filesystem::path(base_dir) / img["_data"].str).string();
        out.dependencies.push_back(full);
        rt_image im{};
        std::string err;
        if (!load_image_file(full, out, im, err)) {
            out.warnings.push_back("Error: Could not load texture " + full + " (" + err + ")");
            im.width = im.height = 0;
        }
        image_ids.add(id, (int)out.images.size());
        out.images.push_back(im);
    });
}

// src/parser.cpp:135-274
void parse_textures(const Value &scene, HostScene &out, const IdIndex &image_ids, IdIndex &texture_ids) {
    if (!scene.has("Textures") || !scene["Textures"].has("TextureMap")) return;
    scene["Textures"]["TextureMap"].each([&](const Value &tex) {
        if (!tex.has("_id")) return;
        rt_texture t{};
        t.image = -1;
        t.decal_mode = RT_DECAL_REPLACE_KD;
        t.interp = RT_INTERP_BILINEAR;
        t.bump_factor = 10.f;
        t.noise_scale = 1.f;
        t.noise_conversion = RT_NOISE_LINEAR;
        t.num_octaves = 1.f;
        if (tex.has("ImageId")) {
            int image_id = int_of(tex["ImageId"]);
            for (auto &kv : image_ids.items)
                if (kv.first == image_id) {
                    t.image = kv.second;
                    break;
                }
        }
        if (tex.has("DecalMode")) {
            const std::string &m = tex["DecalMode"].str;
            const char *names[7] = {"replace_kd", "blend_kd", "replace_ks", "replace_background", "replace_normal", "bump_normal", "replace_all"};
            for (int k = 0; k < 7; ++k)
                if (m == names[k]) t.decal_mode = k;
        }
        if (tex.has("Interpolation")) {
            if (tex["Interpolation"].str == "nearest")
                t.interp = RT_INTERP_NEAREST;
            else if (tex["Interpolation"].str == "bilinear")
                t.interp = RT_INTERP_BILINEAR;
        }
        bool chess = false;
        if (tex.has("BumpFactor")) t.bump_factor = float_of(tex["BumpFactor"]) * 30;   // src/parser.cpp:209
        if (tex.has("NoiseConversion")) {
            if (tex["NoiseConversion"].str == "absval")
                t.noise_conversion = RT_NOISE_ABSVAL;
            else if (tex["NoiseConversion"].str == "linear")
                t.noise_conversion = RT_NOISE_LINEAR;
        }
        if (tex.has("NoiseScale")) t.noise_scale = float_of(tex["NoiseScale"]);
        if (tex.has("NumOctaves")) t.num_octaves = float_of(tex["NumOctaves"]);
        if (tex.has("BlackColor")) {
            chess = true;
            put3(t.black, vec3_of(tex["BlackColor"]));
        }
        if (tex.has("WhiteColor")) {
            chess = true;
            put3(t.white, vec3_of(tex["WhiteColor"]));
        }
        if (tex.has("Scale")) {
            chess = true;
            t.scale = float_of(tex["Scale"]) * 10.0f;   // src/parser.cpp:252
        }
        if (tex.has("Offset")) {
            chess = true;
            t.offset = float_of(tex["Offset"]);
        }
        // texture::sample dispatch order: image, then checker, then Perlin (src/texture.cpp:168-197)
        t.kind = t.image >= 0 ? RT_TEX_IMAGE : (chess ? RT_TEX_CHECKER : RT_TEX_PERLIN);
        texture_ids.add(int_of(tex["_id"]), (int)out.textures.size());
        out.textures.push_back(t);
    });
}

// "Textures": "3 1" -> every texture whose _id matches, in list order (src/parser.cpp:697-712)
void texture_list(const Value &obj, const IdIndex &texture_ids, HostScene &out, int32_t &offset, int32_t &count) {
    offset = (int32_t)out.texture_refs.size();
    count = 0;
    if (!obj.has("Textures")) return;
    for (long long x : ints_of(obj["Textures"].str))
        for (auto &kv : texture_ids.items)
            if (kv.first == (int)x) {
                out.texture_refs.push_back(kv.second);
                ++count;
            }
}

// src/parser.cpp:1292-1489
void parse_lights(const Value &scene, const TransformTable &tt, HostScene &out) {
    if (!scene.has("Lights")) return;
    const Value &ln = scene["Lights"];
    if (ln.has("AmbientLight")) put3(out.ambient_light, vec3_of(ln["AmbientLight"]));
    ln["PointLight"].each([&](const Value &n) {
        rt_light l{};
        l.type = RT_LIGHT_POINT;
        l.samples = 1;
        put3(l.position, xform_point(compose_transformations(n, tt), vec3_of(n["Position"])));
        put3(l.color, vec3_of(n["Intensity"]));
        out.lights.push_back(l);
    });
    ln["AreaLight"].each([&](const Value &n) {
        rt_light l{};
        l.type = RT_LIGHT_AREA;
        l.samples = 16;   // src/parser.cpp:1370
        put3(l.position, xform_point(compose_transformations(n, tt), vec3_of(n["Position"])));
        V3 normal = vec3_of(n["Normal"]);
        put3(l.normal, normal);
        V3 rad(0, 0, 0);
        if (n.has("Radiance"))
            rad = vec3_of(n["Radiance"]);
        else if (n.has("Intensity"))
            rad = vec3_of(n["Intensity"]);
        put3(l.color, rad);
        l.size = float_of(n["Size"]);
        // basis of the square (src/lights.cpp:42-56)
        V3 helper = fabsf(normal.x) > 0.1f ? V3(0, 1, 0) : V3(1, 0, 0);
        V3 u = normalize(cross(helper, normal));
        V3 v = normalize(cross(normal, u));
        put3(l.u, u);
        put3(l.v, v);
        out.lights.push_back(l);
    });
    ln["DirectionalLight"].each([&](const Value &n) {
        rt_light l{};
        l.type = RT_LIGHT_DIRECTIONAL;
        l.samples = 1;
        put3(l.direction, normalize(vec3_of(n["Direction"])));
        put3(l.color, vec3_of(n["Radiance"]));
        out.lights.push_back(l);
    });
    ln["SpotLight"].each([&](const Value &n) {
        rt_light l{};
        l.type = RT_LIGHT_SPOT;
        l.samples = 1;
        put3(l.position, xform_point(compose_transformations(n, tt), vec3_of(n["Position"])));
        put3(l.direction, normalize(vec3_of(n["Direction"])));
        put3(l.color, vec3_of(n["Intensity"]));
        const float pi = 3.14159265359f;
        float cov = (float_of(n["CoverageAngle"]) * pi / 180.0f) / 2.0f;
        float fall = (float_of(n["FalloffAngle"]) * pi / 180.0f) / 2.0f;
        l.cos_coverage = cosf(cov);
        l.cos_falloff = cosf(fall);
        out.lights.push_back(l);
    });
    ln["SphericalDirectionalLight"].each([&](const Value &n) {
        rt_light l{};
        l.type = RT_LIGHT_ENV;
        l.samples = 1;
        int image_id = int_of(n["ImageId"]);
        l.image = (image_id > 0 && image_id <= (int)out.images.size()) ? image_id - 1 : -1;
        l.cosine_sampling = !(n.has("Sampler") && n["Sampler"].str == "uniform");
        l.is_probe = n.has("_type") && n["_type"].str == "probe";
        out.lights.push_back(l);
    });
}

struct MeshGroup {   // one all_mesh_infos (src/shapes.h:81-84); same container type => same iteration order
    std::unordered_map<int, int> instance_of;   // reference id -> index into HostScene::instances
};

rt_instance make_instance(const M4 &model, int material, int ref_id) {
    rt_instance inst{};
    M4 inv = inverse(model);
    M4 nrm = transpose(inv);
    put34(inst.model, model);
    put34(inst.inv_model, inv);
    put34(inst.normal, nrm);
    inst.material = material;
    inst.ref_id = ref_id;
    inst.identity = model.is_identity() ? 1 : 0;
    return inst;
}

void tri_bounds(const rt_triangle &t, const rt_instance &inst, PrimBounds &b) {   // src/shapes.cpp:124-150
    M4 model;
    memcpy(model.m, inst.model, 12 * sizeof(float));
    V3 v0 = xform_point(model, V3(t.c1[0], t.c1[1], t.c1[2]));
    V3 v1 = xform_point(model, V3(t.c2[0], t.c2[1], t.c2[2]));
    V3 v2 = xform_point(model, V3(t.c3[0], t.c3[1], t.c3[2]));
    V3 mn = vmin(vmin(v0, v1), v2), mx = vmax(vmax(v0, v1), v2);
    V3 mb(inst.motionblur[0], inst.motionblur[1], inst.motionblur[2]);
    V3 mn2 = mn + mb, mx2 = mx + mb;
    mn = vmin(mn, mn2);
    mx = vmax(mx, mx2);
    V3 pad(1e-4f, 1e-4f, 1e-4f);
    put3(b.bmin, mn - pad);
    put3(b.bmax, mx + pad);
}

void sphere_bounds(const rt_sphere &s, PrimBounds &b) {   // src/shapes.cpp:344-386
    M4 model;
    memcpy(model.m, s.model, 12 * sizeof(float));
    V3 lo(s.center[0] - s.radius, s.center[1] - s.radius, s.center[2] - s.radius);
    V3 hi(s.center[0] + s.radius, s.center[1] + s.radius, s.center[2] + s.radius);
    V3 mn(FLT_MAX, FLT_MAX, FLT_MAX), mx(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = 0; i < 8; ++i) {
        V3 c((i & 1) ? hi.x : lo.x, (i & 2) ? hi.y : lo.y, (i & 4) ? hi.z : lo.z);
        V3 wc = xform_point(model, c);
        mn = vmin(mn, wc);
        mx = vmax(mx, wc);
    }
    V3 pad(1e-4f, 1e-4f, 1e-4f);
    put3(b.bmin, mn - pad);
    put3(b.bmax, mx + pad);
}

}  // namespace

rt_scene_desc HostScene::desc() const {
    if (mapping) return mapped_desc;
    rt_scene_desc d{};
    d.abi_version = RT_ABI_VERSION;
    d.n_triangles = (int32_t)triangles.size();
    d.triangles = triangles.data();
    d.n_instances = (int32_t)instances.size();
    d.instances = instances.data();
    d.n_spheres = (int32_t)spheres.size();
    d.spheres = spheres.data();
    d.n_planes = (int32_t)planes.size();
    d.planes = planes.data();
    d.n_materials = (int32_t)materials.size();
    d.materials = materials.data();
    d.n_textures = (int32_t)textures.size();
    d.textures = textures.data();
    d.n_texture_refs = (int32_t)texture_refs.size();
    d.texture_refs = texture_refs.data();
    d.n_images = (int32_t)images.size();
    d.images = images.data();
    d.n_texels_u8 = (int64_t)texels_u8.size();
    d.texels_u8 = texels_u8.data();
    d.n_texels_f32 = (int64_t)texels_f32.size();
    d.texels_f32 = texels_f32.data();
    d.n_lights = (int32_t)lights.size();
    d.lights = lights.data();
    d.n_bvh_nodes = (int32_t)bvh_nodes.size();
    d.bvh_nodes = bvh_nodes.data();
    d.n_bvh_prims = (int32_t)bvh_prims.size();
    d.bvh_prims = bvh_prims.data();
    d.bvh_prim_bounds = nullptr;
    if (bvh_nodes.empty() && !build_prims.empty()) {   // no tree built here: the rendering core builds one on the device
        static_assert(sizeof(PrimBounds) == 6 * sizeof(float), "a box is six floats");
        d.n_bvh_prims = (int32_t)build_prims.size();
        d.bvh_prims = build_prims.data();
        d.bvh_prim_bounds = &build_bounds[0].bmin[0];
    }
    d.intersection_epsilon = intersection_epsilon;
    d.shadow_ray_epsilon = shadow_ray_epsilon;
    d.max_depth = max_depth;
    memcpy(d.ambient_light, ambient_light, sizeof ambient_light);
    memcpy(d.background_color, background_color, sizeof background_color);
    d.background_texture = background_texture;
    d.is_probe = is_probe;
    return d;
}

void ensure_host_bvh(HostScene &scene) {
    if (scene.mapping || !scene.bvh_nodes.empty() || scene.build_prims.empty()) return;
    const auto t0 = std::chrono::high_resolution_clock::now();
    build_bvh(scene);
    scene.bvh_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    scene.build_prims = {};   // only the build needed them
    scene.build_bounds = {};
}

bool load_scene(const std::string &json_path, HostScene &out, std::string &err) {
    auto t0 = std::chrono::high_resolution_clock::now();
    if (scene_cache_load(json_path, out)) {
        out.ingest_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
        out.bvh_ms = 0;
        return true;
    }
    Value doc;
    if (!rtjson::parse_file(json_path, doc, err)) return false;
    const Value &scene = doc["Scene"];
    if (!scene.is_object()) {
        err = "no Scene object in " + json_path;
        return false;
    }
    std::string base_dir = std::filesystem::path(json_path).parent_path().string();

    TransformTable tt = parse_transformations(scene);
    parse_cameras(scene, tt, out);

    // src/parser.cpp:414-452
    std::vector<float> vertices = scene.has("VertexData") ? floats_of(scene["VertexData"]["_data"].str) : std::vector<float>();
    vertices.resize(vertices.size() / 3 * 3);
    std::vector<float> uvs = scene.has("TexCoordData") ? floats_of(scene["TexCoordData"]["_data"].str) : std::vector<float>();

    parse_materials(scene, out);
    IdIndex image_ids, texture_ids;
    parse_images(scene, base_dir, out, image_ids);
    parse_textures(scene, out, image_ids, texture_ids);
    for (size_t i = 0; i < out.textures.size(); ++i)   // src/main.cpp:380-388
        if (out.textures[i].decal_mode == RT_DECAL_REPLACE_BACKGROUND) {
            out.background_texture = (int32_t)i;
            break;
        }
    parse_lights(scene, tt, out);

    if (scene.has("IntersectionTestEpsilon")) out.intersection_epsilon = float_of(scene["IntersectionTestEpsilon"]);
    if (scene.has("ShadowRayEpsilon")) out.shadow_ray_epsilon = float_of(scene["ShadowRayEpsilon"]);
    if (scene.has("MaxRecursionDepth")) out.max_depth = float_of(scene["MaxRecursionDepth"]);
    if (scene.has("BackgroundColor")) put3(out.background_color, vec3_of(scene["BackgroundColor"]));

    // ---- objects (src/parser.cpp:658-1237) --------------------------------------------------
    const Value &objects = scene["Objects"];
    std::vector<MeshGroup> groups;
    std::vector<int> tri_group;   // group of every triangle
    auto vertex = [&](size_t i, V3 &v) -> bool {
        if (i * 3 + 2 >= vertices.size()) return false;
        v = V3(vertices[i * 3], vertices[i * 3 + 1], vertices[i * 3 + 2]);
        return true;
    };
    auto uv_at = [&](long long index) -> float { return uvs.empty() ? 0.0f : uvs[(size_t)index % uvs.size()]; };
    auto material_index = [&](const Value &obj, int &mi) -> bool {
        mi = int_of(obj["Material"]) - 1;
        if (mi < 0 || mi >= (int)out.materials.size()) {
            err = "material index out of range";
            return false;
        }
        return true;
    };
    bool ok = true;

    auto process_mesh = [&](const Value &mesh) {
        if (!ok) return;
        if (!mesh.has("Faces") || !(mesh["Faces"].has("_data") || mesh["Faces"].has("_plyFile"))) return;
        int mi;
        if (!material_index(mesh, mi)) {
            ok = false;
            return;
        }
        int mesh_id = int_of(mesh["_id"]);
        M4 model = compose_transformations(mesh, tt);
        rt_instance inst = make_instance(model, mi, mesh_id);
        texture_list(mesh, texture_ids, out, inst.tex_offset, inst.tex_count);
        if (mesh.has("MotionBlur")) put3(inst.motionblur, vec3_of(mesh["MotionBlur"]));
        bool is_light = false;
        if (mesh.has("Radiance")) {
            put3(inst.emission, vec3_of(mesh["Radiance"]));
            is_light = true;
        }
        int inst_index = (int)out.instances.size();
        out.instances.push_back(inst);
        groups.emplace_back();
        int group = (int)groups.size() - 1;
        groups[group].instance_of[mesh_id] = inst_index;

        bool smooth = mesh.has("_shadingMode") && mesh["_shadingMode"].str == "smooth";
        size_t first_tri = out.triangles.size();
        if (mesh["Faces"].has("_plyFile")) {
            std::string full = (std::filesystem::path(base_dir) / mesh["Faces"]["_plyFile"].str).string();
            out.dependencies.push_back(full);
            PlyResult pr;
            if (!load_ply(full, vertices, out, smooth, pr, err)) {
                ok = false;
                return;
            }
        } else {
            auto idx = ints_of(mesh["Faces"]["_data"].str);
            long long offset = mesh["Faces"].has("_vertexOffset") ? int_of(mesh["Faces"]["_vertexOffset"]) : 0;
            size_t ntri = idx.size() / 3;
            size_t nverts = vertices.size() / 3;
            for (size_t k = 0; k < ntri * 3; ++k) {
                idx[k] = idx[k] - 1 + offset;
                if (idx[k] < 0 || (size_t)idx[k] >= nverts) {
                    err = "vertex index out of range in mesh " + std::to_string(mesh_id);
                    ok = false;
                    return;
                }
            }
            std::vector<V3> vn;
            if (smooth) {   // area-weighted vertex normals (src/parser.cpp:763-780)
                vn.assign(nverts, V3(0, 0, 0));
                for (size_t k = 0; k < ntri; ++k) {
                    V3 a, b, c;
                    vertex(idx[k * 3], a);
                    vertex(idx[k * 3 + 1], b);
                    vertex(idx[k * 3 + 2], c);
                    V3 n = cross(b - a, c - a);
                    for (int j = 0; j < 3; ++j) vn[idx[k * 3 + j]] = vn[idx[k * 3 + j]] + n;
                }
                for (auto &n : vn) n = normalize(n);
            }
            for (size_t k = 0; k < ntri; ++k) {
                rt_triangle t{};
                V3 a, b, c;
                vertex(idx[k * 3], a);
                vertex(idx[k * 3 + 1], b);
                vertex(idx[k * 3 + 2], c);
                put3(t.c1, a);
                put3(t.c2, b);
                put3(t.c3, c);
                for (int j = 0; j < 3; ++j) {
                    t.tex_u[j] = uv_at(idx[k * 3 + j] * 2);
                    t.tex_v[j] = uv_at(idx[k * 3 + j] * 2 + 1);
                }
                if (smooth) {
                    t.smooth = 1;
                    put3(t.n1, vn[idx[k * 3]]);
                    put3(t.n2, vn[idx[k * 3 + 1]]);
                    put3(t.n3, vn[idx[k * 3 + 2]]);
                }
                out.triangles.push_back(t);
                if (is_light) {   // TriangleLight (src/lights.cpp:318-336)
                    rt_light l{};
                    l.type = RT_LIGHT_TRIANGLE;
                    l.samples = 16;
                    V3 v0 = xform_point(model, a), v1 = xform_point(model, b), v2 = xform_point(model, c);
                    put3(l.v0, v0);
                    put3(l.v1, v1);
                    put3(l.v2, v2);
                    V3 n = cross(v1 - v0, v2 - v0);
                    l.area = 0.5f * sqrtf(dot(n, n));
                    put3(l.normal, normalize(n));
                    memcpy(l.color, inst.emission, sizeof l.color);
                    out.lights.push_back(l);
                }
            }
        }
        tri_group.resize(out.triangles.size(), group);
        (void)first_tri;
    };
    objects["Mesh"].each(process_mesh);
    objects["LightMesh"].each(process_mesh);

    objects["MeshInstance"].each([&](const Value &mi_node) {   // src/parser.cpp:830-898
        if (!ok) return;
        int id = int_of(mi_node["_id"]);
        int base_id = int_of(mi_node["_baseMeshId"]);
        bool reset = mi_node.has("_resetTransform") && mi_node["_resetTransform"].str == "true";
        int target = -1;
        for (size_t g = 0; g < groups.size(); ++g)
            if (groups[g].instance_of.count(base_id)) {
                target = (int)g;
                break;
            }
        if (target < 0) {
            err = "MeshInstance can't find base mesh.";
            ok = false;
            return;
        }
        const rt_instance base = out.instances[groups[target].instance_of[base_id]];
        M4 model = compose_transformations(mi_node, tt);
        if (!reset) {
            M4 base_model;
            memcpy(base_model.m, base.model, 12 * sizeof(float));
            model = mul(model, base_model);
        }
        int mat = base.material;
        if (mi_node.has("Material")) {
            if (!material_index(mi_node, mat)) {
                ok = false;
                return;
            }
        }
        rt_instance inst = make_instance(model, mat, id);
        texture_list(mi_node, texture_ids, out, inst.tex_offset, inst.tex_count);
        if (mi_node.has("MotionBlur")) put3(inst.motionblur, vec3_of(mi_node["MotionBlur"]));
        auto it = groups[target].instance_of.find(id);
        if (it != groups[target].instance_of.end()) {
            out.instances[it->second] = inst;   // operator[] on an existing key overwrites
        } else {
            groups[target].instance_of[id] = (int)out.instances.size();
            out.instances.push_back(inst);
        }
    });

    objects["Triangle"].each([&](const Value &tri) {   // src/parser.cpp:900-963
        if (!ok) return;
        if (!tri.has("Indices")) return;
        int mi;
        if (!material_index(tri, mi)) {
            ok = false;
            return;
        }
        M4 model = compose_transformations(tri, tt);
        rt_instance inst = make_instance(model, mi, 123456);
        texture_list(tri, texture_ids, out, inst.tex_offset, inst.tex_count);
        auto idx = ints_of(tri["Indices"].str);
        if (idx.size() < 3) return;
        V3 a, b, c;
        if (!vertex(idx[0] - 1, a) || !vertex(idx[1] - 1, b) || !vertex(idx[2] - 1, c)) {
            err = "vertex index out of range in Triangle";
            ok = false;
            return;
        }
        groups.emplace_back();
        groups.back().instance_of[123456] = (int)out.instances.size();
        out.instances.push_back(inst);
        rt_triangle t{};
        put3(t.c1, a);
        put3(t.c2, b);
        put3(t.c3, c);
        for (int j = 0; j < 3; ++j) {
            t.tex_u[j] = uv_at((idx[j] - 1) * 2);
            t.tex_v[j] = uv_at((idx[j] - 1) * 2 + 1);
        }
        out.triangles.push_back(t);
        tri_group.push_back((int)groups.size() - 1);
    });

    auto process_sphere = [&](const Value &sph) {   // src/parser.cpp:965-1018
        if (!ok) return;
        if (!sph.has("Center") || !sph.has("Radius")) return;
        rt_sphere s{};
        V3 center;
        if (!vertex(int_of(sph["Center"]) - 1, center)) {
            err = "sphere centre index out of range";
            ok = false;
            return;
        }
        put3(s.center, center);
        s.radius = float_of(sph["Radius"]);
        M4 model = compose_transformations(sph, tt);
        M4 inv = inverse(model);
        put34(s.model, model);
        put34(s.inv_model, inv);
        put34(s.normal, transpose(inv));
        if (!material_index(sph, s.material)) {
            ok = false;
            return;
        }
        texture_list(sph, texture_ids, out, s.tex_offset, s.tex_count);
        if (sph.has("Radiance")) {
            put3(s.emission, vec3_of(sph["Radiance"]));
            rt_light l{};   // SphereLight (src/lights.cpp:388-461)
            l.type = RT_LIGHT_SPHERE;
            l.samples = 16;
            l.sphere = (int32_t)out.spheres.size();
            put3(l.position, center);
            l.size = s.radius;
            V3 rw = xform_dir(model, V3(s.radius, 0, 0));
            l.area = 4.0f * 3.14159265359f * dot(rw, rw);
            memcpy(l.color, s.emission, sizeof l.color);
            out.lights.push_back(l);
        }
        out.spheres.push_back(s);
    };
    objects["Sphere"].each(process_sphere);
    objects["LightSphere"].each(process_sphere);

    objects["Plane"].each([&](const Value &pl) {   // src/parser.cpp:1020-1067
        if (!ok) return;
        if (!pl.has("Point") || !pl.has("Normal")) return;
        rt_plane p{};
        V3 point;
        if (!vertex(int_of(pl["Point"]) - 1, point)) {
            err = "plane point index out of range";
            ok = false;
            return;
        }
        M4 model = compose_transformations(pl, tt);
        put3(p.point, xform_point(model, point));
        M4 nm = transpose(inverse(model));
        V3 raw_normal = vec3_of(pl["Normal"]);
        put3(p.normal, normalize(xform_dir(nm, raw_normal)));
        p.raw_normal_x = raw_normal.x;
        if (!material_index(pl, p.material)) {
            ok = false;
            return;
        }
        texture_list(pl, texture_ids, out, p.tex_offset, p.tex_count);
        out.planes.push_back(p);
    });
    if (!ok) return false;

    // environment light doubles as the background (src/main.cpp:403-417)
    for (auto &l : out.lights)
        if (l.type == RT_LIGHT_ENV) {
            if (l.image >= 0) {
                rt_texture t{};
                t.kind = RT_TEX_IMAGE;
                t.image = l.image;
                t.decal_mode = RT_DECAL_REPLACE_BACKGROUND;
                t.interp = RT_INTERP_BILINEAR;
                out.background_texture = (int32_t)out.textures.size();
                out.textures.push_back(t);
                out.is_probe = l.is_probe;
            }
            break;
        }

    // ---- BVH build input: for s in shapes, for id in s->get_ids() (src/bvh.cpp:55-73) ----------
    out.build_prims.reserve(out.triangles.size() + out.spheres.size());
    out.build_bounds.reserve(out.triangles.size() + out.spheres.size());
    for (size_t ti = 0; ti < out.triangles.size(); ++ti) {
        const MeshGroup &g = groups[tri_group[ti]];
        for (const auto &kv : g.instance_of) {   // unordered_map iteration order, as triangle::get_ids
            rt_bvh_prim pr{(int32_t)ti, kv.second};
            PrimBounds b;
            tri_bounds(out.triangles[ti], out.instances[kv.second], b);
            out.build_prims.push_back(pr);
            out.build_bounds.push_back(b);
        }
    }
    for (size_t si = 0; si < out.spheres.size(); ++si) {
        rt_bvh_prim pr{(int32_t)(out.triangles.size() + si), -1};
        PrimBounds b;
        sphere_bounds(out.spheres[si], b);
        out.build_prims.push_back(pr);
        out.build_bounds.push_back(b);
    }
    auto t1 = std::chrono::high_resolution_clock::now();
    out.ingest_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    // RT_GPU_BVH=1: no tree is built here; the scene goes to the rendering core with its primitives' boxes and gets a tree
    // built on the device (faster to make, not the reference's: see rt_scene_desc::bvh_prim_bounds)
    if (const char *e = getenv("RT_GPU_BVH"); e && atoi(e) != 0) {
        out.bvh_ms = 0;
        return true;
    }
    ensure_host_bvh(out);
    scene_cache_store(json_path, out);
    return true;
}

}  // namespace rth
