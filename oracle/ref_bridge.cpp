// TEST INFRASTRUCTURE -- the reference-side binding of the rendering core, compiled against the reference.
//
// This is the translation unit a maintainer of the reference would add to bind include/gpu_render.h: the scene as the
// reference's own parser and BVH builder hold it (parser.h, shapes.h, lights.h, texture.h, bvh.h, camera.h) is
// flattened into an rt_scene_desc, and rt_render takes the place of the thread scheduler of src/main.cpp:437-555.
// It links the reference's translation units where they lie under /root/reference/src (everything except main.cpp) and
// is built only by oracle/Makefile into oracle/_ref/ (git-ignored).  Not part of the product: the tests use it to show
// that the C ABI can be fed from the reference's data structures, and that the result is the product's own.
//
//   ref_bridge scene.json out_prefix dump   [camera]   writes every array of the descriptor as <prefix>.<array>.bin
//                                                      (no GPU call: runs anywhere the library loads)
//   ref_bridge scene.json out_prefix render [camera]   rt_scene_create + rt_render; writes <prefix>.hdr.bin
//
// The reference keeps a vec3 as scalars plus an SSE register that its arithmetic works on; the register is what is
// copied here, except where the reference itself reads the scalars (noted below).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unordered_map>
#include <functional>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <random>

#define private public
#define protected public
#include "bvh.h"
#include "ray_trace.h"
#include "parser.h"
#undef private
#undef protected

#include "gpu_render.h"

namespace {

void put3(float *d, const vec3 &v) {
	float t[4];
	_mm_storeu_ps(t, v.get_vec());
	d[0] = t[0], d[1] = t[1], d[2] = t[2];
}
void put_row(float *d, const vec3 &row) { _mm_storeu_ps(d, row.get_vec()); }
void put34(float *d, const mat4 &m) {   // rows 0..2, four entries each
	put_row(d, m.row0);
	put_row(d + 4, m.row1);
	put_row(d + 8, m.row2);
}
bool is_identity(const mat4 &m) {
	float a[16];
	put_row(a, m.row0), put_row(a + 4, m.row1), put_row(a + 8, m.row2), put_row(a + 12, m.row3);
	for (int i = 0; i < 16; ++i)
		if (a[i] != ((i % 5 == 0) ? 1.0f : 0.0f)) return false;
	return true;
}

struct GpuScene {
	std::vector<rt_triangle> tris;
	std::vector<rt_instance> insts;
	std::vector<rt_sphere> spheres;
	std::vector<rt_plane> planes;
	std::vector<rt_material> mats;
	std::vector<rt_texture> texs;
	std::vector<int32_t> tex_refs;
	std::vector<rt_image> images;
	std::vector<float> texels_f32;
	std::vector<uint8_t> texels_u8;
	std::vector<rt_light> lights;
	std::vector<rt_bvh_node> nodes;
	std::vector<rt_bvh_prim> prims;
	rt_scene_desc desc{};
	rt_scene *handle = nullptr;
};

// What main.cpp:357-421 holds for one file, in the order it makes it.
struct RefScene {
	std::vector<camera> *cameras;
	std::vector<material> *materials;
	std::vector<image *> *images;
	std::vector<texture *> *textures;
	std::vector<Light *> *lights;
	std::vector<shape *> *shapes;
	texture *background = nullptr;   // main.cpp:380-417
	bool is_probe = false;
	const bvh *tree;
};

// Called where main.cpp:421 does `new ray_tracer(...)`.
GpuScene *gpu_scene_from(parser &p, const RefScene &r) {
	auto *g = new GpuScene;
	simd_vec3 calc;
	simd_mat4 mcalc(calc);

	// materials, images, textures: one record each, same order
	for (const material &m : *r.materials) {
		rt_material o{};
		o.type = (int32_t)m.mt;
		o.brdf = (int32_t)m.brdfType;
		o.normalized = m.normalized, o.kdfresnel = m.kdfresnel;
		put3(o.ambient, m.AmbientReflectance), put3(o.diffuse, m.DiffuseReflectance), put3(o.specular, m.SpecularReflectance);
		o.phong_exponent = m.PhongExponent;
		put3(o.mirror, m.MirrorReflectance), put3(o.absorption_coefficient, m.AbsorptionCoefficient);
		o.refraction_index = m.RefractionIndex, o.absorption_index = m.AbsorptionIndex, o.roughness = m.roughness;
		g->mats.push_back(o);
	}
	auto material_index = [&](const material *m) { return (int32_t)(m - r.materials->data()); };
	std::unordered_map<const image *, int> image_index;
	for (const image *im : *r.images) {
		rt_image o{};
		o.width = im->width, o.height = im->height, o.is_hdr = im->is_hdr;
		const size_t texels = (size_t)im->width * im->height;
		if (im->is_hdr) {   // float RGBA
			o.offset = (int32_t)g->texels_f32.size();
			g->texels_f32.insert(g->texels_f32.end(), im->hdr_data, im->hdr_data + texels * 4);
		} else {   // 8-bit RGB (texture.cpp:46 asks the decoder for three channels)
			o.offset = (int32_t)g->texels_u8.size();
			g->texels_u8.insert(g->texels_u8.end(), im->data, im->data + texels * 3);
		}
		image_index[im] = (int)g->images.size();
		g->images.push_back(o);
	}
	std::unordered_map<const texture *, int> texture_index;
	auto add_texture = [&](const texture *t) {
		rt_texture o{};
		o.image = t->im ? image_index.at(t->im) : -1;
		o.kind = t->im ? RT_TEX_IMAGE : (t->ischess ? RT_TEX_CHECKER : RT_TEX_PERLIN);   // dispatch order of texture::sample
		o.decal_mode = (int32_t)t->dmode, o.interp = (int32_t)t->interp;
		o.bump_factor = t->BumpFactor, o.noise_scale = t->NoiseScale, o.noise_conversion = (int32_t)t->nc, o.num_octaves = t->NumOctaves;
		put3(o.black, t->BlackColor), put3(o.white, t->WhiteColor);
		o.scale = t->Scale, o.offset = t->Offset;
		if (!t->ischess) o.scale = 0.f, o.offset = 0.f;   // (never read for the other kinds; the reference leaves them unset)
		texture_index[t] = (int)g->texs.size();
		g->texs.push_back(o);
	};
	for (const texture *t : *r.textures) add_texture(t);
	auto texture_list = [&](const std::vector<texture *> &list, int32_t &offset, int32_t &count) {
		offset = (int32_t)g->tex_refs.size();
		count = (int32_t)list.size();
		for (const texture *t : list) g->tex_refs.push_back(texture_index.at(t));
	};

	// shapes in get_shapes() order: triangles, spheres, planes.  One rt_instance per entry of a triangle's all_mesh_infos.
	std::unordered_map<const shape *, int32_t> shape_index;   // into the arrays of its kind
	std::unordered_map<const all_mesh_infos *, std::unordered_map<int, int32_t>> instance_index;
	for (shape *s : *r.shapes) {
		if (s->get_shapetype() == Triangle) {
			auto *t = static_cast<triangle *>(s);
			rt_triangle o{};
			put3(o.c1, t->c1), put3(o.c2, t->c2), put3(o.c3, t->c3);
			put3(o.tex_u, t->u), put3(o.tex_v, t->v);
			o.smooth = t->smooth;
			if (t->smooth) put3(o.n1, t->n1), put3(o.n2, t->n2), put3(o.n3, t->n3);
			shape_index[s] = (int32_t)g->tris.size();
			g->tris.push_back(o);
			if (!instance_index.count(t->m)) {
				auto &ids = instance_index[t->m];
				for (const auto &kv : t->m->mesh_infos) {
					const mesh_info &mi = kv.second;
					rt_instance inst{};
					put34(inst.model, mi.model), put34(inst.inv_model, mi.inv_model), put34(inst.normal, mi.normal);
					put3(inst.motionblur, mi.motionblur), put3(inst.emission, mi.emission);
					inst.material = material_index(mi.mat);
					texture_list(mi.textures, inst.tex_offset, inst.tex_count);
					inst.ref_id = kv.first;
					inst.identity = is_identity(mi.model);
					ids[kv.first] = (int32_t)g->insts.size();
					g->insts.push_back(inst);
				}
			}
		} else if (s->get_shapetype() == Sphere) {
			auto *sp = static_cast<sphere *>(s);
			rt_sphere o{};
			put3(o.center, sp->center);
			o.radius = sp->radius;
			put34(o.model, sp->model), put34(o.inv_model, sp->inv_model), put34(o.normal, sp->normal_m);
			put3(o.emission, sp->emission);
			o.material = material_index(sp->mat);
			texture_list(sp->textures, o.tex_offset, o.tex_count);
			shape_index[s] = (int32_t)g->spheres.size();
			g->spheres.push_back(o);
		} else {
			auto *pl = static_cast<plane *>(s);
			rt_plane o{};
			put3(o.point, pl->point), put3(o.normal, pl->normal);
			o.raw_normal_x = pl->normal.get_x();   // the scalar copy keeps the value of the scene file (shapes.cpp:533,554)
			o.material = material_index(pl->mat);
			texture_list(pl->textures, o.tex_offset, o.tex_count);
			shape_index[s] = (int32_t)g->planes.size();
			g->planes.push_back(o);
		}
	}

	// lights, in the order of the reference's loop (lights.h)
	for (Light *l : *r.lights) {
		rt_light o{};
		o.samples = l->get_sample_count();
		if (auto *pt = dynamic_cast<PointLight *>(l)) {
			o.type = RT_LIGHT_POINT;
			put3(o.position, pt->position), put3(o.color, pt->intensity);
		} else if (auto *ar = dynamic_cast<AreaLight *>(l)) {
			o.type = RT_LIGHT_AREA;
			put3(o.position, ar->position), put3(o.normal, ar->normal), put3(o.u, ar->u_vec), put3(o.v, ar->v_vec), put3(o.color, ar->radiance);
			o.size = ar->size;
		} else if (auto *dl = dynamic_cast<DirectionalLight *>(l)) {
			o.type = RT_LIGHT_DIRECTIONAL;
			put3(o.direction, dl->direction), put3(o.color, dl->radiance);
		} else if (auto *sl = dynamic_cast<SpotLight *>(l)) {
			o.type = RT_LIGHT_SPOT;
			put3(o.position, sl->position), put3(o.direction, sl->direction), put3(o.color, sl->intensity);
			o.cos_coverage = sl->coverage_angle_cos, o.cos_falloff = sl->falloff_angle_cos;
		} else if (auto *env = dynamic_cast<SphericalDirectionalLight *>(l)) {
			o.type = RT_LIGHT_ENV;
			o.image = env->env_map ? image_index.at(env->env_map) : -1;
			o.cosine_sampling = env->use_cosine_sampling, o.is_probe = env->is_probe_map;
		} else if (auto *tl = dynamic_cast<TriangleLight *>(l)) {
			o.type = RT_LIGHT_TRIANGLE;
			vec3 v0, v1, v2;
			tl->tri->getWorldVertices(calc, mcalc, tl->mesh_id, v0, v1, v2);
			put3(o.v0, v0), put3(o.v1, v1), put3(o.v2, v2), put3(o.normal, tl->normal), put3(o.color, tl->radiance);
			o.area = tl->area;
		} else if (auto *sph = dynamic_cast<SphereLight *>(l)) {
			o.type = RT_LIGHT_SPHERE;
			o.sphere = shape_index.at(sph->sph);
			put3(o.position, sph->sph->center), put3(o.color, sph->radiance);
			o.size = sph->sph->radius;
			vec3 r0(sph->sph->radius, 0.0f, 0.0f), rw;   // lights.cpp:393-461 takes the area from the transformed radius
			mcalc.mult_vec(sph->sph->model, r0, rw, true);
			float w[3];
			put3(w, rw);
			o.area = 4.0f * 3.14159265359f * (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
		} else {
			fprintf(stderr, "ref_bridge: a light of a kind this binding does not know\n");
			exit(2);
		}
		g->lights.push_back(o);
	}

	// the reference's tree (bvh.h:12-28), node for node
	for (const auto &n : r.tree->nodes) {
		rt_bvh_node o{};
		put3(o.bmin, n.bounds.min), put3(o.bmax, n.bounds.max);
		o.offset = n.primitivesOffset;
		o.n_prims = n.nPrimitives;
		o.axis = n.nPrimitives > 0 ? 0 : n.axis;
		g->nodes.push_back(o);
	}
	for (const auto &pr : r.tree->ordered_primitives) {
		rt_bvh_prim o{};
		if (pr.s->get_shapetype() == Triangle) {
			o.shape = shape_index.at(pr.s);
			o.instance = instance_index.at(static_cast<triangle *>(pr.s)->m).at(pr.id);
		} else {
			o.shape = (int32_t)g->tris.size() + shape_index.at(pr.s);
			o.instance = -1;
		}
		g->prims.push_back(o);
	}

	rt_scene_desc &d = g->desc;
	d.abi_version = RT_ABI_VERSION;
	d.background_texture = -1;
	if (r.background) {
		if (!texture_index.count(r.background)) add_texture(r.background);   // the environment map's (main.cpp:403-417)
		d.background_texture = texture_index.at(r.background);
	}
	d.is_probe = r.is_probe;
	d.n_triangles = (int32_t)g->tris.size(), d.triangles = g->tris.data();
	d.n_instances = (int32_t)g->insts.size(), d.instances = g->insts.data();
	d.n_spheres = (int32_t)g->spheres.size(), d.spheres = g->spheres.data();
	d.n_planes = (int32_t)g->planes.size(), d.planes = g->planes.data();
	d.n_materials = (int32_t)g->mats.size(), d.materials = g->mats.data();
	d.n_textures = (int32_t)g->texs.size(), d.textures = g->texs.data();
	d.n_texture_refs = (int32_t)g->tex_refs.size(), d.texture_refs = g->tex_refs.data();
	d.n_images = (int32_t)g->images.size(), d.images = g->images.data();
	d.n_texels_u8 = (int64_t)g->texels_u8.size(), d.texels_u8 = g->texels_u8.data();
	d.n_texels_f32 = (int64_t)g->texels_f32.size(), d.texels_f32 = g->texels_f32.data();
	d.n_lights = (int32_t)g->lights.size(), d.lights = g->lights.data();
	d.n_bvh_nodes = (int32_t)g->nodes.size(), d.bvh_nodes = g->nodes.data();
	d.n_bvh_prims = (int32_t)g->prims.size(), d.bvh_prims = g->prims.data();
	d.intersection_epsilon = p.get_intersectionepsilon();
	d.shadow_ray_epsilon = p.get_shadowrayepsilon();
	d.max_depth = p.get_maxrecursiondepth();
	vec3 amb = p.get_ambientlight(), bgc = p.get_backgroundcolor();
	put3(d.ambient_light, amb), put3(d.background_color, bgc);
	return g;
}

// The camera as its constructor left it (camera.h:35-56) and its RenderSettings (camera.h:23-33).
rt_camera_desc camera_desc_of(const camera &cam) {
	rt_camera_desc c{};
	put3(c.position, cam.position), put3(c.u, cam.u), put3(c.v, cam.v), put3(c.w, cam.w), put3(c.center, cam.center);
	c.near_left = cam.near_left, c.near_right = cam.near_right, c.near_bottom = cam.near_bottom, c.near_top = cam.near_top;
	c.resx = cam.resx, c.resy = cam.resy, c.num_samples = cam.num_samples;
	c.focus_distance = cam.focus_distance, c.aperture_size = cam.aperture_size;
	c.is_hdr = cam.is_hdr;
	const RenderSettings &rs = cam.renderSettings;
	c.path_tracing = rs.isPathTracing, c.importance_sampling = rs.importanceSampling, c.next_event_estimation = rs.nextEventEstimation;
	c.russian_roulette = rs.russianRoulette, c.splitting_factor = rs.splittingFactor;
	c.sample_max_val = rs.sampleMaxVal, c.min_recursion_depth = rs.minRecursionDepth;
	c.seed = 0;
	return c;
}

// Replaces main.cpp:437-555 (thread spawn and join) for one camera; hdr_output as main.cpp:453 lays it out.
void gpu_render_camera(GpuScene *g, const camera &cam, float *hdr_output, rt_stats *st) {
	const rt_camera_desc c = camera_desc_of(cam);
	if (rt_render(g->handle, &c, hdr_output, st)) {
		fprintf(stderr, "%s\n", rt_last_error());
		exit(1);
	}
}

template <class T>
void dump(const std::string &prefix, const char *name, const T *p, size_t n) {
	FILE *f = fopen((prefix + "." + name + ".bin").c_str(), "wb");
	if (!f) {
		fprintf(stderr, "cannot write %s.%s.bin\n", prefix.c_str(), name);
		exit(2);
	}
	if (n) fwrite(p, sizeof(T), n, f);
	fclose(f);
}

}   // namespace

int main(int argc, char **argv) {
	if (argc < 4) {
		fprintf(stderr, "usage: ref_bridge scene.json out_prefix dump|render [camera_index]\n");
		return 1;
	}
	const std::string prefix = argv[2], mode = argv[3];
	const int cam_index = argc > 4 ? atoi(argv[4]) : 0;

	// the ingest calls of main.cpp:357-421, in its order
	parser p(argv[1]);
	simd_vec3 calc;
	simd_mat4 mcalc(calc);
	RefScene r;
	auto transformations = p.get_transformations(mcalc);
	r.cameras = p.get_camera(calc, mcalc, transformations);
	auto vertices = p.get_vertices();
	r.materials = p.get_materials();
	r.images = p.get_images();
	r.textures = p.get_textures(r.images);
	for (auto *t : *r.textures)
		if (t->dmode == replace_background) {
			r.background = t;
			break;
		}
	auto uvs = p.get_uvs();
	r.lights = p.get_lights(calc, mcalc, transformations, r.images);
	std::vector<all_mesh_infos *> infos;
	r.shapes = p.get_shapes(calc, mcalc, vertices, r.materials, transformations, r.textures, uvs, &infos, r.lights);
	for (Light *l : *r.lights)
		if (auto *env = dynamic_cast<SphericalDirectionalLight *>(l)) {
			if (env->env_map) {
				r.background = new texture(const_cast<image *>(env->env_map), -1, replace_background, bilinear);
				r.is_probe = env->is_probe_map;
			}
			break;
		}
	ray_tracer rt(r.shapes, p.get_intersectionepsilon(), p.get_shadowrayepsilon(), p.get_ambientlight(), r.lights, p.get_backgroundcolor(),
				  p.get_maxrecursiondepth());
	r.tree = rt.bvhx;

	GpuScene *g = gpu_scene_from(p, r);
	const camera &cam = r.cameras->at(cam_index);
	if (mode == "dump") {
		dump(prefix, "triangles", g->tris.data(), g->tris.size());
		dump(prefix, "instances", g->insts.data(), g->insts.size());
		dump(prefix, "spheres", g->spheres.data(), g->spheres.size());
		dump(prefix, "planes", g->planes.data(), g->planes.size());
		dump(prefix, "materials", g->mats.data(), g->mats.size());
		dump(prefix, "textures", g->texs.data(), g->texs.size());
		dump(prefix, "texture_refs", g->tex_refs.data(), g->tex_refs.size());
		dump(prefix, "images", g->images.data(), g->images.size());
		dump(prefix, "texels_u8", g->texels_u8.data(), g->texels_u8.size());
		dump(prefix, "texels_f32", g->texels_f32.data(), g->texels_f32.size());
		dump(prefix, "lights", g->lights.data(), g->lights.size());
		dump(prefix, "bvh_nodes", g->nodes.data(), g->nodes.size());
		dump(prefix, "bvh_prims", g->prims.data(), g->prims.size());
		const rt_camera_desc c = camera_desc_of(cam);
		dump(prefix, "camera", &c, 1);
		const rt_scene_desc &d = g->desc;
		const float scalars[12] = {d.intersection_epsilon, d.shadow_ray_epsilon, d.max_depth, d.ambient_light[0], d.ambient_light[1], d.ambient_light[2],
								   d.background_color[0], d.background_color[1], d.background_color[2], (float)d.background_texture, (float)d.is_probe, 0.f};
		dump(prefix, "scalars", scalars, 12);
		return 0;
	}
	if (mode != "render") {
		fprintf(stderr, "ref_bridge: mode must be dump or render\n");
		return 1;
	}
	if (rt_scene_create(&g->desc, &g->handle)) {
		fprintf(stderr, "%s\n", rt_last_error());
		return 1;
	}
	std::vector<float> hdr((size_t)cam.resx * cam.resy * 3);
	rt_stats st;
	gpu_render_camera(g, cam, hdr.data(), &st);
	dump(prefix, "hdr", hdr.data(), hdr.size());
	printf("%d %d primary %llu shadow %llu secondary %llu\n", cam.resx, cam.resy, (unsigned long long)st.primary_rays,
		   (unsigned long long)st.shadow_rays, (unsigned long long)st.secondary_rays);
	rt_scene_destroy(g->handle);
	return 0;
}
