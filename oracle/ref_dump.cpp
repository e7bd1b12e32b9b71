// TEST INFRASTRUCTURE — golden-vector generator.  Not part of the product.
//
// Links against the reference's own translation units where they lie under
// /root/reference/src (everything except main.cpp) and dumps, for one scene:
//   <prefix>.meta.txt   resx resy nshapes nnodes nprims nplanes
//   <prefix>.ids.i32    per pixel (row j = 0 is the bottom row): shape index, instance id
//   <prefix>.t.f32      per pixel: primary-hit distance (1e30 on miss)
//   <prefix>.hdr.f32    per pixel: rgb as stored by the reference's process_pixel
//   <prefix>.bvh.bin    the reference's flattened BVH (nodes + ordered primitives)
// Primitive identity = (position of the shape in parser::get_shapes() output, instance id),
// -1/-1 on a miss.  Built only by oracle/Makefile into oracle/_ref/ (git-ignored).
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
#undef private
#undef protected
#include "parser.h"
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "../third_party/stb/stb_image_write.h"

static void write_file(const std::string &path, const void *p, size_t n)
{
	FILE *f = fopen(path.c_str(), "wb");
	if (!f)
	{
		fprintf(stderr, "cannot write %s\n", path.c_str());
		exit(2);
	}
	fwrite(p, 1, n, f);
	fclose(f);
}

int main(int argc, char **argv)
{
	if (argc < 3)
	{
		fprintf(stderr, "usage: ref_dump scene.json out_prefix [camera_index] [noimage]\n");
		return 1;
	}
	const char *scene = argv[1];
	std::string prefix = argv[2];
	int cam_index = argc > 3 ? atoi(argv[3]) : 0;
	bool want_image = !(argc > 4 && strcmp(argv[4], "noimage") == 0);

	parser p(scene);
	simd_vec3 calc;
	simd_mat4 mcalc(calc);
	auto transformations = p.get_transformations(mcalc);
	auto cameras = p.get_camera(calc, mcalc, transformations);
	auto vertices = p.get_vertices();
	auto materials = p.get_materials();
	auto images = p.get_images();
	auto textures = p.get_textures(images);
	texture *bg = nullptr;
	for (auto *t : *textures)
		if (t->dmode == replace_background)
		{
			bg = t;
			break;
		}
	auto uvs = p.get_uvs();
	auto lights = p.get_lights(calc, mcalc, transformations, images);
	std::vector<all_mesh_infos *> infos;
	auto shapes = p.get_shapes(calc, mcalc, vertices, materials, transformations, textures, uvs, &infos, lights);
	bool is_probe = false;
	for (Light *l : *lights)
	{
		auto *env = dynamic_cast<SphericalDirectionalLight *>(l);
		if (env)
		{
			if (env->env_map)
			{
				bg = new texture(const_cast<image *>(env->env_map), -1, replace_background, bilinear);
				is_probe = env->is_probe_map;
			}
			break;
		}
	}
	ray_tracer rt(shapes, p.get_intersectionepsilon(), p.get_shadowrayepsilon(), p.get_ambientlight(), lights,
				  p.get_backgroundcolor(), p.get_maxrecursiondepth());
	const bvh *b = rt.bvhx;

	std::unordered_map<const shape *, int> index_of;
	for (size_t i = 0; i < shapes->size(); ++i)
		index_of[(*shapes)[i]] = (int)i;

	camera &cam = cameras->at(cam_index);
	const int W = cam.resx, H = cam.resy;
	std::vector<int32_t> ids((size_t)W * H * 2);
	std::vector<float> ts((size_t)W * H);
	std::vector<float> hdr((size_t)W * H * 3);
	const float eps = p.get_intersectionepsilon();

	for (int j = 0; j < H; ++j)
		for (int i = 0; i < W; ++i)
		{
			size_t px = (size_t)j * W + i;
			auto samples = cam.get_samples(calc, i, j);
			float t;
			shape *hs = nullptr;
			int id = -1;
			bool hit = b->intersect(calc, mcalc, samples[0].position, samples[0].direction, 0.0f, t, &hs, id, false, eps);
			ids[px * 2 + 0] = hit ? index_of[hs] : -1;
			ids[px * 2 + 1] = hit ? id : -1;
			ts[px] = hit ? t : 1e30f;
			if (want_image)
			{
				float pixelv = 1.0f - ((float)j / (float)(H - 1));
				float pixelu = (float)i / (float)(W - 1);
				float acc[3] = {0, 0, 0};
				for (auto &s : samples)
				{
					float c[3];
					rt.trace(calc, mcalc, s.position, s.direction, s.time, false, cam.is_hdr, bg, is_probe, pixelu, pixelv, c,
							 cam.renderSettings);
					acc[0] += c[0];
					acc[1] += c[1];
					acc[2] += c[2];
				}
				float n = (float)samples.size();
				hdr[px * 3 + 0] = acc[0] / n;
				hdr[px * 3 + 1] = acc[1] / n;
				hdr[px * 3 + 2] = acc[2] / n;
			}
		}

	write_file(prefix + ".ids.i32", ids.data(), ids.size() * 4);
	write_file(prefix + ".t.f32", ts.data(), ts.size() * 4);
	if (want_image)
		write_file(prefix + ".hdr.f32", hdr.data(), hdr.size() * 4);

	// BVH dump: per node 6 floats bounds, int offset, int nprims, int axis; per primitive shape index, id.
	std::vector<float> nb;
	std::vector<int32_t> ni, prims;
	for (auto &n : b->nodes)
	{
		nb.push_back(n.bounds.min.get_x());
		nb.push_back(n.bounds.min.get_y());
		nb.push_back(n.bounds.min.get_z());
		nb.push_back(n.bounds.max.get_x());
		nb.push_back(n.bounds.max.get_y());
		nb.push_back(n.bounds.max.get_z());
		ni.push_back(n.primitivesOffset);
		ni.push_back(n.nPrimitives);
		ni.push_back(n.nPrimitives > 0 ? 0 : n.axis);
	}
	for (auto &pr : b->ordered_primitives)
	{
		prims.push_back(index_of[pr.s]);
		prims.push_back(pr.id);
	}
	{
		FILE *f = fopen((prefix + ".bvh.bin").c_str(), "wb");
		int32_t hdrv[2] = {(int32_t)b->nodes.size(), (int32_t)b->ordered_primitives.size()};
		fwrite(hdrv, 4, 2, f);
		fwrite(nb.data(), 4, nb.size(), f);
		fwrite(ni.data(), 4, ni.size(), f);
		fwrite(prims.data(), 4, prims.size(), f);
		fclose(f);
	}
	{
		FILE *f = fopen((prefix + ".meta.txt").c_str(), "w");
		fprintf(f, "%d %d %zu %zu %zu %zu\n", W, H, shapes->size(), b->nodes.size(), b->ordered_primitives.size(),
				b->plane_shapes.size());
		fclose(f);
	}
	return 0;
}
