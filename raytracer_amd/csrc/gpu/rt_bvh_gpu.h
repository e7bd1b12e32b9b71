// BVH construction on the device (bvh_gpu.hip): interface between the two translation units of librt_gpu.so.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "gpu_render.h"

namespace rt {

struct DeviceBvh {
    float4 *nodes = nullptr;      // two-box records (DevScene::nodes); record 0 is the root
    float4 *prims = nullptr;      // leaf entries (DevScene::prims), one per primitive, in leaf order
    int4 *prim_meta = nullptr;    // shading records (DevScene::prim_meta)
    int n_interior = 0;
    int n_top = 0;                // the first n_top records are the top levels of the tree, level by level (DevScene::n_top)
    int root_ref = 0;
    int depth = 0;                // interior nodes on the longest root-to-leaf path
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
};

// Builds the tree over `n` (shape, placement) pairs with world-space boxes `bounds` (6 floats each: min, max), both in
// host memory; triangles and instances are the scene's arrays already on the device; `id_scale`: DevScene::id_scale.  Returns 0 and the device arrays
// (owned by the caller, hipFree), or an error code with a message.
int build_bvh_on_device(const rt_bvh_prim *prims, const float *bounds, int n, const rt_triangle *d_triangles, int n_triangles,
                        const rt_instance *d_instances, float id_scale, int max_depth, DeviceBvh &out, std::string &err);

}  // namespace rt
