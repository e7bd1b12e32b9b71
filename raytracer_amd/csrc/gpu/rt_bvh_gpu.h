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
    float4 *wide_nodes = nullptr; // four-box records (DevScene::wnodes), record 0 heads the root; nullptr when the tree has none
    int n_wide = 0;               // records allocated (places the tree does not use are zero)
    int n_wide_heads = 0;         // records the tree uses
    int n_wide_top = 0;           // the first n_wide_top of them are the top of the tree, breadth-first (DevScene::n_wtop)
    rt_bvh_node *tree_nodes = nullptr;   // records_of_host_tree with keep_nodes: the scene's node array on the device
    int root_ref = 0;
    int depth = 0;                // interior nodes on the longest root-to-leaf path
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
};

// Builds the tree over `n` (shape, placement) pairs with world-space boxes `bounds` (6 floats each: min, max), both in
// host memory; triangles and instances are the scene's arrays already on the device; `id_scale`: DevScene::id_scale;
// `wide_stack_entries`: the stack a walk of the four-box records has (0: do not make them).  Returns 0 and the device arrays
// (owned by the caller; they come from device_blocks() of rt_devmem.h and go back to it), or an error code with a message.
int build_bvh_on_device(const rt_bvh_prim *prims, const float *bounds, int n, const rt_triangle *d_triangles, int n_triangles,
                        const rt_instance *d_instances, float id_scale, int max_depth, int wide_stack_entries, DeviceBvh &out, std::string &err);

// The records of a tree that came with the scene, made on the device from the reference's node array (src/bvh.h:9-28; nodes,
// top and prims in host memory): the two-box records of the `n_top` interior nodes listed in `top` (the tree's top, breadth
// first, root first) take the first places, every other interior node's record the place n_top + its number among the
// n_interior interior nodes in array order; leaf entries keep the order of `prims`.  Fills out.nodes, out.prims,
// out.prim_meta (owned by the caller, as above).
// `keep_nodes`: the uploaded node array stays on the device (out.tree_nodes, owned by the caller) for wide_records_of_host_tree.
int records_of_host_tree(const rt_bvh_node *nodes, int n_nodes, const int *top, int n_top, int n_interior, const rt_bvh_prim *prims, int n, const rt_triangle *d_triangles,
                         int n_triangles, const rt_instance *d_instances, float id_scale, bool keep_nodes, DeviceBvh &out, std::string &err);

// The four-box records (DevScene::wnodes) of a tree that came with the scene, from its node array on the device: the records
// build_wide_nodes of rt_wide.h makes on the host, in another order below the top.  `top`: the first heads breadth-first (host
// memory, at most 127; record k heads node top[k], top[0] = 0 the root); `stack_entries`: the stack a walk has.  Fills
// out.wide_nodes / n_wide / n_wide_top; leaves them empty (and returns 0) when the tree does not qualify: a box that is not
// finite, or a walk that could need more stack.
int wide_records_of_host_tree(const rt_bvh_node *d_nodes, int n_nodes, const int *top, int n_top, int stack_entries, DeviceBvh &out, std::string &err);

}  // namespace rt
