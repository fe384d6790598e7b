// Device-side acceleration-structure maintenance: launch declarations shared by api.cpp and bvh_gpu.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <utility>
#include <vector>

#include "../../include/sunray_hip.h"

namespace srd {
struct FlatInstance {   // 64 B: what flattening one instance needs
    float o2w[12];      // ObjectToWorld3x4, row-major (EntityTransform, rt_types.slang:101-103)
    uint32_t tri_offset, mesh_slot, _pad[2];
};
}  // namespace srd

int srk_launch_flatten_slots(float4* tris, const float4* shade, const SrMeshInfo* meshes, const srd::FlatInstance* instances, uint32_t n_tris,
                             hipStream_t stream);
int srk_launch_refit(uint32_t* nodes, const float4* tris, float* node_box, const uint32_t* level_nodes, const uint32_t* level_offsets_host,
                     uint32_t n_levels, hipStream_t stream);

struct LbvhArgs {
    const SrMeshInfo* meshes; const srd::FlatInstance* instances; uint32_t n_instances, n_tris;
    float4* nodes; uint32_t node_cap;                    // node_cap x 64 B
    float4 *tris, *shade, *shade_tex;                    // leaf-order records (shade_tex may be null)
    uint32_t* slot_of_gid; float* node_box;              // n_tris / node_cap x 6 floats
    void* scratch; size_t scratch_bytes;
    uint32_t stack_floor, stack_cap;                     // budget = max(stack_floor, binary height); fail above stack_cap
    int ploc;                                            // topology: 0 = binary radix tree (LBVH), r > 0 = PLOC with search radius r
};
struct LbvhResult {
    uint32_t n_nodes = 0, max_stack = 0, max_depth = 0;
    std::vector<std::pair<uint32_t, uint32_t>> level_ranges;   // (first node, count) per level, root level first
};
int srk_lbvh_build(const LbvhArgs& args, LbvhResult* out, hipStream_t stream);
size_t srk_lbvh_scratch_bytes(uint32_t n_tris, uint32_t node_cap);
