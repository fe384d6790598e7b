// Device-side acceleration-structure maintenance: launch declarations shared by api.cpp and bvh_gpu.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

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
