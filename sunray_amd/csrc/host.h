// Host-side (C++) counterparts of the reference's Rust host code for the hot path.
// Names follow the reference: Camera (camera.rs), ResourceManager tables (resource_manager.rs),
// FrameInstanceData (resource_manager.rs:216-267).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/sunray_hip.h"

namespace srh {

// Camera{position, target, fov_y} (camera.rs:4-8)
struct Camera {
    float position[3] = {0.0f, 0.0f, 1.0f};
    float target[3] = {0.0f, 0.0f, 0.0f};
    float fov_y = 45.0f;
    bool as_matrices(uint32_t width, uint32_t height, const float* prev_view_proj16, SrMatrices* out) const;
};

struct HostMesh {
    uint64_t key = 0;
    std::vector<SrVertex> vertices;     // host copy (BVH build reads positions: blas.rs:266-278)
    std::vector<uint32_t> indices;
    uint32_t n_vertices = 0, n_indices = 0;
    SrMaterial material;
    std::vector<uint32_t> emissive_slots;  // resource_manager.rs:437-446
    void* d_vertices = nullptr;
    void* d_indices = nullptr;
};

struct HostInstance {
    uint32_t mesh_slot;   // instance custom index (resource_manager.rs:239-246)
    SrTransform o2w;
    float w2o[9];
    uint32_t tri_offset;
};

struct FrameInstanceData {
    std::vector<HostInstance> instances;                       // as_instances
    std::vector<SrTransform> transforms;                       // transforms
    std::vector<SrEmissiveIndirectionEntry> emissive_entries;  // emissive_entries
    uint32_t n_triangles = 0;
};

void material_new(const float base_color[4], float metallic, float roughness, const float emissive_factor[3],
                  float emissive_strength, float transmission, float ior, SrMaterial* out);
void emissive_triangles_from_mesh(const SrVertex* vertices, const uint32_t* indices, uint32_t n_indices,
                                  const SrMaterial& material, std::vector<SrEmissiveTriangle>& out);
void world_to_object_3x3(const SrTransform& t, float out[9]);
// Host twins of the shader pack helpers (rt_utils.slang:77-94), used to pre-pack per-mesh payload constants.
uint32_t pack_unorm_4x8(float x, float y, float z, float w);
uint32_t pack_half_2x16(float x, float y);
// Per-light constants (DevLight in traverse.h): 16 floats per emissive_indirection entry.
void light_table(const FrameInstanceData& fid, const std::vector<SrEmissiveTriangle>& emissive_tris, std::vector<float>& out);
bool frame_instance_data(const std::vector<HostMesh>& meshes, const std::map<uint64_t, uint32_t>& slots,
                         const uint64_t* keys, const uint32_t* counts, uint32_t n_keys, const SrTransform* xforms,
                         FrameInstanceData& out, std::string& err);

// ---- BVH (bvh_build.cpp) ------------------------------------------------------------------------
// One world-space triangle in the canonical form the kernels intersect.
struct BuildTri {
    float v0[3], e1[3], e2[3];
    uint32_t prim, inst, gid;
};
struct BvhResult {
    std::vector<float> nodes2; // intermediate binary tree: 16 floats per inner node (two child boxes)
    std::vector<uint32_t> nodes;  // the 4-wide quantised tree the kernels traverse: 16 dwords (64 B) per node
    std::vector<float> tris;   // 12 floats (48 B) per triangle, leaf order
    std::vector<uint32_t> order;  // leaf slot -> index into the input triangle list
    uint32_t n_nodes = 0, max_depth = 0, max_stack = 0;
    float sah_cost = 0.0f;
    double build_ms = 0.0;
};
// Flatten every instance's triangles to world space (instance-major order = global triangle index).
void flatten_instances(const std::vector<HostMesh>& meshes, const FrameInstanceData& fid, std::vector<BuildTri>& out);
// Binned-SAH binary tree with depth bounded by `max_depth`, leaves of <= kLeafMax (2) triangles, collapsed into a
// 4-wide BVH (layout: traverse.h). max_stack = worst-case traversal stack entries for this tree.
void build_bvh(const std::vector<BuildTri>& tris, uint32_t max_depth, BvhResult& out);
// The same builder over arbitrary boxes (the instance boxes of a top-level tree): out.nodes + out.order (leaf references
// index `order`: leaf position -> box index); out.tris stays empty.
struct BuildBox { float lo[3], hi[3]; };
void build_bvh_boxes(const std::vector<BuildBox>& boxes, uint32_t max_depth, BvhResult& out);

// Rebuild-vs-update heuristic of one acceleration structure (acceleration_structure/mod.rs:62-148), on the POD
// state the C ABI exposes (SrAsState). Ops: SR_OP_*.
void as_state_initial(uint32_t build_type, SrAsState* out);
uint32_t as_state_next_op(const SrAsState& s, bool inputs_changed);
void as_state_mark_built(SrAsState& s, uint32_t completed_op);
// Breadth-first levels of a 4-wide tree (16 dwords per node, children at dwords 12..15): node indices sorted by depth,
// deepest level first; level_offsets has n_levels + 1 entries.
void tree_levels(const std::vector<uint32_t>& nodes, std::vector<uint32_t>& level_nodes, std::vector<uint32_t>& level_offsets);

// Baseline JPEG -> 8-bit pixels, 1 (greyscale) or 3 (RGB) channels (jpeg_decode.cpp)
bool decode_jpeg(const uint8_t* data, size_t n, uint32_t& width, uint32_t& height, uint32_t& channels, std::vector<uint8_t>& pixels, std::string& err);
// PNG / JPEG by content, as the loader decodes glTF images (gltf_load.cpp)
bool decode_image(const uint8_t* data, size_t n, uint32_t& width, uint32_t& height, uint32_t& channels, std::vector<uint8_t>& pixels, std::string& err);
bool decode_image_rgba8(const uint8_t* data, size_t n, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& err);

// the one thread-local error slot of the library (api.cpp); returns `code`
int set_error(int code, const std::string& msg);
}  // namespace srh
