// ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Parity status: UNPINNED (see orc_math.h).
// Scene tables, host-side data preparation and the ray/triangle queries
// (K2/K3) that stand in for the Vulkan driver's TraceRay: a brute-force O(N) intersector as ground
// truth plus an independent BVH (own builder, own box test) used where brute force is too slow.
#pragma once
#include <cstdint>
#include <map>
#include <vector>

#include "orc_utils.h"
#include "orc_texture.h"

namespace orc {

struct Mesh {
    uint64_t key;
    std::vector<SrVertex> vertices;
    std::vector<uint32_t> indices;
    SrMaterial material;
    std::vector<uint32_t> emissive_slots;  // resource_manager.rs:437-446
};

struct Instance {
    uint32_t mesh_slot;    // instance custom index = mesh-info slot (resource_manager.rs:239-246)
    SrTransform o2w;       // ObjectToWorld3x4
    float w2o[9];          // (float3x3)WorldToObject3x4, row-major
    uint32_t tri_offset;   // first global triangle index
};

// World-space triangle in the canonical form both the oracle and the kernels intersect against:
// v0 and the two edges e1 = v1 - v0, e2 = v2 - v0 of the transformed vertices.
struct WTri {
    V3 v0, e1, e2;
    uint32_t instance, prim;
};

struct BvhNode {
    V3 lo, hi;
    uint32_t left, right;  // children (inner) ...
    uint32_t first, count; // ... or triangle range in `order` (leaf when count > 0)
};

struct Hit {
    float t, u, v;
    uint32_t tri;  // global triangle index, 0xFFFFFFFF on miss
};

struct Counters {
    uint64_t closest = 0, any = 0, boxes = 0, tris = 0;
};

struct Scene {
    std::vector<Mesh> meshes;             // slot order
    std::map<uint64_t, uint32_t> slots;   // key -> mesh-info slot
    std::vector<SrEmissiveTriangle> emissive_tris;  // the emissive arena
    std::vector<Image> images;            // image slot order (Material::*_image)
    std::vector<SrSamplerDesc> samplers;  // sampler slot order (Material::*_sampler)
    // frame data (resource_manager.rs:216-267 + lib.rs:1058-1081)
    std::vector<Instance> instances;
    std::vector<SrTransform> transforms;
    std::vector<SrEmissiveIndirectionEntry> indirection;
    // flattened geometry + BVH
    std::vector<WTri> tris;
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> order;
    bool use_brute_force = false;
    Counters counters;

    int add_mesh(uint64_t key, const SrVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const SrMaterial* m);
    int add_blas(uint64_t key, const SrVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const SrMaterial* m,
                 const SrEmissiveTriangle* et, uint32_t n_et);
    void remove(uint64_t key);
    std::vector<uint32_t> free_mesh_slots, free_emissive_slots;   // LIFO (buffer/arena_core.rs)
    int add_image(const uint8_t* data, uint32_t w, uint32_t h, uint32_t channels);
    int add_sampler(const SrSamplerDesc* d);
    V4 sample_texture(uint32_t image_slot, uint32_t sampler_slot, float s, float t, V4 fallback) const;
    int set_instances(const uint64_t* keys, const uint32_t* counts, uint32_t n_keys, const SrTransform* xf);
    uint32_t num_lights() const { return (uint32_t)indirection.size(); }

    void build_bvh();
    Hit closest_brute(V3 o, V3 d, float tmin, float tmax) const;
    bool any_brute(V3 o, V3 d, float tmin, float tmax) const;
    Hit closest_bvh(V3 o, V3 d, float tmin, float tmax, Counters* c) const;
    bool any_bvh(V3 o, V3 d, float tmin, float tmax, Counters* c) const;
    Hit closest(V3 o, V3 d, float tmin, float tmax, Counters* c) const {
        return use_brute_force ? closest_brute(o, d, tmin, tmax) : closest_bvh(o, d, tmin, tmax, c);
    }
    bool any(V3 o, V3 d, float tmin, float tmax, Counters* c) const {
        return use_brute_force ? any_brute(o, d, tmin, tmax) : any_bvh(o, d, tmin, tmax, c);
    }
    // closest_hit.slang:12-91 / ray_miss.slang:10-13
    SrRayPayload shade_hit(const Hit& h) const;
    bool any_hit_ignores(const Hit& h) const;   // any_hit.slang:11-43
};

// H1/H2: camera.rs:33-63 + lib.rs:1017-1048
void camera_matrices(const float pos[3], const float target[3], float fov_y_deg, uint32_t w, uint32_t h,
                     const float* prev_view_proj16, SrMatrices* out);
// H6: resources/material.rs:52-92 with the NULL-texture resolver of lib.rs:937-943
void material_new(const float base_color[4], float metallic, float roughness, const float emissive_factor[3],
                  float emissive_strength, float transmission, float ior, SrMaterial* out);
// (float3x3) inverse used for WorldToObject
void inverse3x3(const SrTransform& t, float out[9]);

// The canonical Möller–Trumbore test: explicit fmaf in dot/cross, exclusive (tmin,tmax).
// The barycentric bounds are widened by kBaryEps so that a ray through a shared edge hits at least one
// of the two triangles (plain MT leaves cracks there; the Vulkan triangle test is watertight).
constexpr float kBaryEps = 1e-6f;
static inline float dot_fma(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline V3 cross_fma(V3 a, V3 b) {
    return V3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
static inline bool intersect_tri(V3 o, V3 d, const WTri& tr, float tmin, float tmax, float& t, float& u, float& v) {
    V3 pvec = cross_fma(d, tr.e2);
    float det = dot_fma(tr.e1, pvec);
    float inv = 1.0f / det;
    V3 tvec = o - tr.v0;
    u = dot_fma(tvec, pvec) * inv;
    V3 qvec = cross_fma(tvec, tr.e1);
    v = dot_fma(d, qvec) * inv;
    t = dot_fma(tr.e2, qvec) * inv;
    return (u >= -kBaryEps) && (v >= -kBaryEps) && (u + v <= 1.0f + kBaryEps) && (t > tmin) && (t < tmax);
}

struct PassParams {
    const SrRtParams* p;
};
void trace_ris(Scene& s, const SrRtParams& p);
void trace_final(Scene& s, const SrRtParams& p);

}  // namespace orc
