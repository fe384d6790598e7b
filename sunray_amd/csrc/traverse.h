// BVH traversal + ray/triangle intersection for gfx950 — the replacement for the Vulkan driver's
// TraceRay (call sites ray_gen_ris.slang:75,294,332,381 and
// ray_gen_final.slang:80,212,283,316,370).
//
// Data layout in HBM:
//   nodes : 64 B per inner node = 4 x float4
//           n0 = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)   n1 = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//           n2 = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)   n3 = (child0, child1, -, -) as int bits
//           child >= 0: inner node index; child < 0: leaf, ~child = (first_triangle << 3) | count
//   tris  : 48 B per triangle = 3 x float4, stored in leaf order
//           t0 = (v0.x, v0.y, v0.z, e1.x)  t1 = (e1.y, e1.z, e2.x, e2.y)
//           t2 = (e2.z, primitive index, instance index, global triangle index) (last three as bits)
// One lane = one ray. The per-lane traversal stack lives in LDS, interleaved as stack[level][lane]
// so a wave's pushes/pops hit 64 consecutive banks (conflict-free ds_write_b32 / ds_read_b32).
#pragma once
#include "rt_device.h"

namespace srd {

constexpr int kStackDepth = 32;      // the builder bounds the tree depth to this (bvh_build.cpp)
constexpr int kSentinel = 0x7fffffff;

struct DevInstance {   // 96 B
    float o2w[12];     // ObjectToWorld3x4 (row-major)
    float w2o[9];      // (float3x3)WorldToObject3x4
    uint32_t mesh_slot;
    uint32_t tri_offset;
    uint32_t _pad;
};

struct DevScene {
    const float4* nodes;
    const float4* tris;
    const SrMeshInfo* meshes;
    const DevInstance* instances;
    const SrEmissiveTriangle* emissive;
    const SrEmissiveIndirectionEntry* indirection;
    const SrTransform* transforms;
    unsigned long long* counters;  // [0]=closest queries [1]=any queries [2]=boxes [3]=tris
    uint32_t num_lights;
    uint32_t n_tris;
    uint32_t n_instances;
    uint32_t _pad;
};

struct TravHit {
    float t, u, v;
    uint32_t prim, inst, gid;  // gid == 0xFFFFFFFF: miss
};

struct TravStats { uint32_t boxes, tris; };

// The canonical Möller–Trumbore test. fmaf is spelled out so that the oracle's
// scalar code and this kernel round identically; (tmin, tmax) is exclusive on both sides. The
// barycentric bounds are widened by kBaryEps so a ray through a shared edge hits at least one of the
// two triangles (plain MT leaves cracks there; the Vulkan triangle test is watertight).
constexpr float kBaryEps = 1e-6f;
SRD float dot_fma(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
SRD f3 cross_fma(f3 a, f3 b) {
    return mk3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
SRD bool intersect_tri(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float tmin, float tmax, float& t, float& u, float& v) {
    f3 pvec = cross_fma(d, e2);
    float det = dot_fma(e1, pvec);
    float inv = 1.0f / det;
    f3 tvec = o - v0;
    u = dot_fma(tvec, pvec) * inv;
    f3 qvec = cross_fma(tvec, e1);
    v = dot_fma(d, qvec) * inv;
    t = dot_fma(e2, qvec) * inv;
    return (u >= -kBaryEps) && (v >= -kBaryEps) && (u + v <= 1.0f + kBaryEps) && (t > tmin) && (t < tmax);
}

// Conservative slab test: a box is only rejected if no triangle hit with t in (t_lo, t_hi] can lie
// inside it. Planes are taken in the ray's own order (near plane = hi for a negative direction), so
// a zero direction component yields -inf/+inf inside the slab and NaN exactly on a face; v_max/v_min
// (IEEE maxNum/minNum) drop the NaN, i.e. slabs are closed. The far side is inflated (Ize 2013).
struct RaySetup {
    f3 o, inv;
    bool sx, sy, sz;
};
SRD RaySetup ray_setup(f3 o, f3 d) {
    RaySetup r;
    r.o = o;
    r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.sx = (__float_as_uint(r.inv.x) >> 31) != 0u;
    r.sy = (__float_as_uint(r.inv.y) >> 31) != 0u;
    r.sz = (__float_as_uint(r.inv.z) >> 31) != 0u;
    return r;
}
SRD bool slab(float lox, float hix, float loy, float hiy, float loz, float hiz, const RaySetup& r, float t_lo, float t_hi, float& tnear) {
    const float nx = ((r.sx ? hix : lox) - r.o.x) * r.inv.x, fx = ((r.sx ? lox : hix) - r.o.x) * r.inv.x;
    const float ny = ((r.sy ? hiy : loy) - r.o.y) * r.inv.y, fy = ((r.sy ? loy : hiy) - r.o.y) * r.inv.y;
    const float nz = ((r.sz ? hiz : loz) - r.o.z) * r.inv.z, fz = ((r.sz ? loz : hiz) - r.o.z) * r.inv.z;
    const float t0 = fmaxf(fmaxf(nx, ny), fmaxf(nz, t_lo));
    float far = fminf(fminf(fx, fy), fz);
    far = far * (1.0f + copysignf(5e-7f, far));  // away from zero; keeps +-inf (an fma form turns -inf into NaN)
    const float t1 = fminf(far, t_hi);
    tnear = t0;
    return t0 <= t1;
}

// stack: this lane's column of the LDS stack (element k at stack[k * stride]).
template <bool ANY, bool STATS>
SRD bool traverse(const DevScene& sc, f3 o, f3 d, float tmin, float tmax, TravHit& hit, int* stack, int stride, TravStats& st) {
    const float4* __restrict__ nodes = sc.nodes;
    const float4* __restrict__ tris = sc.tris;
    const RaySetup rs = ray_setup(o, d);
    const uint32_t boxes_at_entry = STATS ? st.boxes : 0u;
    // Near bound of the box test: one |tmin| BELOW tmin. Close to the origin the triangle test's t carries
    // an absolute error far above 1e-5*tmin (cancellation in o - v0), so a relative slack is not enough.
    const float t_lo = fminf(tmin, 0.0f) - fabsf(tmin);
    // Box culling bounds are relaxed on both ends; only the triangle test applies the exact (tmin, tmax).
    float cull = fmaf(fabsf(tmax), 1e-5f, tmax);
    float best_t = tmax;
    hit.t = -1.0f; hit.u = 0.0f; hit.v = 0.0f; hit.prim = 0; hit.inst = 0; hit.gid = 0xFFFFFFFFu;
    int sp = 0;
    int node = 0;  // the root is always inner node 0
    while (node != kSentinel) {
        while (node >= 0 && node != kSentinel) {
            const float4 n0 = nodes[node * 4 + 0];
            const float4 n1 = nodes[node * 4 + 1];
            const float4 n2 = nodes[node * 4 + 2];
            const float4 n3 = nodes[node * 4 + 3];
            float tn0, tn1;
            const bool h0 = slab(n0.x, n0.y, n0.z, n0.w, n2.x, n2.y, rs, t_lo, cull, tn0);
            const bool h1 = slab(n1.x, n1.y, n1.z, n1.w, n2.z, n2.w, rs, t_lo, cull, tn1);
            if (STATS) st.boxes += 2;
            int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
            if (h0 && h1) {
                if (tn1 < tn0) { int t = c0; c0 = c1; c1 = t; }
                stack[sp * stride] = c1;
                sp++;
                node = c0;
            } else if (h0) {
                node = c0;
            } else if (h1) {
                node = c1;
            } else {
                if (sp == 0) node = kSentinel;
                else { sp--; node = stack[sp * stride]; }
            }
        }
        if (node == kSentinel) break;
        // leaf
        const uint32_t lv = ~(uint32_t)node;
        const uint32_t first = lv >> 3, cnt = lv & 7u;
        for (uint32_t i = 0; i < cnt; i++) {
            const float4 t0 = tris[(first + i) * 3 + 0];
            const float4 t1 = tris[(first + i) * 3 + 1];
            const float4 t2 = tris[(first + i) * 3 + 2];
            float t, u, v;
            if (STATS) st.tris += 1;
            if (intersect_tri(o, d, mk3(t0.x, t0.y, t0.z), mk3(t0.w, t1.x, t1.y), mk3(t1.z, t1.w, t2.x), tmin, tmax, t, u, v)) {
                if (ANY) return true;
                const uint32_t gid = __float_as_uint(t2.w);
                if (t < best_t || (t == best_t && gid < hit.gid)) {
                    best_t = t;
                    hit.t = t; hit.u = u; hit.v = v;
                    hit.prim = __float_as_uint(t2.y); hit.inst = __float_as_uint(t2.z); hit.gid = gid;
                    cull = fmaf(fabsf(t), 1e-5f, t);
                }
            }
        }
        if (sp == 0) node = kSentinel;
        else { sp--; node = stack[sp * stride]; }
    }
    if (STATS) {   // diagnostics: remember the most expensive ray of the launch
        const uint32_t steps = st.boxes - boxes_at_entry;
        uint32_t* dbg = reinterpret_cast<uint32_t*>(sc.counters) + 16;
        if (steps > 20000u && atomicMax(dbg, steps) < steps) {
            float* r = reinterpret_cast<float*>(dbg + 8);
            r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = tmin; r[4] = d.x; r[5] = d.y; r[6] = d.z; r[7] = tmax;
            dbg[1] = ANY ? 1u : 0u;
        }
    }
    return hit.gid != 0xFFFFFFFFu;
}

// Payload of one closest-hit query in registers (rt_types.slang:9-16).
struct Payload {
    f3 emission;
    float dist;
    uint32_t albedo_packed, normal_packed, material_info, transmission_ior_packed;
};

// closest_hit.slang:12-91 for NULL-texture materials (sample_texture returns its fallback,
// rt_utils.slang:127-129); ray_miss.slang:10-13 for misses.
SRD Payload shade_hit(const DevScene& sc, const TravHit& h) {
    Payload pl;
    pl.emission = splat(0.0f);
    pl.albedo_packed = 0; pl.normal_packed = 0; pl.material_info = 0; pl.transmission_ior_packed = 0;
    if (h.gid == 0xFFFFFFFFu) { pl.dist = -1.0f; return pl; }
    const DevInstance* inst = sc.instances + h.inst;
    const SrMeshInfo* mi = sc.meshes + inst->mesh_slot;
    const SrVertex* verts = (const SrVertex*)mi->vertices;
    const uint32_t* idx = (const uint32_t*)mi->indices;
    const uint32_t io = h.prim * 3;
    const uint32_t i0 = idx[io + 0], i1 = idx[io + 1], i2 = idx[io + 2];
    const f3 bary = mk3(1.0f - h.u - h.v, h.u, h.v);
    const f3 na = ld3(verts[i0].normal), nb = ld3(verts[i1].normal), nc = ld3(verts[i2].normal);
    const f3 normal = na * bary.x + nb * bary.y + nc * bary.z;
    const SrMaterial& m = mi->material;
    const f3 base_color = mk3(m.base_color_value[0], m.base_color_value[1], m.base_color_value[2]);
    const f3 final_emission = mk3(m.emissive_factor[0], m.emissive_factor[1], m.emissive_factor[2]) * m.emissive_factor[3];
    const float* W = inst->w2o;
    const f3 world_normal = norm3(mk3((normal.x * W[0] + normal.y * W[3]) + normal.z * W[6],
                                      (normal.x * W[1] + normal.y * W[4]) + normal.z * W[7],
                                      (normal.x * W[2] + normal.y * W[5]) + normal.z * W[8]));
    pl.dist = h.t;
    pl.emission = final_emission;
    pl.albedo_packed = pack_unorm_4x8(base_color.x, base_color.y, base_color.z, 1.0f);
    pl.normal_packed = pack_normal(world_normal);
    pl.material_info = pack_half_2x16(m.roughness_factor, m.metallic_factor);
    pl.transmission_ior_packed = pack_half_2x16(m.transmission_factor, m.ior);
    return pl;
}

}  // namespace srd
