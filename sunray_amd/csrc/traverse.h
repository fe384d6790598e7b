// BVH traversal + ray/triangle intersection for gfx950 — the replacement for the Vulkan driver's
// TraceRay (call sites ray_gen_ris.slang:75,294,332,381 and
// ray_gen_final.slang:80,212,283,316,370).
//
// Data layout in HBM:
//   nodes : 4-wide BVH, one 128-byte node (= one L2 line) per step: 8 x float4
//           [0] lo.x[4] [1] hi.x[4] [2] lo.y[4] [3] hi.y[4] [4] lo.z[4] [5] hi.z[4] [6] child[4] [7] pad
//           child >= 0: node index; child < 0: leaf, ~child = (first_triangle << 3) | count;
//           an unused child has lo = +inf, hi = -inf (nothing hits it) and an empty leaf reference.
//           A lane reads the NEAR plane of an axis straight from lo or hi by its ray's sign — the
//           loads are per-lane gathers anyway, so ordering the slabs costs no select instructions.
//   tris  : 48 B per triangle = 3 x float4, in leaf order:
//           (v0.x, v0.y, v0.z, e1.x) (e1.y, e1.z, e2.x, e2.y) (e2.z, global triangle index, -, -)
//   shade : 48 B per triangle, same order, read once per committed closest hit:
//           (n0.x, n0.y, n0.z, n1.x) (n1.y, n1.z, n2.x, n2.y) (n2.z, instance index, mesh slot, -)
//           (object-space vertex normals: the only vertex attribute closest_hit needs without textures)
// One lane = one ray. The per-lane traversal stack lives in LDS, interleaved as stack[level][lane]
// so a wave's pushes/pops hit 64 consecutive banks (conflict-free ds_write_b32 / ds_read_b32); the
// rare entries beyond kStackLds spill to a private array.
#pragma once
#include "rt_device.h"

namespace srd {

constexpr int kStackLane = 48;        // LDS stack entries of a per-lane traversal (48 KB per 256 threads)
constexpr int kStackQuad = 64;        // LDS stack entries of a quad traversal (64 quads x 64 x 4 B = 16 KB per 256 threads)
constexpr int kStackMax = 48;         // the scene build fails if a traversal could need more than this
constexpr int kMaxBinaryDepth = 32;  // depth bound of the binary tree the 4-wide tree is collapsed from
constexpr int kSentinel = 0x7fffffff;

struct DevInstance {   // 48 B: (float3x3)WorldToObject3x4, row-major, padded to 3 x float4
    float w2o[12];
};
struct DevMeshConst {  // 32 B: the per-mesh constants of the 32-byte RayPayload (no textures in scope)
    float emission[3];           // emissive_factor.rgb * strength      (closest_hit.slang:43-46)
    uint32_t albedo_packed;      // pack_unorm_4x8(base_color.rgb, 1)   (:76)
    uint32_t material_info;      // pack_half_2x16(roughness, metallic) (:79-89)
    uint32_t transmission_ior_packed;  // (:90)
    uint32_t _pad[2];
};

struct DevScene {
    const float4* nodes;
    const float4* nodes_q;
    const float4* tris;
    const float4* shade;
    const DevMeshConst* mesh_const;
    const DevInstance* instances;
    const SrEmissiveTriangle* emissive;
    const SrEmissiveIndirectionEntry* indirection;
    const SrTransform* transforms;
    const uint32_t* slot_of_gid;   // global triangle index -> leaf-order slot (sr_shade_closest_hit only)
    unsigned long long* counters;  // [0]=closest queries [1]=any queries [2]=boxes [3]=tris
    uint32_t num_lights;
    uint32_t n_tris;
    uint32_t n_instances;
    uint32_t _pad;
};

struct TravHit {
    float t, u, v;
    uint32_t gid, slot;  // gid == 0xFFFFFFFF: miss; slot = position in tris/shade
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

// Conservative slab tests: a box is only rejected if no triangle hit with t in
// (t_lo, t_hi] can lie inside it. Planes are taken in the ray's own order, so a zero direction
// component yields -inf/+inf inside the slab and NaN exactly on a face; v_max/v_min (IEEE
// maxNum/minNum) drop the NaN, i.e. slabs are closed. The far side is inflated (Ize 2013).
struct RaySetup {
    f3 o, inv;
    int nx, ny, nz;   // float4 index of the NEAR plane array of each axis inside a node (far = the other)
    int fx, fy, fz;
};
SRD RaySetup ray_setup(f3 o, f3 d) {
    RaySetup r;
    r.o = o;
    r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const int sx = (int)(__float_as_uint(r.inv.x) >> 31), sy = (int)(__float_as_uint(r.inv.y) >> 31), sz = (int)(__float_as_uint(r.inv.z) >> 31);
    r.nx = 0 + sx; r.fx = 1 - sx;
    r.ny = 2 + sy; r.fy = 3 - sy;
    r.nz = 4 + sz; r.fz = 5 - sz;
    return r;
}
SRD float slab_near(float px, float py, float pz, const RaySetup& r, float t_lo) {
    return fmaxf(fmaxf((px - r.o.x) * r.inv.x, (py - r.o.y) * r.inv.y), fmaxf((pz - r.o.z) * r.inv.z, t_lo));
}
SRD float slab_far(float px, float py, float pz, const RaySetup& r, float t_hi) {
    float far = fminf(fminf((px - r.o.x) * r.inv.x, (py - r.o.y) * r.inv.y), (pz - r.o.z) * r.inv.z);
    far = far * (1.0f + copysignf(5e-7f, far));  // away from zero; keeps +-inf (an fma form turns -inf into NaN)
    return fminf(far, t_hi);
}

// LDS stack helpers: element k of a column lives at base[k * stride]. The builder bounds the number of
// entries a traversal of the tree can need (BvhResult::max_stack <= kStackMax), so there is no overflow path.
#define SR_PUSH(v) do { stack_base[sp * stride] = (v); sp++; } while (0)
#define SR_POP() (sp == 0 ? kSentinel : stack_base[(--sp) * stride])

SRD int pick(int4 c, uint32_t i) { return i == 0u ? c.x : (i == 1u ? c.y : (i == 2u ? c.z : c.w)); }
SRD void cswap(uint32_t& a, uint32_t& b) { const uint32_t lo = min(a, b), hi = max(a, b); a = lo; b = hi; }

template <bool ANY, bool STATS>
SRD bool traverse(const DevScene& sc, f3 o, f3 d, float tmin, float tmax, TravHit& hit, int* stack_base, int stride, TravStats& st) {
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
    hit.t = -1.0f; hit.u = 0.0f; hit.v = 0.0f; hit.gid = 0xFFFFFFFFu; hit.slot = 0u;
    int sp = 0;
    int node = 0;  // the root is always inner node 0
    while (node != kSentinel) {
        while (node >= 0 && node != kSentinel) {
            const float4* n = nodes + (size_t)node * 8;
            const float4 qnx = n[rs.nx], qfx = n[rs.fx], qny = n[rs.ny], qfy = n[rs.fy], qnz = n[rs.nz], qfz = n[rs.fz];
            const float4 qc = n[6];
            const int4 child = make_int4(__float_as_int(qc.x), __float_as_int(qc.y), __float_as_int(qc.z), __float_as_int(qc.w));
            if (STATS) st.boxes += 4;
            const float n0 = slab_near(qnx.x, qny.x, qnz.x, rs, t_lo), f0 = slab_far(qfx.x, qfy.x, qfz.x, rs, cull);
            const float n1 = slab_near(qnx.y, qny.y, qnz.y, rs, t_lo), f1 = slab_far(qfx.y, qfy.y, qfz.y, rs, cull);
            const float n2 = slab_near(qnx.z, qny.z, qnz.z, rs, t_lo), f2 = slab_far(qfx.z, qfy.z, qfz.z, rs, cull);
            const float n3 = slab_near(qnx.w, qny.w, qnz.w, rs, t_lo), f3_ = slab_far(qfx.w, qfy.w, qfz.w, rs, cull);
            if (ANY) {
                // order is irrelevant for an existence query: continue with the first hit child, push the rest
                int next = kSentinel;
                bool have = false;
                if (n0 <= f0) { next = child.x; have = true; }
                if (n1 <= f1) { if (have) SR_PUSH(child.y); else { next = child.y; have = true; } }
                if (n2 <= f2) { if (have) SR_PUSH(child.z); else { next = child.z; have = true; } }
                if (n3 <= f3_) { if (have) SR_PUSH(child.w); else { next = child.w; have = true; } }
                node = have ? next : SR_POP();
            } else {
                // sort the hit children near-to-far: key = entry distance (clamped to >= 0, low 2 mantissa
                // bits replaced by the child slot) — positive floats order like unsigned integers
                uint32_t k0 = (n0 <= f0) ? ((__float_as_uint(fmaxf(n0, 0.0f)) & ~3u) | 0u) : 0xFFFFFFFFu;
                uint32_t k1 = (n1 <= f1) ? ((__float_as_uint(fmaxf(n1, 0.0f)) & ~3u) | 1u) : 0xFFFFFFFFu;
                uint32_t k2 = (n2 <= f2) ? ((__float_as_uint(fmaxf(n2, 0.0f)) & ~3u) | 2u) : 0xFFFFFFFFu;
                uint32_t k3 = (n3 <= f3_) ? ((__float_as_uint(fmaxf(n3, 0.0f)) & ~3u) | 3u) : 0xFFFFFFFFu;
                cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2);
                if (k3 != 0xFFFFFFFFu) SR_PUSH(pick(child, k3 & 3u));
                if (k2 != 0xFFFFFFFFu) SR_PUSH(pick(child, k2 & 3u));
                if (k1 != 0xFFFFFFFFu) SR_PUSH(pick(child, k1 & 3u));
                node = (k0 != 0xFFFFFFFFu) ? pick(child, k0 & 3u) : SR_POP();
            }
        }
        if (node == kSentinel) break;
        // leaf
        const uint32_t lv = ~(uint32_t)node;
        const uint32_t first = lv >> 3, cnt = lv & 7u;
        for (uint32_t i = 0; i < cnt; i++) {
            const float4 t0 = tris[(size_t)(first + i) * 3 + 0];
            const float4 t1 = tris[(size_t)(first + i) * 3 + 1];
            const float4 t2 = tris[(size_t)(first + i) * 3 + 2];
            float t, u, v;
            if (STATS) st.tris += 1;
            if (intersect_tri(o, d, mk3(t0.x, t0.y, t0.z), mk3(t0.w, t1.x, t1.y), mk3(t1.z, t1.w, t2.x), tmin, tmax, t, u, v)) {
                if (ANY) return true;
                const uint32_t gid = __float_as_uint(t2.y);
                if (t < best_t || (t == best_t && gid < hit.gid)) {
                    best_t = t;
                    hit.t = t; hit.u = u; hit.v = v;
                    hit.gid = gid; hit.slot = first + i;
                    cull = fmaf(fabsf(t), 1e-5f, t);
                }
            }
        }
        node = SR_POP();
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

// ---------------------------------------------------------------------------------------------
// Quad-cooperative traversal: FOUR lanes (one DPP quad) trace ONE ray.
//
// A per-lane traversal makes every lane gather its own 112 bytes per node step: 7 dwordx4 loads x 64
// different cache lines per wave instruction — the vector L1 / address path, not HBM and not the ALU,
// bounds it. Here lane q of a quad owns child q of the 4-wide node: the quad reads the
// 128-byte node as 4 x 32 contiguous bytes (two coalesced dwordx4 per lane, ONE line per quad), each
// lane runs one slab test, and the quad shares the four results through DPP quad_perm moves — no LDS,
// no extra latency. Everything that steers the walk (sort, stack, best hit) is computed redundantly
// and identically by the four lanes, so a quad behaves as one ray; a leaf's <= 4 triangles are tested
// one per lane. A wave therefore carries 16 rays, and a wave's ACTIVE rays are packed into batches of
// 16 first — lanes without a ray still help, which removes the idle-lane cost of sparse shadow rays.
// ---------------------------------------------------------------------------------------------
SRD uint32_t quad_bcast_u(uint32_t v, int lane_in_quad) {
    switch (lane_in_quad) {   // compile-time constant at every call site
        case 0: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, false);
        case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xF, 0xF, false);
        case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xF, 0xF, false);
        default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xF, 0xF, false);
    }
}
SRD uint32_t quad_xor1_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }  // [1,0,3,2]
SRD uint32_t quad_xor2_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); }  // [2,3,0,1]
SRD float quad_xor1_f(float v) { return __uint_as_float(quad_xor1_u(__float_as_uint(v))); }
SRD float quad_xor2_f(float v) { return __uint_as_float(quad_xor2_u(__float_as_uint(v))); }

SRD uint32_t pick4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t i) { return i == 0u ? a : (i == 1u ? b : (i == 2u ? c : d)); }

// `valid`, the ray and every output are uniform across the quad. qstack: this QUAD's LDS stack column
// (element k at qstack[k * stride]); all four lanes store the same value to the same address.
template <bool ANY, bool STATS>
SRD bool quad_traverse(const DevScene& sc, bool valid, f3 o, f3 d, float tmin, float tmax, TravHit& hit, int* qstack, int stride, TravStats& st) {
    const float4* __restrict__ nodes = sc.nodes_q;
    const float4* __restrict__ tris = sc.tris;
    const uint32_t q = threadIdx.x & 3u;
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool sx = (__float_as_uint(inv.x) >> 31) != 0u, sy = (__float_as_uint(inv.y) >> 31) != 0u, sz = (__float_as_uint(inv.z) >> 31) != 0u;
    const float t_lo = fminf(tmin, 0.0f) - fabsf(tmin);
    float cull = fmaf(fabsf(tmax), 1e-5f, tmax);
    float best_t = tmax;
    hit.t = -1.0f; hit.u = 0.0f; hit.v = 0.0f; hit.gid = 0xFFFFFFFFu; hit.slot = 0u;
    bool found = false;
    int* stack_base = qstack;
    int sp = 0;
    int node = valid ? 0 : kSentinel;
    while (node != kSentinel) {
        while (node >= 0 && node != kSentinel) {
            const float4* n = nodes + (size_t)node * 8 + q * 2u;
            const float4 a = n[0];   // lo.x lo.y lo.z hi.x
            const float4 b = n[1];   // hi.y hi.z ref  pad
            if (STATS && q == 0u) st.boxes += 4;
            const float tnx = ((sx ? a.w : a.x) - o.x) * inv.x, tfx = ((sx ? a.x : a.w) - o.x) * inv.x;
            const float tny = ((sy ? b.x : a.y) - o.y) * inv.y, tfy = ((sy ? a.y : b.x) - o.y) * inv.y;
            const float tnz = ((sz ? b.y : a.z) - o.z) * inv.z, tfz = ((sz ? a.z : b.y) - o.z) * inv.z;
            const float t0 = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, t_lo));
            float far = fminf(fminf(tfx, tfy), tfz);
            far = far * (1.0f + copysignf(5e-7f, far));
            const bool h = t0 <= fminf(far, cull);
            const uint32_t key = h ? ((__float_as_uint(fmaxf(t0, 0.0f)) & ~3u) | q) : 0xFFFFFFFFu;
            const uint32_t ref = __float_as_uint(b.z);
            uint32_t k0 = quad_bcast_u(key, 0), k1 = quad_bcast_u(key, 1), k2 = quad_bcast_u(key, 2), k3 = quad_bcast_u(key, 3);
            const uint32_t r0 = quad_bcast_u(ref, 0), r1 = quad_bcast_u(ref, 1), r2 = quad_bcast_u(ref, 2), r3 = quad_bcast_u(ref, 3);
            if (!ANY) { cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2); }
            else {   // order is irrelevant for an existence query: just move the hits to the front
                cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2);
            }
            if (k3 != 0xFFFFFFFFu) SR_PUSH((int)pick4(r0, r1, r2, r3, k3 & 3u));
            if (k2 != 0xFFFFFFFFu) SR_PUSH((int)pick4(r0, r1, r2, r3, k2 & 3u));
            if (k1 != 0xFFFFFFFFu) SR_PUSH((int)pick4(r0, r1, r2, r3, k1 & 3u));
            node = (k0 != 0xFFFFFFFFu) ? (int)pick4(r0, r1, r2, r3, k0 & 3u) : SR_POP();
        }
        if (node == kSentinel) break;
        // leaf: lane q tests triangle q
        const uint32_t lv = ~(uint32_t)node;
        const uint32_t first = lv >> 3, cnt = lv & 7u;
        float t = 0.0f, u = 0.0f, v = 0.0f;
        uint32_t gid = 0xFFFFFFFFu;
        bool th = false;
        if (q < cnt) {
            const float4 t0 = tris[(size_t)(first + q) * 3 + 0];
            const float4 t1 = tris[(size_t)(first + q) * 3 + 1];
            const float4 t2 = tris[(size_t)(first + q) * 3 + 2];
            if (STATS) st.tris += 1;
            th = intersect_tri(o, d, mk3(t0.x, t0.y, t0.z), mk3(t0.w, t1.x, t1.y), mk3(t1.z, t1.w, t2.x), tmin, tmax, t, u, v);
            gid = __float_as_uint(t2.y);
        }
        // reduce over the quad: smallest t, ties to the lowest global triangle index
        float bt = th ? t : __builtin_inff();
        uint32_t bg = th ? gid : 0xFFFFFFFFu;
        float bu = u, bv = v;
        uint32_t bslot = first + q;
        {
            const float ot = quad_xor1_f(bt); const uint32_t og = quad_xor1_u(bg);
            const float ou = quad_xor1_f(bu), ov = quad_xor1_f(bv); const uint32_t os = quad_xor1_u(bslot);
            const bool take = (ot < bt) || (ot == bt && og < bg);
            bt = take ? ot : bt; bg = take ? og : bg; bu = take ? ou : bu; bv = take ? ov : bv; bslot = take ? os : bslot;
        }
        {
            const float ot = quad_xor2_f(bt); const uint32_t og = quad_xor2_u(bg);
            const float ou = quad_xor2_f(bu), ov = quad_xor2_f(bv); const uint32_t os = quad_xor2_u(bslot);
            const bool take = (ot < bt) || (ot == bt && og < bg);
            bt = take ? ot : bt; bg = take ? og : bg; bu = take ? ou : bu; bv = take ? ov : bv; bslot = take ? os : bslot;
        }
        if (bg != 0xFFFFFFFFu) {
            if (ANY) { found = true; break; }
            if (bt < best_t || (bt == best_t && bg < hit.gid)) {
                best_t = bt;
                hit.t = bt; hit.u = bu; hit.v = bv; hit.gid = bg; hit.slot = bslot;
                cull = fmaf(fabsf(bt), 1e-5f, bt);
            }
        }
        node = SR_POP();
    }
    return ANY ? found : (hit.gid != 0xFFFFFFFFu);
}

// Payload of one closest-hit query in registers (rt_types.slang:9-16).
struct Payload {
    f3 emission;
    float dist;
    uint32_t albedo_packed, normal_packed, material_info, transmission_ior_packed;
};

// closest_hit.slang:12-91 for NULL-texture materials (sample_texture returns its fallback,
// rt_utils.slang:127-129), ray_miss.slang:10-13 for misses. Without textures the payload needs one
// vertex attribute (the normal) plus per-mesh constants: one 48-byte shade record, the instance's
// WorldToObject and a 32-byte DevMeshConst, instead of the reference's MeshInfo -> indices -> vertices
// pointer chase (4 dependent fetches).
SRD Payload shade_hit(const DevScene& sc, const TravHit& h) {
    Payload pl;
    pl.emission = splat(0.0f);
    pl.albedo_packed = 0; pl.normal_packed = 0; pl.material_info = 0; pl.transmission_ior_packed = 0;
    if (h.gid == 0xFFFFFFFFu) { pl.dist = -1.0f; return pl; }
    const float4 s0 = sc.shade[(size_t)h.slot * 3 + 0];
    const float4 s1 = sc.shade[(size_t)h.slot * 3 + 1];
    const float4 s2 = sc.shade[(size_t)h.slot * 3 + 2];
    const uint32_t inst = __float_as_uint(s2.y), mesh = __float_as_uint(s2.z);
    const f3 bary = mk3(1.0f - h.u - h.v, h.u, h.v);
    const f3 na = mk3(s0.x, s0.y, s0.z), nb = mk3(s0.w, s1.x, s1.y), nc = mk3(s1.z, s1.w, s2.x);
    const f3 normal = na * bary.x + nb * bary.y + nc * bary.z;
    const float* W = sc.instances[inst].w2o;
    const f3 world_normal = norm3(mk3((normal.x * W[0] + normal.y * W[3]) + normal.z * W[6],
                                      (normal.x * W[1] + normal.y * W[4]) + normal.z * W[7],
                                      (normal.x * W[2] + normal.y * W[5]) + normal.z * W[8]));
    const DevMeshConst mc = sc.mesh_const[mesh];
    pl.dist = h.t;
    pl.emission = mk3(mc.emission[0], mc.emission[1], mc.emission[2]);
    pl.albedo_packed = mc.albedo_packed;
    pl.normal_packed = pack_normal(world_normal);
    pl.material_info = mc.material_info;
    pl.transmission_ior_packed = mc.transmission_ior_packed;
    return pl;
}

}  // namespace srd
