// BVH traversal + ray/triangle intersection for gfx950 — the replacement for the Vulkan driver's
// TraceRay (call sites ray_gen_ris.slang:75,294,332,381 and
// ray_gen_final.slang:80,212,283,316,370).
//
// Data layout in HBM:
//   nodes : 4-wide BVH with quantised child boxes, 64 bytes per node = 4 x 16-byte gathers per step
//           (a step lasts as long as the slowest of its lanes' gathers — under load one of them nearly always misses L2 —
//           not as long as its tag lookups or its arithmetic: DESIGN.md §5, round-2 experiments):
//           [0]  origin.xyz | exponents ex | ey<<8 | ez<<16     plane = fmaf(q, 2^e, origin)
//           [16] LX LY LZ HX   one dword per plane set, byte c = child c
//           [32] HY HZ - -
//           [48] child[4]      >= 0: node index; < 0: leaf, ~child = (first_triangle << 3) | count
//           The builder rounds lower planes down / upper planes up and verifies them with the same
//           fmaf, so a decoded box always contains the exact one; unused children decode inverted.
//   tris  : 48 B per triangle = 3 x float4, in leaf order:
//           (v0.x, v0.y, v0.z, e1.x) (e1.y, e1.z, e2.x, e2.y) (e2.z, global triangle index, -, -)
//   shade : 48 B per triangle, same order, read once per committed closest hit:
//           (n0.x, n0.y, n0.z, n1.x) (n1.y, n1.z, n2.x, n2.y) (n2.z, instance index, mesh slot, -)
//           (object-space vertex normals: the only vertex attribute closest_hit needs without textures)
// One lane = one ray. The per-lane traversal stack lives in LDS, interleaved as stack[level][lane]
// so a wave's pushes/pops hit 64 consecutive banks (conflict-free ds_write_b32 / ds_read_b32).
#pragma once
#include "bvh_layout.h"
#include "rt_device.h"

namespace srd {

constexpr int kStackMax = 31;        // stack entries per lane the builders shape a tree for (= kMaxBinaryDepth); with the spare
                                     // level of the branch-free push that is 32 LDS levels (8 KB per wave). Launches size the
                                     // dynamic LDS stack to what the scene's tree actually needs.
constexpr int kMaxBinaryDepth = 31;  // depth bound of the binary tree the 4-wide tree is collapsed from
constexpr int kSentinel = 0x7fffffff;

struct DevInstance {   // 96 B: (float3x3)WorldToObject3x4 in w2o[0..8], (float3x3)ObjectToWorld3x4 in o2w[0..8], row-major
    float w2o[12];
    float o2w[12];         // only the normal-map branch reads it (closest_hit.slang:57-58)
};
struct DevMeshConst {  // 32 B: the per-mesh constants of the 32-byte RayPayload; complete for untextured meshes
    float emission[3];           // emissive_factor.rgb * strength      (closest_hit.slang:43-46)
    uint32_t albedo_packed;      // pack_unorm_4x8(base_color.rgb, 1)   (:76)
    uint32_t material_info;      // pack_half_2x16(roughness, metallic) (:79-89)
    uint32_t transmission_ior_packed;  // (:90)
    uint32_t textured;           // != 0: any of the four sampled texture slots is set -> DevMeshTex + shade_tex apply
    uint32_t _pad;
};
// Texture side of a material (closest_hit.slang:42-46,65-66,82-87): factors (the sample_texture fallbacks) and the
// resolved (image, sampler) slots. Sampler codes: bit0 = mag filter LINEAR, bits1-2 = address mode u, bits3-4 = v.
struct DevMeshTex {    // 72 B
    float base_color[4];
    float emissive_factor[3], emissive_strength;
    float roughness, metallic;
    uint32_t img_base, img_mr;
    uint32_t img_normal, img_emissive;
    uint32_t samplers;           // base | mr << 8 | normal << 16 | emissive << 24
    uint32_t alpha_mode;         // any_hit.slang:23 (0 in every material Material::new builds, material.rs:74)
    float alpha_cutoff;          // any_hit.slang:40
    uint32_t _pad;
};
struct DevTexture {    // 16 B: R8G8B8A8_UNORM, one mip level, R in the low byte (image/mod.rs:96-107)
    const uint32_t* texels;
    uint32_t w, h;
};

// One emissive triangle of the frame's light list (= one emissive_indirection entry), with everything the
// shaders derive from it per use hoisted to scene build time — same operations, same order, same bits
// (ray_gen_ris.slang:192-210,347-362; ray_gen_final.slang:331-351): world vertices via transform_point,
// area = 0.5*length(cross(e1,e2)), normal = normalize(cross(e1,e2)), emission.
struct DevLight {   // 64 B
    float wv0[3], area;
    float wv1[3], nx;
    float wv2[3], ny;
    float emission[3], nz;
};

// Ray counters: every wave of a pass adds its counts once. Atomics of all waves on ONE line serialise at the memory side
// (measured round 3: 4 adds per wave cost final_kernel 0.36 ms of 1.05), so the counts are spread over this many lines.
constexpr uint32_t kCounterSlots = 256, kCounterStride = 8;   // stride in u64

struct DevScene {
    const float4* nodes;
    const float4* tris;
    const float4* shade;
    const float4* shade_tex;       // 6 x float4 per slot: uv[3], normal uv[3], tangent[3], handedness; NULL without textures
    const DevMeshConst* mesh_const;
    const DevMeshTex* mesh_tex;    // [n_meshes]
    const DevTexture* textures;    // [n_images]
    const DevInstance* instances;
    const DevLight* lights;        // [num_lights]
    const uint32_t* slot_of_gid;   // global triangle index -> leaf-order slot (sr_shade_closest_hit only); two-level: per-mesh primitive -> slot tables
    // two-level structure (DevTlInstance below; null / 0 in the flattened form): `nodes` is then the top-level tree over instance
    // boxes, `tris` / `shade` / `shade_tex` hold every mesh's records once, in OBJECT space and leaf order
    const float4* blas_nodes;      // the meshes' trees, one after the other (child references already offset)
    const uint32_t* tl_inst;       // leaf position of the top-level tree -> instance index
    const struct DevTlInstance* tl_instances;
    // kCounterSlots copies of {[0]=closest queries [1]=any queries [2]=boxes [3]=tris [4]=primary hits reused [5]=visibility queries
    // reused, 2 x pad}, one 64-byte line each; a workgroup adds to the copy blockIdx.x selects and the host sums the copies
    unsigned long long* counters;
    uint32_t num_lights;
    uint32_t n_tris;
    uint32_t n_instances;
    uint32_t _pad;
};

struct TravHit {
    float t, u, v;
    uint32_t gid, slot;  // gid == 0xFFFFFFFF: miss; slot = position in tris/shade
    uint32_t inst;       // instance index (set by the two-level walk; the flattened form reads it from the shade record)
};

// One instance of the two-level structure — the counterpart of a VkAccelerationStructureInstanceKHR (resource_manager.rs:236-251:
// transform, custom index, mask 0xFF, cull disabled, BLAS reference) plus what the walk needs to stay conservative.
struct DevTlInstance {   // 128 B
    float w2o[12];        // row-major 3x4 inverse of o2w (computed in double): the object-space ray = w2o * world ray; steers box
                          // culling only — triangles are tested in WORLD space, so a hit has the bits of the flattened form
    float o2w[12];        // the instance transform: vertices go to world space with transform_point's operation order
    uint32_t blas_root;   // root node of the mesh's tree in blas_nodes
    uint32_t tri_offset;  // global triangle index of the instance's primitive 0 (instance-major order)
    float pad_a, pad_b;   // the mesh's boxes are widened by pad_a * |ray origin|_inf + pad_b (object units): bounds the rounding of
                          // the ray transform and of the world-space vertices against the object-space boxes
    uint32_t mesh_slot;
    uint32_t prim_base;   // of the mesh's primitive -> slot table in slot_of_gid
    uint32_t flags;       // bit 0: "baked" — the instance owns a WORLD-space copy of its mesh's tree and triangles (a transform that
                          // cannot be inverted, e.g. a zero scale: its triangles are still real): no ray transform, no vertex transform
    uint32_t _pad;
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

// Conservative slab tests: a box is only rejected if no triangle hit with t in (t_lo, t_hi] can lie
// inside it. Planes are taken in the ray's own order. The plane distance t = (origin + q*2^e - o) / d is evaluated as
// fma(q, A, B) with the per-node, per-axis constants A = 2^e * inv (exact: a power of two times inv) and
// B = (origin - o) * inv, i.e. two VALU instructions per plane (v_cvt_f32_ubyteN + v_fma) instead of four. The
// rounding of B (relative 2^-23 of a distance of at most the node extent when the ray starts near the node, covered by
// the relative cull slack otherwise) is far inside the guard band the builders leave around every quantised plane
// (at least 1/32 grid cell, bvh_build.cpp / bvh_gpu.hip). A zero direction component is replaced by +-2^-100 for the
// box test only, so the slab of an axis-parallel ray is (-huge, +huge) inside, same-signed outside and [0, huge] on a
// face — no 0*inf, no NaN — while the triangle test keeps the true direction. The far side is inflated (Ize 2013).
struct RaySetup {
    f3 inv;            // carries the direction signs (box_dir keeps the sign of a zero component)
};
SRD float box_dir(float d) { return fabsf(d) >= 7.888609e-31f ? d : copysignf(7.888609e-31f, d); }   // 2^-100
SRD RaySetup ray_setup(f3 o, f3 d) {
    RaySetup r;
    r.inv = mk3(1.0f / box_dir(d.x), 1.0f / box_dir(d.y), 1.0f / box_dir(d.z));
    return r;
}
// Byte `C` of a plane dword as a float: v_cvt_f32_ubyteC.
template <int C>
SRD float plane_q(uint32_t packed) { return (float)((packed >> (8 * C)) & 0xFFu); }
struct NodePlanes {
    uint32_t nx, ny, nz, fx, fy, fz;   // near / far plane dwords, already picked by the ray's signs
    float ax, ay, az;                  // 2^e * inv
    float bx, by, bz;                  // (node origin - ray origin) * inv
};
template <int C>
SRD bool child_hit(const NodePlanes& p, float t_lo, float t_hi, float& tnear) {
    const float t0 = fmaxf(fmaxf(fmaf(plane_q<C>(p.nx), p.ax, p.bx), fmaf(plane_q<C>(p.ny), p.ay, p.by)),
                           fmaxf(fmaf(plane_q<C>(p.nz), p.az, p.bz), t_lo));
    float far = fminf(fminf(fmaf(plane_q<C>(p.fx), p.ax, p.bx), fmaf(plane_q<C>(p.fy), p.ay, p.by)), fmaf(plane_q<C>(p.fz), p.az, p.bz));
    far = fmaf(fabsf(far), 5e-7f, far);   // inflate towards +inf whatever the sign (queries with tmin < 0 accept hits at negative t);
                                          // one v_fma with an |x| modifier. -inf gives NaN, which v_min drops: not culled, conservative
    tnear = t0;
    return t0 <= fminf(far, t_hi);
}

// LDS stack: element k of this lane's column lives at stack_base[k * stride]. The builder bounds the number of entries
// a traversal of the tree can need (BvhResult::max_stack <= kStackMax): no overflow path.
SRD int pick(int4 c, uint32_t i) {   // two levels of selects (v_cndmask), no branches
    const int lo = (i & 1u) ? c.y : c.x;
    const int hi = (i & 1u) ? c.w : c.z;
    return (i & 2u) ? hi : lo;
}

// ---------------------------------------------------------------------------------------------
// Traversal with work stealing inside the wave. Rays of one wave differ a lot in length: measured on the bench frame, the
// lanes that enter a query together are busy for only half of the wave's node steps, the rest have finished and wait
// for the longest ray. Here a lane that runs out of work takes the BOTTOM entry of a busy lane's stack (the stacks
// live in LDS, one column per lane, so any lane of the wave can read them) together with that lane's ray, and walks
// the subtree on its own stack. Results meet in LDS: per ray one 64-bit key (ordered t, triangle id) updated with an
// atomic minimum — exactly the tie rule of the sequential walk (smallest t, then lowest id), so the answer does not
// depend on who found it — and the payload (t, u, v, slot) written by whoever holds the minimum. The key also carries
// the best t to every lane working for the ray (culling bound) and, for existence queries, the "found" flag.
// Box culling is conservative, so splitting a walk cannot change its result.
// LDS: kWsRows rows of `stride` ints in front of the stack levels: keys (2 rows), lane exchange (1), payload (4; the two-level
// form adds the instance: 5).
//
// TL = the two-level form of the structure (optional, for instance-heavy scenes; the reference's TLAS over BLASes, tlas.rs:155-191,
// resource_manager.rs:236-251). `nodes` is then the top-level tree over padded world-space instance boxes. Entering an instance
// transforms the ray into the mesh's object space for the BOX tests only (the parameter t is the same in both spaces, the
// direction is not re-normalised). A leaf's triangles are brought to world space on the fly — the same transform_point
// operations that flatten them in the one-level form — and tested against the world-space ray, so a hit carries exactly the bits
// of the one-level form; the tie rule (smallest t, then lowest global triangle index = instance-major) is unchanged. Culling
// stays conservative through the per-instance padding (DevTlInstance::pad_*). A lane's stack holds top-level entries below
// index `lsp` and the current instance's entries from there; the instance is left when the stack falls back to `lsp`; a
// postponed leaf remembers its own instance. Work stealing hands over the bottom entry as before: a top-level one with the
// world-space ray, or — once the donor has none left — one of the instance with its object-space ray.
// ---------------------------------------------------------------------------------------------
constexpr int kWsRows = 7, kWsRowsTl = 8;
constexpr int kEarlyTriWalking = 16, kEarlyTriBlocked = 16;   // round 3 sweep: 8 / 12 / 16 / 24 / 32 walking x 8 / 16 / 20 / 24 / 32 blocked
SRD uint32_t ord_f32(float f) { const uint32_t b = __float_as_uint(f + 0.0f); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
SRD float unord_f32(uint32_t k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }
SRD uint32_t lanes_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
struct NodePlanesTl {
    uint32_t nx, ny, nz, fx, fy, fz;
    float ax, ay, az;
    float bnx, bny, bnz, bfx, bfy, bfz;   // near / far offsets: (origin - o) * inv -/+ pad * |inv|
};
template <int C>
SRD bool child_hit_tl(const NodePlanesTl& p, float t_lo, float t_hi, float& tnear) {
    const float t0 = fmaxf(fmaxf(fmaf(plane_q<C>(p.nx), p.ax, p.bnx), fmaf(plane_q<C>(p.ny), p.ay, p.bny)),
                           fmaxf(fmaf(plane_q<C>(p.nz), p.az, p.bnz), t_lo));
    float far = fminf(fminf(fmaf(plane_q<C>(p.fx), p.ax, p.bfx), fmaf(plane_q<C>(p.fy), p.ay, p.bfy)), fmaf(plane_q<C>(p.fz), p.az, p.bfz));
    far = fmaf(fabsf(far), 5e-7f, far);
    tnear = t0;
    return t0 <= fminf(far, t_hi);
}

// Every lane of the wave that is active at the call site takes part; `want` = this lane has a ray of its own.
template <bool ANY, bool STATS, bool TL = false>
SRD bool traverse_ws(const DevScene& sc, bool want, f3 o, f3 d, float tmin, float tmax, TravHit& hit, int* lds_col, int stride, TravStats& st) {
    const float4* __restrict__ nodes = sc.nodes;
    const float4* __restrict__ tris = sc.tris;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    int* const blk = lds_col - tid;
    unsigned long long* const keys = reinterpret_cast<unsigned long long*>(blk);   // rows 0-1: one u64 per thread
    int* const xch = blk + 2 * stride + (tid & ~63u);                               // row 2: this wave's 64 exchange slots
    int* const pay = blk + 3 * stride;                                              // rows 3-6: t, u, v, slot of the key holder (TL: row 7 = instance)
    int* const stack_base = lds_col + (TL ? kWsRowsTl : kWsRows) * stride;
    keys[tid] = ~0ull;
    uint32_t root = tid;                 // thread whose ray this lane is working for
    RaySetup rs = ray_setup(o, d);       // of the space being walked
    float t_lo = fminf(tmin, 0.0f) - fabsf(tmin);
    float cull = fmaf(fabsf(tmax), 1e-5f, tmax);
    float best_t = tmax;
    uint32_t best_gid = 0xFFFFFFFFu;
    int sp = 0, sb = 0, leaf = 0;
    int node = want ? 0 : kSentinel;
    // two-level state (compiled out of the one-level form)
    bool in_blas = false;
    int lsp = 0;                         // first stack index of the current instance's entries
    uint32_t inst = 0u, leaf_inst = 0u;  // instance being walked; instance of the postponed leaf (the lane may have moved on meanwhile)
    f3 ro = o;                           // origin in the space being walked
    f3 winv = rs.inv;                    // 1/d of the world-space ray, to come back to without dividing again
    f3 pad = splat(0.0f);                // box padding of the current instance times |1/d'| per axis
    auto leave = [&]() { in_blas = false; ro = o; rs.inv = winv; pad = splat(0.0f); };
    // Next entry of this lane's stack. Two-level: an instance is left when its entries are used up.
    auto pop = [&]() -> int {
        if (TL) { if (in_blas && sp == lsp) leave(); }
        return sp == sb ? kSentinel : stack_base[(--sp) * stride];
    };
    for (;;) {
        // ---- phase top: all participating lanes meet here ----
        if (node != kSentinel) {
            const unsigned long long k = keys[root];
            if (ANY) { if (k == 0ull) { node = kSentinel; sp = sb; if (TL) { in_blas = false; } } }        // somebody found an occluder
            else if (k != ~0ull) { const float ts = unord_f32((uint32_t)(k >> 32)); cull = fminf(cull, fmaf(fabsf(ts), 1e-5f, ts)); }
        }
        if (__builtin_amdgcn_ballot_w64(node != kSentinel || (TL && leaf != 0)) == 0ull) break;
        // ---- node phase: wave-uniform loop, so that lanes without work stay in step and can be handed some ----
        for (;;) {
            // `inner`: the lane has a node-phase step to do — an inner node, or (two-level) a top-level leaf to enter
            const bool inner = node != kSentinel && (node >= 0 || (TL && !in_blas));
            // On to the triangles when every walking lane holds a (postponed) leaf — or already when most of the wave is BLOCKED (holds a
            // postponed leaf and has arrived at a second one) and only a few lanes still walk: the blocked lanes would wait for them.
            // Progress: an early exit needs a blocked lane, and the triangle phase consumes its leaf.
            {
                const unsigned long long walking = __builtin_amdgcn_ballot_w64(inner && leaf == 0);
                if (walking == 0ull) break;
                if ((int)__popcll(walking) <= kEarlyTriWalking &&
                    (int)__popcll(__builtin_amdgcn_ballot_w64(node != kSentinel && node < 0 && leaf != 0 && (!TL || in_blas))) >= kEarlyTriBlocked) break;
            }
            const bool idle = node == kSentinel && leaf == 0;
            const unsigned long long idle_mask = __builtin_amdgcn_ballot_w64(idle);
            if (idle_mask != 0ull) {
                // donors: every lane with a node or leaf in hand and a spare stack entry — blocked lanes too (they wait, their subtrees need not)
                const bool can_give = node != kSentinel && sp > sb;
                const unsigned long long donor_mask = __builtin_amdgcn_ballot_w64(can_give);
                if (donor_mask != 0ull) {
                    const uint32_t n = min((uint32_t)__popcll(idle_mask), (uint32_t)__popcll(donor_mask));
                    const uint32_t drank = lanes_below(donor_mask), irank = lanes_below(idle_mask);
                    const bool gives = can_give && drank < n, takes = idle && irank < n;
                    if (gives) xch[drank] = (int)lane;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    const int dl = takes ? xch[irank] : (int)lane;
                    const int d_sb = __shfl(sb, dl);
                    const uint32_t d_root = (uint32_t)__shfl((int)root, dl);
                    const float ox = __shfl(o.x, dl), oy = __shfl(o.y, dl), oz = __shfl(o.z, dl);
                    const float dx = __shfl(d.x, dl), dy = __shfl(d.y, dl), dz = __shfl(d.z, dl);
                    const float d_tmin = __shfl(tmin, dl), d_tmax = __shfl(tmax, dl), d_cull = __shfl(cull, dl);
                    // the donor's 1/d comes along as well: three more shuffles instead of three correctly rounded divisions (and
                    // the zero-direction fix-ups) that the WHOLE wave would issue whenever one lane takes work — the exchange runs
                    // in 28 % / 46 % of the node-loop iterations of the RIS / final pass
                    const float ix = __shfl(rs.inv.x, dl), iy = __shfl(rs.inv.y, dl), iz = __shfl(rs.inv.z, dl);
                    // two-level: is the donor's bottom entry one of its current instance (it has no top-level entries left)?
                    // Then the instance and the object-space ray come along; a top-level entry goes with the world-space ray.
                    bool d_blas = false;
                    uint32_t d_inst = 0u;
                    f3 d_ro = splat(0.0f), d_pad = splat(0.0f), d_winv = splat(0.0f);
                    if (TL) {
                        d_blas = __shfl((int)(in_blas && sb == lsp), dl) != 0;
                        d_inst = (uint32_t)__shfl((int)inst, dl);
                        d_ro = mk3(__shfl(ro.x, dl), __shfl(ro.y, dl), __shfl(ro.z, dl));
                        d_pad = mk3(__shfl(pad.x, dl), __shfl(pad.y, dl), __shfl(pad.z, dl));
                        d_winv = mk3(__shfl(winv.x, dl), __shfl(winv.y, dl), __shfl(winv.z, dl));
                    }
                    if (takes) {
                        node = stack_base[d_sb * stride + (dl - (int)lane)];
                        o = mk3(ox, oy, oz); d = mk3(dx, dy, dz); tmin = d_tmin; tmax = d_tmax; cull = d_cull; root = d_root;
                        rs.inv = mk3(ix, iy, iz);
                        t_lo = fminf(tmin, 0.0f) - fabsf(tmin);
                        best_t = tmax; best_gid = 0xFFFFFFFFu;
                        sp = 0; sb = 0;
                        if (TL) {
                            winv = d_winv; lsp = 0;
                            in_blas = d_blas; inst = d_inst; leaf_inst = d_inst;
                            if (d_blas) { ro = d_ro; pad = d_pad; }
                            else { ro = o; rs.inv = d_winv; pad = splat(0.0f); }
                        }
                        if (node < 0 && (!TL || in_blas)) { leaf = node; node = kSentinel; }   // the stolen entry is a leaf: it waits for the triangle phase
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    if (gives) { if (TL) { if (in_blas && sb == lsp) lsp++; } sb++; }
                }
            }
            if (TL && node != kSentinel && node < 0 && !in_blas) {
                // top-level leaf: enter its first instance, the others wait below as leaves of one
                const uint32_t lv = ~(uint32_t)node;
                const uint32_t first = lv >> 3, cnt = lv & 7u;
                if (cnt == 0u) node = pop();
                else {
                    for (uint32_t k = 1; k < cnt; k++) { stack_base[sp * stride] = (int)~(((first + k) << 3) | 1u); sp++; }
                    inst = sc.tl_inst[first];
                    const float4* q = reinterpret_cast<const float4*>(sc.tl_instances + inst);
                    const float4 r0 = q[0], r1 = q[1], r2 = q[2], m6 = q[6];
                    if ((__float_as_uint(q[7].z) & 1u) == 0u) {
                        ro = mk3(((r0.x * o.x + r0.y * o.y) + r0.z * o.z) + r0.w, ((r1.x * o.x + r1.y * o.y) + r1.z * o.z) + r1.w,
                                 ((r2.x * o.x + r2.y * o.y) + r2.z * o.z) + r2.w);
                        const f3 rd = mk3((r0.x * d.x + r0.y * d.y) + r0.z * d.z, (r1.x * d.x + r1.y * d.y) + r1.z * d.z, (r2.x * d.x + r2.y * d.y) + r2.z * d.z);
                        rs = ray_setup(ro, rd);
                        const float pd = fmaf(m6.z, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)), m6.w);
                        pad = mk3(pd * fabsf(rs.inv.x), pd * fabsf(rs.inv.y), pd * fabsf(rs.inv.z));
                    }   // a baked instance's tree is in world space: the ray stays as it is (ro == o, rs.inv == winv, pad == 0 here)
                    lsp = sp;
                    in_blas = true;
                    node = (int)__float_as_uint(m6.x);
                }
            } else if (node >= 0 && node != kSentinel) {
                const float4* n = ((TL && in_blas) ? sc.blas_nodes : nodes) + (size_t)node * 4;
                const float4 h0 = n[0], q1 = n[1], q2 = n[2], qc = n[3];
                const int4 child = make_int4(__float_as_int(qc.x), __float_as_int(qc.y), __float_as_int(qc.z), __float_as_int(qc.w));
                if (STATS) st.boxes += 4;
                const uint32_t ex = __float_as_uint(h0.w);
                const uint32_t LX = __float_as_uint(q1.x), LY = __float_as_uint(q1.y), LZ = __float_as_uint(q1.z);
                const uint32_t HX = __float_as_uint(q1.w), HY = __float_as_uint(q2.x), HZ = __float_as_uint(q2.y);
                // The ray's direction signs (the NEAR plane of a slab is the upper one for a negative direction) are re-derived
                // from inv here, three v_cmp: kept as lane masks across the loop, which the stealing path updates divergently,
                // they cost the compiler ~18 s_and / s_andn2 / s_or per step (-35 SALU in the kernel, frame -1.5 %).
                const bool sgx = __float_as_int(rs.inv.x) < 0, sgy = __float_as_int(rs.inv.y) < 0, sgz = __float_as_int(rs.inv.z) < 0;
                float n0, n1, n2, n3;
                bool b0, b1, b2, b3;
                if (TL) {
                    NodePlanesTl p;
                    p.nx = sgx ? HX : LX; p.fx = sgx ? LX : HX;
                    p.ny = sgy ? HY : LY; p.fy = sgy ? LY : HY;
                    p.nz = sgz ? HZ : LZ; p.fz = sgz ? LZ : HZ;
                    p.ax = __uint_as_float((ex & 0xFFu) << 23) * rs.inv.x; p.ay = __uint_as_float(((ex >> 8) & 0xFFu) << 23) * rs.inv.y;
                    p.az = __uint_as_float(((ex >> 16) & 0xFFu) << 23) * rs.inv.z;
                    const float bx = (h0.x - ro.x) * rs.inv.x, by = (h0.y - ro.y) * rs.inv.y, bz = (h0.z - ro.z) * rs.inv.z;
                    p.bnx = bx - pad.x; p.bfx = bx + pad.x; p.bny = by - pad.y; p.bfy = by + pad.y; p.bnz = bz - pad.z; p.bfz = bz + pad.z;
                    // An unused child decodes to an inverted box, which nothing can hit — unless the padding exceeds the node's own
                    // extent (an ill-conditioned instance): its reference (-1) is checked as well, or such children would be pushed and
                    // the stack outgrow what the builders sized it for.
                    b0 = child_hit_tl<0>(p, t_lo, cull, n0) && child.x != -1;
                    b1 = child_hit_tl<1>(p, t_lo, cull, n1) && child.y != -1;
                    b2 = child_hit_tl<2>(p, t_lo, cull, n2) && child.z != -1;
                    b3 = child_hit_tl<3>(p, t_lo, cull, n3) && child.w != -1;
                } else {
                    NodePlanes p;
                    p.nx = sgx ? HX : LX; p.fx = sgx ? LX : HX;
                    p.ny = sgy ? HY : LY; p.fy = sgy ? LY : HY;
                    p.nz = sgz ? HZ : LZ; p.fz = sgz ? LZ : HZ;
                    p.ax = __uint_as_float((ex & 0xFFu) << 23) * rs.inv.x; p.ay = __uint_as_float(((ex >> 8) & 0xFFu) << 23) * rs.inv.y;
                    p.az = __uint_as_float(((ex >> 16) & 0xFFu) << 23) * rs.inv.z;
                    p.bx = (h0.x - o.x) * rs.inv.x; p.by = (h0.y - o.y) * rs.inv.y; p.bz = (h0.z - o.z) * rs.inv.z;
                    b0 = child_hit<0>(p, t_lo, cull, n0);
                    b1 = child_hit<1>(p, t_lo, cull, n1);
                    b2 = child_hit<2>(p, t_lo, cull, n2);
                    b3 = child_hit<3>(p, t_lo, cull, n3);
                }
                uint32_t k0, k1, k2, k3;
                if (ANY) {
                    // Existence queries visit the hit child whose entry point is FARTHEST first (closest-hit queries need the
                    // nearest first, for culling). No order changes their answer; this one leaves the near, dense subtrees on
                    // the stack, where idle lanes can take them over, and lets the lane itself finish the far, mostly empty
                    // ones — measured on the bench frame: slot order 1.86 ms, nearest entry first 1.98, nearest exit first
                    // 1.99, longest way through the box first 1.89, farthest exit first 1.81, farthest entry first 1.81.
                    // (the order is a heuristic: the bits of a non-negative float, inverted, order it descending; an entry
                    // point behind the origin gets some rank of its own. Sign bit cleared: every key stays below the "missed" one.)
                    k0 = b0 ? ((~__float_as_uint(n0) & 0x7FFFFFFCu) | 0u) : 0xFFFFFFFFu;
                    k1 = b1 ? ((~__float_as_uint(n1) & 0x7FFFFFFCu) | 1u) : 0xFFFFFFFFu;
                    k2 = b2 ? ((~__float_as_uint(n2) & 0x7FFFFFFCu) | 2u) : 0xFFFFFFFFu;
                    k3 = b3 ? ((~__float_as_uint(n3) & 0x7FFFFFFCu) | 3u) : 0xFFFFFFFFu;
                } else {
                    k0 = b0 ? ((__float_as_uint(fmaxf(n0, 0.0f)) & ~3u) | 0u) : 0xFFFFFFFFu;
                    k1 = b1 ? ((__float_as_uint(fmaxf(n1, 0.0f)) & ~3u) | 1u) : 0xFFFFFFFFu;
                    k2 = b2 ? ((__float_as_uint(fmaxf(n2, 0.0f)) & ~3u) | 2u) : 0xFFFFFFFFu;
                    k3 = b3 ? ((__float_as_uint(fmaxf(n3, 0.0f)) & ~3u) | 3u) : 0xFFFFFFFFu;
                }
                const uint32_t kmin = min(min(k0, k1), min(k2, k3));
                const uint32_t slot = kmin & 3u;
                stack_base[sp * stride] = child.x; sp += (b0 && k0 != kmin) ? 1 : 0;
                stack_base[sp * stride] = child.y; sp += (b1 && k1 != kmin) ? 1 : 0;
                stack_base[sp * stride] = child.z; sp += (b2 && k2 != kmin) ? 1 : 0;
                stack_base[sp * stride] = child.w; sp += (b3 && k3 != kmin) ? 1 : 0;
                node = (kmin != 0xFFFFFFFFu) ? pick(child, slot) : pop();
                if (node < 0 && node != kSentinel && leaf == 0 && (!TL || in_blas)) { leaf = node; leaf_inst = inst; node = pop(); }
            }
        }
        // ---- triangle phase: two triangles of the lane's leaf per round trip (the builders make leaves of <= 2; an iteration is
        // one dependent gather, ~1500 cycles, against ~60 VALU instructions per test) ----
        while (leaf != 0) {
            const uint32_t lv = ~(uint32_t)leaf;
            const uint32_t slot = lv >> 3, cnt = lv & 7u;
            const bool two = cnt >= 2u;
            leaf = cnt > 2u ? (int)~(((slot + 2u) << 3) | (cnt - 2u)) : 0;
            const uint32_t cur_inst = leaf_inst;          // instance of THESE triangles (the chaining below may move on to another leaf)
            if (leaf == 0 && node < 0 && (!TL || in_blas)) { leaf = node; leaf_inst = inst; node = pop(); }
            if (cnt == 0u) continue;
            const uint32_t slot_b = two ? slot + 1u : slot;   // a lone triangle is read twice (same lines), its second result unused
            const float4 t0 = tris[(size_t)slot * 3 + 0], t1 = tris[(size_t)slot * 3 + 1], t2 = tris[(size_t)slot * 3 + 2];
            const float4 s0 = tris[(size_t)slot_b * 3 + 0], s1 = tris[(size_t)slot_b * 3 + 1], s2 = tris[(size_t)slot_b * 3 + 2];
            float ta, ua, va, tb, ub, vb;
            if (STATS) st.tris += two ? 2u : 1u;
            bool hit_a, hit_b;
            uint32_t gid_a, gid_b;
            if (TL) {
                // object-space v0, v1, v2 -> world space with transform_point (rt_utils.slang:278-281), exactly as the one-level form
                // flattens a triangle; the test itself runs on the world-space ray
                const float4* q = reinterpret_cast<const float4*>(sc.tl_instances + cur_inst);
                const float4 m0 = q[3], m1 = q[4], m2 = q[5];
                const bool baked = (__float_as_uint(q[7].z) & 1u) != 0u;                     // its records already hold world-space vertices
                auto to_world = [&](f3 p) {
                    return baked ? p : mk3(((m0.x * p.x + m0.y * p.y) + m0.z * p.z) + m0.w * 1.0f, ((m1.x * p.x + m1.y * p.y) + m1.z * p.z) + m1.w * 1.0f,
                                           ((m2.x * p.x + m2.y * p.y) + m2.z * p.z) + m2.w * 1.0f);
                };
                const f3 wa = to_world(mk3(t0.x, t0.y, t0.z)), wb = to_world(mk3(t0.w, t1.x, t1.y)), wc = to_world(mk3(t1.z, t1.w, t2.x));
                hit_a = intersect_tri(o, d, wa, wb - wa, wc - wa, tmin, tmax, ta, ua, va);
                const f3 xa = to_world(mk3(s0.x, s0.y, s0.z)), xb = to_world(mk3(s0.w, s1.x, s1.y)), xc = to_world(mk3(s1.z, s1.w, s2.x));
                hit_b = intersect_tri(o, d, xa, xb - xa, xc - xa, tmin, tmax, tb, ub, vb) && two;
                const uint32_t off = sc.tl_instances[cur_inst].tri_offset;
                gid_a = off + __float_as_uint(t2.y); gid_b = off + __float_as_uint(s2.y);
            } else {
                hit_a = intersect_tri(o, d, mk3(t0.x, t0.y, t0.z), mk3(t0.w, t1.x, t1.y), mk3(t1.z, t1.w, t2.x), tmin, tmax, ta, ua, va);
                hit_b = intersect_tri(o, d, mk3(s0.x, s0.y, s0.z), mk3(s0.w, s1.x, s1.y), mk3(s1.z, s1.w, s2.x), tmin, tmax, tb, ub, vb) && two;
                gid_a = __float_as_uint(t2.y); gid_b = __float_as_uint(s2.y);
            }
            if (hit_a || hit_b) {
                if (ANY) { keys[root] = 0ull; node = kSentinel; sp = sb; leaf = 0; if (TL) { in_blas = false; } break; }
                // the better of the two by the tie rule (smallest t, then lowest id), then one update as before
                const bool take_b = hit_b && (!hit_a || tb < ta || (tb == ta && gid_b < gid_a));
                const float t = take_b ? tb : ta, u = take_b ? ub : ua, v = take_b ? vb : va;
                const uint32_t gid = take_b ? gid_b : gid_a, hit_slot = take_b ? slot_b : slot;
                if (t < best_t || (t == best_t && gid < best_gid)) {
                    best_t = t; best_gid = gid;
                    cull = fminf(cull, fmaf(fabsf(t), 1e-5f, t));
                    const unsigned long long key = ((unsigned long long)ord_f32(t) << 32) | gid;
                    atomicMin(&keys[root], key);
                    if (keys[root] == key) {   // this lane holds the minimum: its payload stands
                        pay[0 * stride + root] = __float_as_int(t); pay[1 * stride + root] = __float_as_int(u);
                        pay[2 * stride + root] = __float_as_int(v); pay[3 * stride + root] = (int)hit_slot;
                        if (TL) pay[4 * stride + root] = (int)cur_inst;
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const unsigned long long k = keys[tid];
    hit.t = -1.0f; hit.u = 0.0f; hit.v = 0.0f; hit.gid = 0xFFFFFFFFu; hit.slot = 0u; hit.inst = 0u;
    if (ANY) return k == 0ull;
    if (k != ~0ull) {
        hit.gid = (uint32_t)k;
        hit.t = __int_as_float(pay[0 * stride + tid]); hit.u = __int_as_float(pay[1 * stride + tid]);
        hit.v = __int_as_float(pay[2 * stride + tid]); hit.slot = (uint32_t)pay[3 * stride + tid];
        if (TL) hit.inst = (uint32_t)pay[4 * stride + tid];
    }
    return k != ~0ull;
}

// Two-level form: leaf slot and instance of a global triangle index (test hooks only: sr_shade_closest_hit, sr_any_hit_ignores).
SRD void tl_locate(const DevScene& sc, uint32_t gid, uint32_t& slot, uint32_t& inst) {
    uint32_t lo = 0u, hi = sc.n_instances;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (sc.tl_instances[mid].tri_offset <= gid) lo = mid; else hi = mid; }
    inst = lo;
    slot = sc.slot_of_gid[sc.tl_instances[lo].prim_base + (gid - sc.tl_instances[lo].tri_offset)];
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
// `tex.SampleLevel(smp, uv, 0)` (rt_utils.slang:121-133) on a single-mip RGBA8_UNORM image: the magnification
// filter at LOD 0. Same exact fp32 definition as the oracle's (oracle/orc_texture.h): no texture
// unit, no fma.
SRD float wrap_coord(float s, uint32_t mode) {
    if (!(fabsf(s) < 3.0e38f)) s = 0.0f;
    if (mode == 0u) return s - floorf(s);
    if (mode == 1u) return s - 2.0f * floorf(s * 0.5f);
    return fminf(fmaxf(s, -1.0f), 2.0f);
}
SRD uint32_t wrap_index(int i, int n, uint32_t mode) {
    if (mode == 0u) { int m = i % n; return (uint32_t)(m < 0 ? m + n : m); }
    if (mode == 1u) {
        int m = i % (2 * n);
        if (m < 0) m += 2 * n;
        return (uint32_t)(m < n ? m : 2 * n - 1 - m);
    }
    return (uint32_t)(i < 0 ? 0 : (i > n - 1 ? n - 1 : i));
}
SRD float4 texel_unorm(uint32_t p) {
    return make_float4((float)(p & 0xFFu) / 255.0f, (float)((p >> 8) & 0xFFu) / 255.0f, (float)((p >> 16) & 0xFFu) / 255.0f,
                       (float)(p >> 24) / 255.0f);
}
SRD float4 sample_texture(const DevScene& sc, uint32_t image_slot, uint32_t sampler_code, float s, float t, float4 fallback) {
    if (image_slot == 0xFFFFFFFFu) return fallback;
    const DevTexture tx = sc.textures[image_slot];
    const uint32_t mode_u = (sampler_code >> 1) & 3u, mode_v = (sampler_code >> 3) & 3u;
    const int W = (int)tx.w, H = (int)tx.h;
    float u = wrap_coord(s, mode_u) * (float)W;
    float v = wrap_coord(t, mode_v) * (float)H;
    if ((sampler_code & 1u) == 0u) {
        const uint32_t i = wrap_index((int)floorf(u), W, mode_u), j = wrap_index((int)floorf(v), H, mode_v);
        return texel_unorm(tx.texels[(size_t)j * tx.w + i]);
    }
    u = u - 0.5f; v = v - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float a = u - fu, b = v - fv;
    const uint32_t i0 = wrap_index((int)fu, W, mode_u), i1 = wrap_index((int)fu + 1, W, mode_u);
    const uint32_t j0 = wrap_index((int)fv, H, mode_v), j1 = wrap_index((int)fv + 1, H, mode_v);
    const float4 t00 = texel_unorm(tx.texels[(size_t)j0 * tx.w + i0]), t10 = texel_unorm(tx.texels[(size_t)j0 * tx.w + i1]);
    const float4 t01 = texel_unorm(tx.texels[(size_t)j1 * tx.w + i0]), t11 = texel_unorm(tx.texels[(size_t)j1 * tx.w + i1]);
    const float na = 1.0f - a, nb = 1.0f - b;
    return make_float4((t00.x * na + t10.x * a) * nb + (t01.x * na + t11.x * a) * b, (t00.y * na + t10.y * a) * nb + (t01.y * na + t11.y * a) * b,
                       (t00.z * na + t10.z * a) * nb + (t01.z * na + t11.z * a) * b, (t00.w * na + t10.w * a) * nb + (t01.w * na + t11.w * a) * b);
}

// Textured half of closest_hit (closest_hit.slang:34-46,56-72,82-87): replaces emission, albedo, normal (if normal-
// mapped) and material_info. Compiled only into the kernel variants launched for scenes that own textured
// materials (shade_hit<true>), so untextured scenes keep their register budget.
SRD void shade_textured(const DevScene& sc, uint32_t slot, uint32_t inst, uint32_t mesh, f3 bary, f3 world_normal, Payload& pl) {
    const float4* q = sc.shade_tex + (size_t)slot * 6;
    const float4 q0 = q[0], q1 = q[1], q2 = q[2];
    const DevMeshTex mt = sc.mesh_tex[mesh];
    const float uv_s = (q0.x * bary.x + q0.z * bary.y) + q1.x * bary.z;
    const float uv_t = (q0.y * bary.x + q0.w * bary.y) + q1.y * bary.z;
    const float4 base_color = sample_texture(sc, mt.img_base, mt.samplers & 0xFFu, uv_s, uv_t,
                                             make_float4(mt.base_color[0], mt.base_color[1], mt.base_color[2], mt.base_color[3]));
    const float4 emissive = sample_texture(sc, mt.img_emissive, mt.samplers >> 24, uv_s, uv_t,
                                           make_float4(mt.emissive_factor[0], mt.emissive_factor[1], mt.emissive_factor[2], 1.0f));
    pl.emission = mk3(emissive.x, emissive.y, emissive.z) * mt.emissive_strength;
    pl.albedo_packed = pack_unorm_4x8(base_color.x, base_color.y, base_color.z, 1.0f);
    if (mt.img_normal != 0xFFFFFFFFu) {
        const float4 q3 = q[3], q4 = q[4], q5 = q[5];
        const f3 ta = mk3(q3.x, q3.y, q3.z), tb = mk3(q4.x, q4.y, q4.z), tc = mk3(q4.w, q5.x, q5.y);
        const f3 tangent_dir = ta * bary.x + tb * bary.y + tc * bary.z;
        if (len3(tangent_dir) > 0.001f) {
            const float handedness = q3.w;
            const float nuv_s = (q1.z * bary.x + q2.x * bary.y) + q2.z * bary.z;
            const float nuv_t = (q1.w * bary.x + q2.y * bary.y) + q2.w * bary.z;
            const float* M = sc.instances[inst].o2w;
            f3 wt = norm3(mk3((M[0] * tangent_dir.x + M[1] * tangent_dir.y) + M[2] * tangent_dir.z,
                              (M[3] * tangent_dir.x + M[4] * tangent_dir.y) + M[5] * tangent_dir.z,
                              (M[6] * tangent_dir.x + M[7] * tangent_dir.y) + M[8] * tangent_dir.z));
            wt = norm3(wt - world_normal * dot3(wt, world_normal));
            const f3 wb = cross3(world_normal, wt) * handedness;
            const float4 raw = sample_texture(sc, mt.img_normal, (mt.samplers >> 16) & 0xFFu, nuv_s, nuv_t, make_float4(0.5f, 0.5f, 1.0f, 1.0f));
            f3 sn = mk3(raw.x * 2.0f - 1.0f, raw.y * 2.0f - 1.0f, raw.z * 2.0f - 1.0f);
            sn.z = sqrtf(fminf(fmaxf(1.0f - (sn.x * sn.x + sn.y * sn.y), 0.0f), 1.0f));
            sn = norm3(sn);
            pl.normal_packed = pack_normal(norm3(mk3((sn.x * wt.x + sn.y * wb.x) + sn.z * world_normal.x,
                                                     (sn.x * wt.y + sn.y * wb.y) + sn.z * world_normal.y,
                                                     (sn.x * wt.z + sn.y * wb.z) + sn.z * world_normal.z)));
        }
    }
    float roughness = mt.roughness, metallic = mt.metallic;
    if (mt.img_mr != 0xFFFFFFFFu) {
        const float4 mr = sample_texture(sc, mt.img_mr, (mt.samplers >> 8) & 0xFFu, uv_s, uv_t, make_float4(1.0f, 1.0f, 1.0f, 1.0f));
        roughness = roughness * mr.y;
        metallic = metallic * mr.z;
    }
    pl.material_info = pack_half_2x16(roughness, metallic);
}

// any_hit.slang:11-43: true where the shader calls IgnoreHit(). Dead in the reference (OPAQUE geometry, alpha_mode forced
// 0), so the traversal does not call it; sr_any_hit_ignores applies it to hit records (K5 parity hook).
SRD bool any_hit_ignores(const DevScene& sc, uint32_t slot, float u, float v) {
    const float4 s2 = sc.shade[(size_t)slot * 3 + 2];
    const uint32_t mesh = __float_as_uint(s2.z);
    const DevMeshTex mt = sc.mesh_tex[mesh];
    if (mt.alpha_mode == 0u) return false;                                               // :23-25
    float4 base_color = make_float4(mt.base_color[0], mt.base_color[1], mt.base_color[2], mt.base_color[3]);
    if (mt.img_base != 0xFFFFFFFFu) {                                                    // sample_texture's NULL case is its fallback
        const f3 bary = mk3(1.0f - u - v, u, v);                                         // :27-29
        const float4* q = sc.shade_tex + (size_t)slot * 6;
        const float4 q0 = q[0], q1 = q[1];
        const float uv_s = (q0.x * bary.x + q0.z * bary.y) + q1.x * bary.z;              // :36-37
        const float uv_t = (q0.y * bary.x + q0.w * bary.y) + q1.y * bary.z;
        base_color = sample_texture(sc, mt.img_base, mt.samplers & 0xFFu, uv_s, uv_t, base_color);   // :39
    }
    return base_color.w < mt.alpha_cutoff;                                               // :40-42
}

template <bool TEX, bool TL = false>
SRD Payload shade_hit(const DevScene& sc, const TravHit& h) {
    Payload pl;
    pl.emission = splat(0.0f);
    pl.albedo_packed = 0; pl.normal_packed = 0; pl.material_info = 0; pl.transmission_ior_packed = 0;
    if (h.gid == 0xFFFFFFFFu) { pl.dist = -1.0f; return pl; }
    const float4 s0 = sc.shade[(size_t)h.slot * 3 + 0];
    const float4 s1 = sc.shade[(size_t)h.slot * 3 + 1];
    const float4 s2 = sc.shade[(size_t)h.slot * 3 + 2];
    const uint32_t inst = TL ? h.inst : __float_as_uint(s2.y), mesh = __float_as_uint(s2.z);   // two-level: the record belongs to the mesh, not to an instance
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
    if (TEX) { if (mc.textured) shade_textured(sc, h.slot, inst, mesh, bary, world_normal, pl); }
    return pl;
}

}  // namespace srd
