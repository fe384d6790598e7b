// Device side of the acceleration-structure maintenance: what the reference gets from
// vkCmdBuildAccelerationStructuresKHR in UPDATE mode (tlas.rs:124-140, accel.rs:263-267) — new instance
// transforms, same topology — done here as (1) re-flattening the instances' triangles to world space, in the
// same operation order as the host (bvh_build.cpp flatten_instances) so the records are bit-identical to a
// rebuild's, and (2) a bottom-up refit of the quantised 4-wide nodes. Box culling is conservative on every
// side, so query results do not depend on which tree answers them.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include <hipcub/hipcub.hpp>

#include <vector>

#include "bvh_gpu.h"
#include "bvh_layout.h"

namespace srd {

__device__ __forceinline__ void world_triangle(const SrMeshInfo* meshes, const FlatInstance& inst, uint32_t prim, float v0[3], float e1[3], float e2[3]) {
    const SrMeshInfo mi = meshes[inst.mesh_slot];
    const uint32_t* idx = (const uint32_t*)(uintptr_t)mi.indices;
    const SrVertex* vtx = (const SrVertex*)(uintptr_t)mi.vertices;
    const float* m = inst.o2w;
    float w[3][3];
    for (int j = 0; j < 3; j++) {
        const float* q = vtx[idx[3 * prim + j]].position;
        // transform_point (rt_utils.slang:278-281): rows dotted with (p, 1), left to right
        w[j][0] = ((m[0] * q[0] + m[1] * q[1]) + m[2] * q[2]) + m[3] * 1.0f;
        w[j][1] = ((m[4] * q[0] + m[5] * q[1]) + m[6] * q[2]) + m[7] * 1.0f;
        w[j][2] = ((m[8] * q[0] + m[9] * q[1]) + m[10] * q[2]) + m[11] * 1.0f;
    }
    for (int a = 0; a < 3; a++) { v0[a] = w[0][a]; e1[a] = w[1][a] - w[0][a]; e2[a] = w[2][a] - w[0][a]; }
}

// One thread per leaf-order slot: the slot keeps its triangle (global id), only the world-space record changes.
__global__ void flatten_slots_kernel(float4* tris, const float4* shade, const SrMeshInfo* meshes, const FlatInstance* instances, uint32_t n_tris) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_tris) return;
    const uint32_t gid = __float_as_uint(tris[(size_t)slot * 3 + 2].y);   // record: (v0, e1, e2, gid, 0, 0), bvh_build.cpp
    const uint32_t ii = __float_as_uint(shade[(size_t)slot * 3 + 2].y);
    const FlatInstance inst = instances[ii];
    if (gid < inst.tri_offset) return;      // cannot happen for a tree built from this layout; never index out of bounds
    float v0[3], e1[3], e2[3];
    world_triangle(meshes, inst, gid - inst.tri_offset, v0, e1, e2);
    tris[(size_t)slot * 3 + 0] = make_float4(v0[0], v0[1], v0[2], e1[0]);
    tris[(size_t)slot * 3 + 1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
    tris[(size_t)slot * 3 + 2] = make_float4(e2[2], __uint_as_float(gid), 0.0f, 0.0f);
}

// Padded box of one triangle record, as the host builder bounds it (bvh_build.cpp: fattened by the triangle
// test's barycentric slack, then one ulp each way).
__device__ __forceinline__ void tri_box(const float4* tris, uint32_t slot, float lo[3], float hi[3]) {
    const float4 a = tris[(size_t)slot * 3 + 0], b = tris[(size_t)slot * 3 + 1], c = tris[(size_t)slot * 3 + 2];
    const float v0[3] = {a.x, a.y, a.z}, e1[3] = {a.w, b.x, b.y}, e2[3] = {b.z, b.w, c.x};
    for (int k = 0; k < 3; k++) {
        const float p1 = v0[k] + e1[k], p2 = v0[k] + e2[k];
        const float pad = 4e-6f * (fabsf(e1[k]) + fabsf(e2[k]));
        const float l = fminf(v0[k], fminf(p1, p2)) - pad, h = fmaxf(v0[k], fmaxf(p1, p2)) + pad;
        lo[k] = fminf(lo[k], nextafterf(l, -INFINITY));
        hi[k] = fmaxf(hi[k], nextafterf(h, INFINITY));
    }
}

// One thread per node of one tree level (deepest level first): child boxes from the triangles (leaf children) or
// from the already refitted child nodes (node_box), then the same quantisation the host collapser applies
// (bvh_build.cpp Collapser::emit): origin = node min, per-axis power-of-two grid, planes rounded outward and
// verified with the decode expression fmaf(q, 2^e, origin).
constexpr double kGuardCells = 1.0 / 32.0;   // guard band around every quantised plane (see bvh_build.cpp, traverse.h)

__global__ void refit_level_kernel(uint32_t* nodes, const float4* tris, float* node_box, const uint32_t* level_nodes, uint32_t first, uint32_t count) {
    constexpr int W = srl::kBvhWidth;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t node = level_nodes ? level_nodes[first + i] : first + i;   // null list: the level is the index range itself
    uint32_t* q = nodes + (size_t)node * srl::kNodeDwords;
    float lo[W][3], hi[W][3];
    bool real[W];
    float lo_n[3] = {INFINITY, INFINITY, INFINITY}, hi_n[3] = {-INFINITY, -INFINITY, -INFINITY};
    int n_real = 0;
    for (int c = 0; c < W; c++) {
        const int ref = (int)q[srl::kChildOffset + c];
        for (int a = 0; a < 3; a++) { lo[c][a] = INFINITY; hi[c][a] = -INFINITY; }
        real[c] = false;
        if (ref >= 0) {
            const float* b = node_box + (size_t)ref * 6;
            for (int a = 0; a < 3; a++) { lo[c][a] = b[a]; hi[c][a] = b[3 + a]; }
            real[c] = lo[c][0] <= hi[c][0];
        } else {
            const uint32_t v = ~(uint32_t)ref, t0 = v >> 3, cnt = v & 7u;
            for (uint32_t t = 0; t < cnt; t++) tri_box(tris, t0 + t, lo[c], hi[c]);
            real[c] = cnt != 0u;
        }
        if (!real[c]) continue;
        n_real++;
        for (int a = 0; a < 3; a++) { lo_n[a] = fminf(lo_n[a], lo[c][a]); hi_n[a] = fmaxf(hi_n[a], hi[c][a]); }
    }
    uint32_t plane[6 * srl::kPlaneDwords], exps = 0;
    for (int k = 0; k < 6 * srl::kPlaneDwords; k++) plane[k] = 0u;
    float origin[3] = {0.0f, 0.0f, 0.0f};
    if (n_real > 0) {
        for (int a = 0; a < 3; a++) {
            origin[a] = nextafterf(lo_n[a] - (hi_n[a] - lo_n[a]) * (1.0f / 2048.0f), -INFINITY);   // ~1/8 cell below the minimum
            const float ext = hi_n[a] - origin[a];
            int e = -126;
            if (ext > 0.0f && ext < INFINITY && ext / 255.0f > 0.0f) { int fe; (void)frexpf(ext / 255.0f, &fe); e = max(fe, -126); }
            if (!(ext < INFINITY)) e = 127;
            for (; e < 127; e++) {   // grow the grid until every child's upper plane fits in a byte
                const float scale = ldexpf(1.0f, e);
                bool ok = true;
                for (int c = 0; c < W && ok; c++) {
                    if (!real[c]) continue;
                    int qh = (int)ceil(((double)hi[c][a] - (double)origin[a]) / (double)scale + kGuardCells);
                    qh = max(qh, 0);
                    while (qh <= 255 && fmaf((float)qh, scale, origin[a]) < hi[c][a]) qh++;
                    if (qh > 255) ok = false;
                }
                if (ok) break;
            }
            const float scale = ldexpf(1.0f, e);
            exps |= (uint32_t)(e + 127) << (8 * a);
            for (int c = 0; c < W; c++) {
                uint32_t ql = 255u, qh = 0u;   // inverted box for unused children
                if (real[c]) {
                    int l = (int)floor(((double)lo[c][a] - (double)origin[a]) / (double)scale - kGuardCells);
                    l = min(max(l, 0), 255);
                    while (l > 0 && fmaf((float)l, scale, origin[a]) > lo[c][a]) l--;
                    int h = (int)ceil(((double)hi[c][a] - (double)origin[a]) / (double)scale + kGuardCells);
                    h = min(max(h, 0), 255);
                    while (h < 255 && fmaf((float)h, scale, origin[a]) < hi[c][a]) h++;
                    ql = (uint32_t)l; qh = (uint32_t)h;
                }
                plane[a * srl::kPlaneDwords + c / 4] |= ql << (8 * (c % 4));
                plane[(3 + a) * srl::kPlaneDwords + c / 4] |= qh << (8 * (c % 4));
            }
        }
    } else {
        for (int a = 0; a < 3; a++) {
            for (int d = 0; d < srl::kPlaneDwords; d++) { plane[a * srl::kPlaneDwords + d] = 0xFFFFFFFFu; plane[(3 + a) * srl::kPlaneDwords + d] = 0u; }
            exps |= 127u << (8 * a);
        }
    }
    q[0] = __float_as_uint(origin[0]); q[1] = __float_as_uint(origin[1]); q[2] = __float_as_uint(origin[2]); q[3] = exps;
    for (int k = 0; k < 6 * srl::kPlaneDwords; k++) q[srl::kPlaneOffset + k] = plane[k];
    float* b = node_box + (size_t)node * 6;
    for (int a = 0; a < 3; a++) { b[a] = lo_n[a]; b[3 + a] = hi_n[a]; }
}


// ---------------------------------------------------------------------------------------------------------------
// OpType::FastBuild on the device (PREFER_FAST_BUILD, acceleration_structure/mod.rs:33-35): a linear BVH.
//   prims      world-space triangle records + centroids of their padded boxes, scene bounds (wave-reduced atomics)
//   morton     63-bit Morton code of the centroid (21 bits per axis), radix-sorted with the triangle id (hipCUB)
//   hierarchy  binary radix tree over the sorted codes (Karras 2012; equal codes are split by index)
//   fit        bottom-up: box and "binary walk height" of every radix node (second arriver continues upward)
//   collapse   top-down, one launch per level: radix subtrees of <= kLeafMax triangles become leaves (their triangles are
//              consecutive in sorted order = leaf order), inner nodes take up to 4 children by repeatedly opening
//              the child with the largest box while the stack budget allows (same rule as bvh_build.cpp's Collapser)
//   leaves     triangle / shade / shade_tex records in leaf order, slot_of_gid
//   refit      the kernel above computes the boxes of the 4-wide nodes and quantises them, deepest level first
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t find_instance(const FlatInstance* inst, uint32_t n_inst, uint32_t gid) {
    uint32_t lo = 0, hi = n_inst;            // last instance whose tri_offset <= gid
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (inst[mid].tri_offset <= gid) lo = mid; else hi = mid; }
    return lo;
}
__device__ __forceinline__ uint32_t enc_f(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float dec_f(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e); }

__global__ void lbvh_prims_kernel(const SrMeshInfo* meshes, const FlatInstance* instances, uint32_t n_inst, uint32_t n_tris, float4* W, float4* cent,
                                  uint32_t* bounds_enc) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (gid < n_tris) {
        const uint32_t ii = find_instance(instances, n_inst, gid);
        const FlatInstance inst = instances[ii];
        float v0[3], e1[3], e2[3];
        world_triangle(meshes, inst, gid - inst.tri_offset, v0, e1, e2);
        W[(size_t)gid * 3 + 0] = make_float4(v0[0], v0[1], v0[2], e1[0]);
        W[(size_t)gid * 3 + 1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
        W[(size_t)gid * 3 + 2] = make_float4(e2[2], __uint_as_float(gid), 0.0f, 0.0f);
        tri_box(W, gid, lo, hi);
        cent[gid] = make_float4(0.5f * lo[0] + 0.5f * hi[0], 0.5f * lo[1] + 0.5f * hi[1], 0.5f * lo[2] + 0.5f * hi[2], __uint_as_float(ii));
    }
    for (int a = 0; a < 3; a++) {
        float l = lo[a], h = hi[a];
        for (int o = 32; o > 0; o >>= 1) { l = fminf(l, __shfl_xor(l, o)); h = fmaxf(h, __shfl_xor(h, o)); }
        if ((threadIdx.x & 63) == 0) {
            if (l <= h) { atomicMin(bounds_enc + a, enc_f(l)); atomicMax(bounds_enc + 3 + a, enc_f(h)); }
        }
    }
}

__device__ __forceinline__ unsigned long long spread21(uint32_t v) {   // 21 bits -> every third bit
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void lbvh_morton_kernel(const float4* cent, const uint32_t* bounds_enc, uint32_t n_tris, unsigned long long* keys, uint32_t* vals) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_tris) return;
    const float4 c = cent[gid];
    const float p[3] = {c.x, c.y, c.z};
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        const float lo = dec_f(bounds_enc[a]), hi = dec_f(bounds_enc[3 + a]);
        const float ext = hi - lo;
        float t = ext > 0.0f ? (p[a] - lo) / ext : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);                      // NaN -> 0
        q[a] = min((uint32_t)(t * 2097152.0f), 2097151u);
    }
    keys[gid] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
    vals[gid] = gid;
}

// delta(i, j): length of the common prefix of the sorted codes, ties broken by the index (Karras 2012, section 4)
__device__ __forceinline__ int lbvh_delta(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

__global__ void lbvh_hierarchy_kernel(const unsigned long long* keys, int n, int2* children, uint32_t* parent_of_inner, uint32_t* parent_of_leaf, uint2* range) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1) if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    for (int t = l;;) {
        t = (t + 1) >> 1;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    int2 ch;
    ch.x = first == gamma ? ~gamma : gamma;               // leaf k is encoded ~k
    ch.y = last == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    children[i] = ch;
    range[i] = make_uint2((uint32_t)first, (uint32_t)last);
    if (ch.x >= 0) parent_of_inner[ch.x] = (uint32_t)i; else parent_of_leaf[~ch.x] = (uint32_t)i;
    if (ch.y >= 0) parent_of_inner[ch.y] = (uint32_t)i; else parent_of_leaf[~ch.y] = (uint32_t)i;
    if (i == 0) parent_of_inner[0] = 0xFFFFFFFFu;
}

// Bottom-up fit: boxes of the radix nodes and the stack height of a purely binary walk below each (0 for subtrees that
// will become leaves), second arriver at a node continues (the first one's writes are visible after the fence).
__global__ void lbvh_fit_kernel(const float4* W, const uint32_t* sorted_gid, int n, const int2* children, const uint32_t* parent_of_inner,
                                const uint32_t* parent_of_leaf, const uint2* range, float* bin_box, uint32_t* bin_height, uint32_t* bin_size,
                                uint32_t* flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t node = parent_of_leaf[i];
    while (node != 0xFFFFFFFFu) {
        __threadfence();
        if (atomicAdd(flags + node, 1u) == 0u) return;
        __threadfence();
        const int2 ch = children[node];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t h = 0;
        const int cc[2] = {ch.x, ch.y};
        for (int c = 0; c < 2; c++) {
            if (cc[c] < 0) tri_box(W, sorted_gid[~cc[c]], lo, hi);
            else {
                const volatile float* b = bin_box + (size_t)cc[c] * 6;
                for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], b[a]); hi[a] = fmaxf(hi[a], b[3 + a]); }
                h = max(h, ((const volatile uint32_t*)bin_height)[cc[c]]);
            }
        }
        const uint2 r = range[node];
        float* b = bin_box + (size_t)node * 6;
        for (int a = 0; a < 3; a++) { b[a] = lo[a]; b[3 + a] = hi[a]; }
        bin_size[node] = r.y - r.x + 1u;
        bin_height[node] = (r.y - r.x + 1u <= srl::kLeafMax) ? 0u : h + 1u;
        node = parent_of_inner[node];
    }
}

// A child of a 4-wide node during the collapse: `ref` is a binary-tree reference (>= 0: binary node, < 0: single triangle
// ~sorted index); subtrees of at most kLeafMax triangles become leaves.
struct LbvhKid { int ref; uint32_t size; float area; uint32_t need; bool inner; };

__device__ __forceinline__ LbvhKid lbvh_kid(int ref, const uint32_t* bin_size, const float* bin_box, const uint32_t* bin_height) {
    LbvhKid k;
    k.ref = ref; k.area = 0.0f; k.need = 0; k.inner = false;
    if (ref < 0) { k.size = 1; return k; }
    k.size = bin_size[ref];
    if (k.size <= srl::kLeafMax) return k;
    const float* b = bin_box + (size_t)ref * 6;
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    k.area = dx * dy + dy * dz + dz * dx; k.need = bin_height[ref]; k.inner = true;
    return k;
}

// One thread per 4-wide node of the current level. counters: [0] = nodes allocated, [1] = max stack, [2] = overflow flag.
// Leaf slots: a node owns the slot range [first_of_node, first_of_node + size) of the leaf-order arrays; its children take
// consecutive sub-ranges in child order, and the triangles of a leaf child get their slots here (slot_of_sorted), so the
// binary tree's subtrees need not be contiguous in Morton order (they are for the radix tree, not for PLOC).
__global__ void lbvh_collapse_kernel(uint32_t* nodes, uint32_t level_first, uint32_t level_count, uint32_t node_cap, int* bin_of_node, uint32_t* budget_of_node,
                                     uint32_t* prefix_of_node, uint32_t* first_of_node, const int2* children, const uint32_t* bin_size, const float* bin_box,
                                     const uint32_t* bin_height, uint32_t* slot_of_sorted, uint32_t* counters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= level_count) return;
    const uint32_t self = level_first + t;
    const int bin = bin_of_node[self];
    const uint32_t budget = budget_of_node[self];
    constexpr int W = srl::kBvhWidth;
    LbvhKid kids[W];
    int nk = 0;
    const int2 ch = children[bin];
    kids[nk++] = lbvh_kid(ch.x, bin_size, bin_box, bin_height);
    kids[nk++] = lbvh_kid(ch.y, bin_size, bin_box, bin_height);
    while (nk < W) {
        int best = -1; float best_area = -1.0f;
        for (int i = 0; i < nk; i++) if (kids[i].inner && kids[i].area > best_area) { best_area = kids[i].area; best = i; }
        if (best < 0) break;
        const int2 cb = children[kids[best].ref];
        const LbvhKid ka = lbvh_kid(cb.x, bin_size, bin_box, bin_height), kb = lbvh_kid(cb.y, bin_size, bin_box, bin_height);
        bool fits = ka.need + (uint32_t)nk <= budget && kb.need + (uint32_t)nk <= budget;   // with nk+1 children every subtree gets budget - nk entries
        for (int i = 0; i < nk && fits; i++) if (i != best && kids[i].need + (uint32_t)nk > budget) fits = false;
        if (!fits) break;
        kids[best] = ka;
        kids[nk++] = kb;
    }
    uint32_t* q = nodes + (size_t)self * srl::kNodeDwords;
    for (int k = 0; k < srl::kNodeDwords; k++) q[k] = 0u;
    const uint32_t mine = prefix_of_node[self] + (uint32_t)(nk - 1);
    atomicMax(counters + 1, mine);
    uint32_t first = first_of_node[self];
    for (int i = 0; i < W; i++) {
        uint32_t ref = 0xFFFFFFFFu;                              // leaf_ref(0, 0): unused child
        if (i < nk) {
            if (kids[i].inner) {
                const uint32_t idx = atomicAdd(counters + 0, 1u);
                if (idx < node_cap) {
                    bin_of_node[idx] = kids[i].ref;
                    budget_of_node[idx] = budget - (uint32_t)(nk - 1);
                    prefix_of_node[idx] = mine;
                    first_of_node[idx] = first;
                    ref = idx;
                } else { atomicExch(counters + 2, 1u); }
            } else {
                ref = ~((first << 3) | kids[i].size);
                // slots of the leaf's triangles: walk the (at most kLeafMax-triangle) binary subtree
                int stack[8]; int sp = 0; uint32_t slot = first;
                stack[sp++] = kids[i].ref;
                while (sp > 0) {
                    const int r = stack[--sp];
                    if (r < 0) slot_of_sorted[~r] = slot++;
                    else { const int2 c2 = children[r]; if (sp < 7) { stack[sp++] = c2.y; stack[sp++] = c2.x; } }
                }
            }
            first += kids[i].size;
        }
        q[srl::kChildOffset + i] = ref;
    }
}

// Leaf-order records of one triangle (by sorted index): its slot was assigned by the collapse.
__global__ void lbvh_leaves_kernel(const float4* W, const float4* cent, const uint32_t* sorted_gid, const uint32_t* slot_of_sorted, uint32_t n_tris,
                                   const SrMeshInfo* meshes, const FlatInstance* instances, float4* tris, float4* shade, float4* shade_tex, uint32_t* slot_of_gid) {
    const uint32_t s_idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (s_idx >= n_tris) return;
    const uint32_t slot = slot_of_sorted[s_idx];
    const uint32_t gid = sorted_gid[s_idx];
    for (int k = 0; k < 3; k++) tris[(size_t)slot * 3 + k] = W[(size_t)gid * 3 + k];
    slot_of_gid[gid] = slot;
    const uint32_t ii = __float_as_uint(cent[gid].w);
    const FlatInstance inst = instances[ii];
    const SrMeshInfo mi = meshes[inst.mesh_slot];
    const uint32_t* idx = (const uint32_t*)(uintptr_t)mi.indices;
    const SrVertex* vtx = (const SrVertex*)(uintptr_t)mi.vertices;
    const uint32_t prim = gid - inst.tri_offset;
    const SrVertex* v[3] = {vtx + idx[3 * prim], vtx + idx[3 * prim + 1], vtx + idx[3 * prim + 2]};
    shade[(size_t)slot * 3 + 0] = make_float4(v[0]->normal[0], v[0]->normal[1], v[0]->normal[2], v[1]->normal[0]);
    shade[(size_t)slot * 3 + 1] = make_float4(v[1]->normal[1], v[1]->normal[2], v[2]->normal[0], v[2]->normal[1]);
    shade[(size_t)slot * 3 + 2] = make_float4(v[2]->normal[2], __uint_as_float(ii), __uint_as_float(inst.mesh_slot), 0.0f);
    if (shade_tex) {
        float4* q = shade_tex + (size_t)slot * 6;
        q[0] = make_float4(v[0]->base_color_tex_coord[0], v[0]->base_color_tex_coord[1], v[1]->base_color_tex_coord[0], v[1]->base_color_tex_coord[1]);
        q[1] = make_float4(v[2]->base_color_tex_coord[0], v[2]->base_color_tex_coord[1], v[0]->normal_tex_coord[0], v[0]->normal_tex_coord[1]);
        q[2] = make_float4(v[1]->normal_tex_coord[0], v[1]->normal_tex_coord[1], v[2]->normal_tex_coord[0], v[2]->normal_tex_coord[1]);
        q[3] = make_float4(v[0]->tangent[0], v[0]->tangent[1], v[0]->tangent[2], v[0]->tangent[3] >= 0.0f ? 1.0f : -1.0f);
        q[4] = make_float4(v[1]->tangent[0], v[1]->tangent[1], v[1]->tangent[2], v[2]->tangent[0]);
        q[5] = make_float4(v[2]->tangent[1], v[2]->tangent[2], 0.0f, 0.0f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) as the topology of the fast build: bottom-up
// agglomeration of the Morton-ordered clusters. Every iteration each cluster finds, among its `radius` neighbours on
// either side, the one whose union with it has the smallest surface area; mutual nearest neighbours merge into a binary
// node (box, size and binary-walk height are known at once), the survivors are compacted in order, until one is left.
// Build quality is close to the top-down SAH build at a small multiple of the radix tree's cost.
// ---------------------------------------------------------------------------------------------------------------

__global__ void ploc_init_kernel(const float4* W, const uint32_t* sorted_gid, uint32_t n, int* cid, float* cbox) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    tri_box(W, sorted_gid[i], lo, hi);
    cid[i] = ~(int)i;
    float* b = cbox + (size_t)i * 6;
    for (int a = 0; a < 3; a++) { b[a] = lo[a]; b[3 + a] = hi[a]; }
}

__global__ void ploc_nn_kernel(const float* cbox, uint32_t m, uint32_t radius, uint32_t* nn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float* b = cbox + (size_t)i * 6;
    const float lo[3] = {b[0], b[1], b[2]}, hi[3] = {b[3], b[4], b[5]};
    const uint32_t j0 = i > radius ? i - radius : 0u, j1 = min(m - 1u, i + radius);
    float best = INFINITY; uint32_t best_j = i;
    for (uint32_t j = j0; j <= j1; j++) {
        if (j == i) continue;
        const float* c = cbox + (size_t)j * 6;
        const float dx = fmaxf(hi[0], c[3]) - fminf(lo[0], c[0]), dy = fmaxf(hi[1], c[4]) - fminf(lo[1], c[1]), dz = fmaxf(hi[2], c[5]) - fminf(lo[2], c[2]);
        const float area = dx * dy + dy * dz + dz * dx;
        if (area < best) { best = area; best_j = j; }        // ties: the lower index (scan order)
    }
    nn[i] = best_j;
}

// Mutual nearest neighbours merge (the lower index keeps the merged cluster, the higher one is dropped).
__global__ void ploc_merge_kernel(const uint32_t* nn, uint32_t m, int* cid, float* cbox, uint32_t* valid, int2* children, float* bin_box, uint32_t* bin_size,
                                  uint32_t* bin_height, uint32_t* node_counter) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = nn[i];
    if (j == i || nn[j] != i) { valid[i] = 1u; return; }
    if (j < i) { valid[i] = 0u; return; }
    const int ci = cid[i], cj = cid[j];
    const uint32_t node = atomicAdd(node_counter, 1u);
    const uint32_t si = ci < 0 ? 1u : bin_size[ci], sj = cj < 0 ? 1u : bin_size[cj];
    const uint32_t hi_ = ci < 0 ? 0u : bin_height[ci], hj = cj < 0 ? 0u : bin_height[cj];
    children[node] = make_int2(ci, cj);
    bin_size[node] = si + sj;
    bin_height[node] = (si + sj <= srl::kLeafMax) ? 0u : max(hi_, hj) + 1u;
    float* bi = cbox + (size_t)i * 6;
    const float* bj = cbox + (size_t)j * 6;
    float* nb = bin_box + (size_t)node * 6;
    for (int a = 0; a < 3; a++) { bi[a] = fminf(bi[a], bj[a]); bi[3 + a] = fmaxf(bi[3 + a], bj[3 + a]); nb[a] = bi[a]; nb[3 + a] = bi[3 + a]; }
    cid[i] = (int)node;
    valid[i] = 1u;
}

__global__ void ploc_compact_kernel(const int* cid, const float* cbox, const uint32_t* valid, const uint32_t* pos, uint32_t m, int* cid_out, float* cbox_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || !valid[i]) return;
    const uint32_t p = pos[i];
    cid_out[p] = cid[i];
    for (int a = 0; a < 6; a++) cbox_out[(size_t)p * 6 + a] = cbox[(size_t)i * 6 + a];
}

}  // namespace srd

using namespace srd;

int srk_launch_flatten_slots(float4* tris, const float4* shade, const SrMeshInfo* meshes, const FlatInstance* instances, uint32_t n_tris, hipStream_t stream) {
    if (n_tris == 0) return 0;
    flatten_slots_kernel<<<dim3((n_tris + 255) / 256), dim3(256), 0, stream>>>(tris, shade, meshes, instances, n_tris);
    return (int)hipGetLastError();
}

int srk_launch_refit(uint32_t* nodes, const float4* tris, float* node_box, const uint32_t* level_nodes, const uint32_t* level_offsets_host,
                     uint32_t n_levels, hipStream_t stream) {
    for (uint32_t l = 0; l < n_levels; l++) {     // level_offsets_host[l] .. [l+1]: deepest level first
        const uint32_t first = level_offsets_host[l], count = level_offsets_host[l + 1] - first;
        if (count == 0) continue;
        refit_level_kernel<<<dim3((count + 63) / 64), dim3(64), 0, stream>>>(nodes, tris, node_box, level_nodes, first, count);
    }
    return (int)hipGetLastError();
}

// Device LBVH build. All outputs are device buffers owned by the caller; `scratch` is reused across builds. Returns 0, a
// hipError_t (> 0), or -1 when the tree does not fit the limits (caller falls back to the host builder).
int srk_lbvh_build(const LbvhArgs& a, LbvhResult* out, hipStream_t stream) {
    const uint32_t n = a.n_tris;
    const int B = 256;
    const dim3 gt((n + B - 1) / B), bt(B);
    // carve the scratch slab
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = (char*)a.scratch + off; off += (bytes + 255) & ~(size_t)255; return p; };
    float4* W = (float4*)take((size_t)n * 48);
    float4* cent = (float4*)take((size_t)n * 16);
    unsigned long long* keys_a = (unsigned long long*)take((size_t)n * 8);
    unsigned long long* keys_b = (unsigned long long*)take((size_t)n * 8);
    uint32_t* vals_a = (uint32_t*)take((size_t)n * 4);
    uint32_t* vals_b = (uint32_t*)take((size_t)n * 4);
    int2* children = (int2*)take((size_t)n * 8);
    uint2* range = (uint2*)take((size_t)n * 8);             // radix tree only; PLOC: nn | valid
    uint32_t* parent_inner = (uint32_t*)take((size_t)n * 4);   // radix tree only; PLOC: scan output
    uint32_t* parent_leaf = (uint32_t*)take((size_t)n * 4);
    float* bin_box = (float*)take((size_t)n * 24);
    uint32_t* bin_height = (uint32_t*)take((size_t)n * 4);
    uint32_t* bin_size = (uint32_t*)take((size_t)n * 4);
    uint32_t* flags = (uint32_t*)take((size_t)n * 4);
    uint32_t* slot_of_sorted = (uint32_t*)take((size_t)n * 4);
    int* bin_of_node = (int*)take((size_t)a.node_cap * 4);
    uint32_t* budget_of_node = (uint32_t*)take((size_t)a.node_cap * 4);
    uint32_t* prefix_of_node = (uint32_t*)take((size_t)a.node_cap * 4);
    uint32_t* first_of_node = (uint32_t*)take((size_t)a.node_cap * 4);
    uint32_t* small = (uint32_t*)take(256);          // [0..5] bounds, [8..10] counters, [12] PLOC node counter
    // PLOC cluster arrays (double-buffered)
    int* cid[2] = {nullptr, nullptr}; float* cbox[2] = {nullptr, nullptr};
    if (a.ploc) for (int k = 0; k < 2; k++) { cid[k] = (int*)take((size_t)n * 4); cbox[k] = (float*)take((size_t)n * 24); }
    size_t cub_bytes = 0, scan_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, keys_a, keys_b, vals_a, vals_b, (int)n, 0, 63, stream);
    if (e != hipSuccess) return (int)e;
    if (a.ploc && (e = hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, stream)) != hipSuccess) return (int)e;
    void* cub_tmp = take(cub_bytes > scan_bytes ? cub_bytes : scan_bytes);
    if (off > a.scratch_bytes) return (int)hipErrorOutOfMemory;

    const uint32_t init[16] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if ((e = hipMemcpyAsync(small, init, sizeof(init), hipMemcpyHostToDevice, stream)) != hipSuccess) return (int)e;
    lbvh_prims_kernel<<<gt, bt, 0, stream>>>(a.meshes, a.instances, a.n_instances, n, W, cent, small);
    lbvh_morton_kernel<<<gt, bt, 0, stream>>>(cent, small, n, keys_a, vals_a);
    if ((e = hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, keys_a, keys_b, vals_a, vals_b, (int)n, 0, 63, stream)) != hipSuccess) return (int)e;
    int root_bin = 0;
    if (!a.ploc) {
        if ((e = hipMemsetAsync(flags, 0, (size_t)n * 4, stream)) != hipSuccess) return (int)e;
        lbvh_hierarchy_kernel<<<gt, bt, 0, stream>>>(keys_b, (int)n, children, parent_inner, parent_leaf, range);
        lbvh_fit_kernel<<<gt, bt, 0, stream>>>(W, vals_b, (int)n, children, parent_inner, parent_leaf, range, bin_box, bin_height, bin_size, flags);
    } else {
        uint32_t* nn = (uint32_t*)range;
        uint32_t* valid = nn + n;
        uint32_t* pos = parent_inner;
        uint32_t* node_counter = small + 12;
        ploc_init_kernel<<<gt, bt, 0, stream>>>(W, vals_b, n, cid[0], cbox[0]);
        uint32_t m = n;
        int cur = 0, guard = 0;
        while (m > 1) {
            const dim3 gm((m + B - 1) / B);
            ploc_nn_kernel<<<gm, bt, 0, stream>>>(cbox[cur], m, (uint32_t)a.ploc, nn);
            ploc_merge_kernel<<<gm, bt, 0, stream>>>(nn, m, cid[cur], cbox[cur], valid, children, bin_box, bin_size, bin_height, node_counter);
            if ((e = hipcub::DeviceScan::ExclusiveSum(cub_tmp, scan_bytes, valid, pos, (int)m, stream)) != hipSuccess) return (int)e;
            ploc_compact_kernel<<<gm, bt, 0, stream>>>(cid[cur], cbox[cur], valid, pos, m, cid[cur ^ 1], cbox[cur ^ 1]);
            uint32_t tail[2];      // new count = pos[m-1] + valid[m-1]
            if ((e = hipMemcpyAsync(&tail[0], pos + (m - 1), 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return (int)e;
            if ((e = hipMemcpyAsync(&tail[1], valid + (m - 1), 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return (int)e;
            if ((e = hipStreamSynchronize(stream)) != hipSuccess) return (int)e;
            const uint32_t m_new = tail[0] + tail[1];
            if (m_new >= m || ++guard > 4096) return -1;     // no progress: cannot happen (the closest pair is always mutual)
            m = m_new;
            cur ^= 1;
        }
        if ((e = hipMemcpyAsync(&root_bin, cid[cur], 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return (int)e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return (int)e;
        if (root_bin < 0) return -1;
    }
    // root of the 4-wide tree = the binary root; its budget is the binary height, at least the regular stack size
    uint32_t root_height = 0;
    if ((e = hipMemcpyAsync(&root_height, bin_height + root_bin, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return (int)e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return (int)e;
    if (root_height > a.stack_cap) return -1;
    const uint32_t budget = root_height > a.stack_floor ? root_height : a.stack_floor;
    const uint32_t zero = 0;
    (void)hipMemcpyAsync(bin_of_node, &root_bin, 4, hipMemcpyHostToDevice, stream);
    (void)hipMemcpyAsync(budget_of_node, &budget, 4, hipMemcpyHostToDevice, stream);
    (void)hipMemcpyAsync(prefix_of_node, &zero, 4, hipMemcpyHostToDevice, stream);
    (void)hipMemcpyAsync(first_of_node, &zero, 4, hipMemcpyHostToDevice, stream);
    out->level_ranges.clear();
    uint32_t level_first = 0, level_count = 1;
    uint32_t* counters = small + 8;
    while (level_count) {
        out->level_ranges.emplace_back(level_first, level_count);
        lbvh_collapse_kernel<<<dim3((level_count + 63) / 64), dim3(64), 0, stream>>>((uint32_t*)a.nodes, level_first, level_count, a.node_cap, bin_of_node,
                                                                                       budget_of_node, prefix_of_node, first_of_node, children, bin_size, bin_box,
                                                                                       bin_height, slot_of_sorted, counters);
        uint32_t c[3];
        if ((e = hipMemcpyAsync(c, counters, 12, hipMemcpyDeviceToHost, stream)) != hipSuccess) return (int)e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return (int)e;
        if (c[2] || c[0] > a.node_cap) return -1;
        level_first += level_count;
        level_count = c[0] - level_first;
        out->n_nodes = c[0];
        out->max_stack = c[1];
        if (out->level_ranges.size() > 128) return -1;
    }
    out->max_depth = (uint32_t)out->level_ranges.size();
    lbvh_leaves_kernel<<<gt, bt, 0, stream>>>(W, cent, vals_b, slot_of_sorted, n, a.meshes, a.instances, a.tris, a.shade, a.shade_tex, a.slot_of_gid);
    for (size_t l = out->level_ranges.size(); l-- > 0;)
        refit_level_kernel<<<dim3((out->level_ranges[l].second + 63) / 64), dim3(64), 0, stream>>>((uint32_t*)a.nodes, a.tris, a.node_box, nullptr,
                                                                                                     out->level_ranges[l].first, out->level_ranges[l].second);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return (int)e;
    return 0;
}

size_t srk_lbvh_scratch_bytes(uint32_t n_tris, uint32_t node_cap) {
    size_t cub_bytes = 0, scan_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (int)n_tris, 0, 63, nullptr);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)n_tris, nullptr);
    return (size_t)n_tris * (48 + 16 + 16 + 8 + 8 + 8 + 4 + 4 + 24 + 4 + 4 + 4 + 4 + 2 * (4 + 24)) + (size_t)node_cap * 16 + std::max(cub_bytes, scan_bytes) + 96 * 256;
}
