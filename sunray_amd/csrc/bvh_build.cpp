// World-space BVH builder — the replacement for vkCmdBuildAccelerationStructuresKHR
// (src/vulkan_abstraction/acceleration_structure/accel.rs:134-138) with BLAS inputs as in
// blas.rs:266-278 (positions = first 12 bytes of each 96-byte vertex, u32 indices, opaque) and TLAS
// instances as in resource_manager.rs:243-251 (cull-disabled, mask 0xFF).
//
// All instances are flattened to world space (288 GB of HBM makes instancing-by-copy affordable
// and removes the per-instance ray transform from traversal), then a binned-SAH BVH2 is built:
// 16 bins on each of the three axes, leaves of <= kLeafMax (2) triangles, depth bounded so the per-lane LDS
// traversal stack (traverse.h kStackDepth) can never overflow. The top of the tree is built by
// parallel tasks; nodes are emitted in depth-first order so a node's first child follows it.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <future>

#include "bvh_layout.h"
#include "host.h"

namespace srh {

namespace {

struct Aabb {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
    void add(const float* p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void add(const Aabb& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

#ifndef SR_SAH_BINS
#define SR_SAH_BINS 16
#endif
constexpr int kBins = SR_SAH_BINS;
constexpr uint32_t kLeafMax = srl::kLeafMax;
constexpr uint32_t kParallelMin = 1u << 15;

struct Builder {
    std::vector<Aabb> bounds;
    std::vector<float> centroid[3];
    std::vector<uint32_t> order;
    uint32_t max_depth;
    double root_area = 1.0;

    struct Sub {                 // a built subtree with subtree-local node indices
        std::vector<float> nodes;
        uint32_t depth = 0;      // inner levels below (and including) its root
        double cost = 0.0;
    };

    explicit Builder(uint32_t md) : max_depth(md) {}

    static int leaf_ref(uint32_t first, uint32_t count) { return (int)~((first << 3) | count); }

    // Writes child `which`'s box into the 16-float node record (layout: traverse.h).
    static void put_box(float* n, int which, const Aabb& b) {
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) {   // one ulp of padding each way: boxes stay conservative for
            lo[a] = std::nextafter(b.lo[a], -INFINITY);  // v0 + e1 rounding and never have zero width
            hi[a] = std::nextafter(b.hi[a], INFINITY);
        }
        if (which == 0) { n[0] = lo[0]; n[1] = hi[0]; n[2] = lo[1]; n[3] = hi[1]; n[8] = lo[2]; n[9] = hi[2]; }
        else { n[4] = lo[0]; n[5] = hi[0]; n[6] = lo[1]; n[7] = hi[1]; n[10] = lo[2]; n[11] = hi[2]; }
    }
    static void put_child(float* n, int which, int ref) { memcpy(n + 12 + which, &ref, 4); }

    Aabb range_bounds(uint32_t first, uint32_t count) const {
        Aabb b; b.reset();
        for (uint32_t i = first; i < first + count; i++) b.add(bounds[order[i]]);
        return b;
    }

    // Choose the split of [first, first+count); returns the index of the first element of the right half.
    uint32_t partition(uint32_t first, uint32_t count, uint32_t depth) {
        const uint32_t last = first + count;
        // depth guard: a balanced median split needs ceil(log2(count / kLeafMax)) more levels
        uint32_t need = 0;
        for (uint32_t c = (count + kLeafMax - 1) / kLeafMax; c > 1; c = (c + 1) / 2) need++;
        const bool force_median = depth + need + 1 >= max_depth;
        Aabb cb; cb.reset();
        for (uint32_t i = first; i < last; i++) {
            const uint32_t id = order[i];
            const float c[3] = {centroid[0][id], centroid[1][id], centroid[2][id]};
            cb.add(c);
        }
        int best_axis = -1, best_bin = -1;
        float best_cost = INFINITY;
        if (!force_median) {
            for (int ax = 0; ax < 3; ax++) {
                const float lo = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0.0f)) continue;
                const float scale = (float)kBins * (1.0f - 1e-6f) / ext;
                Aabb bb[kBins]; uint32_t bc[kBins];
                for (int b = 0; b < kBins; b++) { bb[b].reset(); bc[b] = 0; }
                for (uint32_t i = first; i < last; i++) {
                    const uint32_t id = order[i];
                    int b = (int)((centroid[ax][id] - lo) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    bb[b].add(bounds[id]); bc[b]++;
                }
                float ra[kBins]; uint32_t rc[kBins];
                Aabb acc; acc.reset(); uint32_t n = 0;
                for (int b = kBins - 1; b >= 1; b--) { acc.add(bb[b]); n += bc[b]; ra[b] = acc.half_area(); rc[b] = n; }
                acc.reset(); n = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.add(bb[b]); n += bc[b];
                    if (n == 0 || rc[b + 1] == 0) continue;
                    const float cost = acc.half_area() * (float)n + ra[b + 1] * (float)rc[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
                }
            }
        }
        if (best_axis >= 0) {
            const float lo = cb.lo[best_axis], ext = cb.hi[best_axis] - cb.lo[best_axis];
            const float scale = (float)kBins * (1.0f - 1e-6f) / ext;
            const std::vector<float>& cen = centroid[best_axis];
            auto mid_it = std::partition(order.begin() + first, order.begin() + last, [&](uint32_t id) {
                int b = (int)((cen[id] - lo) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= best_bin;
            });
            const uint32_t mid = (uint32_t)(mid_it - order.begin());
            if (mid != first && mid != last) return mid;
        }
        // median split along the widest centroid axis (also the depth-guard path)
        int ax = 0;
        float e = cb.hi[0] - cb.lo[0];
        for (int a = 1; a < 3; a++) if (cb.hi[a] - cb.lo[a] > e) { e = cb.hi[a] - cb.lo[a]; ax = a; }
        const uint32_t mid = first + count / 2;
        const std::vector<float>& cen = centroid[ax];
        std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + last,
                         [&](uint32_t a, uint32_t b) { return cen[a] < cen[b] || (cen[a] == cen[b] && a < b); });
        return mid;
    }

    // Builds the subtree over [first, first+count) (count > kLeafMax) into the EMPTY `out`, root at
    // local index 0. Large halves near the top are built by parallel tasks and merged.
    void build_sub(uint32_t first, uint32_t count, uint32_t depth, Sub& out) {
        const uint32_t mid = partition(first, count, depth);
        const uint32_t cnt[2] = {mid - first, first + count - mid};
        const uint32_t beg[2] = {first, mid};
        const bool par = cnt[0] >= kParallelMin && cnt[1] >= kParallelMin && depth < 4;
        if (!par) {
            // partition() already ran for this range: redo it inside the in-place builder would change
            // nothing (the order array is already split), so emit this node here and recurse in place.
            emit_node(beg, cnt, depth, out, out);
            return;
        }
        Sub child[2];
        std::future<void> fut = std::async(std::launch::async, [&] { build_sub(beg[1], cnt[1], depth + 1, child[1]); });
        build_sub(beg[0], cnt[0], depth + 1, child[0]);
        fut.get();
        out.nodes.assign(16, 0.0f);
        uint32_t below = 0;
        for (int c = 0; c < 2; c++) {
            const Aabb cb = range_bounds(beg[c], cnt[c]);
            put_box(&out.nodes[0], c, cb);
            out.cost += cb.half_area() / root_area;
            const size_t at = out.nodes.size();
            const int off = (int)(at / 16);
            out.nodes.insert(out.nodes.end(), child[c].nodes.begin(), child[c].nodes.end());
            for (size_t n = at; n < out.nodes.size(); n += 16)
                for (int k = 0; k < 2; k++) {
                    int ref; memcpy(&ref, &out.nodes[n + 12 + k], 4);
                    if (ref >= 0) { ref += off; memcpy(&out.nodes[n + 12 + k], &ref, 4); }
                }
            std::vector<float>().swap(child[c].nodes);
            out.cost += child[c].cost;
            below = std::max(below, child[c].depth);
            put_child(&out.nodes[0], c, off);
        }
        out.depth = below + 1;
    }
    // Emits the node for an already partitioned range into `arr` and builds both halves in place.
    void emit_node(const uint32_t beg[2], const uint32_t cnt[2], uint32_t depth, Sub& arr, Sub& stats) {
        const size_t self = arr.nodes.size();
        arr.nodes.resize(self + 16, 0.0f);
        uint32_t below = 0;
        for (int c = 0; c < 2; c++) {
            const Aabb cb = range_bounds(beg[c], cnt[c]);
            put_box(&arr.nodes[self], c, cb);
            stats.cost += cb.half_area() / root_area * (cnt[c] <= kLeafMax ? (double)cnt[c] : 1.0);
            if (cnt[c] <= kLeafMax) { put_child(&arr.nodes[self], c, leaf_ref(beg[c], cnt[c])); continue; }
            const int ref = (int)(arr.nodes.size() / 16);
            Sub sub_stats;
            build_sub_inplace(beg[c], cnt[c], depth + 1, arr, sub_stats);
            stats.cost += sub_stats.cost;
            below = std::max(below, sub_stats.depth);
            put_child(&arr.nodes[self], c, ref);
        }
        stats.depth = below + 1;
    }
    // Sequential variant that appends straight into `arr` (node indices are already final within it).
    void build_sub_inplace(uint32_t first, uint32_t count, uint32_t depth, Sub& arr, Sub& stats) {
        const uint32_t mid = partition(first, count, depth);
        const uint32_t cnt[2] = {mid - first, first + count - mid};
        const uint32_t beg[2] = {first, mid};
        emit_node(beg, cnt, depth, arr, stats);
    }
};

}  // namespace

// Collapse the binary tree into a 4-wide tree with QUANTISED child boxes: a node adopts its two
// children, then repeatedly replaces the inner child with the largest surface area by that child's two
// children until it has four (or only leaves are left). Traversal is bound by vector-L1 tag lookups
// (one per 16-byte gather per lane), so a node is packed into 64 bytes = 4 gathers per step:
//   [0]  origin.x origin.y origin.z | exponents ex | ey<<8 | ez<<16   (the node's own lower corner, per-axis 2^e grid)
//   [16] LX LY LZ HX   one dword per plane set: byte c = child c's plane, in grid steps from the origin
//   [32] HY HZ - -
//   [48] child[4]      >= 0: node index; < 0: leaf, ~child = (first_triangle << 3) | count
// Decoding (device): plane = fmaf(q, 2^e, origin). Quantisation rounds lower planes down and upper
// planes up and is verified here with the SAME fmaf, so a decoded box always contains the exact one.
// Unused children decode to an inverted box (lo byte 255, hi byte 0) and carry an empty-leaf reference.
namespace {
struct Child4 { float lo[3], hi[3]; int ref; };
inline Child4 child_of(const float* n2, int which) {
    Child4 c;
    if (which == 0) { c.lo[0] = n2[0]; c.hi[0] = n2[1]; c.lo[1] = n2[2]; c.hi[1] = n2[3]; c.lo[2] = n2[8]; c.hi[2] = n2[9]; }
    else { c.lo[0] = n2[4]; c.hi[0] = n2[5]; c.lo[1] = n2[6]; c.hi[1] = n2[7]; c.lo[2] = n2[10]; c.hi[2] = n2[11]; }
    memcpy(&c.ref, n2 + 12 + which, 4);
    return c;
}
inline float area_of(const Child4& c) {
    const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
    if (!(dx >= 0.0f)) return -1.0f;
    return dx * dy + dy * dz + dz * dx;
}
// Guard band around every quantised plane, in grid cells: the traversal evaluates plane distances as fma(q, 2^e*inv,
// (origin-o)*inv) (traverse.h), whose rounding must stay inside it. The lower plane of a child that touches the node's
// own minimum stays at q = 0: the node origin is nudged down instead.
constexpr double kGuardCells = 1.0 / 32.0;

struct Collapser {
    const std::vector<float>& n2;
    std::vector<uint32_t>& out;   // 16 dwords per node
    std::vector<uint8_t> height;  // of each binary inner node = stack entries a purely binary walk below it needs
    uint32_t max_depth = 0;
    uint32_t height_of(int node2) {
        uint32_t h = 0;
        for (int c = 0; c < 2; c++) {
            int ref; memcpy(&ref, &n2[(size_t)node2 * 16 + 12 + c], 4);
            if (ref >= 0) h = std::max(h, height_of(ref));
        }
        height[node2] = (uint8_t)(h + 1);
        return h + 1;
    }
    uint32_t need_of(const Child4& c) const { return c.ref >= 0 ? height[c.ref] : 0u; }
    // `budget` = stack entries a traversal may use below (and including) this node. A node with k
    // children can leave k-1 entries on the stack while a child is walked, so wide nodes are only
    // formed where the budget allows; the binary tree below always fits (height <= kMaxBinaryDepth).
    // Returns the worst-case number of entries actually needed.
    uint32_t emit(int node2, uint32_t depth, uint32_t budget, int* out_index) {
        constexpr int W = srl::kBvhWidth;
        Child4 kids[W];
        int nk = 0;
        kids[nk++] = child_of(&n2[(size_t)node2 * 16], 0);
        kids[nk++] = child_of(&n2[(size_t)node2 * 16], 1);
        while (nk < W) {
            int best = -1; float best_area = -1.0f;
            for (int i = 0; i < nk; i++) if (kids[i].ref >= 0) { const float a = area_of(kids[i]); if (a > best_area) { best_area = a; best = i; } }
            if (best < 0) break;
            const Child4 ca = child_of(&n2[(size_t)kids[best].ref * 16], 0), cb = child_of(&n2[(size_t)kids[best].ref * 16], 1);
            // with nk+1 children every child subtree must fit in budget - nk entries
            bool fits = need_of(ca) + (uint32_t)nk <= budget && need_of(cb) + (uint32_t)nk <= budget;
            for (int i = 0; i < nk && fits; i++) if (i != best && need_of(kids[i]) + (uint32_t)nk > budget) fits = false;
            if (!fits) break;
            kids[best] = ca;
            kids[nk++] = cb;
        }
        bool real[W];
        int n_real = 0;
        float lo_n[3] = {INFINITY, INFINITY, INFINITY}, hi_n[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < W; i++) real[i] = false;
        for (int i = 0; i < nk; i++) {
            const bool empty_leaf = kids[i].ref < 0 && ((~(uint32_t)kids[i].ref) & 7u) == 0u;
            real[i] = !empty_leaf;
            if (!real[i]) continue;
            n_real++;
            for (int a = 0; a < 3; a++) { lo_n[a] = std::min(lo_n[a], kids[i].lo[a]); hi_n[a] = std::max(hi_n[a], kids[i].hi[a]); }
        }
        const size_t self = out.size();
        *out_index = (int)(self / srl::kNodeDwords);
        out.resize(self + srl::kNodeDwords, 0u);
        max_depth = std::max(max_depth, depth);
        uint32_t plane[6 * srl::kPlaneDwords];   // LX LY LZ HX HY HZ, kPlaneDwords dwords each
        for (auto& v : plane) v = 0u;
        uint32_t exps = 0;
        float origin[3] = {0.0f, 0.0f, 0.0f};
        if (n_real > 0) {
            for (int a = 0; a < 3; a++) {
                origin[a] = std::nextafter(lo_n[a] - (hi_n[a] - lo_n[a]) * (1.0f / 2048.0f), -INFINITY);   // ~1/8 cell below the minimum
                const float ext = hi_n[a] - origin[a];
                int e = -126;
                if (ext > 0.0f && ext / 255.0f > 0.0f) { int fe; (void)std::frexp(ext / 255.0f, &fe); e = std::max(fe, -126); }   // 2^fe >= ext/255
                for (;;) {   // grow the grid until every child's upper plane fits in a byte
                    const float scale = std::ldexp(1.0f, e);
                    bool ok = true;
                    for (int i = 0; i < nk && ok; i++) {
                        if (!real[i]) continue;
                        int qh = (int)std::ceil(((double)kids[i].hi[a] - (double)origin[a]) / (double)scale + kGuardCells);
                        qh = std::max(qh, 0);
                        while (qh <= 255 && fmaf((float)qh, scale, origin[a]) < kids[i].hi[a]) qh++;
                        if (qh > 255) ok = false;
                    }
                    if (ok) break;
                    e++;
                }
                const float scale = std::ldexp(1.0f, e);
                exps |= (uint32_t)(e + 127) << (8 * a);
                for (int i = 0; i < W; i++) {
                    uint32_t ql = 255u, qh = 0u;   // inverted box for unused children
                    if (i < nk && real[i]) {
                        int l = (int)std::floor(((double)kids[i].lo[a] - (double)origin[a]) / (double)scale - kGuardCells);
                        l = std::min(std::max(l, 0), 255);
                        while (l > 0 && fmaf((float)l, scale, origin[a]) > kids[i].lo[a]) l--;
                        int h = (int)std::ceil(((double)kids[i].hi[a] - (double)origin[a]) / (double)scale + kGuardCells);
                        h = std::max(h, 0);
                        while (fmaf((float)h, scale, origin[a]) < kids[i].hi[a]) h++;
                        ql = (uint32_t)l; qh = (uint32_t)h;
                    }
                    plane[a * srl::kPlaneDwords + i / 4] |= ql << (8 * (i % 4));
                    plane[(3 + a) * srl::kPlaneDwords + i / 4] |= qh << (8 * (i % 4));
                }
            }
        } else {
            for (int a = 0; a < 3; a++) {
                for (int d = 0; d < srl::kPlaneDwords; d++) { plane[a * srl::kPlaneDwords + d] = 0xFFFFFFFFu; plane[(3 + a) * srl::kPlaneDwords + d] = 0u; }
                exps |= 127u << (8 * a);
            }
        }
        uint32_t below = 0;
        int refs[W];
        for (int i = 0; i < W; i++) {
            refs[i] = Builder::leaf_ref(0, 0);
            if (i < nk && real[i]) {
                if (kids[i].ref >= 0) {
                    int idx = 0;
                    below = std::max(below, emit(kids[i].ref, depth + 1, budget - (uint32_t)(n_real - 1), &idx));
                    refs[i] = idx;
                } else refs[i] = kids[i].ref;
            }
        }
        uint32_t* q = &out[self];
        memcpy(q + 0, origin, 12); q[3] = exps;
        memcpy(q + srl::kPlaneOffset, plane, sizeof(plane));
        memcpy(q + srl::kChildOffset, refs, sizeof(refs));
        return below + (uint32_t)std::max(n_real - 1, 0);
    }
};
}  // namespace

static void collapse_to_bvh4(BvhResult& res, uint32_t stack_budget) {
    res.nodes.clear();
    Collapser c{res.nodes2, res.nodes, {}, 0};
    c.height.assign(res.nodes2.size() / 16, 0);
    c.height_of(0);
    int root = 0;
    res.max_stack = c.emit(0, 1, stack_budget, &root);
    res.max_depth = c.max_depth;
    res.n_nodes = (uint32_t)(res.nodes.size() / srl::kNodeDwords);
}

void flatten_instances(const std::vector<HostMesh>& meshes, const FrameInstanceData& fid, std::vector<BuildTri>& out) {
    out.clear();
    out.reserve(fid.n_triangles);
    for (uint32_t ii = 0; ii < fid.instances.size(); ii++) {
        const HostInstance& inst = fid.instances[ii];
        const HostMesh& mesh = meshes[inst.mesh_slot];
        const float* m = inst.o2w.m;
        const uint32_t nprim = mesh.n_indices / 3;
        for (uint32_t p = 0; p < nprim; p++) {
            float w[3][3];
            for (int j = 0; j < 3; j++) {
                const float* q = mesh.vertices[mesh.indices[3 * p + j]].position;
                // transform_point (rt_utils.slang:278-281): rows dotted with (p, 1), left to right
                w[j][0] = ((m[0] * q[0] + m[1] * q[1]) + m[2] * q[2]) + m[3] * 1.0f;
                w[j][1] = ((m[4] * q[0] + m[5] * q[1]) + m[6] * q[2]) + m[7] * 1.0f;
                w[j][2] = ((m[8] * q[0] + m[9] * q[1]) + m[10] * q[2]) + m[11] * 1.0f;
            }
            BuildTri t;
            for (int a = 0; a < 3; a++) { t.v0[a] = w[0][a]; t.e1[a] = w[1][a] - w[0][a]; t.e2[a] = w[2][a] - w[0][a]; }
            t.prim = p; t.inst = ii; t.gid = inst.tri_offset + p;
            out.push_back(t);
        }
    }
}

namespace {
// Tree over the boxes already in b.bounds / b.centroid / b.order: binned-SAH binary tree, then the 4-wide collapse.
void build_core(Builder& b, uint32_t n, const Aabb& root, uint32_t max_depth, BvhResult& res) {
    b.root_area = std::max((double)root.half_area(), 1e-30);
    Builder::Sub top;
    if (n <= kLeafMax) {
        // The root must be an inner node: one leaf child + one empty child whose box nothing can hit.
        top.nodes.assign(16, 0.0f);
        if (n > 0) Builder::put_box(top.nodes.data(), 0, root);
        else { float* q = top.nodes.data(); q[0] = q[2] = q[8] = INFINITY; q[1] = q[3] = q[9] = -INFINITY; }
        { float* q = top.nodes.data(); q[4] = q[6] = q[10] = INFINITY; q[5] = q[7] = q[11] = -INFINITY; }
        Builder::put_child(top.nodes.data(), 0, Builder::leaf_ref(0, n));
        Builder::put_child(top.nodes.data(), 1, Builder::leaf_ref(0, 0));
        top.depth = 1;
        top.cost = (double)n;
    } else {
        b.build_sub(0, n, 0, top);
    }
    res.nodes2.swap(top.nodes);
    res.sah_cost = (float)top.cost;
    collapse_to_bvh4(res, max_depth);
    res.order = b.order;
}
}  // namespace

void build_bvh(const std::vector<BuildTri>& tris, uint32_t max_depth, BvhResult& res) {
    const auto t_start = std::chrono::steady_clock::now();
    const uint32_t n = (uint32_t)tris.size();
    Builder b(max_depth);
    b.bounds.resize(n);
    for (int a = 0; a < 3; a++) b.centroid[a].resize(n);
    b.order.resize(n);
    Aabb root; root.reset();
    for (uint32_t i = 0; i < n; i++) {
        const BuildTri& t = tris[i];
        float p1[3], p2[3];
        for (int a = 0; a < 3; a++) { p1[a] = t.v0[a] + t.e1[a]; p2[a] = t.v0[a] + t.e2[a]; }
        Aabb bb; bb.reset(); bb.add(t.v0); bb.add(p1); bb.add(p2);
        // The triangle test accepts barycentrics up to kBaryEps (1e-6) outside the triangle: bound that
        // fattened triangle, with margin for the test's own rounding (traverse.h intersect_tri).
        for (int a = 0; a < 3; a++) {
            const float pad = 4e-6f * (std::fabs(t.e1[a]) + std::fabs(t.e2[a]));
            bb.lo[a] -= pad; bb.hi[a] += pad;
        }
        b.bounds[i] = bb;
        for (int a = 0; a < 3; a++) b.centroid[a][i] = 0.5f * bb.lo[a] + 0.5f * bb.hi[a];
        b.order[i] = i;
        root.add(bb);
    }
    build_core(b, n, root, max_depth, res);
    res.tris.resize((size_t)n * 12);
    for (uint32_t i = 0; i < n; i++) {
        const BuildTri& t = tris[b.order[i]];
        float* q = &res.tris[(size_t)i * 12];
        q[0] = t.v0[0]; q[1] = t.v0[1]; q[2] = t.v0[2]; q[3] = t.e1[0];
        q[4] = t.e1[1]; q[5] = t.e1[2]; q[6] = t.e2[0]; q[7] = t.e2[1];
        q[8] = t.e2[2];
        memcpy(q + 9, &t.gid, 4);
        q[10] = 0.0f; q[11] = 0.0f;
    }
    res.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
}

// The same builder over arbitrary boxes — the instance boxes of a top-level tree (tlas.rs:100-191): leaf references index
// res.order (leaf position -> box index). The boxes are taken as they are (the caller pads them).
void build_bvh_boxes(const std::vector<BuildBox>& boxes, uint32_t max_depth, BvhResult& res) {
    const auto t_start = std::chrono::steady_clock::now();
    const uint32_t n = (uint32_t)boxes.size();
    Builder b(max_depth);
    b.bounds.resize(n);
    for (int a = 0; a < 3; a++) b.centroid[a].resize(n);
    b.order.resize(n);
    Aabb root; root.reset();
    for (uint32_t i = 0; i < n; i++) {
        Aabb bb;
        for (int a = 0; a < 3; a++) { bb.lo[a] = boxes[i].lo[a]; bb.hi[a] = boxes[i].hi[a]; }
        b.bounds[i] = bb;
        for (int a = 0; a < 3; a++) b.centroid[a][i] = 0.5f * bb.lo[a] + 0.5f * bb.hi[a];
        b.order[i] = i;
        root.add(bb);
    }
    build_core(b, n, root, max_depth, res);
    res.tris.clear();
    res.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
}

}  // namespace srh
