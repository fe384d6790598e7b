// HIP kernels of the ray-tracing hot path for gfx950 (MI355X).
//
//   trace_rays_kernel<ANY>                   — ray-list tracers, one wave per 64 rays (K2 / K3)
//   shade_closest_hit_kernel                 — closest_hit.slang / ray_miss.slang on hit records (K4/K6)
//   any_hit_kernel                           — any_hit.slang's alpha test on hit records (K5; never part of a traversal)
//   ris_kernel                               — ray_gen_ris.slang:12-440   (K1, K7, K8, K9)
//   final_kernel                             — ray_gen_final.slang:11-436 (K1, K10)
//
// Launch geometry: the ray-list and shade kernels use 256-thread workgroups; the two per-pixel passes use ONE WAVE per
// workgroup, owning an 8x8 pixel tile (coherent primary rays per wave; a 4-wave workgroup would hold its LDS until its
// slowest wave finished). Workgroups are dealt to screen tiles XCD-aware: blocks b and b+8 share an XCD (round-robin
// dispatch), XCD x owns column band x of the image, so its private 4 MiB L2 holds the part of the BVH under that band;
// inside a band the rows are swept from the expensive end to the cheap end (thread_pixel, tile_order_kernel).
#include <algorithm>
#include "kernels.h"

namespace srd {

constexpr int kBlock = 256;        // ray-queue and shade kernels
constexpr int kPassBlock = 64;     // the two pass megakernels: one wave = one workgroup = one 8x8 pixel tile
constexpr int kPassTile = 8;
static int lds_extra_rows(int two_level) { return two_level ? kWsRowsTl : kWsRows; }   // LDS rows of the work-stealing traversal in front of the stack levels
#ifndef SR_PASS_WAVES
#define SR_PASS_WAVES 4
#endif
constexpr int kPassWaves = SR_PASS_WAVES;      // __launch_bounds__ second argument on HIP: waves per SIMD (4 -> VGPR budget 128)
// The two-level variants (V bit 2) carry ~30 more live registers (instance, object-space ray, padding): at the 128-register budget
// of 4 waves per SIMD final_kernel<6> spills 35-43 of them, and with that much scratch traffic in the stealing loop its results
// depended on unrelated code changes (a no-op edit of the node loop broke test_two_level_passes_equal_oracle[small_atrium]; the
// same source at 3 waves per SIMD, 160 registers, no spills, passes). They run at 3 waves per SIMD.
#ifndef SR_TL_PASS_WAVES
#define SR_TL_PASS_WAVES 3
#endif
constexpr int pass_waves(int v) { return (v & 4) ? (kPassWaves < SR_TL_PASS_WAVES ? kPassWaves : SR_TL_PASS_WAVES) : kPassWaves; }

// Wave-wide sum, then one atomic per wave (rays are counted, not estimated: SURVEY.md §8d).
SRD void flush_counter(unsigned long long* counters, int which, uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(counters + (blockIdx.x & (kCounterSlots - 1u)) * kCounterStride + which, (unsigned long long)v);
}

// ---------------------------------------------------------------------------------------------
// Ray-list tracers: one wave per 64 consecutive rays, four waves to a workgroup, workgroups in list order; the hardware's workgroup
// dispatcher is the load balancer. (Giving each XCD one contiguous eighth of the list instead — neighbouring rays then share nodes in
// one L2 — lost on every ray set tried: primary rays 0.36 ms against 0.24, the eighths differ too much in cost.)
// (Rounds 1-2 ran a machine-sized grid pulling 64 rays per atomicAdd on a queue head: adds of all waves on one address complete one
// every ~14 ns, which capped 2 M rays at 0.47 ms whatever the rays did — DESIGN.md section 5.)
// ---------------------------------------------------------------------------------------------
template <bool ANY, bool STATS, bool TL>
__global__ __launch_bounds__(kBlock) void trace_rays_kernel(DevScene sc, const SrRay* __restrict__ rays, uint32_t n,
                                                            SrHit* __restrict__ hits, uint32_t* __restrict__ occluded) {
    extern __shared__ __attribute__((aligned(16))) int s_stack[];   // [stack_entries][kBlock], sized at launch
    const int lane = threadIdx.x & 63;
    int* stack = s_stack + threadIdx.x;
    uint32_t n_queries = 0;
    TravStats st; st.boxes = 0; st.tris = 0;
    const uint32_t base = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 64u;
    if (base < n) {                                                 // wave-uniform
        const uint32_t i = base + lane;
        const uint32_t ic = i < n ? i : n - 1u;
        const float4 ra = reinterpret_cast<const float4*>(rays)[(size_t)ic * 2 + 0];
        const float4 rb = reinterpret_cast<const float4*>(rays)[(size_t)ic * 2 + 1];
        TravHit h;
        const bool found = traverse_ws<ANY, STATS, TL>(sc, i < n, mk3(ra.x, ra.y, ra.z), mk3(rb.x, rb.y, rb.z), ra.w, rb.w, h, stack, kBlock, st);
        if (i < n) {
            n_queries++;
            if (ANY) occluded[i] = found ? 1u : 0u;
            else {
                float4 o;
                o.x = h.t; o.y = h.u; o.z = h.v; o.w = __uint_as_float(h.gid);
                reinterpret_cast<float4*>(hits)[i] = o;
            }
        }
    }
    flush_counter(sc.counters, (ANY ? 1 : 0), n_queries);
    if (STATS) { flush_counter(sc.counters, 2, st.boxes); flush_counter(sc.counters, 3, st.tris); }
}

__global__ __launch_bounds__(kBlock) void shade_closest_hit_kernel(DevScene sc, const SrHit* __restrict__ hits, uint32_t n,
                                                                   SrRayPayload* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const SrHit h = hits[i];
    TravHit th;
    th.t = h.t; th.u = h.u; th.v = h.v; th.gid = (h.t < 0.0f || h.tri >= sc.n_tris) ? 0xFFFFFFFFu : h.tri;
    th.slot = 0u; th.inst = 0u;
    const bool two_level = sc.tl_instances != nullptr;
    if (th.gid != 0xFFFFFFFFu) { if (two_level) tl_locate(sc, th.gid, th.slot, th.inst); else th.slot = sc.slot_of_gid[th.gid]; }
    const Payload p = two_level ? shade_hit<true, true>(sc, th) : shade_hit<true, false>(sc, th);
    SrRayPayload o;
    o.emission[0] = p.emission.x; o.emission[1] = p.emission.y; o.emission[2] = p.emission.z;
    o.dist = p.dist; o.albedo_packed = p.albedo_packed; o.normal_packed = p.normal_packed;
    o.material_info = p.material_info; o.transmission_ior_packed = p.transmission_ior_packed;
    out[i] = o;
}

__global__ __launch_bounds__(kBlock) void any_hit_kernel(DevScene sc, const SrHit* __restrict__ hits, uint32_t n, uint32_t* __restrict__ ignored) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const SrHit h = hits[i];
    const bool hit = !(h.t < 0.0f) && h.tri < sc.n_tris;
    uint32_t slot = 0u, inst = 0u;
    if (hit) { if (sc.tl_instances) tl_locate(sc, h.tri, slot, inst); else slot = sc.slot_of_gid[h.tri]; }
    ignored[i] = (hit && any_hit_ignores(sc, slot, h.u, h.v)) ? 1u : 0u;
}

// ---------------------------------------------------------------------------------------------
// Per-pixel passes
// ---------------------------------------------------------------------------------------------
struct Row4 { float x, y, z, w; };
SRD float dot4(const float* r, float x, float y, float z, float w) { return ((r[0] * x + r[1] * y) + r[2] * z) + r[3] * w; }

// K1: primary ray of pixel (px,py) (ray_gen_ris.slang:44-53 == ray_gen_final.slang:58-67)
SRD void primary_ray(const SrMatrices& m, uint32_t px, uint32_t py, uint32_t W, uint32_t H, f3& origin, f3& dir, f2& inUV) {
    const float cx = (float)px + 0.5f, cy = (float)py + 0.5f;
    inUV.x = cx / (float)W; inUV.y = cy / (float)H;
    const float dx = inUV.x * 2.0f - 1.0f, dy = inUV.y * 2.0f - 1.0f;
    const float* vi = m.view_inverse;
    const float* pi = m.proj_inverse;
    origin = mk3(dot4(vi + 0, 0.0f, 0.0f, 0.0f, 1.0f), dot4(vi + 4, 0.0f, 0.0f, 0.0f, 1.0f), dot4(vi + 8, 0.0f, 0.0f, 0.0f, 1.0f));
    const f3 target = mk3(dot4(pi + 0, dx, dy, 1.0f, 1.0f), dot4(pi + 4, dx, dy, 1.0f, 1.0f), dot4(pi + 8, dx, dy, 1.0f, 1.0f));
    const f3 tn = norm3(target);
    dir = mk3(dot4(vi + 0, tn.x, tn.y, tn.z, 0.0f), dot4(vi + 4, tn.x, tn.y, tn.z, 0.0f), dot4(vi + 8, tn.x, tn.y, tn.z, 0.0f));
}

struct PixelCtx {
    const PassArgs& a;
    int* stack;
    // Query counts of the lane's pixel in ONE register (the passes sit at their register budget; a register per counter showed up as
    // 20 % of final_kernel): bits 0-15 existence queries, 16-30 closest-hit queries, bit 31 = the one existence query of the
    // reference that was answered from an identical query of the same pixel (kCountReusedAny). The bounds that keep the fields
    // apart (max_bounces, virtual_bounces <= SR_MAX_BOUNCES) are checked by the host before the launch.
    uint32_t n_q;
    TravStats st;
};
constexpr uint32_t kCountAny = 1u, kCountClosest = 1u << 16, kCountReusedAny = 1u << 31;
SRD uint32_t counted_any(uint32_t n_q) { return n_q & 0xFFFFu; }
SRD uint32_t counted_closest(uint32_t n_q) { return (n_q >> 16) & 0x7FFFu; }

// TraceRay of the passes: V bit 0 = traversal statistics, V bit 2 = the two-level form of the structure
template <int V, bool ANY>
SRD bool trace_ray(PixelCtx& cx, bool want, f3 o, f3 d, float tmin, float tmax, TravHit& h) {
    return traverse_ws<ANY, (V & 1) != 0, (V & 4) != 0>(cx.a.sc, want, o, d, tmin, tmax, h, cx.stack, kPassBlock, cx.st);
}
template <int V>
SRD Payload trace_closest_shaded(PixelCtx& cx, f3 o, f3 d, float tmin, float tmax) {
    TravHit h;
    trace_ray<V, false>(cx, true, o, d, tmin, tmax, h);
    cx.n_q += kCountClosest;
    return shade_hit<(V & 2) != 0, (V & 4) != 0>(cx.a.sc, h);
}
// The shadow-ray idiom of every visibility query: prd.dist preset to 1.0, the miss shader writes -1;
// segments <= 0.002 are not traced and count as visible. Returns the resulting prd.dist.
template <int V>
SRD float trace_shadow(PixelCtx& cx, f3 o, f3 d, float dist) {
    if (dist > 0.002f) {
        TravHit h;
        cx.n_q += kCountAny;
        return trace_ray<V, true>(cx, true, o, d, 0.001f, dist - 0.001f, h) ? 1.0f : -1.0f;
    }
    return -1.0f;
}

SRD void store_gbuffer(const PassArgs& a, uint32_t pix, float depth, f3 n, float rough, f3 diffuse, float mx, float my) {
    a.depth_img[pix] = (uint16_t)f32_to_f16_bits(depth);
    a.normal_img[pix] = pack_rgba8_snorm(n.x, n.y, n.z, rough);
    a.diffuse_img[pix] = pack_b10g11r11(diffuse.x, diffuse.y, diffuse.z);
    a.motion_vec_img[pix] = pack_half_2x16(mx, my);
}
SRD f3 load_normal(const PassArgs& a, uint32_t pix) {
    const uint32_t v = a.normal_img[pix];
    return mk3(unsnorm8(v), unsnorm8(v >> 8), unsnorm8(v >> 16));
}
SRD float load_depth(const PassArgs& a, uint32_t pix) { return f16_bits_to_f32(a.depth_img[pix]); }

SRD void zero_reservoir(SrReservoir& r) {
    r.light_pos[0] = r.light_pos[1] = r.light_pos[2] = 0.0f; r.w_sum = 0.0f;
    r.light_normal[0] = r.light_normal[1] = r.light_normal[2] = 0.0f; r.M = 0.0f;
    r.light_idx = 0u; r.W = 0.0f; r.hit_normal_packed = 0u; r.depth = 0.0f;
}
SRD void zero_reservoir_gi(SrReservoirGI& r) {
    r.sample_pos[0] = r.sample_pos[1] = r.sample_pos[2] = 0.0f; r.w_sum = 0.0f;
    r.sample_radiance[0] = r.sample_radiance[1] = r.sample_radiance[2] = 0.0f; r.M = 0.0f;
    r.sample_normal_packed = 0u; r.W = 0.0f; r.hit_normal_packed = 0u; r.depth = 0.0f;
}
// 48-byte reservoir records move as three 16-byte accesses per lane
template <typename T>
SRD T load48(const T* p) {
    T r;
    const float4* s = reinterpret_cast<const float4*>(p);
    float4* d = reinterpret_cast<float4*>(&r);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    return r;
}
template <typename T>
SRD void store48(T* p, const T& v) {
    float4* d = reinterpret_cast<float4*>(p);
    const float4* s = reinterpret_cast<const float4*>(&v);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

// 32-byte RayPayload records of the primary-hit hand-off: two 16-byte accesses per lane
SRD void store_payload(SrRayPayload* p, const Payload& v) {
    float4* d = reinterpret_cast<float4*>(p);
    d[0] = make_float4(v.emission.x, v.emission.y, v.emission.z, v.dist);
    d[1] = make_float4(__uint_as_float(v.albedo_packed), __uint_as_float(v.normal_packed), __uint_as_float(v.material_info), __uint_as_float(v.transmission_ior_packed));
}
SRD Payload load_payload(const SrRayPayload* p) {
    const float4* s = reinterpret_cast<const float4*>(p);
    const float4 a = s[0], b = s[1];
    Payload v;
    v.emission = mk3(a.x, a.y, a.z); v.dist = a.w;
    v.albedo_packed = __float_as_uint(b.x); v.normal_packed = __float_as_uint(b.y);
    v.material_info = __float_as_uint(b.z); v.transmission_ior_packed = __float_as_uint(b.w);
    return v;
}

SRD f3 light_emission(const DevScene& sc, uint32_t indirection_idx) { return ld3(sc.lights[indirection_idx].emission); }
// Emissive triangle `idx` of the light list (ray_gen_ris.slang:191-210,346-362; ray_gen_final.slang:330-351):
// world-space vertices, area and normal come precomputed from the scene build (DevLight). The random
// draws stay at the call sites: their order is semantic.
struct LightTri { f3 wv0, wv1, wv2, emission, normal; float area; };
SRD LightTri fetch_light(const DevScene& sc, uint32_t idx) {
    const float4* q = reinterpret_cast<const float4*>(sc.lights + idx);
    const float4 a = q[0], b = q[1], c = q[2], e = q[3];
    LightTri r;
    r.wv0 = mk3(a.x, a.y, a.z); r.area = a.w;
    r.wv1 = mk3(b.x, b.y, b.z);
    r.wv2 = mk3(c.x, c.y, c.z);
    r.emission = mk3(e.x, e.y, e.z);
    r.normal = mk3(b.w, c.w, e.w);
    return r;
}

// Map this thread to its pixel. Returns false for threads outside the image / tile.
// XCD-aware and load-balanced: the image is cut into 8 COLUMN bands, one per XCD (blocks b and b+8
// share an XCD under round-robin dispatch), walked row by row inside the band. Cost varies mostly with
// image row (distance to the terrain), so column bands give every XCD the same mix of rows, while the
// tiles an XCD works on at any moment stay neighbours and share its 4 MiB L2.
// Both the band boundaries and the order of a band's tiles come from tile_order (tile_order_kernel, derived from the
// measured costs of an earlier launch): bands of equal summed cost instead of equal width (the image centre is more
// expensive than its edges: equal widths left the two outer XCDs 30 % short of work), and each band swept from its
// expensive end to its cheap end, so the launch drains with cheap tiles instead of ending on a front of expensive
// ones (a frame's cost is spatially correlated: sky rows finish in a fraction of the time of terrain rows).
// Scheduling only — results do not depend on it. tile_order holds, per XCD, `order_cap` absolute tile indices
// (ty * tiles_x + tx; 0xFFFFFFFF past the end of the XCD's list); cost slots are absolute tile indices.
SRD bool thread_pixel(const PassArgs& a, uint32_t& px, uint32_t& py, uint32_t& cost_slot) {
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u, k = b >> 3;
    cost_slot = 0xFFFFFFFFu;
    uint32_t tile;
    if (a.tile_order) {
        tile = a.tile_order[xcd * a.order_cap + k];                                         // k < order_cap by the grid size
    } else {                                                                                // no measured costs yet: equal widths, top to bottom
        const uint32_t bx0 = (a.tiles_x * xcd) >> 3, bx1 = (a.tiles_x * (xcd + 1u)) >> 3;
        const uint32_t bw = bx1 - bx0;
        if (bw == 0u || k >= bw * a.tiles_y) return false;
        tile = (k / bw) * a.tiles_x + bx0 + k % bw;
    }
    if (tile >= a.tiles_x * a.tiles_y) return false;
    cost_slot = tile;
    const uint32_t tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const uint32_t l = threadIdx.x;
    px = a.x0 + tx * (uint32_t)kPassTile + (l & 7u);
    py = a.y0 + ty * (uint32_t)kPassTile + (l >> 3);
    return px < a.x1 && py < a.y1;
}
SRD void record_tile_cost(const PassArgs& a, uint32_t cost_slot, unsigned long long t_start) {
    if (a.tile_cost && cost_slot != 0xFFFFFFFFu && threadIdx.x == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_start;
        a.tile_cost[cost_slot] = (uint32_t)(dt > 0xFFFFFFFFull ? 0xFFFFFFFFull : dt);
    }
}

// Widest band the schedule may form, in tile columns (the launch is sized for it): 1.5x the equal share.
SRD uint32_t band_cap_cols_dev(uint32_t tiles_x) { return min(tiles_x, (tiles_x * 3u) / 16u + 2u); }

// One workgroup per XCD band. Every workgroup sums the measured cost per tile column, thread 0 cuts the columns into 8
// bands of (nearly) equal cost (each at most band_cap_cols wide), then the workgroup writes its band's tile list: swept
// row by row from the band's expensive end to its cheap end (first quarter of rows vs last quarter), tiles of a row in
// x order. A monotone sweep keeps concurrently running tiles adjacent (shared BVH nodes in the XCD's L2) and still
// ends the launch on cheap tiles. Measured against the fixed top-to-bottom order on the bench frame: sweep from the
// expensive end +8 %, rows ranked by cost +6 %, tiles ranked by cost +5 %; cost-balanced bands: see DESIGN.md.
constexpr uint32_t kMaxScheduleCols = 1024;   // 8192-pixel-wide images; wider ones keep equal-width bands
__global__ void tile_order_kernel(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t tiles_x, uint32_t tiles_y, uint32_t order_cap) {
    __shared__ unsigned long long s_col[kMaxScheduleCols];
    __shared__ uint32_t s_b[9];
    __shared__ unsigned long long s_sum[2];
    const uint32_t band = blockIdx.x;
    const uint32_t cap_cols = band_cap_cols_dev(tiles_x);
    const bool balance = tiles_x <= kMaxScheduleCols && tiles_x >= 16u;
    if (balance) {
        for (uint32_t x = threadIdx.x; x < tiles_x; x += blockDim.x) {
            unsigned long long c = 0;
            for (uint32_t y = 0; y < tiles_y; y++) c += cost[(size_t)y * tiles_x + x];
            s_col[x] = c + 1ull;                                   // + 1: columns of an unmeasured image still spread evenly
        }
    }
    if (threadIdx.x < 2) s_sum[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        s_b[0] = 0u; s_b[8] = tiles_x;
        if (balance) {
            unsigned long long total = 0;
            for (uint32_t x = 0; x < tiles_x; x++) total += s_col[x];
            unsigned long long run = 0;
            uint32_t x = 0;
            for (uint32_t i = 1; i < 8u; i++) {
                const unsigned long long target = total / 8ull * i;
                while (x < tiles_x && run + s_col[x] / 2ull < target) run += s_col[x++];    // the column goes to the side its centre lies on
                uint32_t cut = x;
                cut = max(cut, s_b[i - 1] + 1u);                                            // at least one column per band
                cut = min(cut, s_b[i - 1] + cap_cols);                                      // at most cap_cols
                cut = max(cut, tiles_x > (8u - i) * cap_cols ? tiles_x - (8u - i) * cap_cols : 0u);   // the bands that are left can cover the rest
                cut = min(cut, tiles_x - (8u - i));                                         // ... and each gets a column
                while (x < cut) run += s_col[x++];
                s_b[i] = cut;
            }
        } else {
            for (uint32_t i = 1; i < 8u; i++) s_b[i] = (tiles_x * i) >> 3;
        }
    }
    __syncthreads();
    const uint32_t bx0 = s_b[band], bx1 = s_b[band + 1];
    const uint32_t bw = bx1 - bx0, n = bw * tiles_y;
    uint32_t* const mine = order + (size_t)band * order_cap;
    const uint32_t q = max(tiles_y / 4u, 1u) * bw;      // tiles in a quarter of the rows
    unsigned long long head = 0, tail = 0;
    for (uint32_t i = threadIdx.x; i < q; i += blockDim.x) {
        const uint32_t r = i / bw, x = bx0 + i % bw;
        head += cost[(size_t)r * tiles_x + x];
        tail += cost[(size_t)(tiles_y - 1u - r) * tiles_x + x];
    }
    atomicAdd(&s_sum[0], head);
    atomicAdd(&s_sum[1], tail);
    __syncthreads();
    const bool bottom_up = s_sum[1] > s_sum[0];
    // Sweep order of the band's rows: the longest waves of a frame sit in the middle of the image (rays that graze the terrain
    // near the horizon: 250-400 us, a quarter of a launch), and a launch that reaches them late drains for as long as they
    // last. So the sweep starts a few rows beyond the row with the longest wave, runs to the band's expensive end, and then
    // takes the remaining rows from there to the cheap end (sky): two monotone sweeps, neighbours in time stay neighbours
    // in space (+0.7 % over one sweep from the expensive end).
    __shared__ unsigned long long s_key[1024];
    __shared__ uint16_t s_row_at[1024];
    if (tiles_y <= 1024u) {
        for (uint32_t r = threadIdx.x; r < tiles_y; r += blockDim.x) {
            unsigned long long k = 0;
            for (uint32_t x = 0; x < bw; x++) k = max(k, (unsigned long long)cost[(size_t)r * tiles_x + bx0 + x]);
            s_key[r] = k;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t rs = 0;
            for (uint32_t r = 1; r < tiles_y; r++) if (s_key[r] > s_key[rs]) rs = r;
            rs = bottom_up ? (rs > 4u ? rs - 4u : 0u) : min(rs + 4u, tiles_y - 1u);
            uint32_t at = 0;
            if (bottom_up) { for (uint32_t r = rs; r < tiles_y; r++) s_row_at[at++] = (uint16_t)r; for (uint32_t r = rs; r-- > 0;) s_row_at[at++] = (uint16_t)r; }
            else { for (uint32_t r = rs + 1; r-- > 0;) s_row_at[at++] = (uint16_t)r; for (uint32_t r = rs + 1; r < tiles_y; r++) s_row_at[at++] = (uint16_t)r; }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < order_cap; i += blockDim.x) {
            uint32_t t = 0xFFFFFFFFu;
            if (i < n) t = (uint32_t)s_row_at[i / bw] * tiles_x + bx0 + i % bw;
            mine[i] = t;
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < order_cap; i += blockDim.x) {
        uint32_t t = 0xFFFFFFFFu;
        if (i < n) {
            const uint32_t r = i / bw, x = i % bw;
            t = (bottom_up ? tiles_y - 1u - r : r) * tiles_x + bx0 + x;
        }
        mine[i] = t;
    }
}

// One query of the flattened passes: reached by every lane of the wave, `want` = this lane has a ray (counted as the
// TraceRay it stands for). Returns found / occluded for the lane's own ray.
template <int V, bool ANY>
SRD bool ws_query(PixelCtx& cx, bool want, f3 o, f3 d, float tmin, float tmax, TravHit& h) {
    if (want) cx.n_q += ANY ? kCountAny : kCountClosest;
    return trace_ray<V, ANY>(cx, want, o, d, tmin, tmax, h);
}

template <int V>
__global__ __launch_bounds__(kPassBlock, pass_waves(V)) void ris_kernel(const PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) int s_stack[];   // [stack_entries][kPassBlock], sized at launch
    PixelCtx cx{a, s_stack + threadIdx.x, 0u, {0u, 0u}};
    const DevScene& sc = a.sc;
    uint32_t px = 0, py = 0, cost_slot = 0;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    const bool active = thread_pixel(a, px, py, cost_slot);
    if (active) {
        const uint32_t W = a.width, H = a.height;
        const uint32_t pix = py * W + px;
        const uint32_t cur_buf = a.frame_count & 1u, hist_buf = cur_buf ^ 1u;
        SrReservoir* reservoir_cur = a.reservoirs[cur_buf];
        const SrReservoir* reservoir_hist = a.reservoirs[hist_buf];
        SrReservoirGI* reservoir_gi_cur = a.reservoirs_gi[cur_buf];
        const SrReservoirGI* reservoir_gi_hist = a.reservoirs_gi[hist_buf];
        const float* pvp = a.mats.prev_view_proj;

        uint32_t rng = init_rng(px, py, a.frame_count, W);
        f3 origin, direction; f2 inUV;
        primary_ray(a.mats, px, py, W, H, origin, direction, inUV);
        f3 rayOrigin = origin, rayDir = direction;

        Payload prd;
        f3 hitPos = splat(0.0f), hit_normal = splat(0.0f), hit_albedo = splat(0.0f);
        float roughness = 0.5f, metallic = 0.0f;
        f3 V_view = splat(0.0f);
        float prev_u = -1.0f, prev_v = -1.0f;
        bool prev_valid = false, found_diffuse_surface = false;
        float virtual_distance = 0.0f;

        for (uint32_t vb = 0; vb < a.cfg.virtual_bounces; vb++) {
            prd = trace_closest_shaded<V>(cx, rayOrigin, rayDir, 0.001f, 10000.0f);
            // the camera ray's payload, for the final pass: ray_gen_final.slang:80 at bounce 0 is this very query
            if (vb == 0u && a.primary_payload) store_payload(a.primary_payload + pix, prd);
            if (prd.dist < 0.0f) break;
            hitPos = rayOrigin + rayDir * prd.dist;
            hit_normal = unpack_normal(prd.normal_packed);
            hit_albedo = unpack_unorm_rgb(prd.albedo_packed);
            const f2 mat_info = unpack_half_2x16(prd.material_info);
            roughness = fmaxf(mat_info.x, 0.01f);
            metallic = clampf(mat_info.y, 0.0f, 1.0f);
            const f2 trans_ior = unpack_half_2x16(prd.transmission_ior_packed);
            const float transmission = trans_ior.x;
            V_view = -rayDir;
            virtual_distance += prd.dist;
            if (transmission > 0.5f) {
                const float ior = fmaxf(trans_ior.y, 1.0f);
                const bool is_inside = dot3(rayDir, hit_normal) > 0.0f;
                const f3 N = is_inside ? -hit_normal : hit_normal;
                const float eta = is_inside ? (ior / 1.0f) : (1.0f / ior);
                const float cos_theta = fminf(dot3(-rayDir, N), 1.0f);
                float R0 = (1.0f - eta) / (1.0f + eta);
                R0 = R0 * R0;
                float fresnel = R0 + (1.0f - R0) * pow5f(1.0f - cos_theta);
                const f3 refracted = refract3(rayDir, N, eta);
                if (len3(refracted) < 0.01f) fresnel = 1.0f;
                if (rnd(rng) < fresnel) rayDir = reflect3(rayDir, N);
                else rayDir = refracted;
                rayOrigin = hitPos + rayDir * 0.001f;
            } else if (metallic > 0.9f && roughness < 0.1f) {
                rayOrigin = hitPos + hit_normal * 0.001f;
                rayDir = reflect3(rayDir, hit_normal);
            } else {
                const f3 vwp = origin + direction * virtual_distance;
                const float clip_x = dot4(pvp + 0, vwp.x, vwp.y, vwp.z, 1.0f);
                const float clip_y = dot4(pvp + 4, vwp.x, vwp.y, vwp.z, 1.0f);
                const float clip_w = dot4(pvp + 12, vwp.x, vwp.y, vwp.z, 1.0f);
                prev_valid = clip_w > 0.01f;
                if (prev_valid) {
                    const float iw = 1.0f / clip_w;
                    prev_u = (clip_x * iw) * 0.5f + 0.5f;
                    prev_v = (clip_y * iw) * 0.5f + 0.5f;
                    prev_valid = (prev_u >= 0.0f && prev_v >= 0.0f) && (prev_u < 1.0f && prev_v < 1.0f);
                }
                const float mvx = prev_valid ? (inUV.x - prev_u) : (inUV.x + 2.0f);
                const float mvy = prev_valid ? (inUV.y - prev_v) : (inUV.y + 2.0f);
                const f3 denoiser_albedo = lerp3(hit_albedo, splat(1.0f), metallic);
                store_gbuffer(a, pix, virtual_distance, hit_normal, roughness, denoiser_albedo, mvx, mvy);
                found_diffuse_surface = true;
                break;
            }
        }

        if (!found_diffuse_surface) {
            store_gbuffer(a, pix, 100000.0f, splat(0.0f), 0.0f, splat(0.0f), 0.0f, 0.0f);
            SrReservoir empty; zero_reservoir(empty);
            store48(reservoir_cur + pix, empty);
        } else {
            SrReservoir current_r; zero_reservoir(current_r);
            const uint32_t num_lights = sc.num_lights;
            if (num_lights > 0 && roughness > 0.2f) {
                for (uint32_t i = 0; i < a.cfg.ris_candidates; i++) {
                    uint32_t cand_idx = (uint32_t)(rnd(rng) * (float)num_lights);
                    if (cand_idx > num_lights - 1) cand_idx = num_lights - 1;
                    const LightTri lt = fetch_light(sc, cand_idx);
                    const float cand_area = lt.area;
                    const float sqr1 = sqrtf(rnd(rng));
                    const float u = 1.0f - sqr1;
                    const float v = rnd(rng) * sqr1;
                    const float w = 1.0f - u - v;
                    const f3 cand_pos = lt.wv0 * u + lt.wv1 * v + lt.wv2 * w;
                    const f3 cand_normal = lt.normal;
                    const f3 f_y = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic, lt.emission, cand_pos, cand_normal);
                    const float p_hat = maxc(f_y);
                    const float p_y = 1.0f / fmaxf((float)num_lights * cand_area, 0.0001f);
                    current_r.w_sum += (p_hat / p_y);
                    current_r.M += 1.0f;
                    if (rnd(rng) < ((p_hat / p_y) / fmaxf(current_r.w_sum, 0.0001f))) {
                        current_r.light_idx = cand_idx;
                        st3(current_r.light_pos, cand_pos);
                        st3(current_r.light_normal, cand_normal);
                    }
                }
                if (current_r.w_sum > 0.0f) {
                    const f3 fw = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                        light_emission(sc, current_r.light_idx), ld3(current_r.light_pos), ld3(current_r.light_normal));
                    current_r.W = current_r.w_sum / fmaxf(current_r.M * maxc(fw), 0.0001f);
                }
                if (a.frame_count > 0 && prev_valid) {
                    const float ppx = prev_u * (float)W, ppy = prev_v * (float)H;
                    const float j0 = rnd(rng), j1 = rnd(rng);
                    const int pcx = (int)(ppx + (j0 - 0.5f)), pcy = (int)(ppy + (j1 - 0.5f));
                    if (pcx >= 0 && pcy >= 0 && pcx < (int)W && pcy < (int)H) {
                        SrReservoir history_r = load48(reservoir_hist + ((uint32_t)pcy * W + (uint32_t)pcx));
                        history_r.M = fminf(history_r.M, 10.0f);
                        history_r.W = fminf(history_r.W, 20.0f);
                        const f3 hist_normal = unpack_normal(history_r.hit_normal_packed);
                        const float normal_conf_di = smoothstepf(0.9f, 0.99f, dot3(hit_normal, hist_normal));
                        const float depth_diff_di = fabsf(virtual_distance - history_r.depth) / fmaxf(virtual_distance, 1e-4f);
                        const float depth_conf_di = 1.0f - smoothstepf(0.05f, 0.20f, depth_diff_di);
                        const float conf_di = normal_conf_di * depth_conf_di;
                        history_r.M *= conf_di;
                        if (history_r.W > 0.0f) {
                            history_r.light_idx = history_r.light_idx < num_lights - 1 ? history_r.light_idx : num_lights - 1;
                            const f3 fh = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                                light_emission(sc, history_r.light_idx), ld3(history_r.light_pos), ld3(history_r.light_normal));
                            const float hist_rand = rnd(rng);
                            merge_reservoirs(current_r, history_r, maxc(fh), hist_rand);
                            const f3 fm = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                                light_emission(sc, current_r.light_idx), ld3(current_r.light_pos), ld3(current_r.light_normal));
                            current_r.W = current_r.w_sum / fmaxf(current_r.M * maxc(fm), 0.0001f);
                        }
                    }
                }
            }
            if (current_r.W > 0.0f) {
                f3 vis_dir = ld3(current_r.light_pos) - hitPos;
                const float vis_dist = fmaxf(len3(vis_dir), 0.0001f);
                vis_dir = vis_dir / vis_dist;
                if (dot3(hit_normal, vis_dir) <= 0.0f) current_r.W = 0.0f;
                else {
                    prd.dist = trace_shadow<V>(cx, hitPos + hit_normal * 0.001f, vis_dir, vis_dist);
                    if (prd.dist >= 0.0f) current_r.W = 0.0f;
                }
            }
            current_r.hit_normal_packed = pack_normal(hit_normal);
            current_r.depth = virtual_distance;
            store48(reservoir_cur + pix, current_r);

            // Phase 3: ReSTIR GI initial sample
            SrReservoirGI gi; zero_reservoir_gi(gi);
            const float gr1 = rnd(rng), gr2 = rnd(rng);
            const f3 gi_dir = get_random_bounce(hit_normal, gr1, gr2);
            const float gi_NdotL = fmaxf(dot3(hit_normal, gi_dir), 0.0f);
            if (gi_NdotL > 0.0f) {
                const f3 gi_origin = hitPos + hit_normal * 0.001f;
                prd = trace_closest_shaded<V>(cx, gi_origin, gi_dir, 0.001f, 10000.0f);
                f3 sample_pos = splat(0.0f), sample_normal = splat(0.0f), sample_radiance = splat(0.0f);
                if (prd.dist > 0.0f) {
                    sample_pos = gi_origin + gi_dir * prd.dist;
                    sample_normal = unpack_normal(prd.normal_packed);
                    const f3 x2_albedo = unpack_unorm_rgb(prd.albedo_packed);
                    sample_radiance = prd.emission;
                    if (num_lights > 0) {
                        uint32_t nee_idx = (uint32_t)(rnd(rng) * (float)num_lights);
                        if (nee_idx > num_lights - 1) nee_idx = num_lights - 1;
                        const LightTri lt = fetch_light(sc, nee_idx);
                        const float sq = sqrtf(rnd(rng));
                        const float nu = 1.0f - sq;
                        const float nv = rnd(rng) * sq;
                        const float nw = 1.0f - nu - nv;
                        const f3 nee_pos = lt.wv0 * nu + lt.wv1 * nv + lt.wv2 * nw;
                        const f3 nee_normal = lt.normal;
                        const float nee_area = lt.area;
                        f3 to_light = nee_pos - sample_pos;
                        const float nee_dist = fmaxf(len3(to_light), 0.0001f);
                        to_light = to_light / nee_dist;
                        const float nee_cos_surf = fmaxf(dot3(sample_normal, to_light), 0.0f);
                        const float nee_cos_light = fmaxf(dot3(nee_normal, -to_light), 0.0f);
                        if (nee_cos_surf > 0.0f && nee_cos_light > 0.0f) {
                            prd.dist = trace_shadow<V>(cx, sample_pos + sample_normal * 0.001f, to_light, nee_dist);
                            if (prd.dist < 0.0f) {
                                const float nee_pdf_sa = (nee_dist * nee_dist) / fmaxf(nee_cos_light * nee_area * (float)num_lights, 0.0001f);
                                sample_radiance = sample_radiance + (lt.emission * x2_albedo * nee_cos_surf) / (nee_pdf_sa * 3.14159f);
                            }
                        }
                    }
                }
                sample_radiance = vmin(sample_radiance, splat(5.0f));
                const float p_hat = gi_target_pdf(hitPos, hit_normal, hit_albedo, metallic, sample_pos, sample_radiance);
                const float pdf = gi_NdotL / 3.14159f;
                gi.M = 1.0f;
                gi.w_sum = (pdf > 0.0f) ? (p_hat / pdf) : 0.0f;
                gi.W = (p_hat > 0.0f) ? (gi.w_sum / (gi.M * p_hat)) : 0.0f;
                st3(gi.sample_pos, sample_pos);
                gi.sample_normal_packed = pack_normal(sample_normal);
                st3(gi.sample_radiance, sample_radiance);
            }
            if (a.frame_count > 0 && prev_valid) {
                const float ppx = prev_u * (float)W, ppy = prev_v * (float)H;
                const float j0 = rnd(rng), j1 = rnd(rng);
                const int gx = (int)(ppx + (j0 - 0.5f)), gy = (int)(ppy + (j1 - 0.5f));
                if (gx >= 0 && gy >= 0 && gx < (int)W && gy < (int)H) {
                    SrReservoirGI hgi = load48(reservoir_gi_hist + ((uint32_t)gy * W + (uint32_t)gx));
                    const f3 gi_hist_normal = unpack_normal(hgi.hit_normal_packed);
                    const float normal_conf = smoothstepf(0.8f, 0.95f, dot3(hit_normal, gi_hist_normal));
                    const float depth_diff = fabsf(virtual_distance - hgi.depth) / fmaxf(virtual_distance, 1e-4f);
                    const float depth_conf = 1.0f - smoothstepf(0.05f, 0.20f, depth_diff);
                    const float conf = normal_conf * depth_conf;
                    hgi.M = fminf(hgi.M, 12.0f) * conf;
                    hgi.W = fminf(hgi.W, 10.0f);
                    if (hgi.W > 0.0f && hgi.M > 0.0f) {
                        const float p_hat_hist = gi_target_pdf(hitPos, hit_normal, hit_albedo, metallic, ld3(hgi.sample_pos), ld3(hgi.sample_radiance));
                        const float mr = rnd(rng);
                        merge_reservoirs_gi(gi, hgi, p_hat_hist, 1.0f, mr);
                        const float p_hat_merged = gi_target_pdf(hitPos, hit_normal, hit_albedo, metallic, ld3(gi.sample_pos), ld3(gi.sample_radiance));
                        gi.W = (p_hat_merged > 1e-6f) ? (gi.w_sum / (gi.M * p_hat_merged)) : 0.0f;
                    }
                }
            }
            gi.hit_normal_packed = pack_normal(hit_normal);
            gi.depth = virtual_distance;
            store48(reservoir_gi_cur + pix, gi);
        }
    }
    record_tile_cost(a, cost_slot, t_start);
    if (!(a.cfg.flags & SR_TRACE_FLAG_UNCOUNTED)) {
        const bool counted = (a.cfg.count_rows == 0u || (py - a.cfg.count_y0) < a.cfg.count_rows) &&      // per lane: its pixel's row
                             (a.cfg.count_cols == 0u || (px - a.cfg.count_x0) < a.cfg.count_cols);       // and column
        flush_counter(sc.counters, 0, counted ? counted_closest(cx.n_q) : 0u);
        flush_counter(sc.counters, 1, counted ? counted_any(cx.n_q) : 0u);
        if (V & 1) { flush_counter(sc.counters, 2, counted ? cx.st.boxes : 0u); flush_counter(sc.counters, 3, counted ? cx.st.tris : 0u); }
    }
}
template <int V>
__global__ __launch_bounds__(kPassBlock, pass_waves(V)) void final_kernel(const PassArgs a) {
    // Flattened form of the pass for the work-stealing traversal: every trace point is reached by ALL lanes of the wave
    // (the control flow around it is predicated, not branched), so lanes without a ray of their own can take over
    // subtrees of the lanes that have one. Same operations in the same order per pixel as the branched form.
    extern __shared__ __attribute__((aligned(16))) int s_stack[];
    PixelCtx cx{a, s_stack + threadIdx.x, 0u, {0u, 0u}};
    const DevScene& sc = a.sc;
    uint32_t px = 0, py = 0, cost_slot = 0;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    const bool active = thread_pixel(a, px, py, cost_slot);
    const uint32_t W = a.width, H = a.height;
    const uint32_t pix = active ? py * W + px : 0u;
    const int ipx = (int)px, ipy = (int)py;
    const uint32_t cur_buf = a.frame_count & 1u;
    const SrReservoir* reservoir_cur = a.reservoirs[cur_buf];
    const SrReservoirGI* reservoir_gi_cur = a.reservoirs_gi[cur_buf];
    const uint32_t num_lights = sc.num_lights;

    uint32_t rng = init_rng(px, py, a.frame_count, W);                                 // :37
    const int BOUNCES = (int)a.cfg.max_bounces;
    const int SHADOW_BOUNCES = (int)a.cfg.shadow_bounces;
    // :44-50 the two blue-noise texels of the pixel are fetched where they are used (the BRDF bounce of bounce 0, :393-396): with
    // ReSTIR on, most walks end at their first rough hit and never get there
    f3 origin, rayDir0; f2 inUV;
    primary_ray(a.mats, px, py, W, H, origin, rayDir0, inUV);                          // :58-67
    f3 rayOrigin = origin, rayDir = rayDir0;
    f3 throughput = splat(1.0f), radiance = splat(0.0f);
    bool restir_evaluated = (a.cfg.enable_restir == 0);
    bool prev_did_nee = false;
    Payload prd;
    TravHit h; bool occ;

    bool in_loop = active;
    for (int bounce = 0; bounce < BOUNCES; bounce++) {                                 // :74
        if (__builtin_amdgcn_ballot_w64(in_loop) == 0ull) break;
        // bounce 0 is the query the RIS pass answered for this pixel a moment ago (same camera ray, same structure): with the
        // hand-off buffer the payload is read back instead of traversed and shaded again (launch-uniform condition)
        const bool reuse_primary = bounce == 0 && a.primary_payload != nullptr;
        if (!reuse_primary) ws_query<V, false>(cx, in_loop, rayOrigin, rayDir, 0.001f, 10000.0f, h);
        bool do_restir = false, do_nee = false, do_bounce = false;
        f3 hit_normal = splat(0.0f), hit_albedo = splat(0.0f), hitPos = splat(0.0f), V_view = splat(0.0f);
        float roughness = 0.5f, metallic = 0.0f;
        if (in_loop) {
            if (reuse_primary) prd = load_payload(a.primary_payload + pix);
            else prd = shade_hit<(V & 2) != 0, (V & 4) != 0>(sc, h);
            if (prd.dist < 0.0f) in_loop = false;                                      // :82-84
            else {
                hit_normal = unpack_normal(prd.normal_packed);
                hit_albedo = unpack_unorm_rgb(prd.albedo_packed);
                hitPos = rayOrigin + rayDir * prd.dist;
                V_view = -rayDir;
                const f2 mat_info = unpack_half_2x16(prd.material_info);
                roughness = fmaxf(mat_info.x, 0.01f);
                metallic = clampf(mat_info.y, 0.0f, 1.0f);
                const f2 trans_ior = unpack_half_2x16(prd.transmission_ior_packed);
                const float transmission = trans_ior.x;
                const float ior = fmaxf(trans_ior.y, 1.0f);
                if (!prev_did_nee) radiance = radiance + prd.emission * throughput;    // :99-101
                prev_did_nee = false;
                const float brightness = maxc(prd.emission);
                if (brightness > 1.0f) in_loop = false;                                // :104
                else if (transmission > 0.5f) {                                        // :106-133 (continue)
                    const bool is_inside = dot3(rayDir, hit_normal) > 0.0f;
                    const f3 N = is_inside ? -hit_normal : hit_normal;
                    const float eta = is_inside ? (ior / 1.0f) : (1.0f / ior);
                    const float cos_theta = fminf(dot3(-rayDir, N), 1.0f);
                    float R0 = (1.0f - eta) / (1.0f + eta);
                    R0 = R0 * R0;
                    float fresnel = R0 + (1.0f - R0) * pow5f(1.0f - cos_theta);
                    const f3 refracted = refract3(rayDir, N, eta);
                    if (len3(refracted) < 0.01f) fresnel = 1.0f;
                    if (rnd(rng) < fresnel) rayDir = reflect3(rayDir, N);
                    else {
                        rayDir = refracted;
                        if (is_inside) {
                            const f3 absorption = 1.0f - hit_albedo;
                            const f3 e = -absorption * prd.dist * 5.0f;
                            throughput = throughput * mk3(exp_pinned(e.x), exp_pinned(e.y), exp_pinned(e.z));
                        } else throughput = throughput * hit_albedo;
                    }
                    rayOrigin = hitPos + rayDir * 0.001f;
                } else {
                    do_bounce = true;
                    if (num_lights > 0 && bounce < SHADOW_BOUNCES) {                   // :135
                        if (!restir_evaluated && roughness > 0.2f) { restir_evaluated = true; do_restir = true; do_bounce = false; }
                        else if (restir_evaluated && roughness > 0.2f) do_nee = true;
                    }
                }
            }
        }

        // ---- :136-327 first rough hit: ReSTIR DI + GI spatial reuse, then the walk stops ----
        if (__builtin_amdgcn_ballot_w64(do_restir) != 0ull) {
            SrReservoir spatial_r; zero_reservoir(spatial_r);
            f3 f_y_winner = splat(0.0f), shadow_dir = splat(0.0f);
            float shadow_dist = 0.0f;
            bool di_pending = false, want_di = false;
            if (do_restir) {
                SrReservoir center_r = load48(reservoir_cur + pix);                    // :139-158
                if (center_r.W > 0.0f && center_r.light_idx < num_lights) {
                    center_r.light_idx = center_r.light_idx < num_lights - 1 ? center_r.light_idx : num_lights - 1;
                    const f3 fc = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                        light_emission(sc, center_r.light_idx), ld3(center_r.light_pos), ld3(center_r.light_normal));
                    const float cr = rnd(rng);
                    merge_reservoirs(spatial_r, center_r, maxc(fc), cr);
                }
                const float current_depth = len3(hitPos - origin);                     // :162
                for (int s = 0; s < 5; s++) {                                          // :164-188 SPATIAL_SAMPLES = 5, RADIUS = 30
                    const float angle = rnd(rng) * 2.0f * 3.14159f;
                    const float radius = sqrtf(rnd(rng)) * 30.0f;
                    float sa, ca; sincos_pinned(angle, sa, ca);
                    const int ncx = ipx + (int)(ca * radius), ncy = ipy + (int)(sa * radius);
                    if (ncx < 0 || ncy < 0 || ncx >= (int)W || ncy >= (int)H) continue;
                    const uint32_t pi_n = (uint32_t)ncy * W + (uint32_t)ncx;
                    const f3 neighbor_normal = load_normal(a, pi_n);
                    const float neighbor_depth = load_depth(a, pi_n);
                    if (dot3(hit_normal, neighbor_normal) < 0.9f) continue;
                    if (fabsf(current_depth - neighbor_depth) > 0.1f * current_depth) continue;
                    SrReservoir nr = load48(reservoir_cur + pi_n);
                    nr.W = fminf(nr.W, 20.0f);
                    nr.M = fminf(nr.M, 10.0f);
                    if (nr.W > 0.0f && nr.light_idx < num_lights) {
                        nr.light_idx = nr.light_idx < num_lights - 1 ? nr.light_idx : num_lights - 1;
                        const f3 fn = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                            light_emission(sc, nr.light_idx), ld3(nr.light_pos), ld3(nr.light_normal));
                        const float nrnd = rnd(rng);
                        merge_reservoirs(spatial_r, nr, maxc(fn), nrnd);
                    }
                }
                if (spatial_r.w_sum > 0.0f) {                                          // :190-205
                    f_y_winner = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                       light_emission(sc, spatial_r.light_idx), ld3(spatial_r.light_pos), ld3(spatial_r.light_normal));
                    spatial_r.W = spatial_r.w_sum / fmaxf(spatial_r.M * maxc(f_y_winner), 1e-3f);
                    spatial_r.W = fminf(spatial_r.W, 50.0f);
                    shadow_dir = ld3(spatial_r.light_pos) - hitPos;
                    shadow_dist = fmaxf(len3(shadow_dir), 0.0001f);
                    shadow_dir = shadow_dir / shadow_dist;
                    if (dot3(hit_normal, shadow_dir) > 0.0f) { di_pending = true; want_di = shadow_dist > 0.002f; }
                }
            }
            occ = ws_query<V, true>(cx, want_di, hitPos, shadow_dir, 0.001f, shadow_dist - 0.001f, h);   // :206-212 (origin = bare hitPos)
            if (di_pending) {
                prd.dist = want_di ? (occ ? 1.0f : -1.0f) : -1.0f;
                if (prd.dist < 0.0f) radiance = radiance + f_y_winner * throughput * spatial_r.W;             // :217-219
                prev_did_nee = true;
            }

            // ReSTIR GI spatial reuse (:224-291)
            SrReservoirGI combined; zero_reservoir_gi(combined);
            float gi_current_depth = 0.0f;
            // true once the combined reservoir's sample is a neighbour's: its visibility ray from this very hit point was traced (or skipped
            // as too short) a moment ago and found unoccluded — else the neighbour would not have been merged (:287)
            bool gi_sample_seen = false;
            if (do_restir) {
                combined = load48(reservoir_gi_cur + pix);
                gi_current_depth = len3(hitPos - origin);
            }
            for (int s = 0; s < 3; s++) {                                              // GI_SPATIAL_SAMPLES = 3, RADIUS = 20
                bool cand = false, want_g = false;
                uint32_t cand_pix = 0u;   // the neighbour's reservoir is read again after the query instead of being kept in registers
                float jacobian = 0.0f, d_new = 0.0f;
                f3 gi_spatial_dir = splat(0.0f);
                if (do_restir) {
                    const float gi_angle = rnd(rng) * 2.0f * 3.14159f;
                    const float gi_radius = sqrtf(rnd(rng)) * 20.0f;
                    float sa, ca; sincos_pinned(gi_angle, sa, ca);
                    const int ncx = ipx + (int)(ca * gi_radius), ncy = ipy + (int)(sa * gi_radius);
                    bool ok = !(ncx == ipx && ncy == ipy);                             // :237
                    ok = ok && !(ncx < 0 || ncy < 0 || ncx >= (int)W || ncy >= (int)H);
                    if (ok) {
                        const uint32_t pi_nn = (uint32_t)ncy * W + (uint32_t)ncx;
                        const f3 neighbor_normal = load_normal(a, pi_nn);
                        const float neighbor_depth = load_depth(a, pi_nn);
                        ok = !(dot3(hit_normal, neighbor_normal) < 0.9f);
                        ok = ok && !(fabsf(gi_current_depth - neighbor_depth) > 0.1f * gi_current_depth);
                        if (ok) {
                            SrReservoirGI nr = load48(reservoir_gi_cur + pi_nn);
                            ok = !(nr.W <= 0.0f);                                      // :248
                            if (ok) {
                                nr.W = fminf(nr.W, 10.0f);
                                nr.M = fminf(nr.M, 10.0f);
                                f3 n_origin, n_dir; f2 n_uv;
                                primary_ray(a.mats, (uint32_t)ncx, (uint32_t)ncy, W, H, n_origin, n_dir, n_uv);   // :253-258
                                const f3 neighbor_x1 = origin + n_dir * neighbor_depth;
                                const f3 nsp = ld3(nr.sample_pos);
                                const f3 w_new = nsp - hitPos;
                                const f3 w_old = nsp - neighbor_x1;
                                d_new = fmaxf(len3(w_new), 1e-4f);
                                const float d_old = fmaxf(len3(w_old), 1e-4f);
                                const f3 n_x2 = unpack_normal(nr.sample_normal_packed);
                                const float cos_new = fmaxf(dot3(n_x2, (-w_new) / d_new), 0.0f);
                                const float cos_old = fmaxf(dot3(n_x2, (-w_old) / d_old), 0.0f);
                                ok = !(cos_new <= 0.0f || cos_old <= 0.0f);            // :267
                                if (ok) {
                                    jacobian = (cos_new * d_old * d_old) / fmaxf(cos_old * d_new * d_new, 1e-4f);
                                    jacobian = clampf(jacobian, 0.0f, 10.0f);
                                    gi_spatial_dir = w_new / d_new;
                                    ok = !(dot3(hit_normal, gi_spatial_dir) <= 0.0f);  // :273
                                    if (ok) { cand = true; cand_pix = pi_nn; want_g = d_new > 0.002f; }
                                }
                            }
                        }
                    }
                }
                occ = ws_query<V, true>(cx, want_g, hitPos, gi_spatial_dir, 0.001f, d_new - 0.001f, h);   // :276-286
                if (cand) {
                    prd.dist = want_g ? (occ ? 1.0f : -1.0f) : -1.0f;
                    if (!(prd.dist >= 0.0f)) {                                         // :287
                        SrReservoirGI nr = load48(reservoir_gi_cur + cand_pix);
                        nr.W = fminf(nr.W, 10.0f);
                        nr.M = fminf(nr.M, 10.0f);
                        const float p_hat_neighbor = gi_target_pdf(hitPos, hit_normal, hit_albedo, metallic, ld3(nr.sample_pos), ld3(nr.sample_radiance));
                        const float gr = rnd(rng);
                        if (merge_reservoirs_gi(combined, nr, p_hat_neighbor, jacobian, gr)) gi_sample_seen = true;
                    }
                }
            }
            bool gif_pending = false, want_gif = false;
            f3 gi_x2_dir = splat(0.0f);
            float gi_x2_dist = 0.0f, gi_NdotL = 0.0f;
            if (do_restir) {                                                           // :293-303
                const float p_hat_final = gi_target_pdf(hitPos, hit_normal, hit_albedo, metallic, ld3(combined.sample_pos), ld3(combined.sample_radiance));
                combined.W = (p_hat_final > 1e-3f) ? (combined.w_sum / fmaxf(combined.M, 1.0f) / p_hat_final) : 0.0f;
                combined.W = fminf(combined.W, 20.0f);
                if (combined.W > 0.0f) {
                    gi_x2_dir = ld3(combined.sample_pos) - hitPos;
                    gi_x2_dist = fmaxf(len3(gi_x2_dir), 0.0001f);
                    gi_x2_dir = gi_x2_dir / gi_x2_dist;
                    gi_NdotL = fmaxf(dot3(hit_normal, gi_x2_dir), 0.0f);
                    if (gi_NdotL > 0.0f) {
                        gif_pending = true; want_gif = gi_x2_dist > 0.002f;
                        // :304-316 would trace hitPos -> combined.sample_pos once more: same origin, and direction and length come from the
                        // same three operations on the same operands as :262-274 — the same query, hence the same answer: not occluded.
                        // (Answered without a traversal unless the caller asks for every query, SR_TRACE_FLAG_TRACE_EVERY_QUERY.)
                        if (gi_sample_seen && !(a.cfg.flags & SR_TRACE_FLAG_TRACE_EVERY_QUERY)) { if (want_gif) cx.n_q |= kCountReusedAny; want_gif = false; }
                    }
                }
            }
            occ = ws_query<V, true>(cx, want_gif, hitPos, gi_x2_dir, 0.001f, gi_x2_dist - 0.001f, h);   // :309-316
            if (gif_pending) {
                prd.dist = want_gif ? (occ ? 1.0f : -1.0f) : -1.0f;
                if (prd.dist < 0.0f) {                                                 // :321-324
                    const f3 gi_f_diffuse = hit_albedo * (1.0f - metallic) / 3.14159f;
                    radiance = radiance + ld3(combined.sample_radiance) * gi_f_diffuse * gi_NdotL * combined.W * throughput;
                }
            }
            if (do_restir) in_loop = false;                                            // :327 break
        }

        // ---- :328-382 later rough bounces: one NEE sample ----
        if (__builtin_amdgcn_ballot_w64(do_nee) != 0ull) {
            bool nee_pending = false, want_nee = false;
            f3 shadow_ray_dir = splat(0.0f), nee_emission = splat(0.0f);
            float light_dist = 0.0f, cos_theta_light = 0.0f, cos_theta_surface = 0.0f, light_area = 0.0f;
            if (do_nee) {
                uint32_t light_idx = (uint32_t)(rnd(rng) * (float)num_lights);
                if (light_idx > num_lights - 1) light_idx = num_lights - 1;
                const LightTri lt = fetch_light(sc, light_idx);
                light_area = lt.area;
                nee_emission = lt.emission;
                const float r1_nee = rnd(rng);
                const float r2_nee = rnd(rng);
                const float sqr1 = sqrtf(r1_nee);
                const float u = 1.0f - sqr1;
                const float v = r2_nee * sqr1;
                const float w = 1.0f - u - v;
                const f3 light_pos = lt.wv0 * u + lt.wv1 * v + lt.wv2 * w;
                const f3 light_normal = lt.normal;
                shadow_ray_dir = light_pos - hitPos;
                light_dist = len3(shadow_ray_dir);
                shadow_ray_dir = shadow_ray_dir / light_dist;
                cos_theta_light = fmaxf(dot3(light_normal, -shadow_ray_dir), 0.0f);
                cos_theta_surface = fmaxf(dot3(hit_normal, shadow_ray_dir), 0.0f);
                if (cos_theta_light > 0.0f && cos_theta_surface > 0.0f) { nee_pending = true; want_nee = light_dist > 0.002f; }
            }
            occ = ws_query<V, true>(cx, want_nee, hitPos, shadow_ray_dir, 0.001f, light_dist - 0.001f, h);   // :363-370
            if (nee_pending) {
                prd.dist = want_nee ? (occ ? 1.0f : -1.0f) : -1.0f;
                if (prd.dist < 0.0f) {                                                 // :375-379
                    const float solid_angle_pdf = (light_dist * light_dist) / fmaxf(cos_theta_light * light_area * (float)num_lights, 1e-4f);
                    const f3 nee_contrib = (nee_emission * hit_albedo * throughput * cos_theta_surface) / (solid_angle_pdf * 3.14159f);
                    radiance = radiance + vmin(nee_contrib, splat(5.0f));
                }
                prev_did_nee = true;
            }
        }

        // ---- :385-427 BRDF bounce ----
        if (in_loop && do_bounce) {
            const f3 N = hit_normal;
            const f3 F0 = lerp3(splat(0.04f), hit_albedo, metallic);
            const float cos_theta = fmaxf(dot3(N, V_view), 0.0f);
            const f3 F = F0 + (1.0f - F0) * pow5f(clampf(1.0f - cos_theta, 0.0f, 1.0f));
            const float p_specular = clampf(maxc(F), 0.05f, 1.0f);
            float r1, r2;
            if (bounce == 0) {
                const int bw = (int)a.blue_noise_w, bh = (int)a.blue_noise_h;      // :44-50
                const int n1x = ipx % bw, n1y = ipy % bh;
                const int n2x = (ipx + 47) % bw, n2y = (ipy + 71) % bh;
                const float bn_1 = (float)a.blue_noise_tex[((size_t)n1y * bw + n1x) * 4] / 255.0f;
                const float bn_2 = (float)a.blue_noise_tex[((size_t)n2y * bw + n2x) * 4] / 255.0f;
                r1 = fracf(bn_1 + (float)(a.frame_count % 1024u) * 0.75487766f);
                r2 = fracf(bn_2 + (float)(a.frame_count % 1024u) * 0.56984029f);
            } else {
                r1 = rnd(rng);
                r2 = rnd(rng);
            }
            if (rnd(rng) < p_specular) {
                const f3 Hh = sample_ggx_vndf(N, V_view, roughness, r1, r2);
                rayDir = reflect3(-V_view, Hh);
                if (dot3(N, rayDir) <= 0.0f) {
                    rayDir = get_random_bounce(N, r1, r2);
                    throughput = throughput * (hit_albedo * (1.0f - metallic) * (1.0f - F) / (1.0f - p_specular));
                } else {
                    const float NdotL_b = fmaxf(dot3(N, rayDir), 0.001f);
                    const float alpha_b = roughness * roughness;
                    const float G1_L = smith_g1_ggx(NdotL_b, alpha_b);
                    throughput = throughput * ((F * G1_L) / p_specular);
                }
            } else {
                rayDir = get_random_bounce(N, r1, r2);
                throughput = throughput * (hit_albedo * (1.0f - metallic) * (1.0f - F) / (1.0f - p_specular));
            }
            const float p = maxc(throughput);
            if (p < 0.001f) in_loop = false;                                           // :420
            else {
                bool cont = true;
                if (bounce > 2) {
                    if (rnd(rng) > p) cont = false;                                    // :423
                    else throughput = throughput / p;
                }
                if (cont) rayOrigin = hitPos + hit_normal * 0.001f;                    // :427
                else in_loop = false;
            }
        }
    }
    if (active) {                                                                      // :430-435
        f3 total_radiance = splat(0.0f) + radiance;
        total_radiance = vmin(total_radiance, splat(10.0f));
        const f3 color = total_radiance / 1.0f;  // / float(SAMPLES)
        float4 o;
        o.x = color.x; o.y = color.y; o.z = color.z; o.w = 1.0f;
        reinterpret_cast<float4*>(a.raw_color)[pix] = o;
    }
    record_tile_cost(a, cost_slot, t_start);
    if (!(a.cfg.flags & SR_TRACE_FLAG_UNCOUNTED)) {
        const bool counted = (a.cfg.count_rows == 0u || (py - a.cfg.count_y0) < a.cfg.count_rows) &&      // per lane: its pixel's row
                             (a.cfg.count_cols == 0u || (px - a.cfg.count_x0) < a.cfg.count_cols);       // and column
        flush_counter(sc.counters, 0, counted ? counted_closest(cx.n_q) : 0u);
        flush_counter(sc.counters, 1, counted ? counted_any(cx.n_q) : 0u);
        // every pixel of the launch reaches bounce 0, and with the hand-off buffer bound that query is the one read back
        flush_counter(sc.counters, 4, (counted && active && BOUNCES > 0 && a.primary_payload != nullptr) ? 1u : 0u);
        flush_counter(sc.counters, 5, counted ? cx.n_q >> 31 : 0u);
        if (V & 1) { flush_counter(sc.counters, 2, counted ? cx.st.boxes : 0u); flush_counter(sc.counters, 3, counted ? cx.st.tris : 0u); }
    }
}

}  // namespace srd

// ---------------------------------------------------------------------------------------------
// Host-side launchers (called from api.cpp)
// ---------------------------------------------------------------------------------------------
using namespace srd;

int srk_launch_trace(const DevScene& sc, const SrRay* rays, uint32_t n, SrHit* hits, uint32_t* occluded,
                     int any, int stats, int two_level, int stack_entries, hipStream_t stream) {
    if (n == 0) return 0;
    dim3 grid((n + kBlock - 1) / kBlock), block(kBlock);
    const size_t lds = (size_t)(stack_entries + lds_extra_rows(two_level)) * kBlock * sizeof(int);
    const int v = (any ? 1 : 0) | (stats ? 2 : 0) | (two_level ? 4 : 0);
    switch (v) {
        case 0: trace_rays_kernel<false, false, false><<<grid, block, lds, stream>>>(sc, rays, n, hits, occluded); break;
        case 1: trace_rays_kernel<true, false, false><<<grid, block, lds, stream>>>(sc, rays, n, hits, occluded); break;
        case 2: trace_rays_kernel<false, true, false><<<grid, block, lds, stream>>>(sc, rays, n, hits, occluded); break;
        case 3: trace_rays_kernel<true, true, false><<<grid, block, lds, stream>>>(sc, rays, n, hits, occluded); break;
        case 4: trace_rays_kernel<false, false, true><<<grid, block, lds, stream>>>(sc, rays, n, hits, occluded); break;
        case 5: trace_rays_kernel<true, false, true><<<grid, block, lds, stream>>>(sc, rays, n, hits, occluded); break;
        case 6: trace_rays_kernel<false, true, true><<<grid, block, lds, stream>>>(sc, rays, n, hits, occluded); break;
        default: trace_rays_kernel<true, true, true><<<grid, block, lds, stream>>>(sc, rays, n, hits, occluded); break;
    }
    return (int)hipGetLastError();
}

int srk_launch_shade(const DevScene& sc, const SrHit* hits, uint32_t n, SrRayPayload* out, hipStream_t stream) {
    if (n == 0) return 0;
    shade_closest_hit_kernel<<<dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream>>>(sc, hits, n, out);
    return (int)hipGetLastError();
}

int srk_launch_any_hit(const DevScene& sc, const SrHit* hits, uint32_t n, uint32_t* ignored, hipStream_t stream) {
    if (n == 0) return 0;
    any_hit_kernel<<<dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream>>>(sc, hits, n, ignored);
    return (int)hipGetLastError();
}

// Kernel variant V: bit 0 = traversal statistics (instrumented build), bit 1 = the scene has textured materials
// (closest_hit's texture half compiled in; untextured scenes run the variant without it), bit 2 = two-level structure.
int srk_launch_pass(const PassArgs& args_in, int which, int stats, int textured, int two_level, int stack_entries, hipStream_t stream) {
    PassArgs args = args_in;
    args.tiles_x = (args.x1 - args.x0 + kPassTile - 1) / kPassTile;
    args.tiles_y = (args.y1 - args.y0 + kPassTile - 1) / kPassTile;
    args.order_cap = srk_pass_order_cap(args.x1 - args.x0, args.y1 - args.y0);   // list length per XCD = blocks per XCD
    const uint32_t n_tiles = args.tiles_x * args.tiles_y;
    if (n_tiles == 0) return 0;
    dim3 grid(args.order_cap * 8), block(kPassBlock);
    const size_t lds = (size_t)(stack_entries + lds_extra_rows(two_level)) * kPassBlock * sizeof(int);
    const int v = (stats ? 1 : 0) | (textured ? 2 : 0) | (two_level ? 4 : 0);
    if (which == 0) {
        switch (v) {
            case 0: ris_kernel<0><<<grid, block, lds, stream>>>(args); break;
            case 1: ris_kernel<1><<<grid, block, lds, stream>>>(args); break;
            case 2: ris_kernel<2><<<grid, block, lds, stream>>>(args); break;
            case 3: ris_kernel<3><<<grid, block, lds, stream>>>(args); break;
            case 4: ris_kernel<4><<<grid, block, lds, stream>>>(args); break;
            case 5: ris_kernel<5><<<grid, block, lds, stream>>>(args); break;
            case 6: ris_kernel<6><<<grid, block, lds, stream>>>(args); break;
            default: ris_kernel<7><<<grid, block, lds, stream>>>(args); break;
        }
    } else {
        switch (v) {
            case 0: final_kernel<0><<<grid, block, lds, stream>>>(args); break;
            case 1: final_kernel<1><<<grid, block, lds, stream>>>(args); break;
            case 2: final_kernel<2><<<grid, block, lds, stream>>>(args); break;
            case 3: final_kernel<3><<<grid, block, lds, stream>>>(args); break;
            case 4: final_kernel<4><<<grid, block, lds, stream>>>(args); break;
            case 5: final_kernel<5><<<grid, block, lds, stream>>>(args); break;
            case 6: final_kernel<6><<<grid, block, lds, stream>>>(args); break;
            default: final_kernel<7><<<grid, block, lds, stream>>>(args); break;
        }
    }
    return (int)hipGetLastError();
}

int srk_lds_rows(int stack_entries, int two_level) { return stack_entries + lds_extra_rows(two_level); }   // LDS rows (of one int per thread) a block needs

uint32_t srk_pass_tile_count(uint32_t width, uint32_t rows) { return ((width + kPassTile - 1) / kPassTile) * ((rows + kPassTile - 1) / kPassTile); }

// Entries per XCD of the tile schedule (= blocks per XCD of a pass launch): the widest band the schedule may form.
uint32_t srk_pass_order_cap(uint32_t width, uint32_t rows) {
    const uint32_t tiles_x = (width + kPassTile - 1) / kPassTile, tiles_y = (rows + kPassTile - 1) / kPassTile;
    const uint32_t cap_cols = std::max(std::min(tiles_x, (tiles_x * 3u) / 16u + 2u), (tiles_x + 7u) / 8u);
    return cap_cols * tiles_y;
}

int srk_launch_tile_order(const uint32_t* tile_cost, uint32_t* tile_order, uint32_t width, uint32_t rows, hipStream_t stream) {
    const uint32_t tiles_x = (width + kPassTile - 1) / kPassTile, tiles_y = (rows + kPassTile - 1) / kPassTile;
    if (tiles_x * tiles_y == 0) return 0;
    tile_order_kernel<<<dim3(8), dim3(256), 0, stream>>>(tile_cost, tile_order, tiles_x, tiles_y, srk_pass_order_cap(width, rows));
    return (int)hipGetLastError();
}
