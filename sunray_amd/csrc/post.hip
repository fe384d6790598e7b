// Post-RT compute chain for gfx950: the three 16x16-tile compute passes the
// reference appends after the ray-tracing passes (src/lib.rs:1576-1615).
//
//   temporal_kernel  <- shaders/temporal_accumulation.slang:60-132
//   denoise_kernel   <- shaders/denoise.slang:29-116   (launched once per a-trous pass)
//   tonemap_kernel   <- shaders/postprocess.slang:22-42
//
// All three are HBM/L2-bandwidth bound pointwise / stencil kernels. Images stay in the reference's
// packed formats (4 B per pixel for colour, 2 B depth) so a 1080p pass moves ~30-60 MB: the temporal
// pass stages its 18x18 colour tile in LDS exactly like the reference; the a-trous taps are gathers
// with stride 1..8 pixels that the L2 absorbs. 256-thread workgroups = one 16x16 tile, like the
// reference's numthreads(16,16,1).
#include <hip/hip_runtime.h>

#include "rt_device.h"

namespace srd {

constexpr int kTile = 16;
SRD float luminance(f3 c) { return dot3(c, mk3(0.2126f, 0.7152f, 0.0722f)); }
SRD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
SRD f3 vmax(f3 a, f3 b) { return mk3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
SRD f3 vdiv(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
// raw_color is fp32 here and B10G11R11 in the reference (lib.rs:1492-1496): quantise where it is read
SRD f3 load_raw_color(const float* raw, size_t i) {
    const float4 v = reinterpret_cast<const float4*>(raw)[i];
    return unpack_b10g11r11(pack_b10g11r11(v.x, v.y, v.z));
}

__global__ __launch_bounds__(256) void temporal_kernel(SrPostParams p) {
    __shared__ float tile[3][18 * 18];   // TILE_FULL x TILE_FULL colours, planar: conflict-free ds_read_b32
    const int W = (int)p.width, H = (int)p.height;
    const int lx = (int)threadIdx.x & 15, ly = (int)threadIdx.x >> 4;
    const int x = (int)blockIdx.x * kTile + lx, y = (int)blockIdx.y * kTile + ly;
    const int bx0 = (int)blockIdx.x * kTile - 1, by0 = (int)blockIdx.y * kTile - 1;
    for (int i = (int)threadIdx.x; i < 18 * 18; i += 256) {                       // :82-86 cooperative tile load
        const int sx = clampi(bx0 + i % 18, 0, W - 1), sy = clampi(by0 + i / 18, 0, H - 1);
        const f3 c = load_raw_color(p.raw_color, (size_t)sy * W + sx);
        tile[0][i] = c.x; tile[1][i] = c.y; tile[2][i] = c.z;
    }
    __syncthreads();
    if (x >= W || y >= H) return;
    const int tc = (ly + 1) * 18 + (lx + 1);
    const f3 current_color = mk3(tile[0][tc], tile[1][tc], tile[2][tc]);
    f3 min_color = current_color, max_color = current_color;
    const float center_luma = luminance(current_color);
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            if (dx == 0 && dy == 0) continue;
            const int t = tc + dy * 18 + dx;
            const f3 nc = mk3(tile[0][t], tile[1][t], tile[2][t]);
            const float neighbor_luma = luminance(nc);
            const float luma_threshold = fmaxf(center_luma * 5.0f, 0.08f);
            if (fabsf(neighbor_luma - center_luma) < luma_threshold) { min_color = vmin(min_color, nc); max_color = vmax(max_color, nc); }
        }
    const size_t i = (size_t)y * W + x;
    const float uvx = ((float)x + 0.5f) / (float)W, uvy = ((float)y + 0.5f) / (float)H;
    const f2 motion = unpack_half_2x16(p.motion_vec_img[i]);
    const float pux = uvx - motion.x, puy = uvy - motion.y;
    f3 accumulated = current_color;
    const bool off = (pux < 0.0f || puy < 0.0f) || (pux > 1.0f || puy > 1.0f);
    if (!off && p.frame_count > 2u) {
        const uint32_t* history = p.accum[(p.frame_count + 1u) % 2u];
        const float px_ = pux * (float)W - 0.5f, py_ = puy * (float)H - 0.5f;   // sample_history_bilinear :42-58
        const int bx = (int)floorf(px_), by = (int)floorf(py_);
        const float fx = px_ - (float)bx, fy = py_ - (float)by;
        const int x0 = clampi(bx, 0, W - 1), x1 = clampi(bx + 1, 0, W - 1), y0 = clampi(by, 0, H - 1), y1 = clampi(by + 1, 0, H - 1);
        const f3 h00 = unpack_b10g11r11(history[(size_t)y0 * W + x0]), h10 = unpack_b10g11r11(history[(size_t)y0 * W + x1]);
        const f3 h01 = unpack_b10g11r11(history[(size_t)y1 * W + x0]), h11 = unpack_b10g11r11(history[(size_t)y1 * W + x1]);
        const f3 history_color = lerp3(lerp3(h00, h10, fx), lerp3(h01, h11, fx), fy);
        const f3 clamped = mk3(clampf(history_color.x, min_color.x, max_color.x), clampf(history_color.y, min_color.y, max_color.y),
                               clampf(history_color.z, min_color.z, max_color.z));
        accumulated = lerp3(clamped, current_color, 0.14f);
    }
    p.accum[p.frame_count % 2u][i] = pack_b10g11r11(accumulated.x, accumulated.y, accumulated.z);
}

// One a-trous pass (denoise.slang:29-116), launched with step_width 1, 2, 4, 8. The 25 taps of a pixel lie on the lattice
// x + i*step, y + j*step, so a workgroup takes a 16x16 tile OF ONE LATTICE (all pixels with the same x mod step, y mod step):
// its taps fall on a 20x20 patch of that lattice, which is decoded ONCE per workgroup into LDS (illumination = colour /
// max(diffuse, 0.001), its luminance, diffuse, normal, depth: planar fp32, conflict-free) instead of 25 times per pixel — the
// per-tap unpacking and the three correctly rounded divisions were 2/3 of the kernel's VALU work. Same operations on the same
// values in the same tap order as the straightforward form, so the output bits do not change.
constexpr int kPatch = kTile + 4;             // 20
constexpr int kPatchN = kPatch * kPatch;      // 400
// LDS image of the patch: one 48-byte record per lattice point = three float4 (illumination.xyz, luminance |
// diffuse.xyz, depth | normal.xyz, inside-the-image flag), so a tap costs three ds_read_b128 instead of twelve ds_read_b32
// (the LDS pipe was a third of the pass). A patch row is padded from 60 to 64 float4: with a row stride that is a multiple of
// 16 slots the 48-byte-strided reads of the two half-rows a ds_read_b128 lane group covers fall on complementary slots
// (3 lx mod 16 over lx = 0-3, 12-15 and over lx = 4-11): conflict-free.
constexpr int kRowSlots = 64;                 // float4 per patch row (20 records x 3, padded)
__global__ __launch_bounds__(256) void denoise_kernel(SrPostParams p, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int step_width) {
    __shared__ float4 patch[kPatch * kRowSlots];                                    // 20 KB
    const int W = (int)p.width, H = (int)p.height, s = step_width;
    const int rx = (int)blockIdx.x % s, ry = (int)blockIdx.y % s;                   // residue class = which lattice
    const int l0x = ((int)blockIdx.x / s) * kTile, l0y = ((int)blockIdx.y / s) * kTile;   // tile origin in lattice coordinates
    for (int i = (int)threadIdx.x; i < kPatchN; i += 256) {
        const int px_ = i % kPatch, py_ = i / kPatch;
        const int sx = rx + s * (l0x - 2 + px_), sy = ry + s * (l0y - 2 + py_);
        const bool ok = sx >= 0 && sy >= 0 && sx < W && sy < H;
        float4* rec = patch + py_ * kRowSlots + px_ * 3;
        if (!ok) { rec[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); continue; }
        const size_t j = (size_t)sy * W + sx;
        const f3 sample_color = unpack_b10g11r11(src[j]);
        const uint32_t sn = p.normal_img[j];
        const f3 sample_diffuse = unpack_b10g11r11(p.diffuse_img[j]);
        const f3 sample_illum = vdiv(sample_color, vmax(sample_diffuse, splat(0.001f)));
        rec[0] = make_float4(sample_illum.x, sample_illum.y, sample_illum.z, luminance(sample_illum));
        rec[1] = make_float4(sample_diffuse.x, sample_diffuse.y, sample_diffuse.z, f16_bits_to_f32(p.depth_img[j]));
        rec[2] = make_float4(unsnorm8(sn), unsnorm8(sn >> 8), unsnorm8(sn >> 16), 1.0f);
    }
    __syncthreads();
    const int lx = (int)threadIdx.x & 15, ly = (int)threadIdx.x >> 4;
    const int x = rx + s * (l0x + lx), y = ry + s * (l0y + ly);
    if (x >= W || y >= H) return;
    const size_t i = (size_t)y * W + x;
    const float4* centre = patch + (ly + 2) * kRowSlots + (lx + 2) * 3;
    const f3 center_color = unpack_b10g11r11(src[i]);
    const float4 c1 = centre[1];
    const float center_depth = c1.w;
    if (center_depth >= 10000.0f) { dst[i] = pack_b10g11r11(center_color.x, center_color.y, center_color.z); return; }
    const float4 c2 = centre[2];
    const f3 center_normal = mk3(c2.x, c2.y, c2.z);
    const float center_roughness = unsnorm8(p.normal_img[i] >> 24);
    const f3 center_diffuse = mk3(c1.x, c1.y, c1.z);
    if (center_roughness < 0.1f) { dst[i] = pack_b10g11r11(center_color.x, center_color.y, center_color.z); return; }
    const float4 c0 = centre[0];
    const f3 center_illum = mk3(c0.x, c0.y, c0.z);
    const float kernel[5] = {1.0f / 16.0f, 4.0f / 16.0f, 6.0f / 16.0f, 4.0f / 16.0f, 1.0f / 16.0f};
    const float center_weight = kernel[2] * kernel[2];
    f3 sum_color = center_illum * center_weight;
    float sum_weight = center_weight;
    const float center_luma = c0.w;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const float4* rec = centre + dy * kRowSlots + dx * 3;
            const float4 r2 = rec[2];
            if (r2.w == 0.0f) continue;                                             // outside the image (denoise.slang:83-86)
            const float4 r0 = rec[0], r1 = rec[1];
            const f3 sample_illum = mk3(r0.x, r0.y, r0.z);
            const f3 sample_diffuse = mk3(r1.x, r1.y, r1.z);
            const f3 sample_normal = mk3(r2.x, r2.y, r2.z);
            const float sample_depth = r1.w;
            const float sample_luma = r0.w;
            const float diffuse_diff = len3(center_diffuse - sample_diffuse);
            const float luma_diff = fabsf(center_luma - sample_luma);
            const float luma_sigma = fmaxf(center_luma, sample_luma) * 0.4f + 0.01f;
            const float luma_ratio = luma_diff / luma_sigma;
            const float combined_power = -fabsf(center_depth - sample_depth) * 8.0f
                                         + (dot3(center_normal, sample_normal) - 1.0f) * 80.0f
                                         - diffuse_diff * 50.0f
                                         - luma_ratio * luma_ratio;
            const float weight = exp_pinned(combined_power) * kernel[dx + 2] * kernel[dy + 2];
            sum_color = sum_color + sample_illum * weight;
            sum_weight += weight;
        }
    }
    const f3 out = (sum_color / fmaxf(sum_weight, 0.0001f)) * center_diffuse;
    dst[i] = pack_b10g11r11(out.x, out.y, out.z);
}

SRD float aces(float x) {
    x = clampf(x, 0.0f, 100.0f);
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return clampf((x * (a * x + b)) / (x * (c * x + d) + e), 0.0f, 1.0f);
}

__global__ __launch_bounds__(256) void tonemap_kernel(SrPostParams p, const uint32_t* __restrict__ src) {
    const size_t n = (size_t)p.width * p.height;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    f3 color = unpack_b10g11r11(src[i]);
    const bool bad = !(color.x == color.x) || !(color.y == color.y) || !(color.z == color.z) || isinf(color.x) || isinf(color.y) || isinf(color.z);
    if (bad) color = splat(0.0f);
    color = color * p.exposure;
    const f3 mapped = mk3(aces(color.x), aces(color.y), aces(color.z));
    const float g = 1.0f / 2.2f;
    p.output_rgba8[i] = pack_unorm_4x8(pow_pinned(mapped.x, g), pow_pinned(mapped.y, g), pow_pinned(mapped.z, g), 1.0f);
}

}  // namespace srd

using namespace srd;

int srk_launch_post_temporal(const SrPostParams& p, hipStream_t stream) {
    dim3 grid((p.width + kTile - 1) / kTile, (p.height + kTile - 1) / kTile), block(256);
    temporal_kernel<<<grid, block, 0, stream>>>(p);
    return (int)hipGetLastError();
}
int srk_launch_post_denoise(const SrPostParams& p, hipStream_t stream) {
    dim3 block(256);
    for (uint32_t pass = 0; pass < p.denoise_passes; pass++) {   // lib.rs:1817-1826
        const uint32_t* src = pass == 0 ? p.accum[p.frame_count % 2u] : (pass % 2u == 1u ? p.denoise[0] : p.denoise[1]);
        uint32_t* dst = (pass == 0 || pass % 2u == 0u) ? p.denoise[0] : p.denoise[1];
        const uint32_t s = 1u << pass;                           // step_width; each of the s*s lattices gets its own 16x16 tiles
        const uint32_t lw = (p.width + s - 1) / s, lh = (p.height + s - 1) / s;
        dim3 grid(s * ((lw + kTile - 1) / kTile), s * ((lh + kTile - 1) / kTile));
        denoise_kernel<<<grid, block, 0, stream>>>(p, src, dst, (int)s);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}
int srk_launch_post_tonemap(const SrPostParams& p, hipStream_t stream) {
    const size_t n = (size_t)p.width * p.height;
    tonemap_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(p, p.denoise[(p.denoise_passes - 1u) % 2u]);
    return (int)hipGetLastError();
}
