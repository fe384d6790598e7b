// ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Parity status: UNPINNED (see orc_math.h).
// CPU restatement of the post-RT compute chain:
//   post_temporal <- shaders/temporal_accumulation.slang:60-132
//   post_denoise  <- shaders/denoise.slang:29-116 (one a-trous pass per call inside the loop of lib.rs:1817-1870)
//   post_tonemap  <- shaders/postprocess.slang:22-42
// Image formats follow the reference (lib.rs:452-461,1492-1516): every load decodes and every store
// encodes B10G11R11 / R16F / RGBA8_SNORM / RG16F / RGBA8_UNORM exactly where the reference's typed
// image accesses do.
#include "orc_utils.h"

namespace orc {

namespace {
inline V3 load_b10g11r11(const uint32_t* img, size_t i) {
    const uint32_t v = img[i];
    return v3(from_ufloat(v & 0x7ffu, 6), from_ufloat((v >> 11) & 0x7ffu, 6), from_ufloat(v >> 22, 5));
}
// raw_color: fp32 here, B10G11R11 in the reference -> quantise on read
inline V3 load_raw_color(const float* raw, size_t i) {
    const uint32_t p = pack_b10g11r11(raw[4 * i], raw[4 * i + 1], raw[4 * i + 2]);
    return v3(from_ufloat(p & 0x7ffu, 6), from_ufloat((p >> 11) & 0x7ffu, 6), from_ufloat(p >> 22, 5));
}
inline float luminance(V3 c) { return dot(c, v3(0.2126f, 0.7152f, 0.0722f)); }
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline V3 max3v(V3 a, V3 b) { return V3{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
}  // namespace

void post_temporal(const SrPostParams& p) {
    const int W = (int)p.width, H = (int)p.height;
    const uint32_t* history = p.accum[(p.frame_count + 1u) % 2u];   // lib.rs:1360-1361
    uint32_t* target = p.accum[p.frame_count % 2u];
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            auto tile = [&](int dx, int dy) {   // the LDS tile: clamp-to-edge reads of raw_rt_color (:82-86)
                return load_raw_color(p.raw_color, (size_t)clampi(y + dy, 0, H - 1) * W + clampi(x + dx, 0, W - 1));
            };
            const V3 current_color = tile(0, 0);
            V3 min_color = current_color, max_color = current_color;
            const float center_luma = luminance(current_color);
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (dx == 0 && dy == 0) continue;
                    const V3 nc = tile(dx, dy);
                    const float neighbor_luma = luminance(nc);
                    const float luma_threshold = max_f(center_luma * 5.0f, 0.08f);
                    if (fabsf(neighbor_luma - center_luma) < luma_threshold) { min_color = min3(min_color, nc); max_color = max3v(max_color, nc); }
                }
            const float uvx = ((float)x + 0.5f) / (float)W, uvy = ((float)y + 0.5f) / (float)H;   // :117
            const V2 motion = unpack_half_2x16(p.motion_vec_img[(size_t)y * W + x]);
            const float pux = uvx - motion.x, puy = uvy - motion.y;
            V3 accumulated = current_color;
            const bool off = (pux < 0.0f || puy < 0.0f) || (pux > 1.0f || puy > 1.0f);               // :123
            if (!off && p.frame_count > 2u) {
                // sample_history_bilinear (:42-58)
                const float px_ = pux * (float)W - 0.5f, py_ = puy * (float)H - 0.5f;
                const int bx = (int)floorf(px_), by = (int)floorf(py_);
                const float fx = px_ - (float)bx, fy = py_ - (float)by;
                auto hist = [&](int ix, int iy) { return load_b10g11r11(history, (size_t)clampi(iy, 0, H - 1) * W + clampi(ix, 0, W - 1)); };
                const V3 h00 = hist(bx, by), h10 = hist(bx + 1, by), h01 = hist(bx, by + 1), h11 = hist(bx + 1, by + 1);
                const V3 history_color = lerp3(lerp3(h00, h10, fx), lerp3(h01, h11, fx), fy);
                const V3 clamped = V3{clamp_f(history_color.x, min_color.x, max_color.x), clamp_f(history_color.y, min_color.y, max_color.y),
                                      clamp_f(history_color.z, min_color.z, max_color.z)};
                accumulated = lerp3(clamped, current_color, 0.14f);                                   // ACCUMULATION_FACTOR
            }
            target[(size_t)y * W + x] = pack_b10g11r11(accumulated.x, accumulated.y, accumulated.z);
        }
}

static void denoise_pass(const SrPostParams& p, const uint32_t* src, uint32_t* dst, int step_width) {
    const int W = (int)p.width, H = (int)p.height;
    static const float kernel[5] = {1.0f / 16.0f, 4.0f / 16.0f, 6.0f / 16.0f, 4.0f / 16.0f, 1.0f / 16.0f};
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t i = (size_t)y * W + x;
            const V3 center_color = load_b10g11r11(src, i);
            const float center_depth = f16_to_f32(p.depth_img[i]);
            if (center_depth >= 10000.0f) { dst[i] = pack_b10g11r11(center_color.x, center_color.y, center_color.z); continue; }   // :46-49
            const uint32_t nv = p.normal_img[i];
            const V3 center_normal = v3(unsnorm8(nv), unsnorm8(nv >> 8), unsnorm8(nv >> 16));
            const float center_roughness = unsnorm8(nv >> 24);
            const V3 center_diffuse = load_b10g11r11(p.diffuse_img, i);
            if (center_roughness < 0.1f) { dst[i] = pack_b10g11r11(center_color.x, center_color.y, center_color.z); continue; }     // :56-59
            const V3 center_illum = center_color / max3v(center_diffuse, v3(0.001f));     // vec / vec
            const float center_weight = kernel[2] * kernel[2];
            V3 sum_color = center_illum * center_weight;
            float sum_weight = center_weight;
            const float center_luma = luminance(center_illum);
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                    const int sx = x + dx * step_width, sy = y + dy * step_width;
                    if (sx < 0 || sy < 0 || sx >= W || sy >= H) continue;
                    const size_t j = (size_t)sy * W + sx;
                    const V3 sample_color = load_b10g11r11(src, j);
                    const float sample_depth = f16_to_f32(p.depth_img[j]);
                    const uint32_t sn = p.normal_img[j];
                    const V3 sample_normal = v3(unsnorm8(sn), unsnorm8(sn >> 8), unsnorm8(sn >> 16));
                    const V3 sample_diffuse = load_b10g11r11(p.diffuse_img, j);
                    const V3 sample_illum = sample_color / max3v(sample_diffuse, v3(0.001f));
                    const float sample_luma = luminance(sample_illum);
                    const float diffuse_diff = length(center_diffuse - sample_diffuse);           // distance()
                    const float luma_diff = fabsf(center_luma - sample_luma);
                    const float luma_sigma = max_f(center_luma, sample_luma) * 0.4f + 0.01f;
                    const float luma_ratio = luma_diff / luma_sigma;
                    const float combined_power = -fabsf(center_depth - sample_depth) * 8.0f
                                                 + (dot(center_normal, sample_normal) - 1.0f) * 80.0f
                                                 - diffuse_diff * 50.0f
                                                 - luma_ratio * luma_ratio;
                    const float weight = exp_f(combined_power) * kernel[dx + 2] * kernel[dy + 2];
                    sum_color += sample_illum * weight;
                    sum_weight += weight;
                }
            const V3 out = (sum_color / max_f(sum_weight, 0.0001f)) * center_diffuse;
            dst[i] = pack_b10g11r11(out.x, out.y, out.z);
        }
}

void post_denoise(const SrPostParams& p) {
    for (uint32_t pass = 0; pass < p.denoise_passes; pass++) {   // lib.rs:1817-1826
        const uint32_t* src = pass == 0 ? p.accum[p.frame_count % 2u] : (pass % 2u == 1u ? p.denoise[0] : p.denoise[1]);
        uint32_t* dst = (pass == 0 || pass % 2u == 0u) ? p.denoise[0] : p.denoise[1];
        denoise_pass(p, src, dst, 1 << pass);
    }
}

static inline float aces(float x) {   // postprocess.slang:14-18, per component
    x = clamp_f(x, 0.0f, 100.0f);
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return clamp_f((x * (a * x + b)) / (x * (c * x + d) + e), 0.0f, 1.0f);
}

void post_tonemap(const SrPostParams& p) {
    const size_t n = (size_t)p.width * p.height;
    const uint32_t* src = p.denoise[(p.denoise_passes - 1u) % 2u];   // lib.rs:1599-1601
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        V3 color = load_b10g11r11(src, (size_t)i);
        const bool bad = !(color.x == color.x) || !(color.y == color.y) || !(color.z == color.z) || std::isinf(color.x) || std::isinf(color.y) || std::isinf(color.z);
        if (bad) color = v3(0.0f);                                    // :33-35
        color = color * p.exposure;
        const V3 mapped = v3(aces(color.x), aces(color.y), aces(color.z));
        const float g = 1.0f / 2.2f;
        const V3 fin = v3(pow_f(mapped.x, g), pow_f(mapped.y, g), pow_f(mapped.z, g));
        p.output_rgba8[i] = pack_unorm_4x8(fin.x, fin.y, fin.z, 1.0f);   // R8G8B8A8_UNORM store
    }
}

}  // namespace orc
