// ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Parity status: UNPINNED (see orc_math.h).
// CPU restatement of shaders/rt_utils.slang. Each function cites the lines it
// follows. Constants are copied digit for digit (pi is 3.14159 in most places and 3.14159265 in
// sample_ggx_vndf — that asymmetry is the reference's).
#pragma once
#include "orc_math.h"
#include "../include/sunray_hip.h"

namespace orc {

// rt_utils.slang:38-45
static inline uint32_t pcg_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct Rng { uint32_t seed; };
// rt_utils.slang:47-52
static inline Rng init_rng(uint32_t px, uint32_t py, uint32_t frame, uint32_t launch_w) {
    uint32_t pixel_idx = py * launch_w + px;
    return Rng{pcg_hash(pixel_idx ^ pcg_hash(frame))};
}
// rt_utils.slang:54-59. The literal 4294967295.0 is 2^32 in fp32, so the range is [0,1] inclusive.
static inline float rnd(Rng& rng) {
    rng.seed = rng.seed * 747796405u + 2891336453u;
    uint32_t word = ((rng.seed >> ((rng.seed >> 28u) + 4u)) ^ rng.seed) * 277803737u;
    uint32_t result = (word >> 22u) ^ word;
    return (float)result / 4294967296.0f;
}

// rt_utils.slang:68-76. round() -> rintf (round-half-even).
static inline uint32_t pack_snorm_2x16(float x, float y) {
    int ix = (int)rintf(clamp_f(x, -1.0f, 1.0f) * 32767.0f);
    int iy = (int)rintf(clamp_f(y, -1.0f, 1.0f) * 32767.0f);
    return ((uint32_t)ix & 0xFFFFu) | (((uint32_t)iy & 0xFFFFu) << 16);
}
static inline V2 unpack_snorm_2x16(uint32_t p) {
    int x = (int)(p << 16) >> 16;
    int y = (int)p >> 16;
    return V2{clamp_f((float)x / 32767.0f, -1.0f, 1.0f), clamp_f((float)y / 32767.0f, -1.0f, 1.0f)};
}
// rt_utils.slang:77-88
static inline uint32_t pack_unorm_4x8(float x, float y, float z, float w) {
    uint32_t cx = (uint32_t)rintf(clamp_f(x, 0.0f, 1.0f) * 255.0f);
    uint32_t cy = (uint32_t)rintf(clamp_f(y, 0.0f, 1.0f) * 255.0f);
    uint32_t cz = (uint32_t)rintf(clamp_f(z, 0.0f, 1.0f) * 255.0f);
    uint32_t cw = (uint32_t)rintf(clamp_f(w, 0.0f, 1.0f) * 255.0f);
    return cx | (cy << 8u) | (cz << 16u) | (cw << 24u);
}
static inline V4 unpack_unorm_4x8(uint32_t p) {
    return V4{(float)((p >> 0u) & 0xFFu) / 255.0f, (float)((p >> 8u) & 0xFFu) / 255.0f,
              (float)((p >> 16u) & 0xFFu) / 255.0f, (float)((p >> 24u) & 0xFFu) / 255.0f};
}
// rt_utils.slang:89-94
static inline uint32_t pack_half_2x16(float x, float y) { return f32_to_f16(x) | (f32_to_f16(y) << 16u); }
static inline V2 unpack_half_2x16(uint32_t p) { return V2{f16_to_f32(p & 0xFFFFu), f16_to_f32(p >> 16u)}; }

// rt_utils.slang:97-114 (octahedral)
static inline uint32_t pack_normal(V3 n) {
    n = n / (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
    float px, py;
    if (n.z >= 0.0f) { px = n.x; py = n.y; }
    else {
        px = (1.0f - fabsf(n.y)) * (n.x >= 0.0f ? 1.0f : -1.0f);
        py = (1.0f - fabsf(n.x)) * (n.y >= 0.0f ? 1.0f : -1.0f);
    }
    return pack_snorm_2x16(px, py);
}
static inline V3 unpack_normal(uint32_t p) {
    V2 v = unpack_snorm_2x16(p);
    V3 n = v3(v.x, v.y, 1.0f - fabsf(v.x) - fabsf(v.y));
    float t = max_f(-n.z, 0.0f);
    n.x += (n.x >= 0.0f) ? -t : t;
    n.y += (n.y >= 0.0f) ? -t : t;
    return normalize(n);
}

// rt_utils.slang:150-156
static inline void build_onb(V3 n, V3& t, V3& b) {
    float sign_n = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = -1.0f / (sign_n + n.z);
    float bb = n.x * n.y * a;
    t = v3(1.0f + sign_n * n.x * n.x * a, sign_n * bb, -sign_n * n.x);
    b = v3(bb, sign_n + n.y * n.y * a, -n.y);
}
// rt_utils.slang:158-163
static inline float smith_v_ggx(float NdotV, float NdotL, float alpha) {
    float a2 = alpha * alpha;
    float ggxV = NdotL * sqrtf(NdotV * NdotV * (1.0f - a2) + a2);
    float ggxL = NdotV * sqrtf(NdotL * NdotL * (1.0f - a2) + a2);
    return 0.5f / max_f(ggxV + ggxL, 0.0001f);
}
// rt_utils.slang:165-169
static inline float smith_g1_ggx(float NdotX, float alpha) {
    float a2 = alpha * alpha;
    float denom = NdotX + sqrtf(a2 + (1.0f - a2) * NdotX * NdotX);
    return 2.0f * NdotX / max_f(denom, 0.0001f);
}
// rt_utils.slang:171-177
static inline V3 get_random_bounce(V3 normal, float r1, float r2) {
    float phi = 2.0f * 3.14159f * r1;
    float r = sqrtf(r2);
    V3 u, v;
    build_onb(normal, u, v);
    float s, c;
    sincos_f(phi, &s, &c);
    return normalize(u * c * r + v * s * r + normal * sqrtf(1.0f - r2));
}
// rt_utils.slang:179-201
static inline V3 sample_ggx_vndf(V3 normal, V3 V_world, float roughness, float r1, float r2) {
    V3 T, B;
    build_onb(normal, T, B);
    V3 Vl = v3(dot(V_world, T), dot(V_world, B), dot(V_world, normal));
    float a = max_f(roughness * roughness, 0.001f);
    V3 Vh = normalize(v3(a * Vl.x, a * Vl.y, Vl.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    V3 T1 = lensq > 0.0f ? v3(-Vh.y, Vh.x, 0.0f) * (1.0f / sqrtf(lensq)) : v3(1.0f, 0.0f, 0.0f);
    V3 T2 = cross(Vh, T1);
    float rr = sqrtf(r1);
    float phi = 2.0f * 3.14159265f * r2;
    float sp, cp;
    sincos_f(phi, &sp, &cp);
    float t1 = rr * cp;
    float t2 = rr * sp;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    V3 Nh = t1 * T1 + t2 * T2 + sqrtf(max_f(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    V3 Hl = normalize(v3(a * Nh.x, a * Nh.y, max_f(0.0f, Nh.z)));
    return T * Hl.x + B * Hl.y + normal * Hl.z;
}
// rt_utils.slang:203-234. `emission` = light.emission.rgb.
static inline V3 eval_unshadowed_light(V3 hit_pos, V3 hit_normal, V3 V_view, V3 hit_albedo,
                                       float roughness, float metallic, V3 emission, V3 light_pos,
                                       V3 light_normal) {
    V3 L = light_pos - hit_pos;
    float dist = max_f(length(L), 0.0001f);
    L /= dist;
    float NdotL = max_f(dot(hit_normal, L), 0.0f);
    float cos_light = max_f(dot(light_normal, -L), 0.0f);
    if (NdotL <= 0.0f || cos_light <= 0.0f) return v3(0.0f);
    V3 H = normalize(V_view + L);
    float NdotH = max_f(dot(hit_normal, H), 0.0f);
    float VdotH = max_f(dot(V_view, H), 0.0f);
    float NdotV = max_f(dot(hit_normal, V_view), 0.001f);
    float a = roughness * roughness;
    float a2 = a * a;
    float denom = (NdotH * NdotH * (a2 - 1.0f) + 1.0f);
    float D = a2 / (3.14159f * denom * denom);
    V3 F0 = lerp3(v3(0.04f), hit_albedo, metallic);
    V3 F = F0 + (1.0f - F0) * pow5(1.0f - VdotH);
    float V_term = smith_v_ggx(NdotV, NdotL, a);
    V3 specular_brdf = D * V_term * F;
    V3 diffuse_brdf = hit_albedo * (1.0f - metallic) * (v3(1.0f) - F) / 3.14159f;
    float geometry = (NdotL * cos_light) / max_f(dist * dist, 0.0001f);
    return emission * (diffuse_brdf + specular_brdf) * geometry;
}
// rt_utils.slang:244-253
static inline void merge_reservoirs(SrReservoir& r, const SrReservoir& new_r, float p_hat_new, float random_val) {
    r.M += new_r.M;
    float weight = p_hat_new * new_r.W * new_r.M;
    r.w_sum += weight;
    if (random_val < (weight / max_f(r.w_sum, 0.0001f))) {
        r.light_idx = new_r.light_idx;
        for (int i = 0; i < 3; i++) { r.light_pos[i] = new_r.light_pos[i]; r.light_normal[i] = new_r.light_normal[i]; }
    }
}
// rt_utils.slang:255-263
static inline float gi_target_pdf(V3 shade_pos, V3 shade_normal, V3 albedo, float metallic, V3 sample_pos, V3 sample_radiance) {
    V3 w = sample_pos - shade_pos;
    float d = max_f(length(w), 0.0001f);
    w /= d;
    float NdotL = max_f(dot(shade_normal, w), 0.0f);
    V3 f_diffuse = albedo * (1.0f - metallic) / 3.14159f;
    V3 contrib = sample_radiance * f_diffuse * NdotL;
    return max_comp(contrib);
}
// rt_utils.slang:265-274
static inline void merge_reservoirs_gi(SrReservoirGI& r, const SrReservoirGI& new_r, float p_hat_new, float jacobian, float random_val) {
    r.M += new_r.M;
    float weight = p_hat_new * new_r.W * new_r.M * jacobian;
    r.w_sum += weight;
    if (random_val < (weight / max_f(r.w_sum, 0.0001f))) {
        for (int i = 0; i < 3; i++) { r.sample_pos[i] = new_r.sample_pos[i]; r.sample_radiance[i] = new_r.sample_radiance[i]; }
        r.sample_normal_packed = new_r.sample_normal_packed;
    }
}
// rt_utils.slang:278-281: rows of a 3x4 row-major transform dotted with (p,1).
static inline V3 transform_point(const SrTransform& x, V3 p) {
    const float* m = x.m;
    return v3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3] * 1.0f,
              ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7] * 1.0f,
              ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11] * 1.0f);
}

// ---- image-format conversions of the reference's G-buffer (lib.rs:1492-1516) --------------------
// R8G8B8A8_SNORM store: clamp to [-1,1], scale by 127, round-half-even; NaN -> 0.
static inline uint32_t snorm8(float x) {
    if (!(x == x)) return 0u;
    int i = (int)rintf(clamp_f(x, -1.0f, 1.0f) * 127.0f);
    return (uint32_t)i & 0xFFu;
}
static inline uint32_t pack_rgba8_snorm(float x, float y, float z, float w) {
    return snorm8(x) | (snorm8(y) << 8) | (snorm8(z) << 16) | (snorm8(w) << 24);
}
static inline float unsnorm8(uint32_t b) { int i = (int)(int8_t)(b & 0xFFu); return max_f((float)i / 127.0f, -1.0f); }
// unsigned small floats of B10G11R11_UFLOAT: 5 exponent bits (bias 15), MANT mantissa bits, no sign.
// Negative and -0 -> 0; NaN -> NaN; +inf -> inf; above the largest finite -> largest finite;
// otherwise round-to-nearest-even (rounding is implementation-defined in Vulkan; fixed here).
static inline uint32_t to_ufloat(float f, int MANT) {
    uint32_t u = f2u(f);
    const uint32_t max_finite = (30u << MANT) | ((1u << MANT) - 1u);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (31u << MANT) | 1u;
    if (u & 0x80000000u) return 0u;
    if (u == 0x7f800000u) return 31u << MANT;
    if (u >= 0x47800000u) return max_finite;  // >= 65536
    uint32_t r;
    if (u >= 0x38800000u) {
        uint32_t t = u - 0x38000000u;         // exponent rebias 127 -> 15
        uint32_t sh = 23u - (uint32_t)MANT;
        t = t + ((1u << (sh - 1u)) - 1u) + ((t >> sh) & 1u);
        r = t >> sh;
    } else {
        if (u < 0x30000000u) return 0u;
        uint32_t e = u >> 23;
        uint32_t m = (u & 0x7fffffu) | 0x800000u;
        // denormal unit = 2^(-14-MANT); result = round(m * 2^(e-150) / 2^(-14-MANT)) = m >> (136 - MANT - e)
        uint32_t s = 136u - (uint32_t)MANT - e;
        if (s > 31u) return 0u;
        r = m >> s;
        uint32_t lower = m & ((1u << s) - 1u);
        uint32_t half = 1u << (s - 1u);
        if (lower > half || (lower == half && (r & 1u))) r++;
    }
    return r > max_finite ? max_finite : r;
}
static inline uint32_t pack_b10g11r11(float r, float g, float b) {
    return to_ufloat(r, 6) | (to_ufloat(g, 6) << 11) | (to_ufloat(b, 5) << 22);
}
static inline float from_ufloat(uint32_t v, int MANT) {
    uint32_t e = v >> MANT, m = v & ((1u << MANT) - 1u);
    if (e == 0) return (float)m * u2f((uint32_t)(127 - 14 - MANT) << 23);
    if (e == 31) return u2f(0x7f800000u | (m << (23 - MANT)));
    return u2f(((e + 112u) << 23) | (m << (23 - MANT)));
}

}  // namespace orc
