// Device-side arithmetic vocabulary of the hot path (gfx950).
//
// Restates shaders/rt_utils.slang for HIP. Numerics contract:
// IEEE fp32, no FMA contraction (-ffp-contract=off) except where fmaf is written, correctly rounded
// divide/sqrt (-fhip-fp32-correctly-rounded-divide-sqrt), vector/scalar division = multiply by the
// rounded reciprocal, sin/cos/exp pinned to one fp32 algorithm. MFMA is not used anywhere on this
// path: it is branchy per-ray scalar math, not a contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/sunray_hip.h"

namespace srd {

struct f3 { float x, y, z; };
struct f2 { float x, y; };

#define SRD __device__ __forceinline__

SRD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
SRD f3 splat(float s) { return mk3(s, s, s); }
SRD f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
SRD void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
SRD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
SRD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
SRD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
SRD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
SRD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
SRD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
SRD f3 operator-(float s, f3 a) { return mk3(s - a.x, s - a.y, s - a.z); }
// vector / scalar := vector * (1 / scalar)
SRD f3 operator/(f3 a, float s) { float r = 1.0f / s; return mk3(a.x * r, a.y * r, a.z * r); }

SRD float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
SRD f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
SRD float len3(f3 a) { return sqrtf(dot3(a, a)); }
SRD f3 norm3(f3 a) { float r = 1.0f / sqrtf(dot3(a, a)); return a * r; }
SRD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
SRD f3 vmin(f3 a, f3 b) { return mk3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
SRD float maxc(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
SRD f3 lerp3(f3 a, f3 b, float t) { return a + (b - a) * t; }
SRD float fracf(float x) { return x - floorf(x); }
SRD float smoothstepf(float a, float b, float x) {
    float t = clampf((x - a) / (b - a), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
SRD f3 reflect3(f3 i, f3 n) { return i - (2.0f * dot3(n, i)) * n; }
SRD f3 refract3(f3 i, f3 n, float eta) {
    float ni = dot3(n, i);
    float k = 1.0f - eta * eta * (1.0f - ni * ni);
    if (k < 0.0f) return splat(0.0f);
    return eta * i - (eta * ni + sqrtf(k)) * n;
}
SRD float pow5f(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }

// sin/cos: Cody–Waite reduction by pi/2 + Cephes minimax polynomials, every step an explicit fma.
SRD void sincos_pinned(float x, float& s, float& c) {
    float kf = rintf(x * 0.63661977236758134f);
    int k = (int)kf;
    float r = fmaf(-kf, 1.5703125f, x);
    r = fmaf(-kf, 4.837512969970703125e-4f, r);
    r = fmaf(-kf, 7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sr = fmaf(r * z, ps, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cr = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    int q = k & 3;
    float s0 = (q & 1) ? cr : sr;
    float c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q == 1) || (q == 2)) ? -c0 : c0;
}
SRD float exp_pinned(float x) {
    if (!(x == x)) return x;
    if (x > 88.72283935546875f) return __builtin_inff();
    if (x < -103.9720840454f) return 0.0f;
    float kf = rintf(x * 1.44269504088896341f);
    float r = fmaf(-kf, 0.693359375f, x);
    r = fmaf(-kf, -2.12194440e-4f, r);
    float z = r * r;
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, z, r) + 1.0f;
    int k = (int)kf;
    int k1 = k / 2, k2 = k - k1;
    // two exact scalings by a power of two, as the oracle's two multiplications (each rounds like the product would, also
    // into the denormal range): v_ldexp_f32 instead of building 2^k and multiplying
    return ldexpf(ldexpf(y, k1), k2);
}

// natural log pinned like sin/cos/exp (Cephes logf, polynomial as explicit fma); pow(x, y) = exp(y * log(x))
// restates the only non-integer pow of the path (postprocess.slang:39, gamma 1/2.2).
SRD float log_pinned(float x) {
    if (!(x == x)) return x;
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    int e = 0;
    if (x < 1.17549435e-38f) { x = x * 33554432.0f; e = -25; }
    const uint32_t u = __float_as_uint(x);
    e += (int)(u >> 23) - 126;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else m = m - 1.0f;
    const float z = m * m;
    float y = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    y = fmaf(y, m, 1.1676998740e-1f);
    y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f);
    y = fmaf(y, m, -1.6668057665e-1f);
    y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f);
    y = fmaf(y, m, 3.3333331174e-1f);
    y = y * m * z;
    const float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

SRD float pow_pinned(float x, float y) { return exp_pinned(y * log_pinned(x)); }

// binary16 conversions: v_cvt_f16_f32 / v_cvt_f32_f16 (round-to-nearest-even, denormals kept)
SRD uint32_t f32_to_f16_bits(float f) { return (uint32_t)__half_as_ushort(__float2half_rn(f)); }
SRD float f16_bits_to_f32(uint32_t h) { return __half2float(__ushort_as_half((unsigned short)(h & 0xFFFFu))); }

// ---- rt_utils.slang:38-59 RNG ---------------------------------------------------------------
SRD uint32_t pcg_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
SRD uint32_t init_rng(uint32_t px, uint32_t py, uint32_t frame, uint32_t launch_w) {
    return pcg_hash((py * launch_w + px) ^ pcg_hash(frame));
}
SRD float rnd(uint32_t& seed) {
    seed = seed * 747796405u + 2891336453u;
    uint32_t word = ((seed >> ((seed >> 28u) + 4u)) ^ seed) * 277803737u;
    uint32_t result = (word >> 22u) ^ word;
    return (float)result * 2.3283064365386963e-10f;  // / 2^32, exact (the fp32 value of 4294967295.0)
}

// ---- rt_utils.slang:68-114 pack / unpack ----------------------------------------------------
SRD uint32_t pack_snorm_2x16(float x, float y) {
    int ix = (int)rintf(clampf(x, -1.0f, 1.0f) * 32767.0f);
    int iy = (int)rintf(clampf(y, -1.0f, 1.0f) * 32767.0f);
    return ((uint32_t)ix & 0xFFFFu) | (((uint32_t)iy & 0xFFFFu) << 16);
}
SRD f2 unpack_snorm_2x16(uint32_t p) {
    int x = (int)(p << 16) >> 16;
    int y = (int)p >> 16;
    f2 r;
    r.x = clampf((float)x / 32767.0f, -1.0f, 1.0f);
    r.y = clampf((float)y / 32767.0f, -1.0f, 1.0f);
    return r;
}
SRD uint32_t pack_unorm_4x8(float x, float y, float z, float w) {
    uint32_t cx = (uint32_t)rintf(clampf(x, 0.0f, 1.0f) * 255.0f);
    uint32_t cy = (uint32_t)rintf(clampf(y, 0.0f, 1.0f) * 255.0f);
    uint32_t cz = (uint32_t)rintf(clampf(z, 0.0f, 1.0f) * 255.0f);
    uint32_t cw = (uint32_t)rintf(clampf(w, 0.0f, 1.0f) * 255.0f);
    return cx | (cy << 8) | (cz << 16) | (cw << 24);
}
SRD f3 unpack_unorm_rgb(uint32_t p) {
    return mk3((float)(p & 0xFFu) / 255.0f, (float)((p >> 8) & 0xFFu) / 255.0f, (float)((p >> 16) & 0xFFu) / 255.0f);
}
SRD uint32_t pack_half_2x16(float x, float y) { return f32_to_f16_bits(x) | (f32_to_f16_bits(y) << 16); }
SRD f2 unpack_half_2x16(uint32_t p) { f2 r; r.x = f16_bits_to_f32(p & 0xFFFFu); r.y = f16_bits_to_f32(p >> 16); return r; }

SRD uint32_t pack_normal(f3 n) {
    n = n / (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
    float px, py;
    if (n.z >= 0.0f) { px = n.x; py = n.y; }
    else {
        px = (1.0f - fabsf(n.y)) * (n.x >= 0.0f ? 1.0f : -1.0f);
        py = (1.0f - fabsf(n.x)) * (n.y >= 0.0f ? 1.0f : -1.0f);
    }
    return pack_snorm_2x16(px, py);
}
SRD f3 unpack_normal(uint32_t p) {
    f2 v = unpack_snorm_2x16(p);
    f3 n = mk3(v.x, v.y, 1.0f - fabsf(v.x) - fabsf(v.y));
    float t = fmaxf(-n.z, 0.0f);
    n.x += (n.x >= 0.0f) ? -t : t;
    n.y += (n.y >= 0.0f) ? -t : t;
    return norm3(n);
}

// ---- G-buffer formats of the reference (lib.rs:1492-1516) -------------------------------------
SRD uint32_t snorm8(float x) {
    if (!(x == x)) return 0u;
    int i = (int)rintf(clampf(x, -1.0f, 1.0f) * 127.0f);
    return (uint32_t)i & 0xFFu;
}
SRD uint32_t pack_rgba8_snorm(float x, float y, float z, float w) {
    return snorm8(x) | (snorm8(y) << 8) | (snorm8(z) << 16) | (snorm8(w) << 24);
}
SRD float unsnorm8(uint32_t b) { int i = (int)(int8_t)(b & 0xFFu); return fmaxf((float)i / 127.0f, -1.0f); }
// unsigned small float (5-bit exponent, MANT-bit mantissa), round-to-nearest-even
template <int MANT>
SRD uint32_t to_ufloat(float f) {
    uint32_t u = __float_as_uint(f);
    const uint32_t max_finite = (30u << MANT) | ((1u << MANT) - 1u);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (31u << MANT) | 1u;
    if (u & 0x80000000u) return 0u;
    if (u == 0x7f800000u) return 31u << MANT;
    if (u >= 0x47800000u) return max_finite;
    uint32_t r;
    if (u >= 0x38800000u) {
        uint32_t t = u - 0x38000000u;
        const uint32_t sh = 23u - MANT;
        t = t + ((1u << (sh - 1u)) - 1u) + ((t >> sh) & 1u);
        r = t >> sh;
    } else {
        if (u < 0x30000000u) return 0u;
        uint32_t e = u >> 23;
        uint32_t m = (u & 0x7fffffu) | 0x800000u;
        uint32_t s = 136u - MANT - e;
        if (s > 31u) return 0u;
        r = m >> s;
        uint32_t lower = m & ((1u << s) - 1u);
        uint32_t half = 1u << (s - 1u);
        if (lower > half || (lower == half && (r & 1u))) r++;
    }
    return r > max_finite ? max_finite : r;
}
SRD uint32_t pack_b10g11r11(float r, float g, float b) {
    return to_ufloat<6>(r) | (to_ufloat<6>(g) << 11) | (to_ufloat<5>(b) << 22);
}

template <int MANT>
SRD float from_ufloat(uint32_t v) {
    const uint32_t e = v >> MANT, m = v & ((1u << MANT) - 1u);
    if (e == 0u) return (float)m * __uint_as_float((uint32_t)(127 - 14 - MANT) << 23);
    if (e == 31u) return __uint_as_float(0x7f800000u | (m << (23 - MANT)));
    return __uint_as_float(((e + 112u) << 23) | (m << (23 - MANT)));
}
SRD f3 unpack_b10g11r11(uint32_t v) { return mk3(from_ufloat<6>(v & 0x7ffu), from_ufloat<6>((v >> 11) & 0x7ffu), from_ufloat<5>(v >> 22)); }

// ---- rt_utils.slang:150-234 BRDF helpers ------------------------------------------------------
SRD void build_onb(f3 n, f3& t, f3& b) {
    float sign_n = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = -1.0f / (sign_n + n.z);
    float bb = n.x * n.y * a;
    t = mk3(1.0f + sign_n * n.x * n.x * a, sign_n * bb, -sign_n * n.x);
    b = mk3(bb, sign_n + n.y * n.y * a, -n.y);
}
SRD float smith_v_ggx(float NdotV, float NdotL, float alpha) {
    float a2 = alpha * alpha;
    float ggxV = NdotL * sqrtf(NdotV * NdotV * (1.0f - a2) + a2);
    float ggxL = NdotV * sqrtf(NdotL * NdotL * (1.0f - a2) + a2);
    return 0.5f / fmaxf(ggxV + ggxL, 0.0001f);
}
SRD float smith_g1_ggx(float NdotX, float alpha) {
    float a2 = alpha * alpha;
    float denom = NdotX + sqrtf(a2 + (1.0f - a2) * NdotX * NdotX);
    return 2.0f * NdotX / fmaxf(denom, 0.0001f);
}
SRD f3 get_random_bounce(f3 normal, float r1, float r2) {
    float phi = 2.0f * 3.14159f * r1;
    float r = sqrtf(r2);
    f3 u, v;
    build_onb(normal, u, v);
    float s, c;
    sincos_pinned(phi, s, c);
    return norm3(u * c * r + v * s * r + normal * sqrtf(1.0f - r2));
}
SRD f3 sample_ggx_vndf(f3 normal, f3 V_world, float roughness, float r1, float r2) {
    f3 T, B;
    build_onb(normal, T, B);
    f3 Vl = mk3(dot3(V_world, T), dot3(V_world, B), dot3(V_world, normal));
    float a = fmaxf(roughness * roughness, 0.001f);
    f3 Vh = norm3(mk3(a * Vl.x, a * Vl.y, Vl.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    f3 T1 = lensq > 0.0f ? mk3(-Vh.y, Vh.x, 0.0f) * (1.0f / sqrtf(lensq)) : mk3(1.0f, 0.0f, 0.0f);
    f3 T2 = cross3(Vh, T1);
    float rr = sqrtf(r1);
    float phi = 2.0f * 3.14159265f * r2;
    float sp, cp;
    sincos_pinned(phi, sp, cp);
    float t1 = rr * cp;
    float t2 = rr * sp;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    f3 Nh = t1 * T1 + t2 * T2 + sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    f3 Hl = norm3(mk3(a * Nh.x, a * Nh.y, fmaxf(0.0f, Nh.z)));
    return T * Hl.x + B * Hl.y + normal * Hl.z;
}
SRD f3 eval_unshadowed_light(f3 hit_pos, f3 hit_normal, f3 V_view, f3 hit_albedo, float roughness, float metallic,
                             f3 emission, f3 light_pos, f3 light_normal) {
    f3 L = light_pos - hit_pos;
    float dist = fmaxf(len3(L), 0.0001f);
    L = L / dist;
    float NdotL = fmaxf(dot3(hit_normal, L), 0.0f);
    float cos_light = fmaxf(dot3(light_normal, -L), 0.0f);
    if (NdotL <= 0.0f || cos_light <= 0.0f) return splat(0.0f);
    f3 H = norm3(V_view + L);
    float NdotH = fmaxf(dot3(hit_normal, H), 0.0f);
    float VdotH = fmaxf(dot3(V_view, H), 0.0f);
    float NdotV = fmaxf(dot3(hit_normal, V_view), 0.001f);
    float a = roughness * roughness;
    float a2 = a * a;
    float denom = (NdotH * NdotH * (a2 - 1.0f) + 1.0f);
    float D = a2 / (3.14159f * denom * denom);
    f3 F0 = lerp3(splat(0.04f), hit_albedo, metallic);
    f3 F = F0 + (1.0f - F0) * pow5f(1.0f - VdotH);
    float V_term = smith_v_ggx(NdotV, NdotL, a);
    f3 specular_brdf = (D * V_term) * F;
    f3 diffuse_brdf = hit_albedo * (1.0f - metallic) * (splat(1.0f) - F) / 3.14159f;
    float geometry = (NdotL * cos_light) / fmaxf(dist * dist, 0.0001f);
    return emission * (diffuse_brdf + specular_brdf) * geometry;
}
SRD float gi_target_pdf(f3 shade_pos, f3 shade_normal, f3 albedo, float metallic, f3 sample_pos, f3 sample_radiance) {
    f3 w = sample_pos - shade_pos;
    float d = fmaxf(len3(w), 0.0001f);
    w = w / d;
    float NdotL = fmaxf(dot3(shade_normal, w), 0.0f);
    f3 f_diffuse = albedo * (1.0f - metallic) / 3.14159f;
    f3 contrib = sample_radiance * f_diffuse * NdotL;
    return maxc(contrib);
}
SRD void merge_reservoirs(SrReservoir& r, const SrReservoir& nr, float p_hat_new, float random_val) {
    r.M += nr.M;
    float weight = p_hat_new * nr.W * nr.M;
    r.w_sum += weight;
    if (random_val < (weight / fmaxf(r.w_sum, 0.0001f))) {
        r.light_idx = nr.light_idx;
        r.light_pos[0] = nr.light_pos[0]; r.light_pos[1] = nr.light_pos[1]; r.light_pos[2] = nr.light_pos[2];
        r.light_normal[0] = nr.light_normal[0]; r.light_normal[1] = nr.light_normal[1]; r.light_normal[2] = nr.light_normal[2];
    }
}
// (returns whether the new sample was taken: the final pass wants to know whose sample its combined reservoir ends up with)
SRD bool merge_reservoirs_gi(SrReservoirGI& r, const SrReservoirGI& nr, float p_hat_new, float jacobian, float random_val) {
    r.M += nr.M;
    float weight = p_hat_new * nr.W * nr.M * jacobian;
    r.w_sum += weight;
    const bool taken = random_val < (weight / fmaxf(r.w_sum, 0.0001f));
    if (taken) {
        r.sample_pos[0] = nr.sample_pos[0]; r.sample_pos[1] = nr.sample_pos[1]; r.sample_pos[2] = nr.sample_pos[2];
        r.sample_radiance[0] = nr.sample_radiance[0]; r.sample_radiance[1] = nr.sample_radiance[1]; r.sample_radiance[2] = nr.sample_radiance[2];
        r.sample_normal_packed = nr.sample_normal_packed;
    }
    return taken;
}
// rt_utils.slang:278-281
SRD f3 transform_point(const float* m, f3 p) {
    return mk3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3] * 1.0f,
               ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7] * 1.0f,
               ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11] * 1.0f);
}

}  // namespace srd
