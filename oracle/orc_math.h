// ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// CPU restatement of the arithmetic vocabulary the reference's Slang shaders use
// (shaders/rt_utils.slang, SURVEY.md appendix "Slang/HLSL intrinsic semantics").
// Parity status: UNPINNED — the reference holds no golden vectors for this path;
// this file is pinned only by the hand-derived known-answer tests in tests/golden/kat_rt_utils.json.
//
// Numerics contract: IEEE fp32, round-to-nearest-even, no FMA contraction except
// where fmaf is spelled out, evaluation strictly left to right as written in the Slang source.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 v3(float s) { return V3{s, s, s}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
// vector / scalar is DEFINED as vector * (1/scalar) with a correctly rounded reciprocal:
// SPIR-V OpFDiv may be lowered either way by a driver; fixing the cheaper form keeps GPU and oracle equal.
static inline V3 operator/(V3 a, float s) { float inv = 1.0f / s; return V3{a.x * inv, a.y * inv, a.z * inv}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
static inline V3 operator-(float s, V3 a) { return V3{s - a.x, s - a.y, s - a.z}; }
static inline V3 operator+(float s, V3 a) { return V3{s + a.x, s + a.y, s + a.z}; }
static inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
static inline V3& operator*=(V3& a, V3 b) { a = a * b; return a; }
static inline V3& operator/=(V3& a, float s) { a = a / s; return a; }

// dot / cross / length / normalize as the SPIR-V a Slang compiler emits would evaluate them in
// plain fp32: a sum of products taken left to right.
static inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float dot4(V4 a, V4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
static inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float length(V3 a) { return sqrtf(dot(a, a)); }
// normalize(0) = 0 * inf = NaN, the behaviour the shaders rely on for sky normals.
static inline V3 normalize(V3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }

static inline float min_f(float a, float b) { return fminf(a, b); }
static inline float max_f(float a, float b) { return fmaxf(a, b); }
static inline float clamp_f(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline V3 min3(V3 a, V3 b) { return V3{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
static inline float max_comp(V3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }  // max(r, max(g, b))
static inline V3 lerp3(V3 a, V3 b, float t) { return a + (b - a) * t; }     // HLSL lerp
static inline float frac(float x) { return x - floorf(x); }
static inline float smoothstep(float a, float b, float x) {
    float t = clamp_f((x - a) / (b - a), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
static inline V3 reflect(V3 i, V3 n) { return i - 2.0f * dot(n, i) * n; }  // i - 2*dot(n,i)*n
static inline V3 refract(V3 i, V3 n, float eta) {
    float ni = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - ni * ni);
    if (k < 0.0f) return v3(0.0f);
    return eta * i - (eta * ni + sqrtf(k)) * n;
}
static inline float pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }  // pow(x, 5.0)

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// ---- sin / cos / exp -----------------------------------------------------------------------
// The reference runs GLSL.std.450 Sin/Cos/Exp on a GPU driver (precision 2^-11 absolute for
// sin/cos per the Vulkan spec), i.e. their last bits are implementation-defined. The oracle and the
// HIP kernels both pin them to the same fp32 algorithm (Cody–Waite reduction + the classic Cephes
// single-precision minimax polynomials, all steps spelled with fmaf), so GPU and oracle agree bit
// for bit and both stay within 2 ulp of libm (tests/test_oracle_math.py).
static inline void sincos_f(float x, float* s, float* c) {
    float kf = rintf(x * 0.63661977236758134f);  // x * 2/pi
    int k = (int)kf;
    float r = fmaf(-kf, 1.5703125f, x);
    r = fmaf(-kf, 4.837512969970703125e-4f, r);
    r = fmaf(-kf, 7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sr = fmaf(r * z, ps, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cr = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    switch (k & 3) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}
static inline float sin_f(float x) { float s, c; sincos_f(x, &s, &c); return s; }
static inline float cos_f(float x) { float s, c; sincos_f(x, &s, &c); return c; }

static inline float exp_f(float x) {
    if (!(x == x)) return x;
    if (x > 88.72283935546875f) return INFINITY;
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
    int k1 = k / 2, k2 = k - k1;  // two exact power-of-two scalings, so denormal results round once
    y = y * u2f((uint32_t)(k1 + 127) << 23);
    return y * u2f((uint32_t)(k2 + 127) << 23);
}

// log (natural), pinned like sin/cos/exp: Cephes logf, polynomial spelled with fmaf. Used by pow(x, 1/2.2)
// of postprocess.slang:39, restated as exp(y * log(x)).
static inline float log_f(float x) {
    if (!(x == x)) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    if (x < 1.17549435e-38f) { x = x * 33554432.0f; e = -25; }   // denormal: scale by 2^25
    uint32_t u = f2u(x);
    e += (int)(u >> 23) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);               // mantissa in [0.5, 1)
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
static inline float pow_f(float x, float y) { return exp_f(y * log_f(x)); }

// ---- IEEE binary16 (f32tof16 / f16tof32, rt_utils.slang:89-94), round-to-nearest-even ---------
static inline uint32_t f32_to_f16(float f) {
    uint32_t u = f2u(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return sign | 0x7e00u | ((u >> 13) & 0x3ffu);  // NaN (quieted)
    if (u >= 0x477ff000u) return sign | 0x7c00u;                         // >= 65520 -> inf
    if (u >= 0x38800000u) {                                              // normal half
        uint32_t t = u - 0x38000000u;
        t = t + 0xfffu + ((t >> 13) & 1u);
        return sign | (t >> 13);
    }
    if (u < 0x33000000u) return sign;  // < 2^-25 -> 0 (2^-25 itself ties to even = 0)
    uint32_t e = u >> 23;              // 102..112
    uint32_t m = (u & 0x7fffffu) | 0x800000u;
    uint32_t s = 126u - e;             // 14..24
    uint32_t h = m >> s;
    uint32_t lower = m & ((1u << s) - 1u);
    uint32_t half = 1u << (s - 1u);
    if (lower > half || (lower == half && (h & 1u))) h++;
    return sign | h;
}
static inline float f16_to_f32(uint32_t h) {
    uint32_t s = (h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0) {
        float v = (float)m * 5.9604644775390625e-8f;  // m * 2^-24, exact
        return u2f(f2u(v) | s);
    }
    if (e == 31) return u2f(s | 0x7f800000u | (m << 13));
    return u2f(s | ((e + 112u) << 23) | (m << 13));
}

}  // namespace orc
