// ORACLE — test infrastructure only (see oracle/README header in orc_api.cpp). CPU restatement of the
// texture fetch the path performs: `tex.SampleLevel(smp, uv, 0.0)` (rt_utils.slang:121-133) on an
// R8G8B8A8_UNORM image with a single mip level (image/mod.rs:96-107) through a sampler with
// min_lod = max_lod = 0 (image/sampler.rs:77-94), i.e. the MAGNIFICATION filter at LOD 0.
//
// The filtering itself is the GPU's fixed-function texture unit in the reference (third-party: the
// vendor's Vulkan driver/hardware; its weight precision is implementation-defined), so this file FIXES
// one exact fp32 definition following the Vulkan texel-filtering equations, and the HIP kernels use
// the same one (rt_device.h sample_texture):
//   s guard      : non-finite s -> 0
//   wrap (coord) : REPEAT s -= floor(s) | MIRRORED_REPEAT s -= 2*floor(s/2) | CLAMP_TO_EDGE s in [-1,2]
//   u = s * width;  NEAREST i = floor(u);  LINEAR u -= 0.5, i0 = floor(u), i1 = i0+1, a = u - i0
//   wrap (index) : REPEAT i mod n | MIRRORED m = i mod 2n, m < n ? m : 2n-1-m | CLAMP clamp(i,0,n-1)
//   texel -> float : byte / 255.0f  (UNORM)
//   LINEAR        : (t00*(1-a) + t10*a)*(1-b) + (t01*(1-a) + t11*a)*b, each operation rounded (no fma)
// parity unpinned: the reference holds no texture-fetch vectors.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "../include/sunray_hip.h"
#include "orc_math.h"

namespace orc {

struct Image {
    uint32_t w = 0, h = 0;
    std::vector<uint32_t> rgba;  // R in the low byte
};

static inline float wrap_coord(float s, uint32_t mode) {
    if (!(fabsf(s) < 3.0e38f)) s = 0.0f;
    if (mode == SR_ADDRESS_REPEAT) return s - floorf(s);
    if (mode == SR_ADDRESS_MIRRORED_REPEAT) return s - 2.0f * floorf(s * 0.5f);
    return fminf(fmaxf(s, -1.0f), 2.0f);
}
static inline uint32_t wrap_index(int i, int n, uint32_t mode) {
    if (mode == SR_ADDRESS_REPEAT) { int m = i % n; return (uint32_t)(m < 0 ? m + n : m); }
    if (mode == SR_ADDRESS_MIRRORED_REPEAT) {
        int m = i % (2 * n);
        if (m < 0) m += 2 * n;
        return (uint32_t)(m < n ? m : 2 * n - 1 - m);
    }
    return (uint32_t)(i < 0 ? 0 : (i > n - 1 ? n - 1 : i));
}
static inline V4 texel_unorm(uint32_t p) {
    V4 r;
    r.x = (float)(p & 0xFFu) / 255.0f; r.y = (float)((p >> 8) & 0xFFu) / 255.0f;
    r.z = (float)((p >> 16) & 0xFFu) / 255.0f; r.w = (float)(p >> 24) / 255.0f;
    return r;
}

static inline V4 sample_image(const Image& img, const SrSamplerDesc& smp, float s, float t) {
    const int W = (int)img.w, H = (int)img.h;
    float u = wrap_coord(s, smp.address_mode_u) * (float)W;
    float v = wrap_coord(t, smp.address_mode_v) * (float)H;
    if (smp.mag_filter == SR_FILTER_NEAREST) {
        uint32_t i = wrap_index((int)floorf(u), W, smp.address_mode_u);
        uint32_t j = wrap_index((int)floorf(v), H, smp.address_mode_v);
        return texel_unorm(img.rgba[(size_t)j * img.w + i]);
    }
    u = u - 0.5f; v = v - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    uint32_t i0 = wrap_index((int)fu, W, smp.address_mode_u), i1 = wrap_index((int)fu + 1, W, smp.address_mode_u);
    uint32_t j0 = wrap_index((int)fv, H, smp.address_mode_v), j1 = wrap_index((int)fv + 1, H, smp.address_mode_v);
    V4 t00 = texel_unorm(img.rgba[(size_t)j0 * img.w + i0]), t10 = texel_unorm(img.rgba[(size_t)j0 * img.w + i1]);
    V4 t01 = texel_unorm(img.rgba[(size_t)j1 * img.w + i0]), t11 = texel_unorm(img.rgba[(size_t)j1 * img.w + i1]);
    float na = 1.0f - a, nb = 1.0f - b;
    V4 r;
    r.x = (t00.x * na + t10.x * a) * nb + (t01.x * na + t11.x * a) * b;
    r.y = (t00.y * na + t10.y * a) * nb + (t01.y * na + t11.y * a) * b;
    r.z = (t00.z * na + t10.z * a) * nb + (t01.z * na + t11.z * a) * b;
    r.w = (t00.w * na + t10.w * a) * nb + (t01.w * na + t11.w * a) * b;
    return r;
}

}  // namespace orc
