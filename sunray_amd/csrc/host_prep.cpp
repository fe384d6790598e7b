// Host-side data preparation of the hot path: what the reference's Rust host
// computes before the two ray-tracing passes run. Pure CPU code, no HIP calls.
#include <cmath>
#include <cstring>

#include "bvh_layout.h"
#include "host.h"

namespace srh {

// 4x4 helpers, row-major storage (= "each float4 is a row", lib.rs:1042-1047)
static void mul44(const float* a, const float* b, float* out) {
    for (int r = 0; r < 4; r++) {
        for (int c = 0; c < 4; c++) {
            float acc = a[4 * r] * b[c];
            acc = acc + a[4 * r + 1] * b[4 + c];
            acc = acc + a[4 * r + 2] * b[8 + c];
            acc = acc + a[4 * r + 3] * b[12 + c];
            out[4 * r + c] = acc;
        }
    }
}

// General 4x4 inverse by cofactors (what nalgebra's try_inverse does for a 4x4, camera.rs:53-54).
static bool invert44(const float* m, float* out) {
    float c[16];
    c[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    c[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    c[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    c[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    c[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    c[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    c[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    c[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    c[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    c[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    c[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    c[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    c[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    c[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    c[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    c[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * c[0] + m[1] * c[4] + m[2] * c[8] + m[3] * c[12];
    if (det == 0.0f) return false;
    const float r = 1.0f / det;
    for (int i = 0; i < 16; i++) out[i] = c[i] * r;
    return true;
}

// Camera::as_matrices (camera.rs:33-63) + the transposed upload with the history matrix injected
// (lib.rs:1017-1048). nalgebra 0.35.0 is not vendored in the reference: look_at_rh / Perspective3 /
// try_inverse are restated from their published definitions.
bool Camera::as_matrices(uint32_t width, uint32_t height, const float* prev_view_proj16, SrMatrices* out) const {
    struct v { float x, y, z; };
    auto sub = [](v a, v b) { return v{a.x - b.x, a.y - b.y, a.z - b.z}; };
    auto dot = [](v a, v b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; };
    auto cross = [](v a, v b) { return v{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
    auto normalize = [&](v a) { float r = 1.0f / sqrtf(dot(a, a)); return v{a.x * r, a.y * r, a.z * r}; };
    const v eye{position[0], position[1], position[2]};
    const v tgt{target[0], target[1], target[2]};
    const v up{0.0f, 1.0f, 0.0f};
    const v f = normalize(sub(tgt, eye));
    const v s = normalize(cross(f, up));
    const v u = cross(s, f);
    const float view[16] = {s.x, s.y, s.z, -dot(s, eye), u.x, u.y, u.z, -dot(u, eye),
                            -f.x, -f.y, -f.z, dot(f, eye), 0.0f, 0.0f, 0.0f, 1.0f};
    const float aspect = (float)width / (float)height;
    const float fovy = fov_y * (3.14159265358979323846f / 180.0f);
    const float znear = 0.1f, zfar = 100.0f;
    const float tan_half = tanf(fovy / 2.0f);
    float proj[16];
    memset(proj, 0, sizeof(proj));
    proj[0] = 1.0f / (aspect * tan_half);
    proj[5] = 1.0f / tan_half;
    proj[10] = (zfar + znear) / (znear - zfar);
    proj[11] = (2.0f * zfar * znear) / (znear - zfar);
    proj[14] = -1.0f;
    proj[5] *= -1.0f;
    if (!invert44(view, out->view_inverse)) return false;
    if (!invert44(proj, out->proj_inverse)) return false;
    mul44(proj, view, out->view_proj);
    if (prev_view_proj16) memcpy(out->prev_view_proj, prev_view_proj16, 64);
    else memset(out->prev_view_proj, 0, 64);
    return true;
}

// Material::new (resources/material.rs:52-92) with the runtime-mesh resolver (lib.rs:937-943)
void material_new(const float base_color[4], float metallic, float roughness, const float emissive_factor[3],
                  float emissive_strength, float transmission, float ior, SrMaterial* out) {
    memset(out, 0, sizeof(*out));
    memcpy(out->base_color_value, base_color, 16);
    out->metallic_factor = metallic;
    out->roughness_factor = roughness;
    memcpy(out->emissive_factor, emissive_factor, 12);
    out->emissive_factor[3] = emissive_strength;
    out->alpha_mode = 0;
    out->alpha_cutoff = 0.0f;
    out->transmission_factor = transmission;
    out->ior = ior;
    uint32_t* tex = &out->base_color_image;
    for (int i = 0; i < 10; i++) tex[i] = SR_NULL_TEXTURE;
}

// Emissive-triangle derivation of Renderer::load_mesh (lib.rs:901-925)
void emissive_triangles_from_mesh(const SrVertex* vertices, const uint32_t* indices, uint32_t n_indices,
                                  const SrMaterial& material, std::vector<SrEmissiveTriangle>& out) {
    const float s = material.emissive_factor[3];
    const float e[3] = {material.emissive_factor[0] * s, material.emissive_factor[1] * s, material.emissive_factor[2] * s};
    if (!(e[0] > 0.0f || e[1] > 0.0f || e[2] > 0.0f)) return;
    for (uint32_t t = 0; t + 2 < n_indices; t += 3) {
        SrEmissiveTriangle et;
        memset(&et, 0, sizeof(et));
        memcpy(et.v0, vertices[indices[t]].position, 12);
        memcpy(et.v1, vertices[indices[t + 1]].position, 12);
        memcpy(et.v2, vertices[indices[t + 2]].position, 12);
        memcpy(et.emission, e, 12);
        out.push_back(et);
    }
}

// (float3x3)WorldToObject3x4 of an instance transform: adjugate / determinant
void world_to_object_3x3(const SrTransform& t, float o[9]) {
    const float* m = t.m;
    const float a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
    const float c00 = a11 * a22 - a12 * a21;
    const float c01 = a12 * a20 - a10 * a22;
    const float c02 = a10 * a21 - a11 * a20;
    const float det = (a00 * c00 + a01 * c01) + a02 * c02;
    const float r = 1.0f / det;
    o[0] = c00 * r; o[1] = (a02 * a21 - a01 * a22) * r; o[2] = (a01 * a12 - a02 * a11) * r;
    o[3] = c01 * r; o[4] = (a00 * a22 - a02 * a20) * r; o[5] = (a02 * a10 - a00 * a12) * r;
    o[6] = c02 * r; o[7] = (a01 * a20 - a00 * a21) * r; o[8] = (a00 * a11 - a01 * a10) * r;
}

// pack_unorm_4x8 / pack_half_2x16 (rt_utils.slang:77-94) on the host: round-half-even, IEEE binary16 RNE
static uint32_t unorm8(float v) {
    const float c = fminf(fmaxf(v, 0.0f), 1.0f);
    return (uint32_t)rintf(c * 255.0f);
}
uint32_t pack_unorm_4x8(float x, float y, float z, float w) { return unorm8(x) | (unorm8(y) << 8) | (unorm8(z) << 16) | (unorm8(w) << 24); }
static uint32_t half_bits(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return sign | 0x7e00u | ((u >> 13) & 0x3ffu);
    if (u >= 0x477ff000u) return sign | 0x7c00u;
    if (u >= 0x38800000u) { uint32_t t = u - 0x38000000u; t += 0xfffu + ((t >> 13) & 1u); return sign | (t >> 13); }
    if (u < 0x33000000u) return sign;
    const uint32_t e = u >> 23, m = (u & 0x7fffffu) | 0x800000u, sh = 126u - e;
    uint32_t h = m >> sh;
    const uint32_t lower = m & ((1u << sh) - 1u), half = 1u << (sh - 1u);
    if (lower > half || (lower == half && (h & 1u))) h++;
    return sign | h;
}
uint32_t pack_half_2x16(float x, float y) { return half_bits(x) | (half_bits(y) << 16); }

// ResourceManager::frame_instance_data (resource_manager.rs:216-267) followed by the dummy-entry
// padding of Renderer::render (lib.rs:1058-1081).
bool frame_instance_data(const std::vector<HostMesh>& meshes, const std::map<uint64_t, uint32_t>& slots,
                         const uint64_t* keys, const uint32_t* counts, uint32_t n_keys, const SrTransform* xforms,
                         FrameInstanceData& out, std::string& err) {
    out.instances.clear();
    out.transforms.clear();
    out.emissive_entries.clear();
    uint32_t next = 0, tri_offset = 0;
    for (uint32_t k = 0; k < n_keys; k++) {
        auto it = slots.find(keys[k]);
        if (it == slots.end()) {
            err = "frame_instance_data: instance references a BLAS key that was never loaded";
            return false;
        }
        const uint32_t mesh_info_slot = it->second;
        const HostMesh& mesh = meshes[mesh_info_slot];
        for (uint32_t c = 0; c < counts[k]; c++) {
            const uint32_t instance_index = (uint32_t)out.transforms.size();
            const SrTransform& xf = xforms[next++];
            out.transforms.push_back(xf);
            HostInstance inst;
            inst.mesh_slot = mesh_info_slot;
            inst.o2w = xf;
            world_to_object_3x3(xf, inst.w2o);
            inst.tri_offset = tri_offset;
            tri_offset += mesh.n_indices / 3;
            out.instances.push_back(inst);
            for (uint32_t tri_slot : mesh.emissive_slots)
                out.emissive_entries.push_back(SrEmissiveIndirectionEntry{tri_slot, instance_index});
        }
    }
    out.n_triangles = tri_offset;
    if (out.transforms.empty()) out.transforms.push_back(SrTransform{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}});
    if (out.emissive_entries.empty()) out.emissive_entries.push_back(SrEmissiveIndirectionEntry{0, 0});
    return true;
}

// DevLight table: what the shaders recompute per use from (emissive triangle, entity transform), in the
// shaders' own operation order so the stored floats equal the ones the device would compute.
void light_table(const FrameInstanceData& fid, const std::vector<SrEmissiveTriangle>& emissive_tris, std::vector<float>& out) {
    out.assign(fid.emissive_entries.size() * 16, 0.0f);
    for (size_t i = 0; i < fid.emissive_entries.size(); i++) {
        const SrEmissiveIndirectionEntry& e = fid.emissive_entries[i];
        const SrEmissiveTriangle& t = emissive_tris[e.blas_tri_index];
        const float* m = fid.transforms[e.entity_id].m;
        float w[3][3];
        const float* v[3] = {t.v0, t.v1, t.v2};
        for (int k = 0; k < 3; k++) {   // transform_point (rt_utils.slang:278-281)
            w[k][0] = ((m[0] * v[k][0] + m[1] * v[k][1]) + m[2] * v[k][2]) + m[3] * 1.0f;
            w[k][1] = ((m[4] * v[k][0] + m[5] * v[k][1]) + m[6] * v[k][2]) + m[7] * 1.0f;
            w[k][2] = ((m[8] * v[k][0] + m[9] * v[k][1]) + m[10] * v[k][2]) + m[11] * 1.0f;
        }
        const float e1[3] = {w[1][0] - w[0][0], w[1][1] - w[0][1], w[1][2] - w[0][2]};
        const float e2[3] = {w[2][0] - w[0][0], w[2][1] - w[0][1], w[2][2] - w[0][2]};
        const float c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const float dd = (c[0] * c[0] + c[1] * c[1]) + c[2] * c[2];
        const float len = sqrtf(dd);
        const float r = 1.0f / len;
        float* q = &out[i * 16];
        q[0] = w[0][0]; q[1] = w[0][1]; q[2] = w[0][2]; q[3] = 0.5f * len;
        q[4] = w[1][0]; q[5] = w[1][1]; q[6] = w[1][2]; q[7] = c[0] * r;
        q[8] = w[2][0]; q[9] = w[2][1]; q[10] = w[2][2]; q[11] = c[1] * r;
        q[12] = t.emission[0]; q[13] = t.emission[1]; q[14] = t.emission[2]; q[15] = c[2] * r;
    }
}

// ---- AsState (acceleration_structure/mod.rs:62-148) ------------------------------------------------------
namespace {
constexpr uint32_t kMaxUpdatesBeforeRebuild = 8;   // MAX_UPDATES_BEFORE_REBUILD
constexpr uint32_t kFramesToSettle = 16;           // FRAMES_TO_SETTLE
}

void as_state_initial(uint32_t build_type, SrAsState* out) {
    out->changing = build_type == SR_BUILD_RAPIDLY_CHANGING ? 1u : 0u;      // AsState::initial (:86-91)
    out->frames_without_changes = 0;
    out->number_of_updates_since_last_rebuild = 0;
    out->_pad = 0;
}

uint32_t as_state_next_op(const SrAsState& s, bool inputs_changed) {           // AsState::next_op (:97-114)
    if (!s.changing) return inputs_changed ? SR_OP_UPDATE : SR_OP_NONE;
    if (inputs_changed) return s.number_of_updates_since_last_rebuild >= kMaxUpdatesBeforeRebuild ? SR_OP_FAST_BUILD : SR_OP_UPDATE;
    return s.frames_without_changes + 1 >= kFramesToSettle ? SR_OP_SLOW_BUILD : SR_OP_NONE;
}

void as_state_mark_built(SrAsState& s, uint32_t completed) {                   // AsState::mark_built (:125-147)
    switch (completed) {
        case SR_OP_UPDATE:
            if (s.changing) { s.number_of_updates_since_last_rebuild += 1; s.frames_without_changes = 0; }
            else { s.changing = 1; s.frames_without_changes = 0; s.number_of_updates_since_last_rebuild = 1; }
            break;
        case SR_OP_FAST_BUILD: s.changing = 1; s.frames_without_changes = 0; s.number_of_updates_since_last_rebuild = 0; break;
        case SR_OP_SLOW_BUILD: s.changing = 0; s.frames_without_changes = 0; s.number_of_updates_since_last_rebuild = 0; break;
        default: if (s.changing) s.frames_without_changes += 1; break;
    }
}

void tree_levels(const std::vector<uint32_t>& nodes, std::vector<uint32_t>& level_nodes, std::vector<uint32_t>& level_offsets) {
    const uint32_t n = (uint32_t)(nodes.size() / srl::kNodeDwords);
    std::vector<std::vector<uint32_t>> levels;
    std::vector<uint32_t> cur{0u}, next;
    while (!cur.empty() && n) {
        levels.push_back(cur);
        next.clear();
        for (uint32_t node : cur)
            for (int c = 0; c < srl::kBvhWidth; c++) { const int ref = (int)nodes[(size_t)node * srl::kNodeDwords + srl::kChildOffset + c]; if (ref >= 0) next.push_back((uint32_t)ref); }
        cur.swap(next);
    }
    level_nodes.clear(); level_offsets.assign(1, 0u);
    for (size_t l = levels.size(); l-- > 0;) {
        level_nodes.insert(level_nodes.end(), levels[l].begin(), levels[l].end());
        level_offsets.push_back((uint32_t)level_nodes.size());
    }
}

}  // namespace srh
