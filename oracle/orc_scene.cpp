// ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Parity status: UNPINNED (see orc_math.h).
#include "orc_scene.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace orc {

// ---------------------------------------------------------------------------------------------
// H6  Material::new (resources/material.rs:52-92), textures resolved to NULL (lib.rs:937-943)
// ---------------------------------------------------------------------------------------------
void material_new(const float base_color[4], float metallic, float roughness, const float emissive_factor[3],
                  float emissive_strength, float transmission, float ior, SrMaterial* out) {
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < 4; i++) out->base_color_value[i] = base_color[i];
    out->metallic_factor = metallic;
    out->roughness_factor = roughness;
    for (int i = 0; i < 3; i++) out->emissive_factor[i] = emissive_factor[i];
    out->emissive_factor[3] = emissive_strength;
    out->alpha_mode = 0;       // material.rs:74
    out->alpha_cutoff = 0.0f;  // material.rs:75
    out->transmission_factor = transmission;
    out->ior = ior;
    out->base_color_image = out->base_color_sampler = SR_NULL_TEXTURE;
    out->metallic_roughness_image = out->metallic_roughness_sampler = SR_NULL_TEXTURE;
    out->normal_image = out->normal_sampler = SR_NULL_TEXTURE;
    out->occlusion_image = out->occlusion_sampler = SR_NULL_TEXTURE;
    out->emissive_image = out->emissive_sampler = SR_NULL_TEXTURE;
}

// ---------------------------------------------------------------------------------------------
// H1/H2  Camera::as_matrices (camera.rs:33-63) + transposed upload (lib.rs:1017-1048).
// nalgebra 0.35.0 (Cargo.lock:3093) is not in the reference tree; its published constructions are
// restated: look_at_rh builds the orthonormal frame (s, u, -f) with translation -R*eye,
// Perspective3::new is the OpenGL-style matrix, try_inverse the general cofactor 4x4 inverse.
// Bits may differ from nalgebra in the last place (it routes the rotation through a quaternion).
// ---------------------------------------------------------------------------------------------
static void mat4_mul(const float a[16], const float b[16], float out[16]) {  // row-major a*b
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            float s = a[r * 4 + 0] * b[0 * 4 + c];
            s = s + a[r * 4 + 1] * b[1 * 4 + c];
            s = s + a[r * 4 + 2] * b[2 * 4 + c];
            s = s + a[r * 4 + 3] * b[3 * 4 + c];
            out[r * 4 + c] = s;
        }
}
static bool mat4_inverse(const float m[16], float out[16]) {
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0f) return false;
    float inv_det = 1.0f / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * inv_det;
    return true;
}

void camera_matrices(const float pos[3], const float target[3], float fov_y_deg, uint32_t w, uint32_t h,
                     const float* prev_view_proj16, SrMatrices* out) {
    V3 eye = v3(pos[0], pos[1], pos[2]);
    V3 tgt = v3(target[0], target[1], target[2]);
    V3 up = v3(0.0f, 1.0f, 0.0f);                   // camera.rs:36
    V3 f = normalize(tgt - eye);
    V3 s = normalize(cross(f, up));
    V3 u = cross(s, f);
    // view (row-major): rows s, u, -f; translation = -R*eye
    float view[16] = {s.x, s.y, s.z, -dot(s, eye), u.x, u.y, u.z, -dot(u, eye),
                      -f.x, -f.y, -f.z, dot(f, eye), 0.0f, 0.0f, 0.0f, 1.0f};
    float aspect = (float)w / (float)h;             // camera.rs:42
    float fovy = fov_y_deg * (3.14159265358979323846f / 180.0f);  // f32::to_radians
    float znear = 0.1f, zfar = 100.0f;              // camera.rs:44-45
    float tan_half = tanf(fovy / 2.0f);
    float proj[16] = {0};
    proj[0] = 1.0f / (aspect * tan_half);
    proj[5] = 1.0f / tan_half;
    proj[10] = (zfar + znear) / (znear - zfar);
    proj[11] = (2.0f * zfar * znear) / (znear - zfar);
    proj[14] = -1.0f;
    proj[5] *= -1.0f;                               // camera.rs:51
    float vi[16], pi[16], vp[16];
    mat4_inverse(view, vi);
    mat4_inverse(proj, pi);
    mat4_mul(proj, view, vp);                       // camera.rs:55
    // Row-major storage here == "each float4 is a row" on the GPU (lib.rs:1042-1047).
    memcpy(out->view_inverse, vi, 64);
    memcpy(out->proj_inverse, pi, 64);
    memcpy(out->view_proj, vp, 64);
    if (prev_view_proj16) memcpy(out->prev_view_proj, prev_view_proj16, 64);
    else memset(out->prev_view_proj, 0, 64);        // lib.rs:410
}

void inverse3x3(const SrTransform& t, float o[9]) {
    const float* m = t.m;
    float a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
    float c00 = a11 * a22 - a12 * a21;
    float c01 = a12 * a20 - a10 * a22;
    float c02 = a10 * a21 - a11 * a20;
    float det = (a00 * c00 + a01 * c01) + a02 * c02;
    float id = 1.0f / det;
    o[0] = c00 * id; o[1] = (a02 * a21 - a01 * a22) * id; o[2] = (a01 * a12 - a02 * a11) * id;
    o[3] = c01 * id; o[4] = (a00 * a22 - a02 * a20) * id; o[5] = (a02 * a10 - a00 * a12) * id;
    o[6] = c02 * id; o[7] = (a01 * a20 - a00 * a21) * id; o[8] = (a00 * a11 - a01 * a10) * id;
}

// ---------------------------------------------------------------------------------------------
// Renderer::load_mesh (lib.rs:873-954) + ResourceManager::add_blas (resource_manager.rs:417-447)
// ---------------------------------------------------------------------------------------------
int Scene::add_mesh(uint64_t key, const SrVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const SrMaterial* m) {
    if (nv == 0 || ni == 0 || (ni % 3) != 0) return -1;           // lib.rs:885-891
    for (uint32_t i = 0; i < ni; i++) if (idx[i] >= nv) return -1;  // lib.rs:892-899
    // lib.rs:901-925: emission = factor * strength; emissive iff any component > 0
    std::vector<SrEmissiveTriangle> ets;
    float e[3] = {m->emissive_factor[0] * m->emissive_factor[3], m->emissive_factor[1] * m->emissive_factor[3],
                  m->emissive_factor[2] * m->emissive_factor[3]};
    if (e[0] > 0.0f || e[1] > 0.0f || e[2] > 0.0f) {
        for (uint32_t t = 0; t + 2 < ni; t += 3) {
            SrEmissiveTriangle et;
            const float* p0 = v[idx[t]].position; const float* p1 = v[idx[t + 1]].position; const float* p2 = v[idx[t + 2]].position;
            for (int k = 0; k < 3; k++) { et.v0[k] = p0[k]; et.v1[k] = p1[k]; et.v2[k] = p2[k]; et.emission[k] = e[k]; }
            et.v0[3] = et.v1[3] = et.v2[3] = 0.0f; et.emission[3] = 0.0f;
            ets.push_back(et);
        }
    }
    return add_blas(key, v, nv, idx, ni, m, ets.data(), (uint32_t)ets.size());
}

// ResourceManager::add_blas (resource_manager.rs:417-447)
int Scene::add_blas(uint64_t key, const SrVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const SrMaterial* m,
                    const SrEmissiveTriangle* et, uint32_t n_et) {
    if (slots.count(key)) return -1;                              // lib.rs:880-884
    if (nv == 0 || ni == 0 || (ni % 3) != 0) return -1;
    for (uint32_t i = 0; i < ni; i++) if (idx[i] >= nv) return -1;
    const uint32_t* tex = &m->base_color_image;                   // five (image, sampler) pairs
    for (int i = 0; i < 10; i += 2)
        if (tex[i] != SR_NULL_TEXTURE && (tex[i] >= images.size() || tex[i + 1] >= samplers.size())) return -1;
    Mesh mesh;
    mesh.key = key;
    mesh.vertices.assign(v, v + nv);
    mesh.indices.assign(idx, idx + ni);
    mesh.material = *m;
    for (uint32_t i = 0; i < n_et; i++) {
        uint32_t es;
        if (!free_emissive_slots.empty()) { es = free_emissive_slots.back(); free_emissive_slots.pop_back(); emissive_tris[es] = et[i]; }
        else { es = (uint32_t)emissive_tris.size(); emissive_tris.push_back(et[i]); }
        mesh.emissive_slots.push_back(es);
    }
    uint32_t slot;
    if (!free_mesh_slots.empty()) { slot = free_mesh_slots.back(); free_mesh_slots.pop_back(); meshes[slot] = std::move(mesh); }
    else { slot = (uint32_t)meshes.size(); meshes.push_back(std::move(mesh)); }
    slots[key] = slot;
    return (int)slot;
}

// ResourceManager::remove (resource_manager.rs:459-487)
void Scene::remove(uint64_t key) {
    auto it = slots.find(key);
    if (it == slots.end()) return;
    Mesh& m = meshes[it->second];
    for (uint32_t es : m.emissive_slots) free_emissive_slots.push_back(es);
    m = Mesh();
    free_mesh_slots.push_back(it->second);
    slots.erase(it);
}

// Image::new_from_data (image/mod.rs:82-111) with utils::realign_data (utils.rs:27-43): R8/RG8/RGB8
// are widened to RGBA8 with 0x00 in the missing channels; UNORM, no sRGB decode.
int Scene::add_image(const uint8_t* data, uint32_t w, uint32_t h, uint32_t channels) {
    if (!data || w == 0 || h == 0 || channels < 1 || channels > 4) return -1;
    Image img;
    img.w = w; img.h = h;
    img.rgba.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        uint32_t p = 0;
        for (uint32_t c = 0; c < channels; c++) p |= (uint32_t)data[i * channels + c] << (8 * c);
        img.rgba[i] = p;
    }
    images.push_back(std::move(img));
    return (int)images.size() - 1;
}
int Scene::add_sampler(const SrSamplerDesc* d) {
    if (!d || d->mag_filter > 1 || d->min_filter > 1 || d->address_mode_u > 2 || d->address_mode_v > 2) return -1;
    samplers.push_back(*d);
    return (int)samplers.size() - 1;
}
// rt_utils.slang:121-133
V4 Scene::sample_texture(uint32_t image_slot, uint32_t sampler_slot, float s, float t, V4 fallback) const {
    if (image_slot == SR_NULL_TEXTURE) return fallback;
    return sample_image(images[image_slot], samplers[sampler_slot], s, t);
}

// ---------------------------------------------------------------------------------------------
// H3  frame_instance_data (resource_manager.rs:216-267) + dummy padding (lib.rs:1058-1081),
// then the flattening that stands in for the TLAS/BLAS build.
// ---------------------------------------------------------------------------------------------
int Scene::set_instances(const uint64_t* keys, const uint32_t* counts, uint32_t n_keys, const SrTransform* xf) {
    for (uint32_t k = 0; k < n_keys; k++) if (!slots.count(keys[k])) return -1;  // resource_manager.rs:227-231
    instances.clear(); transforms.clear(); indirection.clear(); tris.clear();
    uint32_t x = 0;
    for (uint32_t k = 0; k < n_keys; k++) {
        uint32_t slot = slots[keys[k]];
        const Mesh& mesh = meshes[slot];
        for (uint32_t c = 0; c < counts[k]; c++, x++) {
            uint32_t instance_index = (uint32_t)transforms.size();
            transforms.push_back(xf[x]);
            Instance inst;
            inst.mesh_slot = slot;
            inst.o2w = xf[x];
            inverse3x3(xf[x], inst.w2o);
            inst.tri_offset = (uint32_t)tris.size();
            instances.push_back(inst);
            for (uint32_t tri_slot : mesh.emissive_slots)
                indirection.push_back(SrEmissiveIndirectionEntry{tri_slot, instance_index});
            uint32_t nprim = (uint32_t)mesh.indices.size() / 3;
            for (uint32_t p = 0; p < nprim; p++) {
                V3 w[3];
                for (int j = 0; j < 3; j++) {
                    const float* pp = mesh.vertices[mesh.indices[3 * p + j]].position;
                    w[j] = transform_point(xf[x], v3(pp[0], pp[1], pp[2]));
                }
                tris.push_back(WTri{w[0], w[1] - w[0], w[2] - w[0], instance_index, p});
            }
        }
    }
    if (transforms.empty()) transforms.push_back(SrTransform{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}});
    if (indirection.empty()) indirection.push_back(SrEmissiveIndirectionEntry{0, 0});
    if (emissive_tris.empty()) { SrEmissiveTriangle z; memset(&z, 0, sizeof(z)); emissive_tris.push_back(z); }
    build_bvh();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Brute force: the ground truth for K2/K3. Lowest global triangle index wins ties.
// ---------------------------------------------------------------------------------------------
Hit Scene::closest_brute(V3 o, V3 d, float tmin, float tmax) const {
    Hit best{-1.0f, 0.0f, 0.0f, 0xFFFFFFFFu};
    float best_t = tmax;
    for (uint32_t i = 0; i < tris.size(); i++) {
        float t, u, v;
        if (intersect_tri(o, d, tris[i], tmin, tmax, t, u, v) && (best.tri == 0xFFFFFFFFu || t < best_t)) {
            best_t = t;
            best = Hit{t, u, v, i};
        }
    }
    return best;
}
bool Scene::any_brute(V3 o, V3 d, float tmin, float tmax) const {
    for (uint32_t i = 0; i < tris.size(); i++) {
        float t, u, v;
        if (intersect_tri(o, d, tris[i], tmin, tmax, t, u, v)) return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// The oracle's own BVH: recursive binned SAH (12 bins), leaves of <= 4 triangles.
// ---------------------------------------------------------------------------------------------
namespace {
struct Box { V3 lo, hi; };
inline Box empty_box() { return Box{v3(INFINITY), v3(-INFINITY)}; }
inline void grow(Box& b, V3 p) {
    b.lo = V3{fminf(b.lo.x, p.x), fminf(b.lo.y, p.y), fminf(b.lo.z, p.z)};
    b.hi = V3{fmaxf(b.hi.x, p.x), fmaxf(b.hi.y, p.y), fmaxf(b.hi.z, p.z)};
}
inline void grow(Box& b, const Box& o) { grow(b, o.lo); grow(b, o.hi); }
inline float half_area(const Box& b) {
    V3 e = b.hi - b.lo;
    if (!(e.x >= 0.0f)) return 0.0f;
    return e.x * e.y + e.y * e.z + e.z * e.x;
}
inline float axis(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
}  // namespace

void Scene::build_bvh() {
    nodes.clear(); order.clear();
    uint32_t n = (uint32_t)tris.size();
    if (n == 0) return;
    std::vector<Box> tb(n);
    std::vector<V3> cen(n);
    order.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const WTri& t = tris[i];
        Box b = empty_box();
        grow(b, t.v0); grow(b, t.v0 + t.e1); grow(b, t.v0 + t.e2);
        // intersect_tri accepts barycentrics up to kBaryEps outside the triangle (and rounds): the box
        // must contain that fattened triangle, or BVH and brute force disagree at silhouette edges.
        V3 pad = V3{fabsf(t.e1.x) + fabsf(t.e2.x), fabsf(t.e1.y) + fabsf(t.e2.y), fabsf(t.e1.z) + fabsf(t.e2.z)} * (4.0f * kBaryEps);
        b.lo = b.lo - pad - V3{fabsf(b.lo.x), fabsf(b.lo.y), fabsf(b.lo.z)} * 2e-7f;
        b.hi = b.hi + pad + V3{fabsf(b.hi.x), fabsf(b.hi.y), fabsf(b.hi.z)} * 2e-7f;
        tb[i] = b;
        cen[i] = (b.lo + b.hi) * 0.5f;
        order[i] = i;
    }
    struct Task { uint32_t node, first, count; };
    std::vector<Task> stack;
    nodes.push_back(BvhNode{});
    stack.push_back(Task{0, 0, n});
    const int NB = 12;
    while (!stack.empty()) {
        Task tk = stack.back(); stack.pop_back();
        Box nb = empty_box(), cb = empty_box();
        for (uint32_t i = tk.first; i < tk.first + tk.count; i++) { grow(nb, tb[order[i]]); grow(cb, cen[order[i]]); }
        nodes[tk.node].lo = nb.lo; nodes[tk.node].hi = nb.hi;
        nodes[tk.node].left = nodes[tk.node].right = 0;
        nodes[tk.node].first = tk.first; nodes[tk.node].count = tk.count;
        if (tk.count <= 4) continue;
        V3 ext = cb.hi - cb.lo;
        int ax = 0;
        if (ext.y > ext.x) ax = 1;
        if (ext.z > axis(ext, ax)) ax = 2;
        float lo = axis(cb.lo, ax), e = axis(ext, ax);
        uint32_t mid = tk.first + tk.count / 2;
        if (e > 0.0f) {
            Box bb[NB]; uint32_t bc[NB];
            for (int b = 0; b < NB; b++) { bb[b] = empty_box(); bc[b] = 0; }
            float scale = (float)NB / e;
            auto bin_of = [&](uint32_t id) { int b = (int)((axis(cen[id], ax) - lo) * scale); return b < 0 ? 0 : (b >= NB ? NB - 1 : b); };
            for (uint32_t i = tk.first; i < tk.first + tk.count; i++) { int b = bin_of(order[i]); grow(bb[b], tb[order[i]]); bc[b]++; }
            float right_area[NB]; uint32_t right_cnt[NB];
            Box acc = empty_box(); uint32_t cnt = 0;
            for (int b = NB - 1; b > 0; b--) { grow(acc, bb[b]); cnt += bc[b]; right_area[b] = half_area(acc); right_cnt[b] = cnt; }
            acc = empty_box(); cnt = 0;
            float best = INFINITY; int best_b = -1;
            for (int b = 0; b < NB - 1; b++) {
                grow(acc, bb[b]); cnt += bc[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                float cost = half_area(acc) * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best) { best = cost; best_b = b; }
            }
            if (best_b >= 0) {
                auto it = std::partition(order.begin() + tk.first, order.begin() + tk.first + tk.count,
                                         [&](uint32_t id) { return bin_of(id) <= best_b; });
                mid = (uint32_t)(it - order.begin());
            }
        }
        if (mid == tk.first || mid == tk.first + tk.count || e <= 0.0f) {
            mid = tk.first + tk.count / 2;
            std::nth_element(order.begin() + tk.first, order.begin() + mid, order.begin() + tk.first + tk.count,
                             [&](uint32_t a, uint32_t b) { return axis(cen[a], ax) < axis(cen[b], ax) || (axis(cen[a], ax) == axis(cen[b], ax) && a < b); });
        }
        uint32_t l = (uint32_t)nodes.size();
        nodes.push_back(BvhNode{}); nodes.push_back(BvhNode{});
        nodes[tk.node].left = l; nodes[tk.node].right = l + 1; nodes[tk.node].count = 0;
        stack.push_back(Task{l, tk.first, mid - tk.first});
        stack.push_back(Task{l + 1, mid, tk.first + tk.count - mid});
    }
}

// Conservative slab test: never rejects a box a triangle hit with t in (t_lo, t_hi] could lie in.
// Planes are taken in the ray's own order (near plane = hi when the direction is negative), so a
// zero direction component gives -inf/+inf for an origin strictly inside the slab and NaN for an
// origin exactly on a face; fmaxf/fminf drop the NaN, i.e. the slab is closed. The far bound is
// inflated (Ize 2013, "Robust BVH Ray Traversal").
static inline float inflate(float f) { return f * (1.0f + copysignf(5e-7f, f)); }  // away from zero; keeps +-inf
struct RaySigns { bool x, y, z; };
static inline bool box_hit(const BvhNode& n, V3 o, V3 inv, RaySigns sg, float t_lo, float t_hi, float& tnear) {
    float nx = ((sg.x ? n.hi.x : n.lo.x) - o.x) * inv.x, fx = ((sg.x ? n.lo.x : n.hi.x) - o.x) * inv.x;
    float ny = ((sg.y ? n.hi.y : n.lo.y) - o.y) * inv.y, fy = ((sg.y ? n.lo.y : n.hi.y) - o.y) * inv.y;
    float nz = ((sg.z ? n.hi.z : n.lo.z) - o.z) * inv.z, fz = ((sg.z ? n.lo.z : n.hi.z) - o.z) * inv.z;
    float t0 = fmaxf(fmaxf(nx, ny), fmaxf(nz, t_lo));
    float t1 = fminf(inflate(fminf(fminf(fx, fy), fz)), t_hi);
    tnear = t0;
    return t0 <= t1;
}

Hit Scene::closest_bvh(V3 o, V3 d, float tmin, float tmax, Counters* c) const {
    Hit best{-1.0f, 0.0f, 0.0f, 0xFFFFFFFFu};
    if (nodes.empty()) return best;
    const V3 inv = V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    const RaySigns sg{std::signbit(inv.x), std::signbit(inv.y), std::signbit(inv.z)};
    // Near bound of the box test: one |tmin| BELOW tmin. Close to the origin the triangle test's t carries
    // an absolute error far above 1e-5*tmin (cancellation in o - v0), so a relative slack is not enough.
    const float t_lo = fminf(tmin, 0.0f) - fabsf(tmin);
    float best_t = tmax;
    float cull = tmax + fabsf(tmax) * 1e-5f;  // relaxed on both ends; only intersect_tri applies the exact bounds
    uint32_t stack[96]; int sp = 0;
    stack[sp++] = 0;
    float tn;
    if (c) c->boxes++;
    if (!box_hit(nodes[0], o, inv, sg, t_lo, cull, tn)) return best;
    while (sp > 0) {
        const BvhNode& n = nodes[stack[--sp]];
        if (n.count > 0) {
            for (uint32_t i = n.first; i < n.first + n.count; i++) {
                uint32_t id = order[i];
                float t, u, v;
                if (c) c->tris++;
                if (intersect_tri(o, d, tris[id], tmin, tmax, t, u, v)) {
                    if (best.tri == 0xFFFFFFFFu || t < best_t || (t == best_t && id < best.tri)) {
                        best_t = t; best = Hit{t, u, v, id};
                        cull = best_t + fabsf(best_t) * 1e-5f;
                    }
                }
            }
            continue;
        }
        float tl, tr;
        if (c) c->boxes += 2;
        bool hl = box_hit(nodes[n.left], o, inv, sg, t_lo, cull, tl);
        bool hr = box_hit(nodes[n.right], o, inv, sg, t_lo, cull, tr);
        if (hl && hr) {
            if (tl <= tr) { stack[sp++] = n.right; stack[sp++] = n.left; }
            else { stack[sp++] = n.left; stack[sp++] = n.right; }
        } else if (hl) stack[sp++] = n.left;
        else if (hr) stack[sp++] = n.right;
    }
    return best;
}

bool Scene::any_bvh(V3 o, V3 d, float tmin, float tmax, Counters* c) const {
    if (nodes.empty()) return false;
    const V3 inv = V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    const RaySigns sg{std::signbit(inv.x), std::signbit(inv.y), std::signbit(inv.z)};
    // Near bound of the box test: one |tmin| BELOW tmin. Close to the origin the triangle test's t carries
    // an absolute error far above 1e-5*tmin (cancellation in o - v0), so a relative slack is not enough.
    const float t_lo = fminf(tmin, 0.0f) - fabsf(tmin);
    const float t_hi = tmax + fabsf(tmax) * 1e-5f;
    uint32_t stack[96]; int sp = 0;
    stack[sp++] = 0;
    float tn;
    if (c) c->boxes++;
    if (!box_hit(nodes[0], o, inv, sg, t_lo, t_hi, tn)) return false;
    while (sp > 0) {
        const BvhNode& n = nodes[stack[--sp]];
        if (n.count > 0) {
            for (uint32_t i = n.first; i < n.first + n.count; i++) {
                float t, u, v;
                if (c) c->tris++;
                if (intersect_tri(o, d, tris[order[i]], tmin, tmax, t, u, v)) return true;
            }
            continue;
        }
        if (c) c->boxes += 2;
        if (box_hit(nodes[n.left], o, inv, sg, t_lo, t_hi, tn)) stack[sp++] = n.left;
        if (box_hit(nodes[n.right], o, inv, sg, t_lo, t_hi, tn)) stack[sp++] = n.right;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// K4 closest_hit (closest_hit.slang:12-91) and K6 ray_miss (ray_miss.slang:10-13).
// Textures go through Scene::sample_texture (orc_texture.h).
// ---------------------------------------------------------------------------------------------
bool Scene::any_hit_ignores(const Hit& h) const {
    if (h.tri == 0xFFFFFFFFu || h.tri >= tris.size()) return false;
    const WTri& wt = tris[h.tri];
    const Mesh& mesh = meshes[instances[wt.instance].mesh_slot];                      // meshes[InstanceID()], :14-15
    const SrMaterial& m = mesh.material;
    if (m.alpha_mode == 0) return false;                                              // :20-22
    V3 bary = v3(1.0f - h.u - h.v, h.u, h.v);                                         // :24-26
    uint32_t io = wt.prim * 3;                                                        // :28-31
    const SrVertex& a = mesh.vertices[mesh.indices[io + 0]];
    const SrVertex& b = mesh.vertices[mesh.indices[io + 1]];
    const SrVertex& c = mesh.vertices[mesh.indices[io + 2]];
    float uv_s = (a.base_color_tex_coord[0] * bary.x + b.base_color_tex_coord[0] * bary.y) + c.base_color_tex_coord[0] * bary.z;   // :33-36
    float uv_t = (a.base_color_tex_coord[1] * bary.x + b.base_color_tex_coord[1] * bary.y) + c.base_color_tex_coord[1] * bary.z;
    V4 base_color = sample_texture(m.base_color_image, m.base_color_sampler, uv_s, uv_t,
                                   V4{m.base_color_value[0], m.base_color_value[1], m.base_color_value[2], m.base_color_value[3]});    // :38
    return base_color.w < m.alpha_cutoff;                                             // :40-42
}

SrRayPayload Scene::shade_hit(const Hit& h) const {
    SrRayPayload pl;
    memset(&pl, 0, sizeof(pl));
    if (h.tri == 0xFFFFFFFFu) { pl.dist = -1.0f; return pl; }  // ray_miss.slang:11-12
    const WTri& wt = tris[h.tri];
    const Instance& inst = instances[wt.instance];
    const Mesh& mesh = meshes[inst.mesh_slot];
    const SrMaterial& m = mesh.material;
    V3 bary = v3(1.0f - h.u - h.v, h.u, h.v);                   // :15-17
    uint32_t io = wt.prim * 3;                                   // :21
    const SrVertex& a = mesh.vertices[mesh.indices[io + 0]];
    const SrVertex& b = mesh.vertices[mesh.indices[io + 1]];
    const SrVertex& c = mesh.vertices[mesh.indices[io + 2]];
    auto P3 = [](const float* p) { return v3(p[0], p[1], p[2]); };
    V3 normal = P3(a.normal) * bary.x + P3(b.normal) * bary.y + P3(c.normal) * bary.z;           // :31
    V3 tangent_dir = P3(a.tangent) * bary.x + P3(b.tangent) * bary.y + P3(c.tangent) * bary.z;   // :32
    float handedness = a.tangent[3] >= 0.0f ? 1.0f : -1.0f;                                      // :34
    float uv_s = (a.base_color_tex_coord[0] * bary.x + b.base_color_tex_coord[0] * bary.y) + c.base_color_tex_coord[0] * bary.z;  // :36
    float uv_t = (a.base_color_tex_coord[1] * bary.x + b.base_color_tex_coord[1] * bary.y) + c.base_color_tex_coord[1] * bary.z;
    float nuv_s = (a.normal_tex_coord[0] * bary.x + b.normal_tex_coord[0] * bary.y) + c.normal_tex_coord[0] * bary.z;             // :37
    float nuv_t = (a.normal_tex_coord[1] * bary.x + b.normal_tex_coord[1] * bary.y) + c.normal_tex_coord[1] * bary.z;
    V4 base_color = sample_texture(m.base_color_image, m.base_color_sampler, uv_s, uv_t,
                                   V4{m.base_color_value[0], m.base_color_value[1], m.base_color_value[2], m.base_color_value[3]});   // :42
    V4 emissive_sample = sample_texture(m.emissive_image, m.emissive_sampler, uv_s, uv_t,
                                        V4{m.emissive_factor[0], m.emissive_factor[1], m.emissive_factor[2], 1.0f});                // :45
    V3 final_emission = v3(emissive_sample.x, emissive_sample.y, emissive_sample.z) * m.emissive_factor[3];                        // :46
    const float* W = inst.w2o;                                                                   // :49-50
    V3 world_normal = normalize(v3((normal.x * W[0] + normal.y * W[3]) + normal.z * W[6],
                                   (normal.x * W[1] + normal.y * W[4]) + normal.z * W[7],
                                   (normal.x * W[2] + normal.y * W[5]) + normal.z * W[8]));
    V3 final_normal = world_normal;                                                              // :51
    if (length(tangent_dir) > 0.001f && m.normal_image != SR_NULL_TEXTURE) {                     // :56,65 (the TBN has no other use)
        const float* M = inst.o2w.m;                                                             // (float3x3)ObjectToWorld3x4, :57
        V3 world_tangent = normalize(v3((M[0] * tangent_dir.x + M[1] * tangent_dir.y) + M[2] * tangent_dir.z,
                                        (M[4] * tangent_dir.x + M[5] * tangent_dir.y) + M[6] * tangent_dir.z,
                                        (M[8] * tangent_dir.x + M[9] * tangent_dir.y) + M[10] * tangent_dir.z));   // :58
        world_tangent = normalize(world_tangent - world_normal * dot(world_tangent, world_normal));                  // :59
        V3 world_bitangent = cross(world_normal, world_tangent) * handedness;                                        // :60
        V4 raw = sample_texture(m.normal_image, m.normal_sampler, nuv_s, nuv_t, V4{0.5f, 0.5f, 1.0f, 1.0f});         // :66
        V3 sn = v3(raw.x * 2.0f - 1.0f, raw.y * 2.0f - 1.0f, raw.z * 2.0f - 1.0f);                                   // :67 (:68 multiplies xy by 1)
        sn.z = sqrtf(fminf(fmaxf(1.0f - (sn.x * sn.x + sn.y * sn.y), 0.0f), 1.0f));                                  // :69
        sn = normalize(sn);                                                                                          // :70
        final_normal = normalize(v3((sn.x * world_tangent.x + sn.y * world_bitangent.x) + sn.z * world_normal.x,
                                    (sn.x * world_tangent.y + sn.y * world_bitangent.y) + sn.z * world_normal.y,
                                    (sn.x * world_tangent.z + sn.y * world_bitangent.z) + sn.z * world_normal.z));  // :72 mul(v, TBN)
    }
    pl.dist = h.t;                                                                               // :74
    pl.emission[0] = final_emission.x; pl.emission[1] = final_emission.y; pl.emission[2] = final_emission.z;
    pl.albedo_packed = pack_unorm_4x8(base_color.x, base_color.y, base_color.z, 1.0f);           // :76
    pl.normal_packed = pack_normal(final_normal);                                                // :77
    float final_roughness = m.roughness_factor, final_metallic = m.metallic_factor;             // :79-80
    if (m.metallic_roughness_image != SR_NULL_TEXTURE) {                                         // :82-87
        V4 mr = sample_texture(m.metallic_roughness_image, m.metallic_roughness_sampler, uv_s, uv_t, V4{1.0f, 1.0f, 1.0f, 1.0f});
        final_roughness = final_roughness * mr.y;
        final_metallic = final_metallic * mr.z;
    }
    pl.material_info = pack_half_2x16(final_roughness, final_metallic);                          // :89
    pl.transmission_ior_packed = pack_half_2x16(m.transmission_factor, m.ior);                   // :90
    return pl;
}

}  // namespace orc
