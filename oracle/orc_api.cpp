// ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Parity status: UNPINNED (see orc_math.h).
// extern "C" surface of the oracle for the ctypes binding in oracle/binding.py. Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <cstring>

#include <omp.h>

#include "orc_scene.h"

using namespace orc;
namespace orc { void post_temporal(const SrPostParams&); void post_denoise(const SrPostParams&); void post_tonemap(const SrPostParams&); }

extern "C" {

void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int orc_get_max_threads() { return omp_get_max_threads(); }
void* orc_scene_create() { return new Scene(); }
void orc_scene_destroy(void* s) { delete (Scene*)s; }
int orc_scene_add_mesh(void* s, uint64_t key, const SrVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const SrMaterial* m) {
    return ((Scene*)s)->add_mesh(key, v, nv, idx, ni, m);
}
int orc_scene_add_blas(void* s, uint64_t key, const SrVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const SrMaterial* m,
                       const SrEmissiveTriangle* et, uint32_t n_et) {
    return ((Scene*)s)->add_blas(key, v, nv, idx, ni, m, et, n_et);
}
void orc_scene_remove(void* s, uint64_t key) { ((Scene*)s)->remove(key); }
int orc_scene_add_image(void* s, const uint8_t* data, uint32_t w, uint32_t h, uint32_t channels) {
    return ((Scene*)s)->add_image(data, w, h, channels);
}
int orc_scene_add_sampler(void* s, const SrSamplerDesc* d) { return ((Scene*)s)->add_sampler(d); }
// KAT hook: one texture fetch
void orc_sample_texture(void* s, uint32_t image, uint32_t sampler, float u, float v, const float* fallback4, float* out4) {
    V4 r = ((Scene*)s)->sample_texture(image, sampler, u, v, V4{fallback4[0], fallback4[1], fallback4[2], fallback4[3]});
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
int orc_scene_set_instances(void* s, const uint64_t* keys, const uint32_t* counts, uint32_t n_keys, const SrTransform* xf) {
    return ((Scene*)s)->set_instances(keys, counts, n_keys, xf);
}
void orc_scene_set_brute_force(void* s, int on) { ((Scene*)s)->use_brute_force = on != 0; }
void orc_scene_get_tables(void* sp, const SrTransform** transforms, uint32_t* n_instances,
                          const SrEmissiveIndirectionEntry** indirection, uint32_t* num_lights,
                          const SrEmissiveTriangle** emissive, uint32_t* n_emissive, uint32_t* n_triangles) {
    Scene* s = (Scene*)sp;
    *transforms = s->transforms.data(); *n_instances = (uint32_t)s->transforms.size();
    *indirection = s->indirection.data(); *num_lights = (uint32_t)s->indirection.size();
    *emissive = s->emissive_tris.data(); *n_emissive = (uint32_t)s->emissive_tris.size();
    *n_triangles = (uint32_t)s->tris.size();
}
void orc_scene_bvh_info(void* sp, uint64_t* n_nodes, uint64_t* n_tris) {
    Scene* s = (Scene*)sp; *n_nodes = s->nodes.size(); *n_tris = s->tris.size();
}

void orc_trace_closest(void* sp, const SrRay* rays, uint32_t n, SrHit* hits) {
    Scene* s = (Scene*)sp;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const SrRay& r = rays[i];
        Hit h = s->closest(v3(r.origin[0], r.origin[1], r.origin[2]), v3(r.dir[0], r.dir[1], r.dir[2]), r.tmin, r.tmax, nullptr);
        hits[i] = SrHit{h.t, h.u, h.v, h.tri};
    }
}
void orc_trace_any(void* sp, const SrRay* rays, uint32_t n, uint32_t* occluded) {
    Scene* s = (Scene*)sp;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const SrRay& r = rays[i];
        occluded[i] = s->any(v3(r.origin[0], r.origin[1], r.origin[2]), v3(r.dir[0], r.dir[1], r.dir[2]), r.tmin, r.tmax, nullptr) ? 1u : 0u;
    }
}
void orc_shade_closest_hit(void* sp, const SrHit* hits, uint32_t n, SrRayPayload* out) {
    Scene* s = (Scene*)sp;
    for (uint32_t i = 0; i < n; i++) out[i] = s->shade_hit(Hit{hits[i].t, hits[i].u, hits[i].v, hits[i].t < 0.0f ? 0xFFFFFFFFu : hits[i].tri});
}
void orc_trace_ris(void* sp, const SrRtParams* p) { trace_ris(*(Scene*)sp, *p); }
void orc_trace_final(void* sp, const SrRtParams* p) { trace_final(*(Scene*)sp, *p); }
void orc_post_temporal(const SrPostParams* p) { post_temporal(*p); }
void orc_post_denoise(const SrPostParams* p) { post_denoise(*p); }
void orc_post_tonemap(const SrPostParams* p) { post_tonemap(*p); }
float orc_log(float x) { return log_f(x); }
float orc_pow(float x, float y) { return pow_f(x, y); }
void orc_reset_counters(void* sp) { ((Scene*)sp)->counters = Counters{}; }
void orc_read_counters(void* sp, SrRayCounters* out) {
    const Counters& c = ((Scene*)sp)->counters;
    out->closest_queries = c.closest; out->any_queries = c.any; out->boxes_tested = c.boxes; out->tris_tested = c.tris;
    out->reused_visibility_queries = 0;
    out->reused_primary_hits = 0;   // the oracle issues every TraceRay of the reference: SrRtParams.primary_payload is ignored here
}

// ---- host-side helpers ------------------------------------------------------------------------
void orc_camera_matrices(const float* pos, const float* target, float fov, uint32_t w, uint32_t h, const float* prev, SrMatrices* out) {
    camera_matrices(pos, target, fov, w, h, prev, out);
}
void orc_material_new(const float* bc, float metallic, float roughness, const float* ef, float es, float tr, float ior, SrMaterial* out) {
    material_new(bc, metallic, roughness, ef, es, tr, ior, out);
}
void orc_inverse3x3(const SrTransform* t, float* out9) { inverse3x3(*t, out9); }

// ---- known-answer-test hooks (rt_utils.slang) -----------------------------------------------
uint32_t orc_pcg_hash(uint32_t x) { return pcg_hash(x); }
uint32_t orc_init_rng(uint32_t px, uint32_t py, uint32_t frame, uint32_t w) { return init_rng(px, py, frame, w).seed; }
// n draws from `seed`: raw 32-bit results (before the float conversion) and the floats.
uint32_t orc_rnd_stream(uint32_t seed, uint32_t n, uint32_t* words, float* vals) {
    Rng r{seed};
    for (uint32_t i = 0; i < n; i++) {
        Rng t = r;
        t.seed = t.seed * 747796405u + 2891336453u;
        uint32_t word = ((t.seed >> ((t.seed >> 28u) + 4u)) ^ t.seed) * 277803737u;
        words[i] = (word >> 22u) ^ word;
        vals[i] = rnd(r);
    }
    return r.seed;
}
uint32_t orc_pack_normal(float x, float y, float z) { return pack_normal(v3(x, y, z)); }
void orc_unpack_normal(uint32_t p, float* out) { V3 n = unpack_normal(p); out[0] = n.x; out[1] = n.y; out[2] = n.z; }
uint32_t orc_pack_unorm_4x8(float x, float y, float z, float w) { return pack_unorm_4x8(x, y, z, w); }
void orc_unpack_unorm_4x8(uint32_t p, float* o) { V4 v = unpack_unorm_4x8(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
uint32_t orc_pack_half_2x16(float x, float y) { return pack_half_2x16(x, y); }
void orc_unpack_half_2x16(uint32_t p, float* o) { V2 v = unpack_half_2x16(p); o[0] = v.x; o[1] = v.y; }
uint32_t orc_pack_snorm_2x16(float x, float y) { return pack_snorm_2x16(x, y); }
uint32_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_f16_to_f32(uint32_t h) { return f16_to_f32(h); }
uint32_t orc_pack_rgba8_snorm(float x, float y, float z, float w) { return pack_rgba8_snorm(x, y, z, w); }
uint32_t orc_pack_b10g11r11(float r, float g, float b) { return pack_b10g11r11(r, g, b); }
void orc_unpack_b10g11r11(uint32_t v, float* o) { o[0] = from_ufloat(v & 0x7ffu, 6); o[1] = from_ufloat((v >> 11) & 0x7ffu, 6); o[2] = from_ufloat(v >> 22, 5); }
void orc_sincos(float x, float* s, float* c) { sincos_f(x, s, c); }
float orc_exp(float x) { return exp_f(x); }
void orc_build_onb(const float* n, float* t, float* b) { V3 T, B; build_onb(v3(n[0], n[1], n[2]), T, B); t[0] = T.x; t[1] = T.y; t[2] = T.z; b[0] = B.x; b[1] = B.y; b[2] = B.z; }
void orc_get_random_bounce(const float* n, float r1, float r2, float* o) { V3 d = get_random_bounce(v3(n[0], n[1], n[2]), r1, r2); o[0] = d.x; o[1] = d.y; o[2] = d.z; }
void orc_sample_ggx_vndf(const float* n, const float* v, float rough, float r1, float r2, float* o) {
    V3 h = sample_ggx_vndf(v3(n[0], n[1], n[2]), v3(v[0], v[1], v[2]), rough, r1, r2); o[0] = h.x; o[1] = h.y; o[2] = h.z;
}
void orc_eval_unshadowed_light(const float* hp, const float* hn, const float* vv, const float* alb, float rough, float metal,
                               const float* emission, const float* lp, const float* ln, float* o) {
    V3 f = eval_unshadowed_light(v3(hp[0], hp[1], hp[2]), v3(hn[0], hn[1], hn[2]), v3(vv[0], vv[1], vv[2]), v3(alb[0], alb[1], alb[2]),
                                 rough, metal, v3(emission[0], emission[1], emission[2]), v3(lp[0], lp[1], lp[2]), v3(ln[0], ln[1], ln[2]));
    o[0] = f.x; o[1] = f.y; o[2] = f.z;
}
float orc_gi_target_pdf(const float* sp, const float* sn, const float* alb, float metallic, const float* xp, const float* rad) {
    return gi_target_pdf(v3(sp[0], sp[1], sp[2]), v3(sn[0], sn[1], sn[2]), v3(alb[0], alb[1], alb[2]), metallic, v3(xp[0], xp[1], xp[2]), v3(rad[0], rad[1], rad[2]));
}
void orc_merge_reservoirs(SrReservoir* r, const SrReservoir* new_r, float p_hat_new, float random_val) { merge_reservoirs(*r, *new_r, p_hat_new, random_val); }
void orc_merge_reservoirs_gi(SrReservoirGI* r, const SrReservoirGI* new_r, float p_hat_new, float jacobian, float random_val) {
    merge_reservoirs_gi(*r, *new_r, p_hat_new, jacobian, random_val);
}
float orc_smith_v_ggx(float a, float b, float c) { return smith_v_ggx(a, b, c); }
float orc_smith_g1_ggx(float a, float b) { return smith_g1_ggx(a, b); }
float orc_smoothstep(float a, float b, float x) { return smoothstep(a, b, x); }
void orc_refract(const float* i, const float* n, float eta, float* o) { V3 r = refract(v3(i[0], i[1], i[2]), v3(n[0], n[1], n[2]), eta); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void orc_reflect(const float* i, const float* n, float* o) { V3 r = reflect(v3(i[0], i[1], i[2]), v3(n[0], n[1], n[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
// any_hit.slang:11-43 alpha test as a pure helper (never invoked by traversal: geometry is OPAQUE,
// blas.rs:276). Returns 1 if the hit would be ignored.
int orc_any_hit_ignores(uint32_t alpha_mode, float alpha_cutoff, float base_alpha) {
    if (alpha_mode == 0) return 0;
    return base_alpha < alpha_cutoff ? 1 : 0;
}
// The whole shader on hit records (the CPU side of sr_any_hit_ignores): MeshInfo -> indices -> vertices -> uv ->
// sample_texture -> alpha test, any_hit.slang:14-42.
void orc_any_hit(void* sp, const SrHit* hits, uint32_t n, uint32_t* ignored) {
    Scene* s = (Scene*)sp;
    for (uint32_t i = 0; i < n; i++) ignored[i] = s->any_hit_ignores(Hit{hits[i].t, hits[i].u, hits[i].v, hits[i].t < 0.0f ? 0xFFFFFFFFu : hits[i].tri}) ? 1u : 0u;
}
// the canonical triangle test, for direct unit tests
int orc_intersect_tri(const float* o, const float* d, const float* v0, const float* v1, const float* v2, float tmin, float tmax, float* tuv) {
    V3 a = v3(v0[0], v0[1], v0[2]), b = v3(v1[0], v1[1], v1[2]), c = v3(v2[0], v2[1], v2[2]);
    WTri t{a, b - a, c - a, 0, 0};
    return intersect_tri(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), t, tmin, tmax, tuv[0], tuv[1], tuv[2]) ? 1 : 0;
}

}  // extern "C"
