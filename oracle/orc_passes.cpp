// ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Parity status: UNPINNED (see orc_math.h).
// CPU restatement of the two ray-generation shaders, statement by statement:
//   trace_ris   <- shaders/ray_gen_ris.slang:12-440   
//   trace_final <- shaders/ray_gen_final.slang:11-436
// TraceRay(RAY_FLAG_NONE) = Scene::closest + shade_hit (closest_hit / ray_miss);
// TraceRay(ACCEPT_FIRST_HIT_AND_END_SEARCH | SKIP_CLOSEST_HIT_SHADER) = Scene::any.
// Line numbers in comments refer to the Slang file each function follows.
#include <cstring>

#include "orc_scene.h"

namespace orc {

namespace {

struct Mat4 { V4 r0, r1, r2, r3; };
inline Mat4 rows(const float* m) {
    return Mat4{V4{m[0], m[1], m[2], m[3]}, V4{m[4], m[5], m[6], m[7]}, V4{m[8], m[9], m[10], m[11]}, V4{m[12], m[13], m[14], m[15]}};
}
inline V4 mul(const Mat4& m, V4 v) { return V4{dot4(m.r0, v), dot4(m.r1, v), dot4(m.r2, v), dot4(m.r3, v)}; }
inline V3 xyz(V4 v) { return v3(v.x, v.y, v.z); }
inline V3 A3(const float* p) { return v3(p[0], p[1], p[2]); }
inline void S3(float* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

struct Ctx {
    const Scene& s;
    const SrRtParams& p;
    Counters c;
    uint32_t num_lights;
    // TraceRay with RAY_FLAG_NONE
    void trace(V3 o, V3 d, float tmin, float tmax, SrRayPayload& prd) {
        c.closest++;
        Hit h = s.closest(o, d, tmin, tmax, &c);
        prd = s.shade_hit(h);
    }
    // The shadow-ray idiom shared by every visibility query: caller presets prd.dist = 1.0, the miss
    // shader writes -1 (visible); segments <= 0.002 are not traced and count as visible.
    // Returns the resulting prd.dist.
    float shadow(V3 o, V3 d, float dist) {
        if (dist > 0.002f) {
            c.any++;
            return s.any(o, d, 0.001f, dist - 0.001f, &c) ? 1.0f : -1.0f;
        }
        return -1.0f;
    }
    const SrEmissiveTriangle& light_of(uint32_t indirection_idx) const {
        return s.emissive_tris[s.indirection[indirection_idx].blas_tri_index];
    }
};

struct PrimaryRay { V3 origin, dir; V2 inUV; };
// K1 (ray_gen_ris.slang:44-53 == ray_gen_final.slang:58-67)
inline PrimaryRay primary_ray(const Mat4& vi, const Mat4& pi, uint32_t px, uint32_t py, uint32_t W, uint32_t H) {
    V2 pixel_center = V2{(float)px + 0.5f, (float)py + 0.5f};
    V2 inUV = V2{pixel_center.x / (float)W, pixel_center.y / (float)H};
    V2 d = V2{inUV.x * 2.0f - 1.0f, inUV.y * 2.0f - 1.0f};
    V4 origin = mul(vi, V4{0.0f, 0.0f, 0.0f, 1.0f});
    V4 target = mul(pi, V4{d.x, d.y, 1.0f, 1.0f});
    V3 tn = normalize(xyz(target));
    V4 direction = mul(vi, V4{tn.x, tn.y, tn.z, 0.0f});
    return PrimaryRay{xyz(origin), xyz(direction), inUV};
}

inline SrReservoir empty_reservoir() { SrReservoir r; memset(&r, 0, sizeof(r)); return r; }
inline SrReservoirGI empty_reservoir_gi() { SrReservoirGI r; memset(&r, 0, sizeof(r)); return r; }

// normal_img / depth_img / motion / diffuse stores in the reference formats (lib.rs:1492-1516)
inline void store_gbuffer(const SrRtParams& p, uint32_t pi, float depth, V3 n, float rough, V3 diffuse, V2 motion) {
    p.depth_img[pi] = (uint16_t)f32_to_f16(depth);
    p.normal_img[pi] = pack_rgba8_snorm(n.x, n.y, n.z, rough);
    p.diffuse_img[pi] = pack_b10g11r11(diffuse.x, diffuse.y, diffuse.z);
    p.motion_vec_img[pi] = pack_half_2x16(motion.x, motion.y);
}
inline V3 load_normal(const SrRtParams& p, uint32_t pi) {
    uint32_t v = p.normal_img[pi];
    return v3(unsnorm8(v), unsnorm8(v >> 8), unsnorm8(v >> 16));
}
inline float load_depth(const SrRtParams& p, uint32_t pi) { return f16_to_f32(p.depth_img[pi]); }

// ---------------------------------------------------------------------------------------------
void ris_pixel(Ctx& cx, uint32_t px, uint32_t py) {
    const SrRtParams& pc = cx.p;
    const Scene& sc = cx.s;
    const uint32_t W = pc.width, H = pc.height;
    Mat4 mat_view_inverse = rows(pc.matrices->view_inverse);         // :19
    Mat4 mat_proj_inverse = rows(pc.matrices->proj_inverse);         // :20
    Mat4 mat_prev_view_proj = rows(pc.matrices->prev_view_proj);     // :21
    uint32_t cur_buf = pc.frame_count & 1u, hist_buf = (pc.frame_count & 1u) ^ 1u;  // :31-32
    SrReservoir* reservoir_cur = pc.reservoirs[cur_buf];
    const SrReservoir* reservoir_hist = pc.reservoirs[hist_buf];
    SrReservoirGI* reservoir_gi_cur = pc.reservoirs_gi[cur_buf];
    const SrReservoirGI* reservoir_gi_hist = pc.reservoirs_gi[hist_buf];
    const uint32_t pix = py * W + px;                                 // get_pixel_index

    Rng rng = init_rng(px, py, pc.frame_count, W);                   // :42
    PrimaryRay pr = primary_ray(mat_view_inverse, mat_proj_inverse, px, py, W, H);  // :44-53
    V2 inUV = pr.inUV;
    V3 origin = pr.origin, direction = pr.dir;
    V3 rayOrigin = origin, rayDir = direction;

    SrRayPayload prd; memset(&prd, 0, sizeof(prd));                  // :55
    V3 hitPos = v3(0.0f), hit_normal = v3(0.0f), hit_albedo = v3(0.0f);
    float roughness = 0.5f, metallic = 0.0f;
    V3 V_view = v3(0.0f);
    V2 prev_uv = V2{-1.0f, -1.0f};
    bool prev_valid = false, found_diffuse_surface = false;
    float virtual_distance = 0.0f;

    for (uint32_t virtual_bounce = 0; virtual_bounce < pc.config.virtual_bounces; virtual_bounce++) {  // :69
        cx.trace(rayOrigin, rayDir, 0.001f, 10000.0f, prd);          // :70-75
        // not in the reference: the camera ray's payload, exposed so that tests can check what the HIP path's RIS pass
        // hands to its final pass (SrRtParams.primary_payload). The oracle's own final pass never reads it.
        if (virtual_bounce == 0 && pc.primary_payload) pc.primary_payload[pix] = prd;
        if (prd.dist < 0.0f) break;                                  // :77-79
        hitPos = rayOrigin + rayDir * prd.dist;                      // :81
        hit_normal = unpack_normal(prd.normal_packed);
        V4 alb = unpack_unorm_4x8(prd.albedo_packed);
        hit_albedo = v3(alb.x, alb.y, alb.z);
        V2 mat_info = unpack_half_2x16(prd.material_info);           // :85
        roughness = max_f(mat_info.x, 0.01f);
        metallic = clamp_f(mat_info.y, 0.0f, 1.0f);
        V2 trans_ior = unpack_half_2x16(prd.transmission_ior_packed);
        float transmission = trans_ior.x;
        V_view = -rayDir;                                            // :92
        virtual_distance += prd.dist;                                // :93

        if (transmission > 0.5f) {                                   // :95
            float ior = max_f(trans_ior.y, 1.0f);
            bool is_inside = dot(rayDir, hit_normal) > 0.0f;
            V3 N = is_inside ? -hit_normal : hit_normal;
            float eta = is_inside ? (ior / 1.0f) : (1.0f / ior);
            float cos_theta = min_f(dot(-rayDir, N), 1.0f);
            float R0 = (1.0f - eta) / (1.0f + eta);
            R0 = R0 * R0;
            float fresnel = R0 + (1.0f - R0) * pow5(1.0f - cos_theta);
            V3 refracted = refract(rayDir, N, eta);
            if (length(refracted) < 0.01f) fresnel = 1.0f;
            if (rnd(rng) < fresnel) rayDir = reflect(rayDir, N);
            else rayDir = refracted;
            rayOrigin = hitPos + rayDir * 0.001f;                    // :114
        } else if (metallic > 0.9f && roughness < 0.1f) {            // :115
            rayOrigin = hitPos + hit_normal * 0.001f;
            rayDir = reflect(rayDir, hit_normal);
        } else {
            V3 virtual_world_pos = origin + direction * virtual_distance;  // :119
            V4 prev_clip = mul(mat_prev_view_proj, V4{virtual_world_pos.x, virtual_world_pos.y, virtual_world_pos.z, 1.0f});
            const float MIN_PREV_W = 0.01f;
            prev_valid = prev_clip.w > MIN_PREV_W;
            if (prev_valid) {
                float iw = 1.0f / prev_clip.w;                       // float2 / scalar
                V2 prev_ndc = V2{prev_clip.x * iw, prev_clip.y * iw};
                prev_uv = V2{prev_ndc.x * 0.5f + 0.5f, prev_ndc.y * 0.5f + 0.5f};
                prev_valid = (prev_uv.x >= 0.0f && prev_uv.y >= 0.0f) && (prev_uv.x < 1.0f && prev_uv.y < 1.0f);
            }
            V2 motion_vector = prev_valid ? V2{inUV.x - prev_uv.x, inUV.y - prev_uv.y} : V2{inUV.x + 2.0f, inUV.y + 2.0f};
            V3 denoiser_albedo = lerp3(hit_albedo, v3(1.0f), metallic);
            store_gbuffer(pc, pix, virtual_distance, hit_normal, roughness, denoiser_albedo, motion_vector);  // :133-136
            found_diffuse_surface = true;
            break;
        }
    }

    if (!found_diffuse_surface) {                                    // :143-172
        // (sky_motion is computed by the shader, :145-154, but never stored: motion gets 0.)
        store_gbuffer(pc, pix, 100000.0f, v3(0.0f), 0.0f, v3(0.0f), V2{0.0f, 0.0f});
        reservoir_cur[pix] = empty_reservoir();
        return;  // the GI reservoir of a sky pixel is left untouched
    }

    // Phase 2: RIS audition (:174-268)
    SrReservoir current_r = empty_reservoir();
    const uint32_t num_lights = cx.num_lights;                       // :185-186
    const int RIS_CANDIDATES = (int)pc.config.ris_candidates;
    if (num_lights > 0 && roughness > 0.2f) {                        // :189
        for (int i = 0; i < RIS_CANDIDATES; i++) {
            uint32_t cand_idx = (uint32_t)(rnd(rng) * (float)num_lights);
            if (cand_idx > num_lights - 1) cand_idx = num_lights - 1;
            const SrEmissiveIndirectionEntry& entry = sc.indirection[cand_idx];
            const SrEmissiveTriangle& cand_light = sc.emissive_tris[entry.blas_tri_index];
            const SrTransform& xform = sc.transforms[entry.entity_id];
            V3 wv0 = transform_point(xform, A3(cand_light.v0));
            V3 wv1 = transform_point(xform, A3(cand_light.v1));
            V3 wv2 = transform_point(xform, A3(cand_light.v2));
            V3 edge1 = wv1 - wv0, edge2 = wv2 - wv0;
            float cand_area = 0.5f * length(cross(edge1, edge2));
            float sqr1 = sqrtf(rnd(rng));
            float u = 1.0f - sqr1;
            float v = rnd(rng) * sqr1;
            float w = 1.0f - u - v;
            V3 cand_pos = wv0 * u + wv1 * v + wv2 * w;
            V3 cand_normal = normalize(cross(wv1 - wv0, wv2 - wv0));
            V3 f_y = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic, A3(cand_light.emission), cand_pos, cand_normal);
            float p_hat = max_comp(f_y);
            float p_y = 1.0f / max_f((float)num_lights * cand_area, 0.0001f);
            current_r.w_sum += (p_hat / p_y);
            current_r.M += 1.0f;
            if (rnd(rng) < ((p_hat / p_y) / max_f(current_r.w_sum, 0.0001f))) {
                current_r.light_idx = cand_idx;
                S3(current_r.light_pos, cand_pos);
                S3(current_r.light_normal, cand_normal);
            }
        }
        if (current_r.w_sum > 0.0f) {                                // :225-231
            V3 f_y_winner = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                  A3(cx.light_of(current_r.light_idx).emission), A3(current_r.light_pos), A3(current_r.light_normal));
            float p_hat_winner = max_comp(f_y_winner);
            current_r.W = current_r.w_sum / max_f(current_r.M * p_hat_winner, 0.0001f);
        }
        if (pc.frame_count > 0 && prev_valid) {                      // :234-267 temporal reuse
            V2 prev_pixel_f = V2{prev_uv.x * (float)W, prev_uv.y * (float)H};
            float j0 = rnd(rng), j1 = rnd(rng);
            V2 di_jitter = V2{j0 - 0.5f, j1 - 0.5f};
            int pcx = (int)(prev_pixel_f.x + di_jitter.x), pcy = (int)(prev_pixel_f.y + di_jitter.y);
            if (pcx >= 0 && pcy >= 0 && pcx < (int)W && pcy < (int)H) {
                SrReservoir history_r = reservoir_hist[(uint32_t)pcy * W + (uint32_t)pcx];
                history_r.M = min_f(history_r.M, 10.0f);
                history_r.W = min_f(history_r.W, 20.0f);
                V3 hist_normal = unpack_normal(history_r.hit_normal_packed);
                float normal_conf_di = smoothstep(0.9f, 0.99f, dot(hit_normal, hist_normal));
                float depth_diff_di = fabsf(virtual_distance - history_r.depth) / max_f(virtual_distance, 1e-4f);
                float depth_conf_di = 1.0f - smoothstep(0.05f, 0.20f, depth_diff_di);
                float conf_di = normal_conf_di * depth_conf_di;
                history_r.M *= conf_di;
                if (history_r.W > 0.0f) {
                    history_r.light_idx = history_r.light_idx < num_lights - 1 ? history_r.light_idx : num_lights - 1;
                    V3 f_y_hist = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                        A3(cx.light_of(history_r.light_idx).emission), A3(history_r.light_pos), A3(history_r.light_normal));
                    float p_hat_hist = max_comp(f_y_hist);
                    merge_reservoirs(current_r, history_r, p_hat_hist, rnd(rng));
                    V3 f_y_merged = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                          A3(cx.light_of(current_r.light_idx).emission), A3(current_r.light_pos), A3(current_r.light_normal));
                    float p_hat_merged = max_comp(f_y_merged);
                    current_r.W = current_r.w_sum / max_f(current_r.M * p_hat_merged, 0.0001f);
                }
            }
        }
    }

    if (current_r.W > 0.0f) {                                        // :277-302 visibility reuse
        V3 vis_dir = A3(current_r.light_pos) - hitPos;
        float vis_dist = max_f(length(vis_dir), 0.0001f);
        vis_dir /= vis_dist;
        if (dot(hit_normal, vis_dir) <= 0.0f) current_r.W = 0.0f;
        else {
            prd.dist = cx.shadow(hitPos + hit_normal * 0.001f, vis_dir, vis_dist);
            if (prd.dist >= 0.0f) current_r.W = 0.0f;
        }
    }
    current_r.hit_normal_packed = pack_normal(hit_normal);           // :304-309
    current_r.depth = virtual_distance;
    reservoir_cur[pix] = current_r;

    // Phase 3: ReSTIR GI initial sample (:311-439)
    SrReservoirGI current_gi_r = empty_reservoir_gi();
    float gr1 = rnd(rng), gr2 = rnd(rng);                            // argument order of :322
    V3 gi_dir = get_random_bounce(hit_normal, gr1, gr2);
    float gi_NdotL = max_f(dot(hit_normal, gi_dir), 0.0f);
    if (gi_NdotL > 0.0f) {
        V3 gi_origin = hitPos + hit_normal * 0.001f;
        cx.trace(gi_origin, gi_dir, 0.001f, 10000.0f, prd);          // :327-332
        V3 sample_pos = v3(0.0f), sample_normal = v3(0.0f), sample_radiance = v3(0.0f);
        if (prd.dist > 0.0f) {
            sample_pos = gi_origin + gi_dir * prd.dist;
            sample_normal = unpack_normal(prd.normal_packed);
            V4 a2 = unpack_unorm_4x8(prd.albedo_packed);
            V3 x2_albedo = v3(a2.x, a2.y, a2.z);
            sample_radiance = A3(prd.emission);
            uint32_t nee_num_lights = num_lights;
            if (nee_num_lights > 0) {
                uint32_t nee_idx = (uint32_t)(rnd(rng) * (float)nee_num_lights);
                if (nee_idx > nee_num_lights - 1) nee_idx = nee_num_lights - 1;
                const SrEmissiveIndirectionEntry& nee_entry = sc.indirection[nee_idx];
                const SrEmissiveTriangle& nee_light = sc.emissive_tris[nee_entry.blas_tri_index];
                const SrTransform& nee_xform = sc.transforms[nee_entry.entity_id];
                V3 nwv0 = transform_point(nee_xform, A3(nee_light.v0));
                V3 nwv1 = transform_point(nee_xform, A3(nee_light.v1));
                V3 nwv2 = transform_point(nee_xform, A3(nee_light.v2));
                float sq = sqrtf(rnd(rng));
                float nu = 1.0f - sq;
                float nv = rnd(rng) * sq;
                float nw = 1.0f - nu - nv;
                V3 nee_pos = nwv0 * nu + nwv1 * nv + nwv2 * nw;
                V3 nee_normal = normalize(cross(nwv1 - nwv0, nwv2 - nwv0));
                float nee_area = 0.5f * length(cross(nwv1 - nwv0, nwv2 - nwv0));
                V3 to_light = nee_pos - sample_pos;
                float nee_dist = max_f(length(to_light), 0.0001f);
                to_light /= nee_dist;
                float nee_cos_surf = max_f(dot(sample_normal, to_light), 0.0f);
                float nee_cos_light = max_f(dot(nee_normal, -to_light), 0.0f);
                if (nee_cos_surf > 0.0f && nee_cos_light > 0.0f) {
                    prd.dist = cx.shadow(sample_pos + sample_normal * 0.001f, to_light, nee_dist);  // :374-384
                    if (prd.dist < 0.0f) {
                        float nee_pdf_sa = (nee_dist * nee_dist) / max_f(nee_cos_light * nee_area * (float)nee_num_lights, 0.0001f);
                        sample_radiance += (A3(nee_light.emission) * x2_albedo * nee_cos_surf) / (nee_pdf_sa * 3.14159f);
                    }
                }
            }
        }
        const float GI_RADIANCE_CLAMP = 5.0f;                        // :394
        sample_radiance = min3(sample_radiance, v3(GI_RADIANCE_CLAMP));
        float p_hat = gi_target_pdf(hitPos, hit_normal, hit_albedo, metallic, sample_pos, sample_radiance);
        float pdf = gi_NdotL / 3.14159f;
        current_gi_r.M = 1.0f;
        current_gi_r.w_sum = (pdf > 0.0f) ? (p_hat / pdf) : 0.0f;
        current_gi_r.W = (p_hat > 0.0f) ? (current_gi_r.w_sum / (current_gi_r.M * p_hat)) : 0.0f;
        S3(current_gi_r.sample_pos, sample_pos);
        current_gi_r.sample_normal_packed = pack_normal(sample_normal);
        S3(current_gi_r.sample_radiance, sample_radiance);
    }
    if (pc.frame_count > 0 && prev_valid) {                          // :408-432
        V2 prev_pixel_f_gi = V2{prev_uv.x * (float)W, prev_uv.y * (float)H};
        float j0 = rnd(rng), j1 = rnd(rng);
        V2 gi_jitter = V2{j0 - 0.5f, j1 - 0.5f};
        int gx = (int)(prev_pixel_f_gi.x + gi_jitter.x), gy = (int)(prev_pixel_f_gi.y + gi_jitter.y);
        if (gx >= 0 && gy >= 0 && gx < (int)W && gy < (int)H) {
            SrReservoirGI history_gi = reservoir_gi_hist[(uint32_t)gy * W + (uint32_t)gx];
            V3 gi_hist_normal = unpack_normal(history_gi.hit_normal_packed);
            float normal_conf = smoothstep(0.8f, 0.95f, dot(hit_normal, gi_hist_normal));
            float depth_diff = fabsf(virtual_distance - history_gi.depth) / max_f(virtual_distance, 1e-4f);
            float depth_conf = 1.0f - smoothstep(0.05f, 0.20f, depth_diff);
            float conf = normal_conf * depth_conf;
            history_gi.M = min_f(history_gi.M, 12.0f) * conf;
            history_gi.W = min_f(history_gi.W, 10.0f);
            if (history_gi.W > 0.0f && history_gi.M > 0.0f) {
                float p_hat_hist = gi_target_pdf(hitPos, hit_normal, hit_albedo, metallic, A3(history_gi.sample_pos), A3(history_gi.sample_radiance));
                merge_reservoirs_gi(current_gi_r, history_gi, p_hat_hist, 1.0f, rnd(rng));
                float p_hat_merged = gi_target_pdf(hitPos, hit_normal, hit_albedo, metallic, A3(current_gi_r.sample_pos), A3(current_gi_r.sample_radiance));
                current_gi_r.W = (p_hat_merged > 1e-6f) ? (current_gi_r.w_sum / (current_gi_r.M * p_hat_merged)) : 0.0f;
            }
        }
    }
    current_gi_r.hit_normal_packed = pack_normal(hit_normal);        // :434-439
    current_gi_r.depth = virtual_distance;
    reservoir_gi_cur[pix] = current_gi_r;
}

// ---------------------------------------------------------------------------------------------
void final_pixel(Ctx& cx, uint32_t px, uint32_t py) {
    const SrRtParams& pc = cx.p;
    const Scene& sc = cx.s;
    const uint32_t W = pc.width, H = pc.height;
    Mat4 mat_view_inverse = rows(pc.matrices->view_inverse);
    Mat4 mat_proj_inverse = rows(pc.matrices->proj_inverse);
    uint32_t cur_buf = pc.frame_count & 1u;
    const SrReservoir* reservoir_cur = pc.reservoirs[cur_buf];
    const SrReservoirGI* reservoir_gi_cur = pc.reservoirs_gi[cur_buf];
    const uint32_t pix = py * W + px;
    const int ipx = (int)px, ipy = (int)py;

    Rng rng = init_rng(px, py, pc.frame_count, W);                   // :37
    V3 total_radiance = v3(0.0f);
    const int SAMPLES = 1;                                           // :40
    const int BOUNCES = (int)pc.config.max_bounces;                  // :41
    const int SHADOW_BOUNCES = (int)pc.config.shadow_bounces;        // :42
    // :44-50 blue noise: Load() of the .r channel of an RGBA8 UNORM texel
    int bw = (int)pc.blue_noise_w, bh = (int)pc.blue_noise_h;
    int n1x = ipx % bw, n1y = ipy % bh;
    int n2x = (ipx + 47) % bw, n2y = (ipy + 71) % bh;
    float bn_1 = (float)pc.blue_noise_tex[((size_t)n1y * bw + n1x) * 4] / 255.0f;
    float bn_2 = (float)pc.blue_noise_tex[((size_t)n2y * bw + n2x) * 4] / 255.0f;
    const uint32_t num_lights = cx.num_lights;                       // :52-53
    SrRayPayload prd; memset(&prd, 0, sizeof(prd));

    for (int sample_i = 0; sample_i < SAMPLES; sample_i++) {
        PrimaryRay pr = primary_ray(mat_view_inverse, mat_proj_inverse, px, py, W, H);  // :58-67
        V3 origin = pr.origin;
        V3 rayOrigin = pr.origin, rayDir = pr.dir;
        V3 throughput = v3(1.0f), radiance = v3(0.0f);
        bool restir_evaluated = (pc.config.enable_restir == 0);      // :71 (knob: SrTraceConfig.enable_restir)
        bool prev_did_nee = false;

        for (int bounce = 0; bounce < BOUNCES; bounce++) {           // :74
            cx.trace(rayOrigin, rayDir, 0.001f, 10000.0f, prd);      // :75-80
            if (prd.dist < 0.0f) break;
            V3 hit_normal = unpack_normal(prd.normal_packed);
            V4 alb = unpack_unorm_4x8(prd.albedo_packed);
            V3 hit_albedo = v3(alb.x, alb.y, alb.z);
            V3 hitPos = rayOrigin + rayDir * prd.dist;
            V3 V_view = -rayDir;
            V2 mat_info = unpack_half_2x16(prd.material_info);
            float roughness = max_f(mat_info.x, 0.01f);
            float metallic = clamp_f(mat_info.y, 0.0f, 1.0f);
            V2 trans_ior = unpack_half_2x16(prd.transmission_ior_packed);
            float transmission = trans_ior.x;
            float ior = max_f(trans_ior.y, 1.0f);
            V3 emission = A3(prd.emission);
            if (!prev_did_nee) radiance += emission * throughput;    // :99-101
            prev_did_nee = false;
            float brightness = max_comp(emission);
            if (brightness > 1.0f) break;                            // :104

            if (transmission > 0.5f) {                               // :106-133
                bool is_inside = dot(rayDir, hit_normal) > 0.0f;
                V3 N = is_inside ? -hit_normal : hit_normal;
                float eta = is_inside ? (ior / 1.0f) : (1.0f / ior);
                float cos_theta = min_f(dot(-rayDir, N), 1.0f);
                float R0 = (1.0f - eta) / (1.0f + eta);
                R0 = R0 * R0;
                float fresnel = R0 + (1.0f - R0) * pow5(1.0f - cos_theta);
                V3 refracted = refract(rayDir, N, eta);
                if (length(refracted) < 0.01f) fresnel = 1.0f;
                if (rnd(rng) < fresnel) rayDir = reflect(rayDir, N);
                else {
                    rayDir = refracted;
                    if (is_inside) {
                        V3 absorption = 1.0f - hit_albedo;
                        V3 e = -absorption * prd.dist * 5.0f;
                        throughput *= v3(exp_f(e.x), exp_f(e.y), exp_f(e.z));
                    } else throughput *= hit_albedo;
                }
                rayOrigin = hitPos + rayDir * 0.001f;
                continue;
            }

            if (num_lights > 0 && bounce < SHADOW_BOUNCES) {         // :135
                if (!restir_evaluated && roughness > 0.2f) {         // :136
                    restir_evaluated = true;
                    SrReservoir center_r = reservoir_cur[pix];       // :139-140
                    SrReservoir spatial_r = empty_reservoir();
                    if (center_r.W > 0.0f && center_r.light_idx < num_lights) {  // :151-158
                        center_r.light_idx = center_r.light_idx < num_lights - 1 ? center_r.light_idx : num_lights - 1;
                        V3 f_y_center = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                              A3(cx.light_of(center_r.light_idx).emission), A3(center_r.light_pos), A3(center_r.light_normal));
                        float p_hat_center = max_comp(f_y_center);
                        merge_reservoirs(spatial_r, center_r, p_hat_center, rnd(rng));
                    }
                    const int SPATIAL_SAMPLES = 5;
                    const float SPATIAL_RADIUS = 30.0f;
                    float current_depth = length(hitPos - origin);   // :162
                    for (int s = 0; s < SPATIAL_SAMPLES; s++) {      // :164-188
                        float angle = rnd(rng) * 2.0f * 3.14159f;
                        float radius = sqrtf(rnd(rng)) * SPATIAL_RADIUS;
                        float sa, ca; sincos_f(angle, &sa, &ca);
                        int ncx = ipx + (int)(ca * radius), ncy = ipy + (int)(sa * radius);
                        if (ncx < 0 || ncy < 0 || ncx >= (int)W || ncy >= (int)H) continue;
                        uint32_t pi_n = (uint32_t)ncy * W + (uint32_t)ncx;
                        V3 neighbor_normal = load_normal(pc, pi_n);
                        float neighbor_depth = load_depth(pc, pi_n);
                        if (dot(hit_normal, neighbor_normal) < 0.9f) continue;
                        if (fabsf(current_depth - neighbor_depth) > 0.1f * current_depth) continue;
                        SrReservoir neighbor_r = reservoir_cur[pi_n];
                        neighbor_r.W = min_f(neighbor_r.W, 20.0f);
                        neighbor_r.M = min_f(neighbor_r.M, 10.0f);
                        if (neighbor_r.W > 0.0f && neighbor_r.light_idx < num_lights) {
                            neighbor_r.light_idx = neighbor_r.light_idx < num_lights - 1 ? neighbor_r.light_idx : num_lights - 1;
                            V3 f_y_neighbor = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                                    A3(cx.light_of(neighbor_r.light_idx).emission), A3(neighbor_r.light_pos), A3(neighbor_r.light_normal));
                            float p_hat_neighbor = max_comp(f_y_neighbor);
                            merge_reservoirs(spatial_r, neighbor_r, p_hat_neighbor, rnd(rng));
                        }
                    }
                    if (spatial_r.w_sum > 0.0f) {                    // :190-222
                        V3 f_y_winner = eval_unshadowed_light(hitPos, hit_normal, V_view, hit_albedo, roughness, metallic,
                                                              A3(cx.light_of(spatial_r.light_idx).emission), A3(spatial_r.light_pos), A3(spatial_r.light_normal));
                        float p_hat_winner = max_comp(f_y_winner);
                        spatial_r.W = spatial_r.w_sum / max_f(spatial_r.M * p_hat_winner, 1e-3f);
                        spatial_r.W = min_f(spatial_r.W, 50.0f);
                        V3 shadow_dir = A3(spatial_r.light_pos) - hitPos;
                        float shadow_dist = max_f(length(shadow_dir), 0.0001f);
                        shadow_dir /= shadow_dist;
                        if (dot(hit_normal, shadow_dir) > 0.0f) {
                            prd.dist = cx.shadow(hitPos, shadow_dir, shadow_dist);  // origin = bare hitPos (:208)
                            if (prd.dist < 0.0f) radiance += f_y_winner * throughput * spatial_r.W;
                            prev_did_nee = true;
                        }
                    }
                    // ReSTIR GI spatial reuse (:224-327)
                    SrReservoirGI combined = reservoir_gi_cur[pix];
                    const int GI_SPATIAL_SAMPLES = 3;
                    const float GI_SPATIAL_RADIUS = 20.0f;
                    float gi_current_depth = length(hitPos - origin);
                    for (int s = 0; s < GI_SPATIAL_SAMPLES; s++) {
                        float gi_angle = rnd(rng) * 2.0f * 3.14159f;
                        float gi_radius = sqrtf(rnd(rng)) * GI_SPATIAL_RADIUS;
                        float sa, ca; sincos_f(gi_angle, &sa, &ca);
                        int ncx = ipx + (int)(ca * gi_radius), ncy = ipy + (int)(sa * gi_radius);
                        if (ncx == ipx && ncy == ipy) continue;      // :237
                        if (ncx < 0 || ncy < 0 || ncx >= (int)W || ncy >= (int)H) continue;
                        uint32_t pi_nn = (uint32_t)ncy * W + (uint32_t)ncx;
                        V3 neighbor_normal = load_normal(pc, pi_nn);
                        float neighbor_depth = load_depth(pc, pi_nn);
                        if (dot(hit_normal, neighbor_normal) < 0.9f) continue;
                        if (fabsf(gi_current_depth - neighbor_depth) > 0.1f * gi_current_depth) continue;
                        SrReservoirGI neighbor_r = reservoir_gi_cur[pi_nn];
                        if (neighbor_r.W <= 0.0f) continue;
                        neighbor_r.W = min_f(neighbor_r.W, 10.0f);
                        neighbor_r.M = min_f(neighbor_r.M, 10.0f);
                        // :253-258 re-derive the neighbour's primary hit point from its depth
                        PrimaryRay npr = primary_ray(mat_view_inverse, mat_proj_inverse, (uint32_t)ncx, (uint32_t)ncy, W, H);
                        V3 neighbor_x1 = origin + npr.dir * neighbor_depth;
                        V3 w_new = A3(neighbor_r.sample_pos) - hitPos;
                        V3 w_old = A3(neighbor_r.sample_pos) - neighbor_x1;
                        float d_new = max_f(length(w_new), 1e-4f);
                        float d_old = max_f(length(w_old), 1e-4f);
                        V3 n_x2 = unpack_normal(neighbor_r.sample_normal_packed);
                        float cos_new = max_f(dot(n_x2, -w_new / d_new), 0.0f);
                        float cos_old = max_f(dot(n_x2, -w_old / d_old), 0.0f);
                        if (cos_new <= 0.0f || cos_old <= 0.0f) continue;
                        float jacobian = (cos_new * d_old * d_old) / max_f(cos_old * d_new * d_new, 1e-4f);
                        jacobian = clamp_f(jacobian, 0.0f, 10.0f);
                        V3 gi_spatial_dir = w_new / d_new;
                        if (dot(hit_normal, gi_spatial_dir) <= 0.0f) continue;
                        prd.dist = cx.shadow(hitPos, gi_spatial_dir, d_new);  // :276-286
                        if (prd.dist >= 0.0f) continue;
                        float p_hat_neighbor = gi_target_pdf(hitPos, hit_normal, hit_albedo, metallic, A3(neighbor_r.sample_pos), A3(neighbor_r.sample_radiance));
                        merge_reservoirs_gi(combined, neighbor_r, p_hat_neighbor, jacobian, rnd(rng));
                    }
                    float p_hat_final = gi_target_pdf(hitPos, hit_normal, hit_albedo, metallic, A3(combined.sample_pos), A3(combined.sample_radiance));
                    combined.W = (p_hat_final > 1e-3f) ? (combined.w_sum / max_f(combined.M, 1.0f) / p_hat_final) : 0.0f;
                    combined.W = min_f(combined.W, 20.0f);
                    if (combined.W > 0.0f) {                         // :297-326
                        V3 gi_x2_dir = A3(combined.sample_pos) - hitPos;
                        float gi_x2_dist = max_f(length(gi_x2_dir), 0.0001f);
                        gi_x2_dir /= gi_x2_dist;
                        float gi_NdotL = max_f(dot(hit_normal, gi_x2_dir), 0.0f);
                        if (gi_NdotL > 0.0f) {
                            prd.dist = cx.shadow(hitPos, gi_x2_dir, gi_x2_dist);
                            if (prd.dist < 0.0f) {
                                V3 gi_f_diffuse = hit_albedo * (1.0f - metallic) / 3.14159f;
                                radiance += A3(combined.sample_radiance) * gi_f_diffuse * gi_NdotL * combined.W * throughput;
                            }
                        }
                    }
                    break;  // :327
                } else if (restir_evaluated && roughness > 0.2f) {   // :328-382 plain NEE
                    uint32_t light_idx = (uint32_t)(rnd(rng) * (float)num_lights);
                    if (light_idx > num_lights - 1) light_idx = num_lights - 1;
                    const SrEmissiveIndirectionEntry& entry = sc.indirection[light_idx];
                    const SrEmissiveTriangle& light = sc.emissive_tris[entry.blas_tri_index];
                    const SrTransform& xform = sc.transforms[entry.entity_id];
                    V3 wv0 = transform_point(xform, A3(light.v0));
                    V3 wv1 = transform_point(xform, A3(light.v1));
                    V3 wv2 = transform_point(xform, A3(light.v2));
                    V3 edge1 = wv1 - wv0, edge2 = wv2 - wv0;
                    float light_area = 0.5f * length(cross(edge1, edge2));
                    float r1_nee = rnd(rng);
                    float r2_nee = rnd(rng);
                    float sqr1 = sqrtf(r1_nee);
                    float u = 1.0f - sqr1;
                    float v = r2_nee * sqr1;
                    float w = 1.0f - u - v;
                    V3 light_pos = wv0 * u + wv1 * v + wv2 * w;
                    V3 light_normal = normalize(cross(wv1 - wv0, wv2 - wv0));
                    V3 shadow_ray_dir = light_pos - hitPos;
                    float light_dist = length(shadow_ray_dir);
                    shadow_ray_dir /= light_dist;
                    float cos_theta_light = max_f(dot(light_normal, -shadow_ray_dir), 0.0f);
                    float cos_theta_surface = max_f(dot(hit_normal, shadow_ray_dir), 0.0f);
                    if (cos_theta_light > 0.0f && cos_theta_surface > 0.0f) {
                        prd.dist = cx.shadow(hitPos, shadow_ray_dir, light_dist);
                        if (prd.dist < 0.0f) {
                            float solid_angle_pdf = (light_dist * light_dist) / max_f(cos_theta_light * light_area * (float)num_lights, 1e-4f);
                            V3 nee_contrib = (A3(light.emission) * hit_albedo * throughput * cos_theta_surface) / (solid_angle_pdf * 3.14159f);
                            radiance += min3(nee_contrib, v3(5.0f));
                        }
                        prev_did_nee = true;
                    }
                }
            }

            // BRDF bounce (:385-427)
            V3 N = hit_normal;
            V3 F0 = lerp3(v3(0.04f), hit_albedo, metallic);
            float cos_theta = max_f(dot(N, V_view), 0.0f);
            V3 F = F0 + (1.0f - F0) * pow5(clamp_f(1.0f - cos_theta, 0.0f, 1.0f));
            float p_specular = clamp_f(max_comp(F), 0.05f, 1.0f);
            float r1, r2;
            if (bounce == 0) {
                r1 = frac(bn_1 + (float)(pc.frame_count % 1024u) * 0.75487766f);
                r2 = frac(bn_2 + (float)(pc.frame_count % 1024u) * 0.56984029f);
            } else {
                r1 = rnd(rng);
                r2 = rnd(rng);
            }
            if (rnd(rng) < p_specular) {
                V3 Hh = sample_ggx_vndf(N, V_view, roughness, r1, r2);
                rayDir = reflect(-V_view, Hh);
                if (dot(N, rayDir) <= 0.0f) {
                    rayDir = get_random_bounce(N, r1, r2);
                    throughput *= hit_albedo * (1.0f - metallic) * (1.0f - F) / (1.0f - p_specular);
                } else {
                    float NdotL_b = max_f(dot(N, rayDir), 0.001f);
                    float alpha_b = roughness * roughness;
                    float G1_L = smith_g1_ggx(NdotL_b, alpha_b);
                    throughput *= (F * G1_L) / p_specular;
                }
            } else {
                rayDir = get_random_bounce(N, r1, r2);
                throughput *= hit_albedo * (1.0f - metallic) * (1.0f - F) / (1.0f - p_specular);
            }
            float p = max_comp(throughput);
            if (p < 0.001f) break;
            if (bounce > 2) {
                if (rnd(rng) > p) break;
                throughput /= p;
            }
            rayOrigin = hitPos + hit_normal * 0.001f;                // :427
        }
        total_radiance += radiance;                                  // :430-431
        total_radiance = min3(total_radiance, v3(10.0f));
    }
    V3 current_frame_color = total_radiance / (float)SAMPLES;        // :434
    float* out = pc.raw_color + (size_t)pix * 4;
    out[0] = current_frame_color.x; out[1] = current_frame_color.y; out[2] = current_frame_color.z; out[3] = 1.0f;
}

template <typename F>
void run_pass(Scene& s, const SrRtParams& p, F pixel_fn) {
    uint32_t y0 = p.tile_h ? p.tile_y0 : 0, y1 = p.tile_h ? p.tile_y0 + p.tile_h : p.height;
    if (y1 > p.height) y1 = p.height;
    uint32_t x0 = p.tile_w ? p.tile_x0 : 0, x1 = p.tile_w ? p.tile_x0 + p.tile_w : p.width;
    if (x1 > p.width) x1 = p.width;
    Counters total;
#pragma omp parallel
    {
        Ctx cx{s, p, Counters{}, s.num_lights()};
        Counters mine;
#pragma omp for schedule(dynamic, 1)
        for (int64_t y = y0; y < (int64_t)y1; y++) {
            // SR_TRACE_FLAG_UNCOUNTED / count_y0, count_rows, count_x0, count_cols: halo pixels do not count (include/sunray_hip.h)
            const bool row_counted = !(p.config.flags & SR_TRACE_FLAG_UNCOUNTED) &&
                                     (p.config.count_rows == 0u || ((uint32_t)y - p.config.count_y0) < p.config.count_rows);
            for (uint32_t x = x0; x < x1; x++) {
                cx.c = Counters{};
                pixel_fn(cx, x, (uint32_t)y);
                if (row_counted && (p.config.count_cols == 0u || (x - p.config.count_x0) < p.config.count_cols)) {
                    mine.closest += cx.c.closest; mine.any += cx.c.any; mine.boxes += cx.c.boxes; mine.tris += cx.c.tris;
                }
            }
        }
#pragma omp critical
        { total.closest += mine.closest; total.any += mine.any; total.boxes += mine.boxes; total.tris += mine.tris; }
    }
    s.counters.closest += total.closest; s.counters.any += total.any;
    s.counters.boxes += total.boxes; s.counters.tris += total.tris;
}

}  // namespace

void trace_ris(Scene& s, const SrRtParams& p) { run_pass(s, p, ris_pixel); }
void trace_final(Scene& s, const SrRtParams& p) { run_pass(s, p, final_pixel); }

}  // namespace orc
