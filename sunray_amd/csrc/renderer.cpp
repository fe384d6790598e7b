// Renderer facade: the reference's `Renderer<K>` method surface for the built
// path, layered purely on this library's own C ABI (sr_scene_*, sr_trace_*, sr_post_*):
//   new / resize / load_mesh / unload_mesh / render / wait_frame / render_to_host_memory
// (src/lib.rs:212-446, 586-639, 873-973, 984-1238, 1908-1934). Keys are u64 (the reference's generic
// ResourceKey, lib.rs:54-58). One Renderer = one GPU, one caller thread (the reference is !Send).
#include <hip/hip_runtime.h>

#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "host.h"


struct SrRenderer {
    int device = 0;
    SrScene* scene = nullptr;
    uint32_t width = 0, height = 0;
    // frame buffers (the reference's transient G-buffer images + temporal resources, lib.rs:320-331,1492-1516)
    // MAX_FRAMES_IN_FLIGHT = 2 (lib.rs:71): the images one frame writes and reads are double-buffered, so raytracing_ris of
    // frame f+1 (own stream) overlaps raytracing_final + the post chain of frame f. The reservoir, accumulation and denoise
    // ping-pongs carry history from frame to frame and stay single sets.
    float* raw_color[2] = {nullptr, nullptr};
    uint16_t* depth[2] = {nullptr, nullptr};
    uint32_t *normal[2] = {nullptr, nullptr}, *diffuse[2] = {nullptr, nullptr}, *motion[2] = {nullptr, nullptr};
    SrReservoir* reservoirs[2] = {nullptr, nullptr};
    SrReservoirGI* reservoirs_gi[2] = {nullptr, nullptr};
    uint32_t *accum[2] = {nullptr, nullptr}, *denoise[2] = {nullptr, nullptr};
    uint32_t* output[2] = {nullptr, nullptr};
    SrRayPayload* primary[2] = {nullptr, nullptr};   // primary-hit hand-off RIS -> final (SrRtParams.primary_payload), part of the per-frame set
    int primary_reuse = 1;                           // SR_PRIMARY_REUSE=0 in the environment: the final pass traces its camera ray itself (A/B)
    hipStream_t s_ris = nullptr, s_final = nullptr;
    hipEvent_t ev_in = nullptr, ev_ris[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    int last_set = 0;
    uint8_t* blue_noise = nullptr;
    uint32_t noise_w = 128, noise_h = 128;
    // per-frame state
    float prev_view_proj[16];       // zero on the first frame (lib.rs:410), NOT reset by resize
    uint32_t relative_frame_count = 0;
    uint64_t absolute_frame_count = 0;
    SrTraceConfig config;
    std::vector<uint64_t> last_keys;
    std::vector<uint32_t> last_counts;
    std::vector<SrTransform> last_transforms;
    bool instances_valid = false;
    // asset groups of load_scene (lib.rs:802-828): group -> BLAS keys and image slots (both freed by unload_scene)
    uint64_t next_group = 0;
    std::map<uint64_t, std::vector<uint64_t>> scene_groups;
    std::map<uint64_t, std::vector<uint32_t>> scene_images;
    // frame / resize callbacks (lib.rs:537-554): (due frame, fn, user); start-of-frame and end-of-frame ones run once
    struct FrameCb { uint64_t frame; SrFrameCallback fn; void* user; };
    std::vector<FrameCb> start_of_frame_callbacks, end_of_frame_callbacks;
    std::vector<std::pair<SrResizeCallback, void*>> resize_callbacks;
    uint64_t frame_of_set[2] = {0, 0};       // absolute frame number last rendered into image set k (its completion = ev_done[k])
    uint64_t completed_frame = 0;            // highest frame known complete on the GPU (frames complete in order)
    std::map<std::array<uint32_t, 4>, uint32_t> sampler_slots;   // dedup like ResourceManager::sampler_slot (resource_manager.rs:491-499)
    int default_sampler = -1;                                    // LINEAR / CLAMP_TO_EDGE (resource_manager.rs:128-136)
};

struct SrLoadedScene {
    uint64_t group = 0;
    std::vector<uint64_t> keys;
    std::vector<uint32_t> counts;
    std::vector<SrTransform> transforms;
};

namespace {

int rfail(int code, const std::string& msg) { return srh::set_error(code, msg); }
#define R_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return rfail(e_ == hipErrorOutOfMemory ? SR_ERR_OOM : SR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

void free_images(SrRenderer* r) {
    void* ptrs[] = {r->raw_color[0], r->raw_color[1], r->depth[0], r->depth[1], r->normal[0], r->normal[1], r->diffuse[0], r->diffuse[1],
                    r->motion[0], r->motion[1], r->reservoirs[0], r->reservoirs[1], r->reservoirs_gi[0], r->reservoirs_gi[1], r->accum[0],
                    r->accum[1], r->denoise[0], r->denoise[1], r->output[0], r->output[1], r->primary[0], r->primary[1]};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int k = 0; k < 2; k++) { r->primary[k] = nullptr; r->raw_color[k] = nullptr; r->depth[k] = nullptr; r->normal[k] = r->diffuse[k] = r->motion[k] = nullptr; r->output[k] = nullptr; }
    r->reservoirs[0] = r->reservoirs[1] = nullptr; r->reservoirs_gi[0] = r->reservoirs_gi[1] = nullptr;
    r->accum[0] = r->accum[1] = r->denoise[0] = r->denoise[1] = nullptr;
}

template <typename T>
int alloc_zero(T** p, size_t n) {
    R_HIP(hipMalloc((void**)p, n * sizeof(T)));
    R_HIP(hipMemset(*p, 0, n * sizeof(T)));
    return SR_OK;
}

int alloc_images(SrRenderer* r, uint32_t w, uint32_t h) {
    const size_t n = (size_t)w * h;
    int rc;
    for (int i = 0; i < 2; i++)
        if ((rc = alloc_zero(&r->raw_color[i], n * 4)) || (rc = alloc_zero(&r->depth[i], n)) || (rc = alloc_zero(&r->normal[i], n)) ||
            (rc = alloc_zero(&r->diffuse[i], n)) || (rc = alloc_zero(&r->motion[i], n)) || (rc = alloc_zero(&r->output[i], n)) ||
            (r->primary_reuse && (rc = alloc_zero(&r->primary[i], n)))) return rc;
    for (int i = 0; i < 2; i++)
        if ((rc = alloc_zero(&r->reservoirs[i], n)) || (rc = alloc_zero(&r->reservoirs_gi[i], n)) ||
            (rc = alloc_zero(&r->accum[i], n)) || (rc = alloc_zero(&r->denoise[i], n))) return rc;
    r->width = w; r->height = h;
    return SR_OK;
}

uint32_t pcg_hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

}  // namespace

extern "C" {

// Stand-in for the reference's embedded 128x128 blue-noise PNG (lib.rs:281-309), which is an asset and is
// not copied: a hashed white-noise RGBA8 texture (grey replicated to rgb, alpha 255). Same generator as
// sunray_amd.scenes.white_noise_rgba8.
int sr_default_noise_texture(uint32_t w, uint32_t h, uint32_t seed, uint8_t* out_rgba8) {
    if (!out_rgba8 || w == 0 || h == 0) return rfail(SR_ERR_INVALID_ARG, "sr_default_noise_texture: bad argument");
    const uint32_t salt = (uint32_t)(((uint64_t)seed * 2654435761ull) & 0xFFFFFFFFull);
    for (uint32_t i = 0; i < h; i++)
        for (uint32_t j = 0; j < w; j++) {
            const uint8_t v = (uint8_t)(pcg_hash((i * w + j) ^ salt) >> 24);
            uint8_t* q = out_rgba8 + ((size_t)i * w + j) * 4;
            q[0] = q[1] = q[2] = v; q[3] = 255;
        }
    return SR_OK;
}

// Renderer::new((w, h), RGBA8_UNORM) (lib.rs:212-446)
int sr_renderer_create(int device, uint32_t width, uint32_t height, SrRenderer** out) {
    if (!out || width == 0 || height == 0) return rfail(SR_ERR_INVALID_ARG, "Renderer::new: empty extent or null out");
    SrRenderer* r = new SrRenderer();
    r->device = device;
    memset(r->prev_view_proj, 0, sizeof(r->prev_view_proj));
    sr_trace_config_default(&r->config);
    if (const char* ev = getenv("SR_PRIMARY_REUSE")) r->primary_reuse = atoi(ev) != 0;
    int rc = sr_scene_create(device, &r->scene);
    if (rc != SR_OK) { delete r; return rc; }
    if ((rc = alloc_images(r, width, height)) != SR_OK) { free_images(r); sr_scene_destroy(r->scene); delete r; return rc; }
    std::vector<uint8_t> noise(128 * 128 * 4);
    sr_default_noise_texture(128, 128, 7, noise.data());
    if (hipMalloc((void**)&r->blue_noise, noise.size()) != hipSuccess || hipMemcpy(r->blue_noise, noise.data(), noise.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipStreamCreateWithFlags(&r->s_ris, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&r->s_final, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_ris[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->ev_ris[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->ev_done[1], hipEventDisableTiming) != hipSuccess) {
        free_images(r); sr_scene_destroy(r->scene); delete r;
        return rfail(SR_ERR_HIP, "Renderer::new: blue-noise upload failed");
    }
    *out = r;
    return SR_OK;
}

int sr_renderer_set_blue_noise(SrRenderer* r, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    if (!r || !rgba8 || w == 0 || h == 0 || w > 16384 || h > 16384) return rfail(SR_ERR_INVALID_ARG, "sr_renderer_set_blue_noise: bad argument");
    if (hipSetDevice(r->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return rfail(SR_ERR_HIP, "sr_renderer_set_blue_noise: device");
    uint8_t* fresh = nullptr;
    const size_t bytes = (size_t)w * h * 4;
    if (hipMalloc((void**)&fresh, bytes) != hipSuccess) return rfail(SR_ERR_HIP, "sr_renderer_set_blue_noise: out of device memory");
    if (hipMemcpy(fresh, rgba8, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(fresh); return rfail(SR_ERR_HIP, "sr_renderer_set_blue_noise: upload failed"); }
    if (r->blue_noise) (void)hipFree(r->blue_noise);
    r->blue_noise = fresh; r->noise_w = w; r->noise_h = h;
    return SR_OK;
}

int sr_renderer_destroy(SrRenderer* r) {
    if (!r) return SR_OK;
    (void)hipSetDevice(r->device);
    (void)hipDeviceSynchronize();
    free_images(r);
    if (r->blue_noise) (void)hipFree(r->blue_noise);
    for (hipEvent_t e : {r->ev_in, r->ev_ris[0], r->ev_ris[1], r->ev_done[0], r->ev_done[1]}) if (e) (void)hipEventDestroy(e);
    if (r->s_ris) (void)hipStreamDestroy(r->s_ris);
    if (r->s_final) (void)hipStreamDestroy(r->s_final);
    sr_scene_destroy(r->scene);
    delete r;
    return SR_OK;
}

// Renderer::resize (lib.rs:586-639): new temporal resources, relative_frame_count = 0
int sr_renderer_resize(SrRenderer* r, uint32_t width, uint32_t height) {
    if (!r || width == 0 || height == 0) return rfail(SR_ERR_INVALID_ARG, "Renderer::resize: bad argument");
    if (!(width == r->width && height == r->height)) {            // resize_internal_images returns early for an unchanged extent (lib.rs:598-600)
        R_HIP(hipSetDevice(r->device));
        R_HIP(hipDeviceSynchronize());                            // device_wait_idle (lib.rs:604)
        free_images(r);
        int rc = alloc_images(r, width, height);
        if (rc != SR_OK) return rc;
        r->relative_frame_count = 0;
    }
    for (auto& cb : r->resize_callbacks) cb.first(cb.second, width, height);   // every resize call, changed extent or not (lib.rs:590-592)
    return SR_OK;
}

// Renderer::add_start_of_frame_callback / add_end_of_frame_callback / add_resize_callback (lib.rs:537-554)
int sr_renderer_add_start_of_frame_callback(SrRenderer* r, SrFrameCallback fn, void* user) {
    if (!r || !fn) return rfail(SR_ERR_INVALID_ARG, "add_start_of_frame_callback: null argument");
    r->start_of_frame_callbacks.push_back({r->absolute_frame_count + 1, fn, user});
    return SR_OK;
}
int sr_renderer_add_end_of_frame_callback(SrRenderer* r, SrFrameCallback fn, void* user) {
    if (!r || !fn) return rfail(SR_ERR_INVALID_ARG, "add_end_of_frame_callback: null argument");
    r->end_of_frame_callbacks.push_back({r->absolute_frame_count + 1, fn, user});
    return SR_OK;
}
int sr_renderer_add_resize_callback(SrRenderer* r, SrResizeCallback fn, void* user) {
    if (!r || !fn) return rfail(SR_ERR_INVALID_ARG, "add_resize_callback: null argument");
    r->resize_callbacks.push_back({fn, user});
    return SR_OK;
}

namespace {
// run_start_of_frame_callbacks (lib.rs:558-568) and run_due_end_of_frame_callbacks (:572-583), in registration order.
// A callback may register further callbacks (they are tagged with a later frame and stay queued).
void run_frame_callbacks(SrRenderer* r, uint64_t upcoming_frame) {
    for (size_t i = 0; i < r->start_of_frame_callbacks.size();) {
        if (r->start_of_frame_callbacks[i].frame <= upcoming_frame) {
            const SrRenderer::FrameCb cb = r->start_of_frame_callbacks[i];
            r->start_of_frame_callbacks.erase(r->start_of_frame_callbacks.begin() + (long)i);
            cb.fn(cb.user);
        } else i++;
    }
    // the frame timeline: frames complete in order; a set's ev_done is the completion of the frame last rendered into it
    for (int k = 0; k < 2; k++)
        if (r->frame_of_set[k] > r->completed_frame && hipEventQuery(r->ev_done[k]) == hipSuccess) r->completed_frame = r->frame_of_set[k];
    for (size_t i = 0; i < r->end_of_frame_callbacks.size();) {
        if (r->end_of_frame_callbacks[i].frame <= r->completed_frame) {
            const SrRenderer::FrameCb cb = r->end_of_frame_callbacks[i];
            r->end_of_frame_callbacks.erase(r->end_of_frame_callbacks.begin() + (long)i);
            cb.fn(cb.user);
        } else i++;
    }
}
}  // namespace

// Renderer::load_mesh (lib.rs:873-954)
int sr_renderer_load_mesh(SrRenderer* r, uint64_t key, const SrVertex* vertices, uint32_t n_vertices, const uint32_t* indices,
                          uint32_t n_indices, const SrMaterial* material) {
    if (!r) return rfail(SR_ERR_INVALID_ARG, "load_mesh: renderer is null");
    r->instances_valid = false;
    return sr_scene_add_mesh(r->scene, key, vertices, n_vertices, indices, n_indices, material, nullptr);
}

int sr_renderer_set_config(SrRenderer* r, const SrTraceConfig* cfg) {
    if (!r || !cfg) return rfail(SR_ERR_INVALID_ARG, "sr_renderer_set_config: null argument");
    r->config = *cfg;
    return SR_OK;
}

// Renderer::render (lib.rs:984-1232): one frame = TLAS (re)build when the instance list changed ->
// raytracing_ris -> raytracing_final -> temporal_accumulation -> denoise_0..3 -> postprocess, enqueued on the
// renderer's own two streams after whatever the caller has enqueued on `stream`; returns the frame number to wait
// on. Two frames may be in flight (MAX_FRAMES_IN_FLIGHT, lib.rs:71): submit frame f+1 before waiting for frame f and
// its RIS pass overlaps frame f's final pass and post chain. Results equal back-to-back execution.
int sr_renderer_render(SrRenderer* r, const float cam_pos[3], const float cam_target[3], float fov_y, const uint64_t* keys,
                       const uint32_t* counts, uint32_t n_keys, const SrTransform* transforms, void* stream, uint64_t* out_frame) {
    if (!r || !cam_pos || !cam_target) return rfail(SR_ERR_INVALID_ARG, "Renderer::render: null argument");
    R_HIP(hipSetDevice(r->device));
    run_frame_callbacks(r, r->absolute_frame_count + 1);            // start of frame (lib.rs:1004-1010)
    // instances: the acceleration structure follows the caller's list — update in place / fast rebuild / settle as
    // AsState decides (tlas.rs:155-191, acceleration_structure/mod.rs:62-148); an identical list is a quiet frame
    uint32_t n_xf = 0;
    for (uint32_t k = 0; k < n_keys; k++) n_xf += counts[k];
    const bool same = r->instances_valid && r->last_keys.size() == n_keys && r->last_transforms.size() == n_xf &&
                      (n_keys == 0 || (memcmp(r->last_keys.data(), keys, n_keys * 8) == 0 && memcmp(r->last_counts.data(), counts, n_keys * 4) == 0)) &&
                      (n_xf == 0 || memcmp(r->last_transforms.data(), transforms, n_xf * sizeof(SrTransform)) == 0);
    if (!same) {
        int rc = sr_scene_set_instances(r->scene, keys, counts, n_keys, transforms);
        if (rc != SR_OK) return rc;
        r->last_keys.assign(keys, keys + n_keys);
        r->last_counts.assign(counts, counts + n_keys);
        r->last_transforms.assign(transforms, transforms + n_xf);
        r->instances_valid = true;
    } else {
        int rc = sr_scene_end_frame(r->scene);      // quiet frame: the heuristic may settle with a quality rebuild
        if (rc != SR_OK) return rc;
    }
    SrMatrices m;
    int rc = sr_camera_matrices(cam_pos, cam_target, fov_y, r->width, r->height, r->prev_view_proj, &m);   // lib.rs:1017-1048
    if (rc != SR_OK) return rc;
    memcpy(r->prev_view_proj, m.view_proj, sizeof(r->prev_view_proj));                                     // history for the NEXT frame
    // image set of this frame; the orderings that remain: RIS(f) -> final(f) -> post(f), RIS(f) -> RIS(f+1) and post(f-2) -> RIS(f)
    const int k = (int)(r->relative_frame_count & 1u);
    R_HIP(hipEventRecord(r->ev_in, (hipStream_t)stream));           // whatever the caller enqueued on `stream` comes first
    R_HIP(hipStreamWaitEvent(r->s_ris, r->ev_in, 0));
    R_HIP(hipStreamWaitEvent(r->s_ris, r->ev_done[k], 0));          // frame f-2 no longer reads this image set
    SrRtParams p;
    memset(&p, 0, sizeof(p));
    p.scene = r->scene;
    p.raw_color = r->raw_color[k]; p.depth_img = r->depth[k]; p.normal_img = r->normal[k]; p.diffuse_img = r->diffuse[k]; p.motion_vec_img = r->motion[k];
    p.matrices = &m;
    p.blue_noise_tex = r->blue_noise; p.blue_noise_w = r->noise_w; p.blue_noise_h = r->noise_h;
    p.reservoirs[0] = r->reservoirs[0]; p.reservoirs[1] = r->reservoirs[1];
    p.reservoirs_gi[0] = r->reservoirs_gi[0]; p.reservoirs_gi[1] = r->reservoirs_gi[1];
    p.primary_payload = r->primary[k];
    p.frame_count = r->relative_frame_count;
    p.width = r->width; p.height = r->height;
    p.config = r->config;
    if (p.config.enable_restir && (rc = sr_trace_ris(&p, r->s_ris)) != SR_OK) return rc;
    R_HIP(hipEventRecord(r->ev_ris[k], r->s_ris));
    R_HIP(hipStreamWaitEvent(r->s_final, r->ev_ris[k], 0));
    if ((rc = sr_trace_final(&p, r->s_final)) != SR_OK) return rc;
    SrPostParams q;
    memset(&q, 0, sizeof(q));
    q.raw_color = r->raw_color[k]; q.motion_vec_img = r->motion[k]; q.depth_img = r->depth[k]; q.normal_img = r->normal[k]; q.diffuse_img = r->diffuse[k];
    q.accum[0] = r->accum[0]; q.accum[1] = r->accum[1]; q.denoise[0] = r->denoise[0]; q.denoise[1] = r->denoise[1];
    q.output_rgba8 = r->output[k];
    q.frame_count = r->relative_frame_count; q.width = r->width; q.height = r->height;
    q.exposure = 1.0f;        // EXPOSURE (lib.rs:44)
    q.denoise_passes = 4;     // DENOISE_PASSES (lib.rs:42)
    if ((rc = sr_post_temporal(&q, r->s_final)) != SR_OK) return rc;
    if ((rc = sr_post_denoise(&q, r->s_final)) != SR_OK) return rc;
    if ((rc = sr_post_tonemap(&q, r->s_final)) != SR_OK) return rc;
    R_HIP(hipEventRecord(r->ev_done[k], r->s_final));
    r->last_set = k;
    r->relative_frame_count += 1;                                    // lib.rs:1438-1439
    r->absolute_frame_count += 1;
    r->frame_of_set[k] = r->absolute_frame_count;
    if (out_frame) *out_frame = r->absolute_frame_count;
    return SR_OK;
}

// Renderer::wait_frame (lib.rs:1234-1238): frames complete in order; waits for the latest submitted one.
int sr_renderer_wait_frame(SrRenderer* r, uint64_t frame) {
    if (!r) return rfail(SR_ERR_INVALID_ARG, "wait_frame: renderer is null");
    if (frame > r->absolute_frame_count) return rfail(SR_ERR_INVALID_ARG, "wait_frame: frame was never submitted");
    if (frame + 1 >= r->absolute_frame_count && r->absolute_frame_count > 0) {
        // the last submitted frame used image set last_set, the one before it the other set; older frames are long done
        const int k = frame == r->absolute_frame_count ? r->last_set : (r->last_set ^ 1);
        R_HIP(hipEventSynchronize(r->ev_done[k]));
    }
    return SR_OK;
}

// Renderer::render_to_host_memory (lib.rs:1908-1934): WARMUP_FRAMES = 16 x (render + wait_frame), then the
// RGBA8 image without padding (W*H*4 bytes).
int sr_renderer_render_to_host_memory(SrRenderer* r, const float cam_pos[3], const float cam_target[3], float fov_y, const uint64_t* keys,
                                      const uint32_t* counts, uint32_t n_keys, const SrTransform* transforms, uint8_t* out_rgba8) {
    if (!r || !out_rgba8) return rfail(SR_ERR_INVALID_ARG, "render_to_host_memory: null argument");
    const uint32_t WARMUP_FRAMES = 16;
    for (uint32_t i = 0; i < WARMUP_FRAMES; i++) {
        uint64_t frame = 0;
        int rc = sr_renderer_render(r, cam_pos, cam_target, fov_y, keys, counts, n_keys, transforms, nullptr, &frame);
        if (rc != SR_OK) return rc;
        if ((rc = sr_renderer_wait_frame(r, frame)) != SR_OK) return rc;
    }
    R_HIP(hipMemcpy(out_rgba8, r->output[r->last_set], (size_t)r->width * r->height * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

// ---- scene loading (lib.rs:779-857, resource_manager.rs:372-413) -------------------------------------------
namespace {
int sampler_slot(SrRenderer* r, const SrSamplerDesc& d, uint32_t* out) {
    const std::array<uint32_t, 4> k = {d.min_filter, d.mag_filter, d.address_mode_u, d.address_mode_v};
    auto it = r->sampler_slots.find(k);
    if (it != r->sampler_slots.end()) { *out = it->second; return SR_OK; }
    int rc = sr_scene_add_sampler(r->scene, &d, out);
    if (rc == SR_OK) r->sampler_slots[k] = *out;
    return rc;
}
}  // namespace

// Renderer::load_scene: uploads the parsed scene's images, samplers and BLASes under fresh keys of a new group.
int sr_renderer_load_scene(SrRenderer* r, const SrGltf* g, SrLoadedScene** out) {
    if (!r || !g || !out) return rfail(SR_ERR_INVALID_ARG, "load_scene: null argument");
    R_HIP(hipSetDevice(r->device));
    R_HIP(hipDeviceSynchronize());                       // device_wait_idle (lib.rs:800)
    uint32_t n_blases = 0, n_instances = 0, n_images = 0, n_samplers = 0, n_textures = 0;
    int rc = sr_gltf_counts(g, &n_blases, &n_instances, &n_images, &n_samplers, &n_textures);
    if (rc != SR_OK) return rc;
    const uint64_t group = r->next_group++;
    std::vector<uint32_t> image_slots(n_images), smp_slots(n_samplers);
    std::vector<uint32_t>& group_images = r->scene_images[group];     // freed again by unload_scene (also after a failed load)
    for (uint32_t i = 0; i < n_images; i++) {
        const uint8_t* px; uint32_t w, h, ch;
        if ((rc = sr_gltf_image(g, i, &px, &w, &h, &ch)) != SR_OK || (rc = sr_scene_add_image(r->scene, px, w, h, ch, &image_slots[i])) != SR_OK) {
            for (uint32_t sl : group_images) sr_scene_remove_image(r->scene, sl);
            r->scene_images.erase(group);
            return rc;
        }
        group_images.push_back(image_slots[i]);
    }
    for (uint32_t i = 0; i < n_samplers; i++) {
        SrSamplerDesc d;
        if ((rc = sr_gltf_sampler(g, i, &d)) != SR_OK || (rc = sampler_slot(r, d, &smp_slots[i])) != SR_OK) return rc;
    }
    if (r->default_sampler < 0) {
        const SrSamplerDesc d = {SR_FILTER_LINEAR, SR_FILTER_LINEAR, SR_ADDRESS_CLAMP_TO_EDGE, SR_ADDRESS_CLAMP_TO_EDGE};
        uint32_t slot;
        if ((rc = sampler_slot(r, d, &slot)) != SR_OK) return rc;
        r->default_sampler = (int)slot;
    }
    SrLoadedScene* ls = new SrLoadedScene();
    ls->group = group;
    std::vector<uint64_t> group_keys;
    for (uint32_t b = 0; b < n_blases; b++) {
        const SrVertex* v; const uint32_t* idx; const SrEmissiveTriangle* et; uint32_t nv, ni, ne;
        SrMaterial m;
        if ((rc = sr_gltf_blas(g, b, &v, &nv, &idx, &ni, &m, &et, &ne)) != SR_OK) { delete ls; return rc; }
        uint32_t* slots = &m.base_color_image;             // Material::new's `resolve` (resource_manager.rs:388-398)
        for (int k = 0; k < 10; k += 2) {
            if (slots[k] == SR_NULL_TEXTURE) continue;
            int32_t smp; uint32_t src;
            if ((rc = sr_gltf_texture(g, slots[k], &smp, &src)) != SR_OK) { delete ls; return rc; }
            slots[k] = image_slots[src];
            slots[k + 1] = smp >= 0 ? smp_slots[smp] : (uint32_t)r->default_sampler;
        }
        const uint64_t key = (group << 32) | (uint64_t)b;   // ResourceKey{group, index}
        if ((rc = sr_scene_add_blas(r->scene, key, v, nv, idx, ni, &m, et, ne, nullptr)) != SR_OK) {
            for (uint64_t k : group_keys) sr_scene_remove(r->scene, k);
            delete ls;
            return rc;
        }
        group_keys.push_back(key);
        ls->keys.push_back(key);
    }
    // group the per-instance transforms by BLAS key, preserving order (lib.rs:830-834)
    std::vector<std::vector<SrTransform>> grouped(n_blases);
    for (uint32_t i = 0; i < n_instances; i++) {
        uint32_t b; SrTransform t;
        if ((rc = sr_gltf_instance(g, i, &b, &t)) != SR_OK) { delete ls; return rc; }
        grouped[b].push_back(t);
    }
    for (uint32_t b = 0; b < n_blases; b++) {
        ls->counts.push_back((uint32_t)grouped[b].size());
        ls->transforms.insert(ls->transforms.end(), grouped[b].begin(), grouped[b].end());
    }
    r->scene_groups[group] = group_keys;
    r->instances_valid = false;
    *out = ls;
    return SR_OK;
}

// Renderer::load_gltf (lib.rs:779-786)
int sr_renderer_load_gltf(SrRenderer* r, const char* path, SrLoadedScene** out) {
    if (!r || !path || !out) return rfail(SR_ERR_INVALID_ARG, "load_gltf: null argument");
    SrGltf* g = nullptr;
    int rc = sr_gltf_open(path, &g);
    if (rc != SR_OK) return rc;
    rc = sr_renderer_load_scene(r, g, out);
    sr_gltf_close(g);
    return rc;
}

int sr_loaded_scene_get(const SrLoadedScene* ls, uint64_t* group, const uint64_t** keys, const uint32_t** counts, uint32_t* n_keys,
                        const SrTransform** transforms, uint32_t* n_transforms) {
    if (!ls) return rfail(SR_ERR_INVALID_ARG, "sr_loaded_scene_get: null argument");
    if (group) *group = ls->group;
    if (keys) *keys = ls->keys.data();
    if (counts) *counts = ls->counts.data();
    if (n_keys) *n_keys = (uint32_t)ls->keys.size();
    if (transforms) *transforms = ls->transforms.data();
    if (n_transforms) *n_transforms = (uint32_t)ls->transforms.size();
    return SR_OK;
}
int sr_loaded_scene_destroy(SrLoadedScene* ls) { delete ls; return SR_OK; }

// Renderer::unload_scene (lib.rs:849-857): frees every asset the load created — the BLASes and, as
// ResourceManager::remove does (resource_manager.rs:459-472), the images; instances of those keys must no longer be
// passed to render. Loading and unloading a scene repeatedly leaks no HBM.
int sr_renderer_unload_scene(SrRenderer* r, uint64_t group) {
    if (!r) return rfail(SR_ERR_INVALID_ARG, "unload_scene: renderer is null");
    R_HIP(hipSetDevice(r->device));
    R_HIP(hipDeviceSynchronize());                                   // device_wait_idle (lib.rs:850)
    auto it = r->scene_groups.find(group);
    if (it != r->scene_groups.end()) {
        for (uint64_t k : it->second) { int rc = sr_scene_remove(r->scene, k); if (rc != SR_OK) return rc; }
        r->scene_groups.erase(it);
    }
    auto im = r->scene_images.find(group);
    if (im != r->scene_images.end()) {
        for (uint32_t sl : im->second) { int rc = sr_scene_remove_image(r->scene, sl); if (rc != SR_OK) return rc; }
        r->scene_images.erase(im);
    }
    r->instances_valid = false;
    return SR_OK;
}

// Renderer::unload_mesh (lib.rs:965-973). The reference defers the removal past the frames in flight; here the
// scene waits for the device, so it is immediate.
int sr_renderer_unload_mesh(SrRenderer* r, uint64_t key) {
    if (!r) return rfail(SR_ERR_INVALID_ARG, "unload_mesh: renderer is null");
    r->instances_valid = false;
    return sr_scene_remove(r->scene, key);
}

// Access for harnesses: the scene (counters, stats), the device output image and the frame counter.
int sr_renderer_get(SrRenderer* r, SrScene** scene, const uint32_t** output_rgba8_device, const float** raw_color_device, uint32_t* relative_frame_count) {
    if (!r) return rfail(SR_ERR_INVALID_ARG, "sr_renderer_get: renderer is null");
    if (scene) *scene = r->scene;
    if (output_rgba8_device) *output_rgba8_device = r->output[r->last_set];     // of the last submitted frame
    if (raw_color_device) *raw_color_device = r->raw_color[r->last_set];
    if (relative_frame_count) *relative_frame_count = r->relative_frame_count;
    return SR_OK;
}

}  // extern "C"
