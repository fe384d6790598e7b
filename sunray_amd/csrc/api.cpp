// C ABI of the MI355X ray-tracing hot path (include/sunray_hip.h). Each entry point cites the
// reference interface it replaces in the header; this file is the thin host layer between that ABI,
// the host-side data preparation (host_prep.cpp, bvh_build.cpp) and the HIP kernels (kernels.hip).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <chrono>
#include <vector>

#include "host.h"
#include "bvh_gpu.h"
#include "bvh_layout.h"
#include "kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
}  // namespace
namespace srh {
int set_error(int code, const std::string& msg) { return fail(code, msg); }   // shared with renderer.cpp
}
namespace {
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? SR_ERR_OOM : SR_ERR_HIP,                            \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                             \
    } while (0)

enum PassKind { kRis = 0, kFinal = 1, kClosest = 2, kAny = 3, kNumKinds = 4 };

// Named profiler ranges around every pass, the counterpart of the debug-utils label the reference's render graph puts around
// each pass (render_graph/graph.rs:1097-1118: a named scope in a capture, a no-op otherwise): rocTX push / pop around the
// enqueue, visible in `rocprofv3 --marker-trace`. The rocTX library is looked up at first use (no link-time dependency);
// without it, or with SR_PASS_LABELS=0 in the environment, the ranges are no-ops.
struct PassLabel {
    typedef int (*PushFn)(const char*);
    typedef int (*PopFn)(void);
    static void resolve(PushFn& push, PopFn& pop) {
        static PushFn s_push = nullptr;
        static PopFn s_pop = nullptr;
        static bool tried = false;
        if (!tried) {
            tried = true;
            const char* ev = getenv("SR_PASS_LABELS");
            if (!(ev && atoi(ev) == 0)) {
                void* h = nullptr;
                for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"})   // rocprofv3 listens to the first
                    if ((h = dlopen(name, RTLD_LAZY | RTLD_LOCAL))) break;
                if (h) { s_push = (PushFn)dlsym(h, "roctxRangePushA"); s_pop = (PopFn)dlsym(h, "roctxRangePop"); }
                if (!s_push || !s_pop) { s_push = nullptr; s_pop = nullptr; }
            }
        }
        push = s_push; pop = s_pop;
    }
    PopFn pop = nullptr;
    explicit PassLabel(const char* name) {
        PushFn push;
        resolve(push, pop);
        if (push) push(name);
    }
    ~PassLabel() { if (pop) pop(); }
};

struct DeviceBuffer {
    void* p = nullptr;
    size_t bytes = 0;
    int upload(const void* src, size_t n) {
        if (n > bytes || p == nullptr) {
            if (p) (void)hipFree(p);
            p = nullptr; bytes = 0;
            HIP_TRY(hipMalloc(&p, n ? n : 16));
            bytes = n ? n : 16;
        }
        if (n) HIP_TRY(hipMemcpy(p, src, n, hipMemcpyHostToDevice));
        return SR_OK;
    }
    int reserve(size_t n) {
        if (n > bytes || p == nullptr) {
            if (p) (void)hipFree(p);
            p = nullptr; bytes = 0;
            HIP_TRY(hipMalloc(&p, n ? n : 16));
            bytes = n ? n : 16;
        }
        return SR_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

}  // namespace

constexpr size_t kCounterBytes = (size_t)srd::kCounterSlots * srd::kCounterStride * 8;   // the copies of the ray counters (traverse.h)

struct SrScene {
    int device = 0;
    std::vector<srh::HostMesh> meshes;
    std::map<uint64_t, uint32_t> slots;
    std::vector<SrEmissiveTriangle> emissive_tris;       // the emissive arena (resource_manager.rs:433-443)
    std::vector<SrEmissiveTriangle> emissive_table;      // what a frame sees: the arena, or one zero entry if it is empty
    std::vector<uint32_t> free_mesh_slots, free_emissive_slots;   // LIFO reuse, like the reference's arenas (buffer/arena_core.rs)
    std::vector<SrMeshInfo> mesh_infos;
    srh::FrameInstanceData fid;
    std::vector<srh::BuildTri> world_tris;
    struct DeviceImage { void* d_texels = nullptr; uint32_t w = 0, h = 0; };
    std::vector<DeviceImage> images;        // image slot order (Material::*_image); a removed image leaves d_texels == nullptr
    std::vector<uint32_t> free_image_slots; // slots of removed images, reused by the next sr_scene_add_image
    std::vector<SrSamplerDesc> samplers;    // sampler slot order (Material::*_sampler)
    DeviceBuffer d_nodes, d_tris, d_shade, d_mesh_const, d_slot_of_gid, d_instances, d_lights, d_misc;
    DeviceBuffer d_shade_tex, d_mesh_tex, d_textures;
    // two-level form (optional): one tree per mesh in object space (built once per mesh, kept on the host until the set of
    // meshes changes) + a top-level tree over the instances, rebuilt from the instance list — nothing here scales with
    // instances x triangles
    struct HostBlas {
        bool valid = false;
        std::vector<uint32_t> nodes;        // 4-wide quantised tree, references local to the mesh
        std::vector<float> tris, shade, shade_tex;   // 12 / 12 / 24 floats per triangle, leaf order
        std::vector<uint32_t> slot_of_prim;
        uint32_t n_tris = 0, n_nodes = 0, max_stack = 0, max_depth = 0;
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};   // root box, object space
        float max_edge_sum = 0.0f, max_abs_vertex = 0.0f;
        double build_ms = 0.0;
    };
    std::vector<HostBlas> blases;           // by mesh slot
    std::vector<uint32_t> blas_node_base, blas_tri_base;   // of the concatenated device arrays, by mesh slot
    bool blas_device_current = false;       // the concatenated arrays match the current set of meshes
    bool any_textured_tl = false;
    uint32_t blas_stack = 0;
    bool tl_baked = false;                  // the device arrays hold baked (world-space) copies of some instances' meshes
    std::vector<uint32_t> tl_baked_node_base, tl_baked_tri_base;
    uint64_t tl_blas_nodes = 0, tl_blas_tris = 0;
    DeviceBuffer d_blas_nodes, d_tl_inst, d_tl_instances;
    int instancing = SR_INSTANCING_AUTO;    // sr_scene_set_instancing / SR_INSTANCING in the environment
    bool two_level = false;                 // form of the structure that is built right now
    // acceleration-structure maintenance (update in place): per-level node lists, exact node boxes, flatten inputs
    DeviceBuffer d_level_nodes, d_node_box, d_mesh_infos, d_flat_instances, d_scratch;
    // cost-ordered tile schedules of the two passes (kernels.hip thread_pixel), one per launch geometry
    struct TileSchedule { int which = -1; uint32_t width = 0 /* columns of the launch rectangle */, y0 = 0, y1 = 0, x0 = 0; DeviceBuffer cost, order; bool have_order = false; uint64_t last_use = 0; uint32_t uses = 0; };
    std::vector<TileSchedule> schedules;
    uint64_t schedule_clock = 0;
    int tile_scheduling = 1;                // SR_TILE_SCHEDULING=0 in the environment disables it (A/B)
    int fast_build_ploc = 16;               // device fast build: PLOC with this search radius (default), 0 = radix tree (SR_FAST_BUILD=lbvh | ploc<r>)
    uint32_t forced_op = SR_OP_NONE;        // sr_scene_force_next_op (test / bench hook)
    bool last_build_on_device = false;
    std::vector<uint32_t> level_offsets;
    std::vector<uint32_t> shape;            // mesh slot of every instance of the built tree: an UPDATE needs the same layout
    SrAsState as_state{0, 0, 0, 0};         // SometimesChanges -> Optimal (resource_manager.rs:119-126, mod.rs:86-91)
    uint32_t last_op = SR_OP_NONE;
    srd::DevScene dev{};
    SrBvhStats stats{};
    bool built = false;
    bool built_once = false;
    int instrumented = 0;
    int timing = 0;
    int n_cus = 256;
    int stack_entries = 8;   // LDS traversal-stack entries per lane this scene's tree needs (multiple of 4)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events[kNumKinds];
    size_t events_used[kNumKinds] = {0, 0, 0, 0};
};

namespace {

int bind_device(const SrScene* s) {
    HIP_TRY(hipSetDevice(s->device));
    return SR_OK;
}

// Record a start/stop event pair around a launch when timing is on (bench.py's roofline leg).
struct ScopedTiming {
    SrScene* s; int kind; hipStream_t stream; hipEvent_t stop = nullptr;
    ScopedTiming(SrScene* s_, int kind_, hipStream_t st) : s(s_), kind(kind_), stream(st) {
        if (!s->timing) return;
        auto& pool = s->events[kind];
        size_t& used = s->events_used[kind];
        if (used == pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            pool.emplace_back(a, b);
        }
        (void)hipEventRecord(pool[used].first, stream);
        stop = pool[used].second;
        used++;
    }
    ~ScopedTiming() { if (stop) (void)hipEventRecord(stop, stream); }
};

}  // namespace

extern "C" {

const char* sr_last_error(void) { return g_last_error.c_str(); }
int sr_version(void) { return 1; }

int sr_camera_matrices(const float position[3], const float target[3], float fov_y_degrees, uint32_t width,
                       uint32_t height, const float* prev_view_proj16, SrMatrices* out) {
    if (!position || !target || !out || width == 0 || height == 0) return fail(SR_ERR_INVALID_ARG, "sr_camera_matrices: null argument or empty extent");
    srh::Camera cam;
    memcpy(cam.position, position, 12);
    memcpy(cam.target, target, 12);
    cam.fov_y = fov_y_degrees;
    if (!cam.as_matrices(width, height, prev_view_proj16, out)) return fail(SR_ERR_INVALID_ARG, "sr_camera_matrices: singular view/projection matrix");
    return SR_OK;
}

int sr_material_new(const float base_color[4], float metallic, float roughness, const float emissive_factor[3],
                    float emissive_strength, float transmission, float ior, SrMaterial* out) {
    if (!base_color || !emissive_factor || !out) return fail(SR_ERR_INVALID_ARG, "sr_material_new: null argument");
    srh::material_new(base_color, metallic, roughness, emissive_factor, emissive_strength, transmission, ior, out);
    return SR_OK;
}

int sr_emissive_triangles_from_mesh(const SrVertex* vertices, uint32_t n_vertices, const uint32_t* indices,
                                    uint32_t n_indices, const SrMaterial* material, SrEmissiveTriangle* out,
                                    uint32_t cap, uint32_t* out_count) {
    if (!vertices || !indices || !material || !out_count) return fail(SR_ERR_INVALID_ARG, "sr_emissive_triangles_from_mesh: null argument");
    if (!out && cap > 0) return fail(SR_ERR_INVALID_ARG, "sr_emissive_triangles_from_mesh: out is null but cap > 0 (pass cap = 0 to query the count)");
    for (uint32_t i = 0; i < n_indices; i++)
        if (indices[i] >= n_vertices) return fail(SR_ERR_INVALID_ARG, "sr_emissive_triangles_from_mesh: index out of range");
    std::vector<SrEmissiveTriangle> v;
    srh::emissive_triangles_from_mesh(vertices, indices, n_indices, *material, v);
    *out_count = (uint32_t)v.size();
    if (out) memcpy(out, v.data(), sizeof(SrEmissiveTriangle) * std::min<size_t>(cap, v.size()));
    return SR_OK;
}

void sr_trace_config_default(SrTraceConfig* out) {
    if (!out) return;
    memset(out, 0, sizeof(*out));
    out->max_bounces = 10;
    out->shadow_bounces = 5;
    out->ris_candidates = 16;
    out->virtual_bounces = 20;
    out->enable_restir = 1;
}

int sr_scene_create(int device, SrScene** out) {
    if (!out) return fail(SR_ERR_INVALID_ARG, "sr_scene_create: out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(SR_ERR_INVALID_ARG, "sr_scene_create: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    SrScene* s = new SrScene();
    s->device = device;
    if (const char* ev = getenv("SR_TILE_SCHEDULING")) s->tile_scheduling = atoi(ev) != 0;
    if (const char* ev = getenv("SR_INSTANCING")) s->instancing = !strcmp(ev, "two_level") ? SR_INSTANCING_TWO_LEVEL : (!strcmp(ev, "flat") ? SR_INSTANCING_FLAT : SR_INSTANCING_AUTO);
    if (const char* ev = getenv("SR_FAST_BUILD")) s->fast_build_ploc = !strcmp(ev, "lbvh") ? 0 : (!strncmp(ev, "ploc", 4) && atoi(ev + 4) > 0 ? atoi(ev + 4) : 16);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->n_cus = prop.multiProcessorCount;
    // the copies of the ray counters, in one small allocation of their own
    const std::vector<unsigned char> zeros(kCounterBytes, 0);
    int rc = s->d_misc.upload(zeros.data(), zeros.size());
    if (rc != SR_OK) { delete s; return rc; }
    *out = s;
    return SR_OK;
}

int sr_scene_destroy(SrScene* s) {
    if (!s) return SR_OK;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    for (auto& m : s->meshes) { if (m.d_vertices) (void)hipFree(m.d_vertices); if (m.d_indices) (void)hipFree(m.d_indices); }
    for (auto& im : s->images) if (im.d_texels) (void)hipFree(im.d_texels);
    s->d_shade_tex.release(); s->d_mesh_tex.release(); s->d_textures.release();
    s->d_level_nodes.release(); s->d_node_box.release(); s->d_mesh_infos.release(); s->d_flat_instances.release(); s->d_scratch.release();
    s->d_nodes.release(); s->d_tris.release(); s->d_shade.release(); s->d_mesh_const.release(); s->d_slot_of_gid.release(); s->d_instances.release();
    s->d_lights.release(); s->d_misc.release();
    s->d_blas_nodes.release(); s->d_tl_inst.release(); s->d_tl_instances.release();
    for (auto& ts : s->schedules) { ts.cost.release(); ts.order.release(); }
    for (auto& pool : s->events) for (auto& e : pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete s;
    return SR_OK;
}

// ResourceManager::add_blas (resource_manager.rs:417-447): the mesh-info slot becomes the instance custom index, the
// local emissive triangles go to the emissive arena.
int sr_scene_add_blas(SrScene* s, uint64_t key, const SrVertex* vertices, uint32_t n_vertices, const uint32_t* indices,
                      uint32_t n_indices, const SrMaterial* material, const SrEmissiveTriangle* emissive, uint32_t n_emissive,
                      uint32_t* out_slot) {
    if (!s || !vertices || !indices || !material || (n_emissive && !emissive)) return fail(SR_ERR_INVALID_ARG, "load_mesh: null argument");
    if (s->slots.count(key)) return fail(SR_ERR_INVALID_ARG, "load_mesh: an asset is already registered under this key");
    if (n_vertices == 0 || n_indices == 0 || (n_indices % 3) != 0) {
        char buf[200];
        snprintf(buf, sizeof(buf), "load_mesh: invalid mesh (%u vertices, %u indices — need non-empty vertices and a triangle-list index count)", n_vertices, n_indices);
        return fail(SR_ERR_INVALID_ARG, buf);
    }
    for (uint32_t i = 0; i < n_indices; i++)
        if (indices[i] >= n_vertices) {
            char buf[120];
            snprintf(buf, sizeof(buf), "load_mesh: index %u out of range for %u vertices", indices[i], n_vertices);
            return fail(SR_ERR_INVALID_ARG, buf);
        }
    // Vulkan treats a triangle with a NaN position as inactive; the builders quantise positions (a non-finite one would
    // be undefined behaviour there), so such meshes are refused instead
    for (uint32_t i = 0; i < n_vertices; i++)
        if (!std::isfinite(vertices[i].position[0]) || !std::isfinite(vertices[i].position[1]) || !std::isfinite(vertices[i].position[2])) {
            char buf[120];
            snprintf(buf, sizeof(buf), "load_mesh: vertex %u has a non-finite position", i);
            return fail(SR_ERR_INVALID_ARG, buf);
        }
    const uint32_t* tex = &material->base_color_image;   // five (image, sampler) slot pairs (resources/material.rs:33-42)
    for (int i = 0; i < 10; i += 2)
        if (tex[i] != SR_NULL_TEXTURE && (tex[i] >= s->images.size() || tex[i + 1] >= s->samplers.size() || !s->images[tex[i]].d_texels))
            return fail(SR_ERR_INVALID_ARG, "load_mesh: material refers to an image or sampler slot that was never added");
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    srh::HostMesh m;
    m.key = key;
    m.vertices.assign(vertices, vertices + n_vertices);
    m.indices.assign(indices, indices + n_indices);
    m.n_vertices = n_vertices; m.n_indices = n_indices;
    m.material = *material;
    {
        hipError_t e = hipMalloc(&m.d_vertices, sizeof(SrVertex) * (size_t)n_vertices);
        if (e == hipSuccess) e = hipMemcpy(m.d_vertices, vertices, sizeof(SrVertex) * (size_t)n_vertices, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&m.d_indices, sizeof(uint32_t) * (size_t)n_indices);
        if (e == hipSuccess) e = hipMemcpy(m.d_indices, indices, sizeof(uint32_t) * (size_t)n_indices, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (m.d_vertices) (void)hipFree(m.d_vertices);
            if (m.d_indices) (void)hipFree(m.d_indices);
            return fail(e == hipErrorOutOfMemory ? SR_ERR_OOM : SR_ERR_HIP, std::string("load_mesh: geometry upload failed: ") + hipGetErrorString(e));
        }
    }
    for (uint32_t i = 0; i < n_emissive; i++) {
        uint32_t es;
        if (!s->free_emissive_slots.empty()) { es = s->free_emissive_slots.back(); s->free_emissive_slots.pop_back(); s->emissive_tris[es] = emissive[i]; }
        else { es = (uint32_t)s->emissive_tris.size(); s->emissive_tris.push_back(emissive[i]); }
        m.emissive_slots.push_back(es);
    }
    SrMeshInfo mi;
    mi.vertices = (uint64_t)(uintptr_t)m.d_vertices;
    mi.indices = (uint64_t)(uintptr_t)m.d_indices;
    mi.material = *material;
    uint32_t slot;
    if (!s->free_mesh_slots.empty()) {
        slot = s->free_mesh_slots.back(); s->free_mesh_slots.pop_back();
        s->mesh_infos[slot] = mi;
        s->meshes[slot] = std::move(m);
    } else {
        slot = (uint32_t)s->meshes.size();
        s->mesh_infos.push_back(mi);
        s->meshes.push_back(std::move(m));
    }
    s->slots[key] = slot;
    s->built = false;
    if (s->blases.size() < s->meshes.size()) s->blases.resize(s->meshes.size());
    s->blases[slot] = SrScene::HostBlas();
    s->blas_device_current = false;
    if (out_slot) *out_slot = slot;
    return SR_OK;
}

int sr_scene_add_mesh(SrScene* s, uint64_t key, const SrVertex* vertices, uint32_t n_vertices, const uint32_t* indices,
                      uint32_t n_indices, const SrMaterial* material, uint32_t* out_slot) {
    if (!s || !vertices || !indices || !material) return fail(SR_ERR_INVALID_ARG, "load_mesh: null argument");
    std::vector<SrEmissiveTriangle> et;
    if (n_indices % 3 == 0) {
        bool in_range = true;
        for (uint32_t i = 0; i < n_indices && in_range; i++) in_range = indices[i] < n_vertices;
        if (in_range) srh::emissive_triangles_from_mesh(vertices, indices, n_indices, *material, et);   // lib.rs:901-925
    }
    return sr_scene_add_blas(s, key, vertices, n_vertices, indices, n_indices, material, et.data(), (uint32_t)et.size(), out_slot);
}

// ResourceManager::remove (resource_manager.rs:459-487) for a BLAS key: frees the mesh-info slot and the emissive
// slots (reused LIFO by later loads) and the geometry. The reference defers the reclaim by MAX_FRAMES_IN_FLIGHT
// frames; here the device is idle-waited instead. Instances of the key must no longer be passed to set_instances.
int sr_scene_remove(SrScene* s, uint64_t key) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "ResourceManager::remove: scene is null");
    auto it = s->slots.find(key);
    if (it == s->slots.end()) return SR_OK;     // removing an unknown key is a no-op in the reference
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    const uint32_t slot = it->second;
    srh::HostMesh& m = s->meshes[slot];
    if (m.d_vertices) (void)hipFree(m.d_vertices);
    if (m.d_indices) (void)hipFree(m.d_indices);
    for (uint32_t es : m.emissive_slots) s->free_emissive_slots.push_back(es);
    m = srh::HostMesh();
    if (slot < s->blases.size()) s->blases[slot] = SrScene::HostBlas();
    s->blas_device_current = false;
    s->free_mesh_slots.push_back(slot);
    s->slots.erase(it);
    s->built = false;
    return SR_OK;
}

int sr_scene_add_image(SrScene* s, const uint8_t* data, uint32_t width, uint32_t height, uint32_t channels, uint32_t* out_image_slot) {
    if (!s || !data || width == 0 || height == 0) return fail(SR_ERR_INVALID_ARG, "Image::new_from_data: null argument or empty extent");
    if (channels < 1 || channels > 4) return fail(SR_ERR_INVALID_ARG, "Image::new_from_data: 1..4 channels of 8 bits are supported");
    if ((uint64_t)width * height >= (1ull << 30)) return fail(SR_ERR_UNSUPPORTED, "Image::new_from_data: image too large");
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    const size_t n = (size_t)width * height;
    std::vector<uint32_t> rgba(n);
    for (size_t i = 0; i < n; i++) {       // utils::realign_data (utils.rs:27-43): missing channels are 0x00
        uint32_t p = 0;
        for (uint32_t c = 0; c < channels; c++) p |= (uint32_t)data[i * channels + c] << (8 * c);
        rgba[i] = p;
    }
    SrScene::DeviceImage im;
    im.w = width; im.h = height;
    HIP_TRY(hipMalloc(&im.d_texels, n * 4));
    const hipError_t ce = hipMemcpy(im.d_texels, rgba.data(), n * 4, hipMemcpyHostToDevice);
    if (ce != hipSuccess) { (void)hipFree(im.d_texels); return fail(SR_ERR_HIP, std::string("Image::new_from_data: ") + hipGetErrorString(ce)); }
    uint32_t slot;
    if (!s->free_image_slots.empty()) { slot = s->free_image_slots.back(); s->free_image_slots.pop_back(); s->images[slot] = im; }
    else { slot = (uint32_t)s->images.size(); s->images.push_back(im); }
    if (out_image_slot) *out_image_slot = slot;
    return SR_OK;
}

// ResourceManager::remove drops a group's images with its BLASes (resource_manager.rs:472). The caller must have removed
// (or be about to remove) every mesh whose material names the slot; meshes still naming it are rejected at the next
// sr_scene_set_instances. Waits for the device: a launch in flight may still sample the texels.
int sr_scene_remove_image(SrScene* s, uint32_t image_slot) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "sr_scene_remove_image: scene is null");
    if (image_slot >= s->images.size() || !s->images[image_slot].d_texels) return fail(SR_ERR_INVALID_ARG, "sr_scene_remove_image: no image in this slot");
    for (const auto& m : s->meshes) {
        if (m.n_vertices == 0) continue;
        const uint32_t* tex = &m.material.base_color_image;
        for (int i = 0; i < 10; i += 2)
            if (tex[i] == image_slot) return fail(SR_ERR_STATE, "sr_scene_remove_image: a registered mesh still uses this image (remove the mesh first)");
    }
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(s->images[image_slot].d_texels));
    s->images[image_slot] = SrScene::DeviceImage();
    s->free_image_slots.push_back(image_slot);
    return SR_OK;
}

int sr_scene_add_sampler(SrScene* s, const SrSamplerDesc* d, uint32_t* out_sampler_slot) {
    if (!s || !d) return fail(SR_ERR_INVALID_ARG, "Sampler::new: null argument");
    if (d->min_filter > SR_FILTER_LINEAR || d->mag_filter > SR_FILTER_LINEAR || d->address_mode_u > SR_ADDRESS_CLAMP_TO_EDGE ||
        d->address_mode_v > SR_ADDRESS_CLAMP_TO_EDGE)
        return fail(SR_ERR_INVALID_ARG, "Sampler::new: filter must be NEAREST/LINEAR, address mode REPEAT/MIRRORED_REPEAT/CLAMP_TO_EDGE");
    s->samplers.push_back(*d);
    if (out_sampler_slot) *out_sampler_slot = (uint32_t)s->samplers.size() - 1;
    return SR_OK;
}

namespace {

// Per-instance device tables (closest_hit's WorldToObject / ObjectToWorld, the flatten inputs) and the light table:
// everything that follows the instance transforms without touching the tree.
int upload_instance_tables(SrScene* s) {
    int rc;
    std::vector<srd::DevInstance> dinst(s->fid.instances.size() ? s->fid.instances.size() : 1);
    memset(dinst.data(), 0, dinst.size() * sizeof(srd::DevInstance));
    std::vector<srd::FlatInstance> flat(dinst.size());
    memset(flat.data(), 0, flat.size() * sizeof(srd::FlatInstance));
    for (size_t i = 0; i < s->fid.instances.size(); i++) {
        memcpy(dinst[i].w2o, s->fid.instances[i].w2o, 36);
        const float* M = s->fid.instances[i].o2w.m;   // (float3x3)ObjectToWorld3x4
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) dinst[i].o2w[3 * r + c] = M[4 * r + c];
        memcpy(flat[i].o2w, M, 48);
        flat[i].tri_offset = s->fid.instances[i].tri_offset;
        flat[i].mesh_slot = s->fid.instances[i].mesh_slot;
    }
    if ((rc = s->d_instances.upload(dinst.data(), dinst.size() * sizeof(srd::DevInstance))) != SR_OK) return rc;
    if ((rc = s->d_flat_instances.upload(flat.data(), flat.size() * sizeof(srd::FlatInstance))) != SR_OK) return rc;
    std::vector<float> lights;
    srh::light_table(s->fid, s->emissive_table, lights);
    if ((rc = s->d_lights.upload(lights.data(), lights.size() * 4)) != SR_OK) return rc;
    s->dev.instances = (const srd::DevInstance*)s->d_instances.p;
    s->dev.lights = (const srd::DevLight*)s->d_lights.p;
    s->dev.num_lights = (uint32_t)s->fid.emissive_entries.size();
    s->dev.n_instances = (uint32_t)s->fid.instances.size();
    return SR_OK;
}

// Per-mesh payload constants, the texture side of the materials and the image table (independent of the tree).
int upload_mesh_tables(SrScene* s, bool* any_textured_out) {
    int rc;
    std::vector<srd::DevMeshConst> mconst(s->meshes.size() ? s->meshes.size() : 1);
    memset(mconst.data(), 0, mconst.size() * sizeof(srd::DevMeshConst));
    for (size_t i = 0; i < s->meshes.size(); i++) {
        if (s->meshes[i].n_vertices == 0) continue;      // freed slot (sr_scene_remove)
        const SrMaterial& m = s->meshes[i].material;
        for (int k = 0; k < 3; k++) mconst[i].emission[k] = m.emissive_factor[k] * m.emissive_factor[3];
        mconst[i].albedo_packed = srh::pack_unorm_4x8(m.base_color_value[0], m.base_color_value[1], m.base_color_value[2], 1.0f);
        mconst[i].material_info = srh::pack_half_2x16(m.roughness_factor, m.metallic_factor);
        mconst[i].transmission_ior_packed = srh::pack_half_2x16(m.transmission_factor, m.ior);
    }
    // texture side: per-mesh factors + resolved slots, per-slot uv/tangent records, the image table. Occlusion
    // textures are carried in SrMaterial but no shader on the path samples them.
    std::vector<srd::DevMeshTex> mtex(mconst.size());
    memset(mtex.data(), 0, mtex.size() * sizeof(srd::DevMeshTex));
    bool any_textured = false;
    auto sampler_code = [&](uint32_t image, uint32_t sampler) -> uint32_t {
        if (image == SR_NULL_TEXTURE) return 0u;
        const SrSamplerDesc& d = s->samplers[sampler];
        return (d.mag_filter & 1u) | (d.address_mode_u << 1) | (d.address_mode_v << 3);
    };
    for (size_t i = 0; i < s->meshes.size(); i++) {
        if (s->meshes[i].n_vertices == 0) continue;
        const SrMaterial& m = s->meshes[i].material;
        srd::DevMeshTex& t = mtex[i];
        memcpy(t.base_color, m.base_color_value, 16);
        memcpy(t.emissive_factor, m.emissive_factor, 12);
        t.emissive_strength = m.emissive_factor[3];
        t.roughness = m.roughness_factor; t.metallic = m.metallic_factor;
        t.alpha_mode = m.alpha_mode; t.alpha_cutoff = m.alpha_cutoff;
        t.img_base = m.base_color_image; t.img_mr = m.metallic_roughness_image; t.img_normal = m.normal_image; t.img_emissive = m.emissive_image;
        t.samplers = sampler_code(m.base_color_image, m.base_color_sampler) | (sampler_code(m.metallic_roughness_image, m.metallic_roughness_sampler) << 8) |
                     (sampler_code(m.normal_image, m.normal_sampler) << 16) | (sampler_code(m.emissive_image, m.emissive_sampler) << 24);
        mconst[i].textured = (t.img_base != SR_NULL_TEXTURE || t.img_mr != SR_NULL_TEXTURE || t.img_normal != SR_NULL_TEXTURE || t.img_emissive != SR_NULL_TEXTURE) ? 1u : 0u;
        any_textured = any_textured || mconst[i].textured;
    }
    std::vector<srd::DevTexture> textures(s->images.size() ? s->images.size() : 1);
    memset(textures.data(), 0, textures.size() * sizeof(srd::DevTexture));
    for (size_t i = 0; i < s->images.size(); i++) { textures[i].texels = (const uint32_t*)s->images[i].d_texels; textures[i].w = s->images[i].w; textures[i].h = s->images[i].h; }
    if ((rc = s->d_mesh_const.upload(mconst.data(), mconst.size() * sizeof(srd::DevMeshConst))) != SR_OK) return rc;
    if ((rc = s->d_mesh_tex.upload(mtex.data(), mtex.size() * sizeof(srd::DevMeshTex))) != SR_OK) return rc;
    if ((rc = s->d_textures.upload(textures.data(), textures.size() * sizeof(srd::DevTexture))) != SR_OK) return rc;
    s->dev.mesh_const = (const srd::DevMeshConst*)s->d_mesh_const.p;
    s->dev.mesh_tex = (const srd::DevMeshTex*)s->d_mesh_tex.p;
    s->dev.textures = (const srd::DevTexture*)s->d_textures.p;
    *any_textured_out = any_textured;
    return SR_OK;
}

// OpType::Update: same instance layout, new transforms. The triangles are re-flattened on the device into their
// existing leaf slots and the quantised nodes are refitted bottom-up; topology, shade records and mesh tables stay.
int update_in_place(SrScene* s) {
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipDeviceSynchronize());
    int rc = upload_instance_tables(s);
    if (rc != SR_OK) return rc;
    int e = srk_launch_flatten_slots((float4*)s->d_tris.p, (const float4*)s->d_shade.p, (const SrMeshInfo*)s->d_mesh_infos.p,
                                     (const srd::FlatInstance*)s->d_flat_instances.p, s->fid.n_triangles, nullptr);
    if (e != 0) return fail(SR_ERR_HIP, std::string("flatten launch failed: ") + hipGetErrorString((hipError_t)e));
    e = srk_launch_refit((uint32_t*)s->d_nodes.p, (const float4*)s->d_tris.p, (float*)s->d_node_box.p, (const uint32_t*)s->d_level_nodes.p,
                         s->level_offsets.data(), (uint32_t)s->level_offsets.size() - 1, nullptr);
    if (e != 0) return fail(SR_ERR_HIP, std::string("refit launch failed: ") + hipGetErrorString((hipError_t)e));
    HIP_TRY(hipDeviceSynchronize());
    s->stats.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SR_OK;
}

int full_build(SrScene* s);

// ---- two-level form -------------------------------------------------------------------------------------------------
// The reference instances BLASes through a TLAS it rebuilds or updates every frame (tlas.rs:155-191,
// resource_manager.rs:236-251). The one-level form copies every instance's triangles into one world-space tree — memory and
// update cost O(instances x triangles). This form keeps one tree per MESH in object space and a top-level tree over padded
// instance boxes: a changed instance list costs a top-level rebuild (host, O(instances log instances)) and one 128-byte record
// per instance, whatever the meshes hold. Hits are those of the one-level form bit for bit (traverse.h: traverse_ws with TL = true).
constexpr uint32_t kTlStackCap = 47;     // LDS stack entries a two-level walk may need (top-level + pending instances of a leaf + marker + mesh
                                         // tree): with the spare level and the 8 work rows, 56 rows = 56 KB for the 256-thread queue tracers

constexpr double kTlMaxCondition = 100.0;    // ||W2O||_inf * ||O2W||_inf above which an instance is baked (rotation + uniform scale: <= 3)

uint32_t min_depth_for(uint64_t n_items) {   // binary depth a median split needs to reach leaves of <= 2 items
    uint32_t need = 2;
    for (uint64_t c = (n_items + 1) / 2; c > 1; c = (c + 1) / 2) need++;
    return need;
}
uint32_t depth_for(uint64_t n_items) {       // ... with slack for SAH splits (the collapse widens nodes until the budget is used)
    return std::max(6u, std::min((uint32_t)srd::kMaxBinaryDepth, min_depth_for(n_items) + 6u));
}

// Object-space tree of one mesh (OpType::SlowBuild of a BLAS, blas.rs:178): same builder, same triangle padding. With `bake` the
// positions are first taken to world space by that transform (transform_point's operation order, as the one-level form flattens):
// the private copy an instance gets whose transform cannot be inverted (see two_level_build).
int build_blas(const srh::HostMesh& mesh, uint32_t mesh_slot, const SrTransform* bake, SrScene::HostBlas& b) {
    const uint32_t n = mesh.n_indices / 3;
    std::vector<float> pos((size_t)mesh.n_vertices * 3);
    for (uint32_t i = 0; i < mesh.n_vertices; i++) {
        const float* q = mesh.vertices[i].position;
        float* w = &pos[(size_t)i * 3];
        if (bake) {
            const float* m = bake->m;
            w[0] = ((m[0] * q[0] + m[1] * q[1]) + m[2] * q[2]) + m[3] * 1.0f;
            w[1] = ((m[4] * q[0] + m[5] * q[1]) + m[6] * q[2]) + m[7] * 1.0f;
            w[2] = ((m[8] * q[0] + m[9] * q[1]) + m[10] * q[2]) + m[11] * 1.0f;
        } else { w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; }
    }
    std::vector<srh::BuildTri> tris(n);
    float max_edge = 0.0f, max_abs = 0.0f;
    for (uint32_t p = 0; p < n; p++) {
        const float* v[3];
        for (int j = 0; j < 3; j++) v[j] = &pos[(size_t)mesh.indices[3 * p + j] * 3];
        srh::BuildTri& t = tris[p];
        for (int a = 0; a < 3; a++) {
            t.v0[a] = v[0][a]; t.e1[a] = v[1][a] - v[0][a]; t.e2[a] = v[2][a] - v[0][a];
            max_edge = std::max(max_edge, std::fabs(t.e1[a]) + std::fabs(t.e2[a]));
            for (int j = 0; j < 3; j++) max_abs = std::max(max_abs, std::fabs(v[j][a]));
        }
        t.prim = p; t.inst = 0; t.gid = p;
    }
    srh::BvhResult bvh;
    srh::build_bvh(tris, std::min(depth_for(n), 26u), bvh);          // leaves the top-level tree at least 18 of the kTlStackCap entries
    b.nodes.swap(bvh.nodes);
    b.n_tris = n; b.n_nodes = bvh.n_nodes; b.max_stack = bvh.max_stack; b.max_depth = bvh.max_depth; b.build_ms = bvh.build_ms;
    b.max_edge_sum = max_edge; b.max_abs_vertex = max_abs;
    b.tris.assign((size_t)n * 12, 0.0f);
    b.shade.assign((size_t)n * 12, 0.0f);
    b.slot_of_prim.assign(n ? n : 1, 0u);
    const bool textured = [&] { const uint32_t* tex = &mesh.material.base_color_image; for (int i = 0; i < 10; i += 2) if (tex[i] != SR_NULL_TEXTURE) return true; return false; }();
    if (textured) b.shade_tex.assign((size_t)n * 24, 0.0f); else b.shade_tex.clear();
    for (int a = 0; a < 3; a++) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
    for (uint32_t sl = 0; sl < n; sl++) {
        const uint32_t p = bvh.order[sl];
        const SrVertex* v[3];
        const float* w[3];
        for (int j = 0; j < 3; j++) { v[j] = &mesh.vertices[mesh.indices[3 * p + j]]; w[j] = &pos[(size_t)mesh.indices[3 * p + j] * 3]; }
        float* q = &b.tris[(size_t)sl * 12];
        for (int j = 0; j < 3; j++) memcpy(q + 3 * j, w[j], 12);                    // v0, v1, v2 (object space; world space when baked)
        memcpy(q + 9, &p, 4);                                                       // primitive index
        float* sh = &b.shade[(size_t)sl * 12];
        for (int j = 0; j < 3; j++) memcpy(sh + 3 * j, v[j]->normal, 12);
        memcpy(sh + 10, &mesh_slot, 4);                                             // mesh slot; the instance comes from the walk
        if (textured) {
            float* tx = &b.shade_tex[(size_t)sl * 24];
            for (int j = 0; j < 3; j++) { memcpy(tx + 2 * j, v[j]->base_color_tex_coord, 8); memcpy(tx + 6 + 2 * j, v[j]->normal_tex_coord, 8); }
            memcpy(tx + 12, v[0]->tangent, 12);
            tx[15] = v[0]->tangent[3] >= 0.0f ? 1.0f : -1.0f;
            memcpy(tx + 16, v[1]->tangent, 12);
            memcpy(tx + 19, v[2]->tangent, 12);
        }
        b.slot_of_prim[p] = sl;
        const srh::BuildTri& t = tris[p];
        for (int a = 0; a < 3; a++) {                                               // the padded triangle box, as the builder bounds it
            const float pad = 4e-6f * (std::fabs(t.e1[a]) + std::fabs(t.e2[a]));
            const float lo = std::min(w[0][a], std::min(w[1][a], w[2][a])) - pad;
            const float hi = std::max(w[0][a], std::max(w[1][a], w[2][a])) + pad;
            b.lo[a] = std::min(b.lo[a], std::nextafter(lo, -INFINITY)); b.hi[a] = std::max(b.hi[a], std::nextafter(hi, INFINITY));
        }
    }
    b.valid = true;
    return SR_OK;
}

// Appends one tree with its triangle / shade / primitive -> slot records to the concatenated host arrays (references become global).
struct BlasCat {
    std::vector<uint32_t> nodes, slot_of_prim;
    std::vector<float> tris, shade, shade_tex;
    bool textured = false;
    void append(const SrScene::HostBlas& b, uint32_t* node_base, uint32_t* tri_base) {
        const uint32_t nb = (uint32_t)(nodes.size() / srl::kNodeDwords), tb = (uint32_t)(tris.size() / 12);
        *node_base = nb; *tri_base = tb;
        nodes.resize(nodes.size() + (size_t)b.n_nodes * srl::kNodeDwords);
        for (uint32_t i = 0; i < b.n_nodes; i++) {
            uint32_t* q = &nodes[((size_t)nb + i) * srl::kNodeDwords];
            memcpy(q, &b.nodes[(size_t)i * srl::kNodeDwords], srl::kNodeBytes);
            for (int c = 0; c < srl::kBvhWidth; c++) {
                const int ref = (int)q[srl::kChildOffset + c];
                if (ref >= 0) q[srl::kChildOffset + c] = (uint32_t)(ref + (int)nb);
                else { const uint32_t lv = ~(uint32_t)ref; const uint32_t cnt = lv & 7u; if (cnt) q[srl::kChildOffset + c] = ~((((lv >> 3) + tb) << 3) | cnt); }
            }
        }
        tris.insert(tris.end(), b.tris.begin(), b.tris.end());
        shade.insert(shade.end(), b.shade.begin(), b.shade.end());
        if (textured) {
            shade_tex.resize((size_t)tb * 24, 0.0f);
            if (!b.shade_tex.empty()) shade_tex.insert(shade_tex.end(), b.shade_tex.begin(), b.shade_tex.end());
            else shade_tex.resize(((size_t)tb + b.n_tris) * 24, 0.0f);
        }
        for (uint32_t p = 0; p < b.n_tris; p++) slot_of_prim.push_back(tb + b.slot_of_prim[p]);
    }
};

int two_level_build(SrScene* s) {
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    const size_t nm = s->meshes.size();
    s->blases.resize(nm);
    bool any_textured = false;
    if ((rc = upload_mesh_tables(s, &any_textured)) != SR_OK) return rc;
    // instance records + padded world-space boxes
    const size_t ni = s->fid.instances.size();
    std::vector<srd::DevTlInstance> recs(ni ? ni : 1);
    memset(recs.data(), 0, recs.size() * sizeof(srd::DevTlInstance));
    std::vector<srh::BuildBox> boxes;
    std::vector<uint32_t> box_inst;
    boxes.reserve(ni); box_inst.reserve(ni);
    std::vector<SrScene::HostBlas> baked;     // private world-space copies: instances whose transform cannot be inverted (well)
    std::vector<uint32_t> baked_inst;
    const double eps = std::ldexp(1.0, -24);
    uint32_t blas_stack = 0;
    for (size_t i = 0; i < ni; i++) {
        const srh::HostInstance& in = s->fid.instances[i];
        const srh::HostMesh& mesh = s->meshes[in.mesh_slot];
        SrScene::HostBlas& b = s->blases[in.mesh_slot];
        if (!b.valid) { b = SrScene::HostBlas(); if ((rc = build_blas(mesh, in.mesh_slot, nullptr, b)) != SR_OK) return rc; s->blas_device_current = false; }
        srd::DevTlInstance& r = recs[i];
        const float* M = in.o2w.m;
        memcpy(r.o2w, M, 48);
        r.tri_offset = in.tri_offset;
        r.mesh_slot = in.mesh_slot;
        if (b.n_tris == 0) continue;
        // inverse of the affine transform, in double
        const double a00 = M[0], a01 = M[1], a02 = M[2], a10 = M[4], a11 = M[5], a12 = M[6], a20 = M[8], a21 = M[9], a22 = M[10];
        const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
        const double det = a00 * c00 + a01 * c01 + a02 * c02;
        const double id = 1.0 / det;
        const double R[9] = {c00 * id, (a02 * a21 - a01 * a22) * id, (a01 * a12 - a02 * a11) * id,
                             c01 * id, (a00 * a22 - a02 * a20) * id, (a02 * a10 - a00 * a12) * id,
                             c02 * id, (a01 * a20 - a00 * a21) * id, (a00 * a11 - a01 * a10) * id};
        const double T[3] = {M[3], M[7], M[11]};
        double r_norm = 0.0, m_norm = 0.0, t_max = 0.0;
        bool finite = std::isfinite(id) && det != 0.0;
        for (int row = 0; row < 3; row++) {
            for (int c = 0; c < 3; c++) r.w2o[4 * row + c] = (float)R[3 * row + c];
            r.w2o[4 * row + 3] = (float)(-(R[3 * row] * T[0] + R[3 * row + 1] * T[1] + R[3 * row + 2] * T[2]));
            r_norm = std::max(r_norm, std::fabs(R[3 * row]) + std::fabs(R[3 * row + 1]) + std::fabs(R[3 * row + 2]));
            m_norm = std::max(m_norm, std::fabs((double)M[4 * row]) + std::fabs((double)M[4 * row + 1]) + std::fabs((double)M[4 * row + 2]));
            t_max = std::max(t_max, std::fabs(T[row]));
            for (int c = 0; c < 4; c++) finite = finite && std::isfinite(r.w2o[4 * row + c]);
        }
        // A transform of rank 2 still yields real (flat) world-space triangles, and a badly conditioned one stretches object space
        // against world space: whatever the fp32 triangle test's own rounding moves a hit by in world space (on sliver triangles
        // that is far more than a box's padding: the round-3 fuzzer found hits 8e-3 off their triangle) is multiplied by
        // ||W2O|| on the way into the mesh's boxes. Such an instance gets a private world-space copy of its mesh's tree and is walked
        // without a ray transform: there the boxes see exactly what the one-level form's boxes see.
        if (!finite || !(r_norm * m_norm < kTlMaxCondition)) {
            memset(r.w2o, 0, sizeof(r.w2o));
            r.w2o[0] = r.w2o[5] = r.w2o[10] = 1.0f;
            r.flags = 1u;
            baked.emplace_back();
            if ((rc = build_blas(mesh, in.mesh_slot, &in.o2w, baked.back())) != SR_OK) return rc;
            baked_inst.push_back((uint32_t)i);
            const SrScene::HostBlas& wb = baked.back();
            bool box_ok = true;
            srh::BuildBox bx;
            for (int a = 0; a < 3; a++) { bx.lo[a] = wb.lo[a]; bx.hi[a] = wb.hi[a]; box_ok = box_ok && std::isfinite(bx.lo[a]) && std::isfinite(bx.hi[a]); }
            blas_stack = std::max(blas_stack, wb.max_stack);
            if (box_ok) { boxes.push_back(bx); box_inst.push_back((uint32_t)i); }
            continue;
        }
        blas_stack = std::max(blas_stack, b.max_stack);
        // world box: the 8 corners of the mesh's (already padded) box, then padding for the rounding of the transformed vertices
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, p_max = 0.0;
        for (int corner = 0; corner < 8; corner++) {
            const double x = (corner & 1) ? b.hi[0] : b.lo[0], y = (corner & 2) ? b.hi[1] : b.lo[1], z = (corner & 4) ? b.hi[2] : b.lo[2];
            for (int row = 0; row < 3; row++) {
                const double w = (double)M[4 * row] * x + (double)M[4 * row + 1] * y + (double)M[4 * row + 2] * z + (double)M[4 * row + 3];
                lo[row] = std::min(lo[row], w); hi[row] = std::max(hi[row], w);
                p_max = std::max(p_max, std::fabs(w));
            }
        }
        const double pad_w = 64.0 * eps * (p_max + m_norm * b.max_abs_vertex + t_max) + 8e-6 * m_norm * b.max_edge_sum;
        srh::BuildBox bx;
        for (int a = 0; a < 3; a++) { bx.lo[a] = std::nextafter((float)(lo[a] - pad_w), -INFINITY); bx.hi[a] = std::nextafter((float)(hi[a] + pad_w), INFINITY); }
        // widening of the mesh's object-space boxes: rounding of the ray transform (grows with the ray origin) and of the world-space
        // vertices, plus the barycentric slack of the world-space triangle test seen from object space
        r.pad_a = (float)(64.0 * eps * r_norm);
        r.pad_b = (float)(r_norm * (64.0 * eps * (p_max + t_max + m_norm * b.max_abs_vertex) + 8e-6 * m_norm * b.max_edge_sum));
        boxes.push_back(bx);
        box_inst.push_back((uint32_t)i);
    }
    // every live mesh's records, one after the other (cached on the host per mesh), then the baked copies of this instance list
    const bool had_baked = s->tl_baked;
    if (!s->blas_device_current || !s->two_level || had_baked || !baked.empty()) {
        BlasCat cat;
        for (size_t m = 0; m < nm; m++) {
            if (s->meshes[m].n_vertices == 0) continue;
            if (!s->blases[m].valid) { s->blases[m] = SrScene::HostBlas(); if ((rc = build_blas(s->meshes[m], (uint32_t)m, nullptr, s->blases[m])) != SR_OK) return rc; }
            cat.textured = cat.textured || !s->blases[m].shade_tex.empty();
        }
        s->blas_node_base.assign(nm, 0u); s->blas_tri_base.assign(nm, 0u);
        for (size_t m = 0; m < nm; m++) if (s->meshes[m].n_vertices) cat.append(s->blases[m], &s->blas_node_base[m], &s->blas_tri_base[m]);
        s->tl_baked_node_base.assign(baked.size(), 0u); s->tl_baked_tri_base.assign(baked.size(), 0u);
        for (size_t k = 0; k < baked.size(); k++) cat.append(baked[k], &s->tl_baked_node_base[k], &s->tl_baked_tri_base[k]);
        if (cat.tris.size() / 12 >= (1ull << 28) || cat.nodes.size() / srl::kNodeDwords >= (1ull << 31)) return fail(SR_ERR_UNSUPPORTED, "the meshes together exceed 2^28 triangles (leaf reference encoding)");
        if (cat.textured) cat.shade_tex.resize(cat.tris.size() / 12 * 24, 0.0f);
        if (cat.slot_of_prim.empty()) cat.slot_of_prim.push_back(0u);
        HIP_TRY(hipDeviceSynchronize());
        if ((rc = s->d_blas_nodes.upload(cat.nodes.data(), cat.nodes.size() * 4)) != SR_OK) return rc;
        if ((rc = s->d_tris.upload(cat.tris.data(), cat.tris.size() * 4)) != SR_OK) return rc;
        if ((rc = s->d_shade.upload(cat.shade.data(), cat.shade.size() * 4)) != SR_OK) return rc;
        if (cat.textured) { if ((rc = s->d_shade_tex.upload(cat.shade_tex.data(), cat.shade_tex.size() * 4)) != SR_OK) return rc; }
        else s->d_shade_tex.release();
        if ((rc = s->d_slot_of_gid.upload(cat.slot_of_prim.data(), cat.slot_of_prim.size() * 4)) != SR_OK) return rc;
        s->any_textured_tl = cat.textured;
        s->tl_blas_nodes = cat.nodes.size() / srl::kNodeDwords; s->tl_blas_tris = cat.tris.size() / 12;
        s->blas_device_current = true;
        s->tl_baked = !baked.empty();
    }
    for (size_t i = 0; i < ni; i++) {
        srd::DevTlInstance& r = recs[i];
        if (r.flags & 1u) continue;
        r.blas_root = s->blas_node_base[r.mesh_slot];
        r.prim_base = s->blas_tri_base[r.mesh_slot];
    }
    for (size_t k = 0; k < baked.size(); k++) { recs[baked_inst[k]].blas_root = s->tl_baked_node_base[k]; recs[baked_inst[k]].prim_base = s->tl_baked_tri_base[k]; }
    s->blas_stack = blas_stack;
    // the stack budget of the top-level tree is what the deepest mesh tree leaves of the walk's LDS stack
    const uint32_t left = kTlStackCap > s->blas_stack + srl::kLeafMax + 1u ? kTlStackCap - s->blas_stack - srl::kLeafMax - 1u : 0u;
    if (left < min_depth_for(boxes.size())) return fail(SR_ERR_UNSUPPORTED, "two-level structure: instance count and mesh size together need a deeper traversal stack than the kernels provide");
    srh::BvhResult tl;
    srh::build_bvh_boxes(boxes, std::min(depth_for(boxes.size()), left), tl);
    std::vector<uint32_t> tl_inst(tl.order.size() ? tl.order.size() : 1, 0u);
    for (size_t k = 0; k < tl.order.size(); k++) tl_inst[k] = box_inst[tl.order[k]];
    const uint32_t need = tl.max_stack + srl::kLeafMax + s->blas_stack + 1u;     // top-level entries + pending instances of a leaf + mesh tree + 1
    if (need > kTlStackCap) return fail(SR_ERR_STATE, "two-level structure needs a deeper traversal stack than the kernels provide");
    HIP_TRY(hipDeviceSynchronize());
    if ((rc = upload_instance_tables(s)) != SR_OK) return rc;
    if ((rc = s->d_nodes.upload(tl.nodes.data(), tl.nodes.size() * 4)) != SR_OK) return rc;
    if ((rc = s->d_tl_inst.upload(tl_inst.data(), tl_inst.size() * 4)) != SR_OK) return rc;
    if ((rc = s->d_tl_instances.upload(recs.data(), recs.size() * sizeof(srd::DevTlInstance))) != SR_OK) return rc;
    s->dev.nodes = (const float4*)s->d_nodes.p;
    s->dev.blas_nodes = (const float4*)s->d_blas_nodes.p;
    s->dev.tl_inst = (const uint32_t*)s->d_tl_inst.p;
    s->dev.tl_instances = (const srd::DevTlInstance*)s->d_tl_instances.p;
    s->dev.tris = (const float4*)s->d_tris.p;
    s->dev.shade = (const float4*)s->d_shade.p;
    s->dev.shade_tex = (const float4*)s->d_shade_tex.p;
    s->dev.slot_of_gid = (const uint32_t*)s->d_slot_of_gid.p;
    s->dev.counters = (unsigned long long*)s->d_misc.p;
    s->dev.n_tris = s->fid.n_triangles;
    s->stats.n_triangles = s->fid.n_triangles;
    s->stats.n_nodes = tl.n_nodes + s->tl_blas_nodes;
    s->stats.node_bytes = s->stats.n_nodes * srl::kNodeBytes;
    s->stats.tri_bytes = s->tl_blas_tris * 48;
    s->stats.max_depth = tl.max_depth;
    s->stats.max_stack = need;
    s->stack_entries = (int)((std::max(need, 3u) + 1u + 3u) & ~3u);
    s->stats.sah_cost = tl.sah_cost;
    s->stats.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    s->shape.clear();
    s->built = true;
    s->two_level = true;
    s->last_build_on_device = false;
    return SR_OK;
}

// The form this instance list is built in: on request, or (auto) where instancing really shares geometry and the flattened copy
// would be large — more than 2^24 flattened triangles and at least four times the meshes' own.
bool wants_two_level(const SrScene* s) {
    if (s->instancing == SR_INSTANCING_TWO_LEVEL) return true;
    if (s->instancing == SR_INSTANCING_FLAT) return false;
    uint64_t own = 0;
    std::vector<char> seen(s->meshes.size(), 0);
    for (const auto& in : s->fid.instances) if (!seen[in.mesh_slot]) { seen[in.mesh_slot] = 1; own += s->meshes[in.mesh_slot].n_indices / 3; }
    return s->fid.n_triangles > (1u << 24) && (uint64_t)s->fid.n_triangles >= 4ull * own;
}

// OpType::FastBuild: linear BVH built on the device (bvh_gpu.hip). Falls back to the host builder for small scenes
// and for trees that would need a deeper traversal stack than one workgroup's LDS share.
constexpr uint32_t kDeviceBuildMinTris = 4096;
constexpr uint32_t kDeviceStackCap = 47;
int fast_build(SrScene* s) {
    const uint32_t n = s->fid.n_triangles;
    if (n < kDeviceBuildMinTris) return full_build(s);
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    HIP_TRY(hipDeviceSynchronize());
    bool any_textured = false;
    if ((rc = upload_mesh_tables(s, &any_textured)) != SR_OK) return rc;
    if ((rc = upload_instance_tables(s)) != SR_OK) return rc;
    if ((rc = s->d_mesh_infos.upload(s->mesh_infos.data(), s->mesh_infos.size() * sizeof(SrMeshInfo))) != SR_OK) return rc;
    const uint32_t node_cap = n + 1024;   // inner nodes of a tree with >= 2 children per node and >= 1 triangle per leaf: < n
    if ((rc = s->d_nodes.reserve((size_t)node_cap * srl::kNodeBytes)) != SR_OK || (rc = s->d_node_box.reserve((size_t)node_cap * 24)) != SR_OK ||
        (rc = s->d_tris.reserve((size_t)n * 48)) != SR_OK || (rc = s->d_shade.reserve((size_t)n * 48)) != SR_OK ||
        (rc = s->d_slot_of_gid.reserve((size_t)n * 4)) != SR_OK) return rc;
    if (any_textured) { if ((rc = s->d_shade_tex.reserve((size_t)n * 96)) != SR_OK) return rc; }
    else s->d_shade_tex.release();
    const size_t scratch = srk_lbvh_scratch_bytes(n, node_cap);
    if ((rc = s->d_scratch.reserve(scratch)) != SR_OK) return rc;
    LbvhArgs a;
    a.meshes = (const SrMeshInfo*)s->d_mesh_infos.p; a.instances = (const srd::FlatInstance*)s->d_flat_instances.p;
    a.n_instances = (uint32_t)s->fid.instances.size(); a.n_tris = n;
    a.nodes = (float4*)s->d_nodes.p; a.node_cap = node_cap;
    a.tris = (float4*)s->d_tris.p; a.shade = (float4*)s->d_shade.p; a.shade_tex = (float4*)s->d_shade_tex.p;
    a.slot_of_gid = (uint32_t*)s->d_slot_of_gid.p; a.node_box = (float*)s->d_node_box.p;
    a.scratch = s->d_scratch.p; a.scratch_bytes = s->d_scratch.bytes;
    a.stack_floor = (uint32_t)srd::kStackMax; a.stack_cap = kDeviceStackCap;
    a.ploc = s->fast_build_ploc;
    LbvhResult r;
    const int e = srk_lbvh_build(a, &r, nullptr);
    if (e > 0) return fail(SR_ERR_HIP, std::string("device BVH build failed: ") + hipGetErrorString((hipError_t)e));
    if (e < 0) return full_build(s);                      // tree outside the limits: quality build on the host instead
    std::vector<uint32_t> level_nodes;
    s->level_offsets.assign(1, 0u);
    for (size_t l = r.level_ranges.size(); l-- > 0;) {
        for (uint32_t k = 0; k < r.level_ranges[l].second; k++) level_nodes.push_back(r.level_ranges[l].first + k);
        s->level_offsets.push_back((uint32_t)level_nodes.size());
    }
    if ((rc = s->d_level_nodes.upload(level_nodes.data(), level_nodes.size() * 4)) != SR_OK) return rc;
    s->shape.resize(s->fid.instances.size());
    for (size_t i = 0; i < s->shape.size(); i++) s->shape[i] = s->fid.instances[i].mesh_slot;
    s->dev.nodes = (const float4*)s->d_nodes.p;
    s->dev.tris = (const float4*)s->d_tris.p;
    s->dev.shade = (const float4*)s->d_shade.p;
    s->dev.shade_tex = (const float4*)s->d_shade_tex.p;
    s->dev.slot_of_gid = (const uint32_t*)s->d_slot_of_gid.p;
    s->dev.counters = (unsigned long long*)s->d_misc.p;
    s->dev.n_tris = n;
    s->stats.n_triangles = n;
    s->stats.n_nodes = r.n_nodes;
    s->stats.node_bytes = (uint64_t)r.n_nodes * srl::kNodeBytes;
    s->stats.tri_bytes = (uint64_t)n * 48;
    s->stats.max_depth = r.max_depth;
    s->stats.max_stack = r.max_stack;
    s->stack_entries = (int)((std::max(r.max_stack, 3u) + 1u + 3u) & ~3u);
    s->stats.sah_cost = 0.0f;
    s->stats.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    s->built = true;
    s->last_build_on_device = true;
    return SR_OK;
}

}  // namespace

int sr_scene_set_instances(SrScene* s, const uint64_t* keys, const uint32_t* counts, uint32_t n_keys, const SrTransform* transforms) {
    if (!s || (n_keys && (!keys || !counts))) return fail(SR_ERR_INVALID_ARG, "frame_instance_data: null argument");
    // sizes first, in 64 bits: the per-instance triangle offsets are 32-bit sums and a leaf reference holds 28 bits
    uint64_t n_inst = 0, n_tri = 0;
    for (uint32_t k = 0; k < n_keys; k++) {
        n_inst += counts[k];
        auto it = s->slots.find(keys[k]);
        if (it != s->slots.end()) n_tri += (uint64_t)counts[k] * (s->meshes[it->second].n_indices / 3);
    }
    if (n_inst && !transforms) return fail(SR_ERR_INVALID_ARG, "frame_instance_data: transforms is null but instances were given");
    if (n_tri >= 0xFFFFFFFFull) return fail(SR_ERR_UNSUPPORTED, "scene exceeds 2^32 - 1 triangles (32-bit global triangle index)");
    if (n_tri >= (1ull << 28) && s->instancing == SR_INSTANCING_FLAT) return fail(SR_ERR_UNSUPPORTED, "scene exceeds 2^28 triangles (leaf reference encoding of the one-level form)");
    if (n_inst >= (1ull << 28)) return fail(SR_ERR_UNSUPPORTED, "scene exceeds 2^28 instances");
    for (uint64_t i = 0; i < n_inst; i++)
        for (int c = 0; c < 12; c++)
            if (!std::isfinite(transforms[i].m[c])) return fail(SR_ERR_INVALID_ARG, "frame_instance_data: instance transform holds a non-finite value");
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    std::string err;
    srh::FrameInstanceData fid;
    if (!srh::frame_instance_data(s->meshes, s->slots, keys, counts, n_keys, transforms, fid, err)) return fail(SR_ERR_INVALID_ARG, err);
    s->fid = std::move(fid);
    s->emissive_table = s->emissive_tris;
    if (s->emissive_table.empty()) { SrEmissiveTriangle z; memset(&z, 0, sizeof(z)); s->emissive_table.push_back(z); }
    // Two-level form (a tree per mesh + a top-level tree over the instances): on request or where the flattened copy would be
    // large. A changed instance list is then a top-level rebuild, reported as a fast build (Tlas::queue_build rebuilds in kind).
    if (wants_two_level(s) || s->fid.n_triangles >= (1u << 28)) {
        const uint32_t op = s->built_once ? SR_OP_FAST_BUILD : SR_OP_SLOW_BUILD;
        s->forced_op = SR_OP_NONE;
        if (!s->two_level) s->built = false;
        rc = two_level_build(s);
        if (rc != SR_OK) { s->built = false; return rc; }
        if (s->built_once) srh::as_state_mark_built(s->as_state, op);
        s->built_once = true;
        s->last_op = op;
        return SR_OK;
    }
    if (s->two_level) {                       // back to the one-level form: everything is rebuilt
        s->two_level = false; s->built = false;
        s->dev.blas_nodes = nullptr; s->dev.tl_inst = nullptr; s->dev.tl_instances = nullptr;
        s->forced_op = s->forced_op == SR_OP_NONE ? SR_OP_SLOW_BUILD : s->forced_op;
    }
    // Tlas::queue_build (tlas.rs:155-191): the instance data is new, so the heuristic is asked with inputs_changed =
    // true; an UPDATE needs the same instance layout (here: the same mesh per instance and unchanged meshes),
    // anything else is a rebuild. The very first build is the quality build (Tlas::new).
    bool can_update = s->built && s->shape.size() == s->fid.instances.size() && s->fid.n_triangles > 0;
    for (size_t i = 0; can_update && i < s->shape.size(); i++) can_update = s->shape[i] == s->fid.instances[i].mesh_slot;
    uint32_t op;
    if (!s->built_once) op = SR_OP_SLOW_BUILD;
    else {
        op = srh::as_state_next_op(s->as_state, true);
        if (op == SR_OP_UPDATE && !can_update) op = SR_OP_FAST_BUILD;
    }
    if (s->forced_op != SR_OP_NONE) { op = (s->forced_op == SR_OP_UPDATE && !can_update) ? SR_OP_FAST_BUILD : s->forced_op; s->forced_op = SR_OP_NONE; }
    rc = op == SR_OP_UPDATE ? update_in_place(s) : op == SR_OP_FAST_BUILD ? fast_build(s) : full_build(s);
    if (rc != SR_OK) { s->built = false; return rc; }
    if (s->built_once) srh::as_state_mark_built(s->as_state, op);
    s->built_once = true;
    s->last_op = op;
    return SR_OK;
}

int sr_scene_end_frame(SrScene* s) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "sr_scene_end_frame: scene is null");
    if (!s->built) { s->last_op = SR_OP_NONE; return SR_OK; }
    const uint32_t op = srh::as_state_next_op(s->as_state, false);
    if (op == SR_OP_SLOW_BUILD) {
        int rc = bind_device(s);
        if (rc != SR_OK) return rc;
        if ((rc = s->two_level ? two_level_build(s) : full_build(s)) != SR_OK) { s->built = false; return rc; }
    }
    srh::as_state_mark_built(s->as_state, op);
    s->last_op = op;
    return SR_OK;
}

int sr_scene_set_instancing(SrScene* s, uint32_t mode) {
    if (!s || mode > SR_INSTANCING_TWO_LEVEL) return fail(SR_ERR_INVALID_ARG, "sr_scene_set_instancing: bad argument");
    s->instancing = (int)mode;
    return SR_OK;
}
int sr_scene_instancing(const SrScene* s, uint32_t* mode, uint32_t* two_level_now) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "sr_scene_instancing: scene is null");
    if (mode) *mode = (uint32_t)s->instancing;
    if (two_level_now) *two_level_now = (s->built && s->two_level) ? 1u : 0u;
    return SR_OK;
}

int sr_scene_force_next_op(SrScene* s, uint32_t op) {
    if (!s || op > SR_OP_UPDATE) return fail(SR_ERR_INVALID_ARG, "sr_scene_force_next_op: bad argument");
    s->forced_op = op;
    return SR_OK;
}

int sr_scene_as_state(const SrScene* s, SrAsState* state, uint32_t* last_op) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "sr_scene_as_state: scene is null");
    if (state) *state = s->as_state;
    if (last_op) *last_op = s->last_op;
    return SR_OK;
}

void sr_as_state_initial(uint32_t build_type, SrAsState* out) { if (out) srh::as_state_initial(build_type, out); }
uint32_t sr_as_state_next_op(const SrAsState* state, int inputs_changed) { return state ? srh::as_state_next_op(*state, inputs_changed != 0) : SR_OP_NONE; }
void sr_as_state_mark_built(SrAsState* state, uint32_t completed_op) { if (state) srh::as_state_mark_built(*state, completed_op); }

int sr_bvh_layout(uint32_t* width, uint32_t* node_dwords, uint32_t* plane_offset, uint32_t* child_offset) {
    if (width) *width = (uint32_t)srl::kBvhWidth;
    if (node_dwords) *node_dwords = (uint32_t)srl::kNodeDwords;
    if (plane_offset) *plane_offset = (uint32_t)srl::kPlaneOffset;
    if (child_offset) *child_offset = (uint32_t)srl::kChildOffset;
    return SR_OK;
}

int sr_scene_read_bvh(const SrScene* s, uint32_t* nodes_out, float* tris_out) {
    if (!s || !s->built) return fail(SR_ERR_STATE, "sr_scene_read_bvh: scene not built");
    if (s->two_level) return fail(SR_ERR_UNSUPPORTED, "sr_scene_read_bvh: the scene is built in the two-level form");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    if (nodes_out) HIP_TRY(hipMemcpy(nodes_out, s->d_nodes.p, (size_t)s->stats.n_nodes * srl::kNodeBytes, hipMemcpyDeviceToHost));
    if (tris_out && s->fid.n_triangles) HIP_TRY(hipMemcpy(tris_out, s->d_tris.p, (size_t)s->fid.n_triangles * 48, hipMemcpyDeviceToHost));
    return SR_OK;
}

namespace {
// OpType::SlowBuild / FastBuild: the whole structure from the current instance list.
int full_build(SrScene* s) {
    int rc;
    HIP_TRY(hipDeviceSynchronize());
    srh::flatten_instances(s->meshes, s->fid, s->world_tris);
    srh::BvhResult bvh;
    srh::build_bvh(s->world_tris, (uint32_t)srd::kMaxBinaryDepth, bvh);
    if (bvh.max_stack > (uint32_t)srd::kStackMax) return fail(SR_ERR_STATE, "BVH needs a deeper traversal stack than the kernels provide");
    // shade records (object-space vertex normals + instance + mesh slot) in leaf order, slot lookup
    const uint32_t n_tris = s->fid.n_triangles;
    std::vector<float> shade((size_t)n_tris * 12, 0.0f);
    std::vector<uint32_t> slot_of_gid(n_tris ? n_tris : 1, 0u);
    for (uint32_t slot = 0; slot < n_tris; slot++) {
        const srh::BuildTri& t = s->world_tris[bvh.order[slot]];
        const srh::HostInstance& inst = s->fid.instances[t.inst];
        const srh::HostMesh& mesh = s->meshes[inst.mesh_slot];
        float* q = &shade[(size_t)slot * 12];
        for (int j = 0; j < 3; j++) memcpy(q + 3 * j, mesh.vertices[mesh.indices[3 * t.prim + j]].normal, 12);
        memcpy(q + 9, &t.inst, 4);
        memcpy(q + 10, &inst.mesh_slot, 4);
        slot_of_gid[t.gid] = slot;
    }
    bool any_textured = false;
    if ((rc = upload_mesh_tables(s, &any_textured)) != SR_OK) return rc;
    std::vector<float> shade_tex;
    if (any_textured) {
        shade_tex.assign((size_t)n_tris * 24, 0.0f);
        for (uint32_t slot = 0; slot < n_tris; slot++) {
            const srh::BuildTri& t = s->world_tris[bvh.order[slot]];
            const srh::HostMesh& mesh = s->meshes[s->fid.instances[t.inst].mesh_slot];
            float* q = &shade_tex[(size_t)slot * 24];
            const SrVertex* v[3];
            for (int j = 0; j < 3; j++) v[j] = &mesh.vertices[mesh.indices[3 * t.prim + j]];
            for (int j = 0; j < 3; j++) {
                memcpy(q + 2 * j, v[j]->base_color_tex_coord, 8);
                memcpy(q + 6 + 2 * j, v[j]->normal_tex_coord, 8);
            }
            memcpy(q + 12, v[0]->tangent, 12);
            q[15] = v[0]->tangent[3] >= 0.0f ? 1.0f : -1.0f;   // handedness from the first vertex only (closest_hit.slang:34)
            memcpy(q + 16, v[1]->tangent, 12);
            memcpy(q + 19, v[2]->tangent, 12);
        }
    }
    // device upload (synchronous, like the reference's scene-load BLAS build: blas.rs:178)
    HIP_TRY(hipDeviceSynchronize());
    if ((rc = upload_instance_tables(s)) != SR_OK) return rc;
    // inputs of later in-place updates: per-level node lists, exact node boxes (filled by the first refit), mesh table
    std::vector<uint32_t> level_nodes;
    srh::tree_levels(bvh.nodes, level_nodes, s->level_offsets);
    if ((rc = s->d_level_nodes.upload(level_nodes.data(), level_nodes.size() * 4)) != SR_OK) return rc;
    std::vector<float> node_box((size_t)bvh.n_nodes * 6, 0.0f);
    if ((rc = s->d_node_box.upload(node_box.data(), node_box.size() * 4)) != SR_OK) return rc;
    if ((rc = s->d_mesh_infos.upload(s->mesh_infos.data(), s->mesh_infos.size() * sizeof(SrMeshInfo))) != SR_OK) return rc;
    s->shape.resize(s->fid.instances.size());
    for (size_t i = 0; i < s->shape.size(); i++) s->shape[i] = s->fid.instances[i].mesh_slot;
    if ((rc = s->d_nodes.upload(bvh.nodes.data(), bvh.nodes.size() * 4)) != SR_OK) return rc;
    if ((rc = s->d_tris.upload(bvh.tris.data(), bvh.tris.size() * 4)) != SR_OK) return rc;
    if ((rc = s->d_shade.upload(shade.data(), shade.size() * 4)) != SR_OK) return rc;
    if ((rc = s->d_slot_of_gid.upload(slot_of_gid.data(), slot_of_gid.size() * 4)) != SR_OK) return rc;
    if (any_textured) { if ((rc = s->d_shade_tex.upload(shade_tex.data(), shade_tex.size() * 4)) != SR_OK) return rc; }
    else s->d_shade_tex.release();
    s->dev.nodes = (const float4*)s->d_nodes.p;
    s->dev.tris = (const float4*)s->d_tris.p;
    s->dev.shade = (const float4*)s->d_shade.p;
    s->dev.shade_tex = (const float4*)s->d_shade_tex.p;
    s->dev.slot_of_gid = (const uint32_t*)s->d_slot_of_gid.p;
    s->dev.counters = (unsigned long long*)s->d_misc.p;
    s->dev.n_tris = s->fid.n_triangles;
    s->stats.n_triangles = s->fid.n_triangles;
    s->stats.n_nodes = bvh.n_nodes;
    s->stats.node_bytes = (uint64_t)bvh.n_nodes * srl::kNodeBytes;
    s->stats.tri_bytes = (uint64_t)s->fid.n_triangles * 48;
    s->stats.max_depth = bvh.max_depth;
    s->stats.max_stack = bvh.max_stack;
    s->stack_entries = (int)((std::max(bvh.max_stack, 3u) + 1u + 3u) & ~3u);   // + the spare level of the branch-free push
    s->stats.sah_cost = bvh.sah_cost;
    s->stats.build_ms = bvh.build_ms;
    s->built = true;
    s->last_build_on_device = false;
    return SR_OK;
}
}  // namespace

int sr_scene_get_tables(const SrScene* s, const SrTransform** transforms, uint32_t* n_instances,
                        const SrEmissiveIndirectionEntry** indirection, uint32_t* num_lights,
                        const SrEmissiveTriangle** emissive_triangles, uint32_t* n_emissive,
                        const SrMeshInfo** meshes_info, uint32_t* n_meshes) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "sr_scene_get_tables: scene is null");
    if (!s->built) return fail(SR_ERR_STATE, "sr_scene_get_tables: call sr_scene_set_instances first");
    if (transforms) *transforms = s->fid.transforms.data();
    if (n_instances) *n_instances = (uint32_t)s->fid.transforms.size();
    if (indirection) *indirection = s->fid.emissive_entries.data();
    if (num_lights) *num_lights = (uint32_t)s->fid.emissive_entries.size();
    if (emissive_triangles) *emissive_triangles = s->emissive_table.data();
    if (n_emissive) *n_emissive = (uint32_t)s->emissive_table.size();
    if (meshes_info) *meshes_info = s->mesh_infos.data();
    if (n_meshes) *n_meshes = (uint32_t)s->mesh_infos.size();
    return SR_OK;
}

int sr_scene_bvh_stats(const SrScene* s, SrBvhStats* out) {
    if (!s || !out) return fail(SR_ERR_INVALID_ARG, "sr_scene_bvh_stats: null argument");
    if (!s->built) return fail(SR_ERR_STATE, "sr_scene_bvh_stats: call sr_scene_set_instances first");
    *out = s->stats;
    return SR_OK;
}

int sr_scene_resolve_triangle(const SrScene* s, uint32_t tri, uint32_t* instance, uint32_t* primitive) {
    if (!s || !s->built) return fail(SR_ERR_STATE, "sr_scene_resolve_triangle: scene not built");
    if (tri >= s->fid.n_triangles) return fail(SR_ERR_INVALID_ARG, "sr_scene_resolve_triangle: triangle index out of range");
    size_t lo = 0, hi = s->fid.instances.size();
    while (hi - lo > 1) { size_t mid = (lo + hi) / 2; if (s->fid.instances[mid].tri_offset <= tri) lo = mid; else hi = mid; }
    if (instance) *instance = (uint32_t)lo;
    if (primitive) *primitive = tri - s->fid.instances[lo].tri_offset;
    return SR_OK;
}

static int trace_list(SrScene* s, const SrRay* rays, uint32_t n, SrHit* hits, uint32_t* occluded, int any, void* stream) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "sr_trace: scene is null");
    if (!s->built) return fail(SR_ERR_STATE, "sr_trace: call sr_scene_set_instances first (TLAS not built)");
    if (n && (!rays || (any ? (void*)occluded : (void*)hits) == nullptr)) return fail(SR_ERR_INVALID_ARG, "sr_trace: null ray/output pointer");
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    if (n > (1u << 31)) return fail(SR_ERR_UNSUPPORTED, "sr_trace: more than 2^31 rays in one call (ray indices are 32-bit)");
    hipStream_t st = (hipStream_t)stream;
    ScopedTiming tm(s, any ? kAny : kClosest, st);
    int e = srk_launch_trace(s->dev, rays, n, hits, occluded, any, s->instrumented, s->two_level ? 1 : 0, s->stack_entries, st);
    if (e != 0) return fail(SR_ERR_HIP, std::string("trace kernel launch: ") + hipGetErrorString((hipError_t)e));
    return SR_OK;
}

int sr_trace_closest(const SrScene* s, const SrRay* rays, uint32_t n, SrHit* hits, void* stream) {
    return trace_list(const_cast<SrScene*>(s), rays, n, hits, nullptr, 0, stream);
}
int sr_trace_any(const SrScene* s, const SrRay* rays, uint32_t n, uint32_t* occluded, void* stream) {
    return trace_list(const_cast<SrScene*>(s), rays, n, nullptr, occluded, 1, stream);
}

int sr_shade_closest_hit(const SrScene* s, const SrHit* hits, uint32_t n, SrRayPayload* payloads, void* stream) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "sr_shade_closest_hit: scene is null");
    if (!s->built) return fail(SR_ERR_STATE, "sr_shade_closest_hit: scene not built");
    if (n && (!hits || !payloads)) return fail(SR_ERR_INVALID_ARG, "sr_shade_closest_hit: null pointer");
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    int e = srk_launch_shade(s->dev, hits, n, payloads, (hipStream_t)stream);
    if (e != 0) return fail(SR_ERR_HIP, std::string("shade kernel launch: ") + hipGetErrorString((hipError_t)e));
    return SR_OK;
}

int sr_any_hit_ignores(const SrScene* s, const SrHit* hits, uint32_t n, uint32_t* ignored, void* stream) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "sr_any_hit_ignores: scene is null");
    if (!s->built) return fail(SR_ERR_STATE, "sr_any_hit_ignores: scene not built");
    if (n && (!hits || !ignored)) return fail(SR_ERR_INVALID_ARG, "sr_any_hit_ignores: null pointer");
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    int e = srk_launch_any_hit(s->dev, hits, n, ignored, (hipStream_t)stream);
    if (e != 0) return fail(SR_ERR_HIP, std::string("any_hit kernel launch: ") + hipGetErrorString((hipError_t)e));
    return SR_OK;
}

static int run_pass(const SrRtParams* p, int which, void* stream) {
    const char* name = which == 0 ? "raytracing_ris" : "raytracing_final";
    if (!p || !p->scene) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": params or scene is null");
    SrScene* s = const_cast<SrScene*>(p->scene);
    if (!s->built) return fail(SR_ERR_STATE, std::string(name) + ": TLAS not built (call sr_scene_set_instances)");
    if (p->width == 0 || p->height == 0) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": trace_extent not set");
    if (!p->matrices) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": matrices is null");
    const bool need_restir = p->config.enable_restir != 0;
    if (which == 0 || need_restir) {
        if (!p->depth_img || !p->normal_img || !p->diffuse_img || !p->motion_vec_img)
            return fail(SR_ERR_INVALID_ARG, std::string(name) + ": G-buffer image pointer is null");
        if (!p->reservoirs[0] || !p->reservoirs[1] || !p->reservoirs_gi[0] || !p->reservoirs_gi[1])
            return fail(SR_ERR_INVALID_ARG, std::string(name) + ": reservoir buffer pointer is null");
    }
    if (which == 1) {
        if (!p->raw_color) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": raw_color is null");
        if (!p->blue_noise_tex || p->blue_noise_w == 0 || p->blue_noise_h == 0)
            return fail(SR_ERR_INVALID_ARG, std::string(name) + ": blue-noise texture missing");
    }
    if ((uint64_t)p->width * p->height >= (1ull << 31)) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": extent too large");
    if (p->config.max_bounces > SR_MAX_BOUNCES || p->config.virtual_bounces > SR_MAX_BOUNCES)
        return fail(SR_ERR_INVALID_ARG, std::string(name) + ": max_bounces / virtual_bounces above SR_MAX_BOUNCES");
    uint32_t y0 = 0, y1 = p->height;
    if (p->tile_h) {
        if (p->tile_y0 >= p->height) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": tile outside the image");
        y0 = p->tile_y0;
        y1 = std::min(p->height, p->tile_y0 + p->tile_h);
    }
    uint32_t x0 = 0, x1 = p->width;
    if (p->tile_w) {
        if (p->tile_x0 >= p->width) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": tile outside the image");
        x0 = p->tile_x0;
        x1 = std::min(p->width, p->tile_x0 + p->tile_w);
    }
    const uint32_t cols = x1 - x0;
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    srd::PassArgs a;
    memset(&a, 0, sizeof(a));
    a.sc = s->dev;
    a.mats = *p->matrices;
    a.raw_color = p->raw_color; a.depth_img = p->depth_img; a.normal_img = p->normal_img;
    a.diffuse_img = p->diffuse_img; a.motion_vec_img = p->motion_vec_img;
    a.blue_noise_tex = p->blue_noise_tex; a.blue_noise_w = p->blue_noise_w; a.blue_noise_h = p->blue_noise_h;
    a.reservoirs[0] = p->reservoirs[0]; a.reservoirs[1] = p->reservoirs[1];
    a.reservoirs_gi[0] = p->reservoirs_gi[0]; a.reservoirs_gi[1] = p->reservoirs_gi[1];
    // primary-hit hand-off: written by the RIS pass at virtual bounce 0, read by the final pass at bounce 0 — only where a RIS pass ran
    a.primary_payload = (need_restir && p->config.virtual_bounces > 0) ? p->primary_payload : nullptr;
    a.frame_count = p->frame_count;
    a.width = p->width; a.height = p->height;
    a.y0 = y0; a.y1 = y1; a.x0 = x0; a.x1 = x1;
    a.cfg = p->config;
    hipStream_t st = (hipStream_t)stream;
    // tile schedule of this launch geometry: order from the previous launch's costs, costs of this launch for the next
    SrScene::TileSchedule* sched = nullptr;
    if (s->tile_scheduling) {
        for (auto& ts : s->schedules) if (ts.which == which && ts.width == cols && ts.x0 == x0 && ts.y0 == y0 && ts.y1 == y1) sched = &ts;
        if (!sched) {
            if (s->schedules.size() < 8) s->schedules.emplace_back();
            sched = &s->schedules[0];
            for (auto& ts : s->schedules) if (ts.which < 0 || ts.last_use < sched->last_use) sched = &ts;
            if (sched->which >= 0) HIP_TRY(hipDeviceSynchronize());            // recycling an entry a launch on ANY stream may still read
            const size_t bytes = (size_t)srk_pass_tile_count(cols, y1 - y0) * 4;
            if ((rc = sched->cost.reserve(bytes)) != SR_OK || (rc = sched->order.reserve((size_t)srk_pass_order_cap(cols, y1 - y0) * 8 * 4)) != SR_OK) return rc;
            HIP_TRY(hipMemsetAsync(sched->cost.p, 0, bytes, st));
            sched->which = which; sched->width = cols; sched->x0 = x0; sched->y0 = y0; sched->y1 = y1; sched->have_order = false; sched->uses = 0;
        }
        sched->last_use = ++s->schedule_clock;
        a.tile_cost = (uint32_t*)sched->cost.p;
        a.tile_order = sched->have_order ? (const uint32_t*)sched->order.p : nullptr;
    }
    int e;
    PassLabel label(name);
    {
        ScopedTiming tm(s, which == 0 ? kRis : kFinal, st);      // times the pass kernel only
        e = srk_launch_pass(a, which, s->instrumented, s->dev.shade_tex != nullptr, s->two_level ? 1 : 0, s->stack_entries, st);
    }
    if (e != 0) return fail(SR_ERR_HIP, std::string(name) + " launch: " + hipGetErrorString((hipError_t)e));
    // The schedule changes rarely (it follows where the expensive rows and columns are): re-derive it after the first launches
    // of a geometry and then every 64th, not after every launch (the kernel is eight workgroups of mostly serial work,
    // 70-100 us at 1080p, in the stream between two passes: every 16th launch cost 1.3 % of the bench's frame time).
    if (sched && (sched->uses++ < 4 || (sched->uses & 63u) == 0u)) {
        e = srk_launch_tile_order((const uint32_t*)sched->cost.p, (uint32_t*)sched->order.p, cols, y1 - y0, st);
        if (e != 0) return fail(SR_ERR_HIP, std::string(name) + " tile schedule: " + hipGetErrorString((hipError_t)e));
        sched->have_order = true;
    }
    return SR_OK;
}

// Measured cost (shader cycles summed over the tiles of each 8-pixel tile row) of the last launch of pass `which` with
// this geometry: what the tile schedule is derived from; tile-parallel hosts use it to cut strips of equal cost.
int sr_scene_read_tile_row_costs(SrScene* s, int which, uint32_t width, uint32_t y0, uint32_t rows, double* out, uint32_t cap, uint32_t* n_tile_rows) {
    if (!s || !out || !n_tile_rows) return fail(SR_ERR_INVALID_ARG, "sr_scene_read_tile_row_costs: null argument");
    SrScene::TileSchedule* sched = nullptr;
    for (auto& ts : s->schedules) if (ts.which == which && ts.width == width && ts.x0 == 0 && ts.y0 == y0 && ts.y1 == y0 + rows) sched = &ts;
    if (!sched) return fail(SR_ERR_STATE, "sr_scene_read_tile_row_costs: no launch of this pass with this geometry yet");
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    const uint32_t n_tiles = srk_pass_tile_count(width, rows);
    const uint32_t tiles_y = srk_pass_tile_count(1, rows), tiles_x = n_tiles / std::max(tiles_y, 1u);
    *n_tile_rows = tiles_y;
    if (cap < tiles_y) return fail(SR_ERR_INVALID_ARG, "sr_scene_read_tile_row_costs: output too small");
    std::vector<uint32_t> cost(n_tiles);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cost.data(), sched->cost.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < tiles_y; r++) {                    // cost slots are absolute tile indices (thread_pixel, kernels.hip)
        double sum = 0.0;
        for (uint32_t x = 0; x < tiles_x; x++) sum += (double)cost[(size_t)r * tiles_x + x];
        out[r] = sum;
    }
    return SR_OK;
}

// The same data per tile, row-major (ty * tiles_x + tx): tuning diagnostics.
int sr_scene_read_tile_costs(SrScene* s, int which, uint32_t width, uint32_t y0, uint32_t rows, uint32_t* out, uint32_t cap, uint32_t* n_tiles_out) {
    if (!s || !out || !n_tiles_out) return fail(SR_ERR_INVALID_ARG, "sr_scene_read_tile_costs: null argument");
    SrScene::TileSchedule* sched = nullptr;
    for (auto& ts : s->schedules) if (ts.which == which && ts.width == width && ts.x0 == 0 && ts.y0 == y0 && ts.y1 == y0 + rows) sched = &ts;
    if (!sched) return fail(SR_ERR_STATE, "sr_scene_read_tile_costs: no launch of this pass with this geometry yet");
    const uint32_t n_tiles = srk_pass_tile_count(width, rows);
    *n_tiles_out = n_tiles;
    if (cap < n_tiles) return fail(SR_ERR_INVALID_ARG, "sr_scene_read_tile_costs: output too small");
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, sched->cost.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_trace_ris(const SrRtParams* params, void* stream) { return run_pass(params, 0, stream); }
int sr_trace_final(const SrRtParams* params, void* stream) { return run_pass(params, 1, stream); }

static int check_post(const SrPostParams* p, const char* name, bool need_rt, bool need_gbuffer) {
    if (!p) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": params is null");
    if (p->width == 0 || p->height == 0) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": extent not set");
    if ((uint64_t)p->width * p->height >= (1ull << 31)) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": extent too large");
    if (!p->accum[0] || !p->accum[1] || !p->denoise[0] || !p->denoise[1]) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": ping-pong image pointer is null");
    if (need_rt && (!p->raw_color || !p->motion_vec_img)) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": raw_color / motion_vec_img is null");
    if (need_gbuffer && (!p->depth_img || !p->normal_img || !p->diffuse_img)) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": G-buffer image pointer is null");
    if (p->denoise_passes == 0 || p->denoise_passes > 8) return fail(SR_ERR_INVALID_ARG, std::string(name) + ": denoise_passes must be 1..8");
    return SR_OK;
}

int sr_post_temporal(const SrPostParams* p, void* stream) {
    int rc = check_post(p, "temporal_accumulation", true, false);
    if (rc != SR_OK) return rc;
    PassLabel label("temporal_accumulation");
    int e = srk_launch_post_temporal(*p, (hipStream_t)stream);
    if (e != 0) return fail(SR_ERR_HIP, std::string("temporal_accumulation launch: ") + hipGetErrorString((hipError_t)e));
    return SR_OK;
}
int sr_post_denoise(const SrPostParams* p, void* stream) {
    int rc = check_post(p, "denoise", false, true);
    if (rc != SR_OK) return rc;
    PassLabel label("denoise");                                    // the reference's denoise_0 .. denoise_3 (lib.rs:1800-1870), one call here
    int e = srk_launch_post_denoise(*p, (hipStream_t)stream);
    if (e != 0) return fail(SR_ERR_HIP, std::string("denoise launch: ") + hipGetErrorString((hipError_t)e));
    return SR_OK;
}
int sr_post_tonemap(const SrPostParams* p, void* stream) {
    int rc = check_post(p, "postprocess", false, false);
    if (rc != SR_OK) return rc;
    if (!p->output_rgba8) return fail(SR_ERR_INVALID_ARG, "postprocess: output image pointer is null");
    PassLabel label("postprocess");
    int e = srk_launch_post_tonemap(*p, (hipStream_t)stream);
    if (e != 0) return fail(SR_ERR_HIP, std::string("postprocess launch: ") + hipGetErrorString((hipError_t)e));
    return SR_OK;
}

int sr_scene_reset_counters(SrScene* s, void* stream) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "sr_scene_reset_counters: scene is null");
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    HIP_TRY(hipMemsetAsync(s->d_misc.p, 0, kCounterBytes, (hipStream_t)stream));
    return SR_OK;
}

int sr_scene_read_counters(SrScene* s, void* stream, SrRayCounters* out) {
    if (!s || !out) return fail(SR_ERR_INVALID_ARG, "sr_scene_read_counters: null argument");
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    std::vector<unsigned long long> copies(kCounterBytes / 8);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(copies.data(), s->d_misc.p, kCounterBytes, hipMemcpyDeviceToHost));
    unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t slot = 0; slot < srd::kCounterSlots; slot++)
        for (int k = 0; k < 6; k++) v[k] += copies[(size_t)slot * srd::kCounterStride + k];
    out->closest_queries = v[0]; out->any_queries = v[1]; out->boxes_tested = v[2]; out->tris_tested = v[3]; out->reused_primary_hits = v[4]; out->reused_visibility_queries = v[5];
    return SR_OK;
}

int sr_scene_set_instrumented(SrScene* s, int on) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "sr_scene_set_instrumented: scene is null");
    s->instrumented = on ? 1 : 0;
    return SR_OK;
}

int sr_scene_enable_timing(SrScene* s, int enable) {
    if (!s) return fail(SR_ERR_INVALID_ARG, "sr_scene_enable_timing: scene is null");
    if (enable && !s->timing) for (int k = 0; k < kNumKinds; k++) s->events_used[k] = 0;
    s->timing = enable ? 1 : 0;
    return SR_OK;
}

int sr_scene_read_timing(SrScene* s, int kind, double* total_ms, uint32_t* n_launches) {
    if (!s || kind < 0 || kind >= kNumKinds) return fail(SR_ERR_INVALID_ARG, "sr_scene_read_timing: bad argument");
    int rc = bind_device(s);
    if (rc != SR_OK) return rc;
    double total = 0.0;
    for (size_t i = 0; i < s->events_used[kind]; i++) {
        HIP_TRY(hipEventSynchronize(s->events[kind][i].second));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, s->events[kind][i].first, s->events[kind][i].second));
        total += ms;
    }
    if (total_ms) *total_ms = total;
    if (n_launches) *n_launches = (uint32_t)s->events_used[kind];
    s->events_used[kind] = 0;
    return SR_OK;
}

struct SrHostBvhImpl { srh::BvhResult r; uint32_t n; };

int sr_host_bvh_build(const float* v, uint32_t n, SrHostBvh** out) {
    if (!out || (n && !v)) return fail(SR_ERR_INVALID_ARG, "sr_host_bvh_build: null argument");
    if (n >= (1u << 28)) return fail(SR_ERR_UNSUPPORTED, "sr_host_bvh_build: too many triangles");
    std::vector<srh::BuildTri> t(n);
    for (uint32_t i = 0; i < n; i++) {
        memcpy(t[i].v0, v + (size_t)i * 9, 12); memcpy(t[i].e1, v + (size_t)i * 9 + 3, 12); memcpy(t[i].e2, v + (size_t)i * 9 + 6, 12);
        t[i].prim = i; t[i].inst = 0; t[i].gid = i;
    }
    auto* h = new SrHostBvhImpl();
    h->n = n;
    srh::build_bvh(t, (uint32_t)srd::kMaxBinaryDepth, h->r);
    *out = reinterpret_cast<SrHostBvh*>(h);
    return SR_OK;
}
int sr_host_bvh_get(const SrHostBvh* bvh, const uint32_t** nodes, uint32_t* n_nodes, const float** tris, uint32_t* n_triangles, uint32_t* max_depth, uint32_t* max_stack) {
    if (!bvh) return fail(SR_ERR_INVALID_ARG, "sr_host_bvh_get: null handle");
    const auto* h = reinterpret_cast<const SrHostBvhImpl*>(bvh);
    if (nodes) *nodes = h->r.nodes.data();
    if (n_nodes) *n_nodes = h->r.n_nodes;
    if (tris) *tris = h->r.tris.data();
    if (n_triangles) *n_triangles = h->n;
    if (max_depth) *max_depth = h->r.max_depth;
    if (max_stack) *max_stack = h->r.max_stack;
    return SR_OK;
}
int sr_host_bvh_destroy(SrHostBvh* bvh) { delete reinterpret_cast<SrHostBvhImpl*>(bvh); return SR_OK; }

}  // extern "C"
