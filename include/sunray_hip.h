/*
 * sunray_hip.h — C ABI of the MI355X-native ray-tracing hot path
 * (ray_gen -> BVH traversal -> ray/triangle intersect -> closest_hit / any_hit / miss shading).
 *
 * This header is the drop-in boundary. Every entry point replaces one interface
 * of the reference (kalsifer-742/sunray, paths relative to its repo root); the reference-side
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no C++/torch types cross this boundary.
 *   - Every function returns an int status: 0 = SR_OK, negative = SrStatus. The text of the last
 *     error on the calling thread is available from sr_last_error() (mirrors SrError{source,
 *     description}, src/error.rs:6-46). Nothing throws or aborts across the ABI.
 *   - "device pointer" = HIP device memory of the GPU the scene was created on (hipMalloc, a torch
 *     tensor's data_ptr(), ...). The caller owns every frame buffer; the library owns only the
 *     scene (mesh tables, BVH) — the ownership split of src/lib.rs:131-153 / 788-793.
 *   - All launches are asynchronous on the caller's hipStream_t (passed as void*); the caller
 *     synchronises, as the reference's caller does with wait_frame (src/lib.rs:1227-1229).
 *   - One context/scene per GPU; single caller thread per scene (the reference's Renderer is
 *     !Send, src/vulkan_abstraction/core/mod.rs:25).
 *
 * Struct layouts T1..T9 are byte-exact mirrors of shaders/rt_types.slang and of the #[repr(C)]
 * Rust structs that feed them; static asserts at the bottom pin the sizes.
 */
#ifndef SUNRAY_HIP_H
#define SUNRAY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* Status codes — mirror ErrorSource (src/error.rs:11-22)                                       */
/* ------------------------------------------------------------------------------------------ */
typedef enum SrStatus {
    SR_OK = 0,
    SR_ERR_INVALID_ARG = -1, /* ErrorSource::Custom: builder/loader misuse (pass_builder.rs:258-294, lib.rs:880-901) */
    SR_ERR_HIP = -2,         /* ErrorSource::Vulkan counterpart: a HIP runtime call failed            */
    SR_ERR_OOM = -3,         /* ErrorSource::GpuAllocator                                            */
    SR_ERR_STATE = -4,       /* ErrorSource::RenderGraph: call order violated (e.g. trace before build) */
    SR_ERR_UNSUPPORTED = -5  /* input outside the built scope (16-bit images, CMYK JPEG, > 2^28 triangles) */
} SrStatus;

#define SR_NULL_TEXTURE 0xFFFFFFFFu /* rt_types.slang:192, resources/material.rs:49 */

/* ------------------------------------------------------------------------------------------ */
/* T1  VertexAttributes (rt_types.slang:24-36) / Vertex (gltf/vertex.rs:1-35) — 96 B            */
/* ------------------------------------------------------------------------------------------ */
typedef struct SrVertex {
    float position[3];
    float _pad0;
    float normal[3];
    float _pad1;
    float tangent[4];
    float base_color_tex_coord[2];
    float metallic_roughness_tex_coord[2];
    float normal_tex_coord[2];
    float occlusion_tex_coord[2];
    float emissive_tex_coord[2];
    float _pad3[2];
} SrVertex;

/* Material (resources/material.rs:15-44) — 112 B, the inlined material_* half of MeshInfo. */
typedef struct SrMaterial {
    float base_color_value[4];
    float metallic_factor;
    float roughness_factor;
    float _pad_mid[2];
    float emissive_factor[4]; /* rgb + strength */
    uint32_t alpha_mode;      /* always 0 in the reference (material.rs:74) */
    float alpha_cutoff;
    float transmission_factor;
    float ior;
    uint32_t base_color_image, base_color_sampler;
    uint32_t metallic_roughness_image, metallic_roughness_sampler;
    uint32_t normal_image, normal_sampler;
    uint32_t occlusion_image, occlusion_sampler;
    uint32_t emissive_image, emissive_sampler;
    uint32_t _pad_end[2];
} SrMaterial;

/* T2  MeshInfo (rt_types.slang:61-86) / EntityGpuData (resources/entity.rs:8-13) — 128 B */
typedef struct SrMeshInfo {
    uint64_t vertices; /* device address of SrVertex[]  */
    uint64_t indices;  /* device address of uint32_t[]  */
    SrMaterial material;
} SrMeshInfo;

/* T3  EmissiveTriangle (rt_types.slang:89-94, gltf/emissive_triangle.rs:6-13) — 64 B */
typedef struct SrEmissiveTriangle {
    float v0[4], v1[4], v2[4]; /* local space, w unused */
    float emission[4];         /* rgb = factor * strength */
} SrEmissiveTriangle;

/* T4  EmissiveIndirectionEntry (rt_types.slang:96-99) — 8 B */
typedef struct SrEmissiveIndirectionEntry {
    uint32_t blas_tri_index; /* slot in the emissive-triangle table */
    uint32_t entity_id;      /* instance index into the transform table */
} SrEmissiveIndirectionEntry;

/* Sampler parameters the path can observe (image/sampler.rs:14-22,77-94; scene.rs:68-83). The
 * enumerators are VkFilter / VkSamplerAddressMode values. Every texture has ONE mip level and the
 * shaders call SampleLevel(.., 0) with min_lod = max_lod = 0 (rt_utils.slang:121-133), so the
 * MAGNIFICATION filter is the one applied; min_filter is carried for completeness only. — 16 B */
#define SR_FILTER_NEAREST 0u
#define SR_FILTER_LINEAR 1u
#define SR_ADDRESS_REPEAT 0u
#define SR_ADDRESS_MIRRORED_REPEAT 1u
#define SR_ADDRESS_CLAMP_TO_EDGE 2u
typedef struct SrSamplerDesc {
    uint32_t min_filter, mag_filter;
    uint32_t address_mode_u, address_mode_v;
} SrSamplerDesc;

/* T5  EntityTransform (rt_types.slang:101-103) = VkTransformMatrixKHR, row-major 3x4 (utils.rs:67-74) — 48 B */
typedef struct SrTransform {
    float m[12];
} SrTransform;

/* T6  Matrices (rt_types.slang:115-120) — 256 B. Each float[16] holds the four ROWS of the
 * matrix (the host uploads the transpose of nalgebra's column-major storage, lib.rs:1023-1047). */
typedef struct SrMatrices {
    float view_inverse[16];
    float proj_inverse[16];
    float view_proj[16];
    float prev_view_proj[16];
} SrMatrices;

/* T7  Reservoir / ReservoirGI (rt_types.slang:123-143, resources/reservoir.rs:6-54) — 48 B each */
typedef struct SrReservoir {
    float light_pos[3];
    float w_sum;
    float light_normal[3];
    float M;
    uint32_t light_idx;
    float W;
    uint32_t hit_normal_packed;
    float depth;
} SrReservoir;

typedef struct SrReservoirGI {
    float sample_pos[3];
    float w_sum;
    float sample_radiance[3];
    float M;
    uint32_t sample_normal_packed;
    float W;
    uint32_t hit_normal_packed;
    float depth;
} SrReservoirGI;

/* T8  RayPayload (rt_types.slang:9-16) — 32 B */
typedef struct SrRayPayload {
    float emission[3];
    float dist;
    uint32_t albedo_packed;
    uint32_t normal_packed;
    uint32_t material_info;
    uint32_t transmission_ior_packed;
} SrRayPayload;

/* RayDesc as passed to TraceRay (ray_gen_ris.slang:70-75) — 32 B */
typedef struct SrRay {
    float origin[3];
    float tmin;
    float dir[3];
    float tmax;
} SrRay;

/* Committed-hit attributes of one closest-hit query — 16 B.
 * t < 0 (exactly -1.0f) = miss (ray_miss.slang:10-13). tri = global triangle index in
 * instance-major order (instance i's triangles follow instance i-1's); u,v = barycentric weights
 * of vertex 1 and 2 (BuiltInTriangleIntersectionAttributes, closest_hit.slang:15-17). */
typedef struct SrHit {
    float t;
    float u;
    float v;
    uint32_t tri;
} SrHit;

/* Compile-time constants of the reference exposed as knobs (SURVEY.md §5 "Config / flags").
 * sr_trace_config_default() fills the reference values. max_bounces and virtual_bounces above
 * SR_MAX_BOUNCES are refused (SR_ERR_INVALID_ARG): the passes count a pixel's queries in packed fields. */
#define SR_MAX_BOUNCES 8192u
typedef struct SrTraceConfig {
    uint32_t max_bounces;     /* BOUNCES = 10            ray_gen_final.slang:41  */
    uint32_t shadow_bounces;  /* SHADOW_BOUNCES = 5      ray_gen_final.slang:42  */
    uint32_t ris_candidates;  /* RIS_CANDIDATES = 16     ray_gen_ris.slang:187   */
    uint32_t virtual_bounces; /* 20                      ray_gen_ris.slang:69    */
    uint32_t enable_restir;   /* 1 = reference behaviour. 0 = ReSTIR block disabled: the final pass
                                 starts with restir_evaluated = true, so every rough bounce takes the
                                 plain NEE branch (ray_gen_final.slang:328-382); the RIS pass is not
                                 needed (BASELINE.json configs 2, 3, 5).                           */
    uint32_t flags;           /* SR_TRACE_FLAG_* */
    uint32_t count_y0;        /* with count_rows != 0: only pixels of rows [count_y0, count_y0 + count_rows) add  */
    uint32_t count_rows;      /* their rays to the scene's counters — a strip traced together with its halo rows  */
    uint32_t count_x0;        /* same for columns: with count_cols != 0 only pixels of columns                    */
    uint32_t count_cols;      /* [count_x0, count_x0 + count_cols) count (column strips)              */
} SrTraceConfig;

/* Do not add this launch's rays to the scene's ray counters: used for the halo rows a GPU re-traces
 * for its neighbours' spatial reuse in tile-parallel rendering, so that counted rays are
 * exactly the rays of the equivalent single-GPU frame. */
#define SR_TRACE_FLAG_UNCOUNTED 1u
/* Trace every query the reference issues, also where its answer is known from an identical query of the same pixel in the same
 * pass (SrRayCounters.reused_visibility_queries stays 0). Same results either way; for accounting and A/B. */
#define SR_TRACE_FLAG_TRACE_EVERY_QUERY 2u

typedef struct SrScene SrScene; /* opaque: mesh tables + instance tables + BVH ("TLAS") on one GPU */

/* T9  RaytracingPC (rt_types.slang:151-190) / RaytracingHeapPushConstant
 * (pipelines/ray_tracing_pipeline.rs:28-60), with every heap handle replaced by a pointer and the
 * dispatch extent (pass_builder.rs:311-320 trace_extent) made explicit. The same params struct is
 * handed to both passes, as the reference pushes identical bytes to both (lib.rs:1626-1652).
 *
 * Image storage follows the reference's formats (lib.rs:1492-1516), one element per pixel, row 0 =
 * top of the image:
 *   raw_color      float[4]  fp32 RGBA (reference: B10G11R11_UFLOAT; kept fp32 here)
 *   depth_img      uint16    R16_SFLOAT bits
 *   normal_img     uint32    R8G8B8A8_SNORM, normal.xyz + roughness in .a
 *   diffuse_img    uint32    B10G11R11_UFLOAT_PACK32
 *   motion_vec_img uint32    R16G16_SFLOAT
 * meshes_info / emissive_triangles / emissive_indirection / entity_transforms / tlas of the
 * reference struct are owned by `scene` (sr_scene_* below). */
typedef struct SrRtParams {
    const SrScene* scene;
    float* raw_color;                /* device, 4*W*H floats  */
    uint16_t* depth_img;             /* device, W*H           */
    uint32_t* normal_img;            /* device, W*H           */
    uint32_t* diffuse_img;           /* device, W*H           */
    uint32_t* motion_vec_img;        /* device, W*H           */
    const SrMatrices* matrices;      /* HOST pointer, copied into the launch like push data */
    const uint8_t* blue_noise_tex;   /* device, RGBA8 texels, blue_noise_w*blue_noise_h*4 bytes (lib.rs:281-309) */
    uint32_t blue_noise_w, blue_noise_h;
    SrReservoir* reservoirs[2];      /* device, W*H each; ping-pong by frame_count & 1 (rt_utils.slang:241-242) */
    SrReservoirGI* reservoirs_gi[2]; /* device, W*H each */
    /* Primary-hit hand-off (optional, device, W*H records, caller-owned like the G-buffer images). Both reference passes
     * start with the SAME query: ray_gen_final.slang:80 at bounce 0 is ray_gen_ris.slang:75 at virtual bounce 0 (same
     * pixel, same matrices, same TLAS), and TraceRay is a pure function of its arguments. With a buffer here sr_trace_ris
     * stores that query's 32-byte RayPayload (T8, closest_hit / miss applied) per pixel and sr_trace_final, enqueued
     * after it for the same frame, reads it back instead of traversing and shading again — same bits, one traversal
     * fewer per pixel (counted in SrRayCounters.reused_primary_hits, not in closest_queries). NULL: sr_trace_final traces
     * the query itself. Ignored when config.enable_restir == 0 (no RIS pass ran) or config.virtual_bounces == 0. Part of
     * the per-frame image set: double-buffer it with the G-buffer when two frames are in flight. */
    SrRayPayload* primary_payload;
    uint32_t frame_count;            /* relative_frame_count (lib.rs:1355,1386) */
    uint32_t use_srgb;               /* carried, unused by the shaders */
    uint32_t width, height;          /* trace_extent[0], [1]; also the image size */
    /* Sub-rectangle of the image this launch covers, for tile-parallel multi-GPU.
     * Pixels are keyed by their GLOBAL coordinates (RNG, camera), so any tiling gives identical
     * values. tile_h == 0 means all rows, tile_w == 0 all columns. Buffers are always full-size W*H. */
    uint32_t tile_y0, tile_h;
    uint32_t tile_x0, tile_w;
    SrTraceConfig config;
} SrRtParams;

/* Ray counters accumulated by the kernels (rays are counted, not estimated). */
typedef struct SrRayCounters {
    uint64_t closest_queries; /* TraceRay(RAY_FLAG_NONE) traversed         */
    uint64_t any_queries;     /* TraceRay(ACCEPT_FIRST_HIT|SKIP_CLOSEST) traversed */
    uint64_t boxes_tested;    /* child boxes tested (32 B each), only in the instrumented build */
    uint64_t tris_tested;     /* triangle records tested (48 B each), only in the instrumented build */
    uint64_t reused_primary_hits; /* TraceRay(RAY_FLAG_NONE) calls of the reference's final pass answered from
                                     SrRtParams.primary_payload without a traversal: the reference issues
                                     closest_queries + reused_primary_hits closest-hit queries */
    uint64_t reused_visibility_queries; /* TraceRay(ACCEPT_FIRST_HIT) calls of the reference's final pass answered without a traversal:
                                     ray_gen_final.slang:304-316 when the combined GI reservoir's sample is a spatial neighbour's, whose
                                     identical ray (:276-286) was found unoccluded a moment before. The reference issues
                                     any_queries + reused_visibility_queries existence queries */
} SrRayCounters;

/* ------------------------------------------------------------------------------------------ */
/* Errors                                                                                       */
/* ------------------------------------------------------------------------------------------ */
const char* sr_last_error(void); /* SrError::description of the last failure on this thread */
int sr_version(void);

/* ------------------------------------------------------------------------------------------ */
/* Host-side data preparation — pure CPU, no GPU needed                     */
/* ------------------------------------------------------------------------------------------ */

/* H1/H2: Camera::as_matrices (src/camera.rs:33-63) followed by the transposed upload of
 * Renderer::render (src/lib.rs:1017-1048). prev_view_proj16 = the previous frame's view_proj rows
 * as returned in out->view_proj by the previous call, or NULL on the first frame (zero matrix,
 * lib.rs:410). */
int sr_camera_matrices(const float position[3], const float target[3], float fov_y_degrees,
                       uint32_t width, uint32_t height, const float* prev_view_proj16,
                       SrMatrices* out);

/* H6: Material::new for a runtime mesh (resources/material.rs:52-92 with the NULL-texture
 * resolver of lib.rs:937-943): factors copied, alpha_mode = 0, alpha_cutoff = 0, all textures NULL. */
int sr_material_new(const float base_color[4], float metallic, float roughness,
                    const float emissive_factor[3], float emissive_strength, float transmission,
                    float ior, SrMaterial* out);

/* H5: emissive-triangle derivation of Renderer::load_mesh (src/lib.rs:901-925).
 * Writes at most cap entries, returns the total count through *out_count. */
int sr_emissive_triangles_from_mesh(const SrVertex* vertices, uint32_t n_vertices,
                                    const uint32_t* indices, uint32_t n_indices,
                                    const SrMaterial* material, SrEmissiveTriangle* out,
                                    uint32_t cap, uint32_t* out_count);

void sr_trace_config_default(SrTraceConfig* out);

/* ------------------------------------------------------------------------------------------ */
/* Scene = ResourceManager + BLAS/TLAS (resource_manager.rs, acceleration_structure/)           */
/* ------------------------------------------------------------------------------------------ */

/* Renderer::new's resource half (lib.rs:212-446, resource_manager.rs:84-155) on HIP device `device`. */
int sr_scene_create(int device, SrScene** out);
int sr_scene_destroy(SrScene* scene);

/* Renderer::load_mesh (src/lib.rs:873-954) + ResourceManager::add_blas (resource_manager.rs:417-447):
 * validates like the reference (non-empty, index count % 3 == 0, indices in range, key unused),
 * uploads vertices/indices, assigns the next mesh-info slot (the instance custom index,
 * closest_hit.slang:18-19) and appends the mesh's emissive triangles to the emissive table.
 * vertices/indices are HOST pointers. *out_slot may be NULL. */
int sr_scene_add_mesh(SrScene* scene, uint64_t key, const SrVertex* vertices, uint32_t n_vertices,
                      const uint32_t* indices, uint32_t n_indices, const SrMaterial* material,
                      uint32_t* out_slot);

/* ------------------------------------------------------------------------------------------ */
/* Acceleration-structure maintenance                                            */
/* ------------------------------------------------------------------------------------------ */
/* BuildType / OpType / AsState (acceleration_structure/mod.rs:22-148): the rebuild-vs-update heuristic shared by
 * the reference's BLAS and TLAS — update in place at most 8 times, then a fast rebuild; after 16 quiet frames one
 * quality rebuild and back to Optimal. Pure logic, exposed for the host that drives the scene. */
#define SR_BUILD_RAPIDLY_CHANGING 0u
#define SR_BUILD_SOMETIMES_CHANGES 1u
#define SR_BUILD_STATIC 2u
#define SR_OP_NONE 0u
#define SR_OP_SLOW_BUILD 1u
#define SR_OP_FAST_BUILD 2u
#define SR_OP_UPDATE 3u
typedef struct SrAsState {
    uint32_t changing; /* 0 = AsState::Optimal, 1 = AsState::Changing(Dynamic) */
    uint32_t frames_without_changes;
    uint32_t number_of_updates_since_last_rebuild;
    uint32_t _pad;
} SrAsState;
void sr_as_state_initial(uint32_t build_type, SrAsState* out);
uint32_t sr_as_state_next_op(const SrAsState* state, int inputs_changed);
void sr_as_state_mark_built(SrAsState* state, uint32_t completed_op);
/* The scene's own state (it is built as SometimesChanges, like the reference's TLAS, resource_manager.rs:119-126)
 * and the operation sr_scene_set_instances / sr_scene_end_frame last performed (SR_OP_*). */
int sr_scene_as_state(const SrScene* scene, SrAsState* state, uint32_t* last_op);
/* A frame whose instance list did not change (Tlas::mark_built(None) + the settle rebuild, tlas.rs:155-191 with
 * inputs_changed = false): advances the quiet-frame counter and, when AsState asks for it, performs the quality
 * rebuild. The Renderer facade calls it for every frame that skips sr_scene_set_instances. */
int sr_scene_end_frame(SrScene* scene);
/* Test / bench hook: the next sr_scene_set_instances performs `op` (SR_OP_SLOW_BUILD = host binned-SAH build,
 * SR_OP_FAST_BUILD = device LBVH build, SR_OP_UPDATE = in-place update if the layout allows) instead of the
 * heuristic's choice. The heuristic state is still advanced with the op performed. */
int sr_scene_force_next_op(SrScene* scene, uint32_t op);
/* Form of the acceleration structure. The reference instances BLASes through a TLAS (tlas.rs:155-191, resource_manager.rs:236-251).
 * SR_INSTANCING_FLAT copies every instance's triangles into ONE world-space tree (no ray transform in the walk, work stealing inside the
 * wave; memory and update cost grow with instances x triangles); SR_INSTANCING_TWO_LEVEL keeps one tree per mesh in object space plus a
 * top-level tree over the instances (a changed instance list costs a top-level rebuild whatever the meshes hold; the walk transforms
 * the ray per instance). Both answer every query with the same bits: in the two-level walk the object-space ray only steers box
 * culling, triangles are tested in world space. SR_INSTANCING_AUTO (default): two-level where the flattened copy would exceed 2^24
 * triangles and at least four times the meshes' own, or 2^28 in any case. Takes effect at the next sr_scene_set_instances.
 * SR_INSTANCING = flat | two_level | auto in the environment sets the initial mode. */
#define SR_INSTANCING_AUTO 0u
#define SR_INSTANCING_FLAT 1u
#define SR_INSTANCING_TWO_LEVEL 2u
int sr_scene_set_instancing(SrScene* scene, uint32_t mode);
int sr_scene_instancing(const SrScene* scene, uint32_t* mode, uint32_t* two_level_now);
/* Node layout of the quantised wide BVH this build uses (csrc/bvh_layout.h): children per node, dwords per node, first
 * plane dword, first child dword. */
int sr_bvh_layout(uint32_t* width, uint32_t* node_dwords, uint32_t* plane_offset, uint32_t* child_offset);
/* Debug read-back of the device tree: n_nodes x node_dwords dwords, n_triangles x 12 floats (either may be NULL). */
int sr_scene_read_bvh(const SrScene* scene, uint32_t* nodes_out, float* tris_out);

/* ResourceManager::add_blas (resource_manager.rs:417-447): sr_scene_add_mesh with the local-space emissive
 * triangles supplied by the caller instead of derived from the material — the glTF path marks a primitive
 * emissive under a different rule than load_mesh (gltf/mod.rs:272 vs lib.rs:907). */
int sr_scene_add_blas(SrScene* scene, uint64_t key, const SrVertex* vertices, uint32_t n_vertices,
                      const uint32_t* indices, uint32_t n_indices, const SrMaterial* material,
                      const SrEmissiveTriangle* emissive, uint32_t n_emissive, uint32_t* out_slot);
/* ResourceManager::remove (resource_manager.rs:459-487): frees the mesh-info slot and the emissive slots of
 * `key` (later loads reuse them LIFO, as the reference's arenas do). Unknown key: no-op. Waits for the device. */
int sr_scene_remove(SrScene* scene, uint64_t key);

/* Image::new_from_data (image/mod.rs:82-111): `channels` = 1..4 bytes per texel; fewer than 4 are
 * widened to R8G8B8A8_UNORM with the missing channels 0x00 (utils.rs:27-43), no sRGB decode. Host
 * pointer, w*h*channels bytes. Returns the image slot materials refer to (Material::*_image). */
int sr_scene_add_image(SrScene* scene, const uint8_t* data, uint32_t width, uint32_t height, uint32_t channels,
                       uint32_t* out_image_slot);
/* Frees an image added with sr_scene_add_image (ResourceManager::remove drops a key's images with its BLAS,
 * resource_manager.rs:459-472); the slot is handed out again by a later sr_scene_add_image. SR_ERR_STATE while a registered
 * mesh's material still names the slot. Waits for the device. */
int sr_scene_remove_image(SrScene* scene, uint32_t image_slot);

/* Sampler::new (image/sampler.rs:44-67). Returns the sampler slot (Material::*_sampler). */
int sr_scene_add_sampler(SrScene* scene, const SrSamplerDesc* desc, uint32_t* out_sampler_slot);
/* ResourceManager::frame_instance_data (resource_manager.rs:216-267) + the dummy-entry padding of
 * Renderer::render (lib.rs:1058-1081) + the TLAS build it queues (resource_manager.rs:346-363):
 * keys[i] is instanced counts[i] times with the next counts[i] row-major 3x4 transforms taken from
 * `transforms`. Instance order, transform table and emissive-indirection order follow the
 * reference loop exactly. Builds the world-space BVH over every instance's triangles (the
 * replacement for vkCmdBuildAccelerationStructuresKHR, accel.rs:134-138) and uploads it.
 * Unknown key -> SR_ERR_INVALID_ARG (resource_manager.rs:227-231). */
int sr_scene_set_instances(SrScene* scene, const uint64_t* keys, const uint32_t* counts,
                           uint32_t n_keys, const SrTransform* transforms);

/* Introspection of the tables frame_instance_data produced (host copies; pointers valid until the
 * next sr_scene_set_instances / destroy). num_lights is the emissive_indirection length the
 * shaders read through GetDimensions (ray_gen_ris.slang:185-186), i.e. >= 1 because of the dummy. */
int sr_scene_get_tables(const SrScene* scene, const SrTransform** transforms, uint32_t* n_instances,
                        const SrEmissiveIndirectionEntry** indirection, uint32_t* num_lights,
                        const SrEmissiveTriangle** emissive_triangles, uint32_t* n_emissive,
                        const SrMeshInfo** meshes_info, uint32_t* n_meshes);

/* BVH statistics for roofline accounting. */
typedef struct SrBvhStats {
    uint64_t n_triangles;
    uint64_t n_nodes;     /* 4-wide nodes with quantised child boxes, 64 B each */
    uint64_t node_bytes;
    uint64_t tri_bytes;   /* 48 B per triangle record */
    uint32_t max_depth;   /* of the 4-wide tree */
    float sah_cost;
    uint32_t max_stack;   /* worst-case traversal stack entries (sizes the kernels' LDS stack) */
    uint32_t _pad;
    double build_ms;
} SrBvhStats;
int sr_scene_bvh_stats(const SrScene* scene, SrBvhStats* out);

/* global triangle index (SrHit.tri) -> (instance index, primitive index) */
int sr_scene_resolve_triangle(const SrScene* scene, uint32_t tri, uint32_t* instance,
                              uint32_t* primitive);

/* Host-only access to the BVH builder (no GPU needed): builds the same BVH sr_scene_set_instances
 * would build over n world-space triangles given as 9 floats each (v0, e1, e2), for structural
 * checks on machines without a device. nodes: 16 dwords per 4-wide quantised node, tris: 12 floats
 * per triangle in leaf order (layout: sunray_amd/csrc/traverse.h). max_stack: worst-case traversal
 * stack entries. */
typedef struct SrHostBvh SrHostBvh;
int sr_host_bvh_build(const float* v0_e1_e2, uint32_t n_triangles, SrHostBvh** out);
int sr_host_bvh_get(const SrHostBvh* bvh, const uint32_t** nodes, uint32_t* n_nodes, const float** tris,
                    uint32_t* n_triangles, uint32_t* max_depth, uint32_t* max_stack);
int sr_host_bvh_destroy(SrHostBvh* bvh);

/* ------------------------------------------------------------------------------------------ */
/* The hot path                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* TraceRay(tlas, RAY_FLAG_NONE, 0xFF, 0,0,0, ray, prd) minus the closest-hit shader
 * (ray_gen_ris.slang:75,332; ray_gen_final.slang:80): nearest triangle with tmin < t < tmax,
 * two-sided (resource_manager.rs:249), all geometry opaque (blas.rs:276).
 * rays/hits are device pointers to n elements. */
int sr_trace_closest(const SrScene* scene, const SrRay* rays, uint32_t n, SrHit* hits, void* stream);

/* TraceRay(tlas, ACCEPT_FIRST_HIT_AND_END_SEARCH | SKIP_CLOSEST_HIT_SHADER, ...)
 * (ray_gen_ris.slang:285-300,372-385; ray_gen_final.slang:203-216,274-287,304-319,361-373):
 * occluded[i] = 1 if any triangle has tmin < t < tmax, else 0 (the miss shader ran). */
int sr_trace_any(const SrScene* scene, const SrRay* rays, uint32_t n, uint32_t* occluded, void* stream);

/* closest_hit (closest_hit.slang:12-91) / ray_miss (ray_miss.slang:10-13) applied to hit records:
 * payloads[i] = the 32-byte RayPayload the reference shader would produce. Device pointers. */
int sr_shade_closest_hit(const SrScene* scene, const SrHit* hits, uint32_t n, SrRayPayload* payloads,
                         void* stream);

/* any_hit (any_hit.slang:11-43) applied to hit records: ignored[i] = 1 where the shader would call IgnoreHit() (the
 * mesh's material has alpha_mode != 0 and the base-colour alpha sampled at the hit's interpolated uv is below
 * alpha_cutoff), else 0. The traversal never runs it: every BLAS geometry carries the OPAQUE flag (blas.rs:276) and
 * Material::new forces alpha_mode = 0 (material.rs:74) — this hook exists so that the shader's lines have a device
 * counterpart with a parity test. Misses (t < 0) give 0. Device pointers. */
int sr_any_hit_ignores(const SrScene* scene, const SrHit* hits, uint32_t n, uint32_t* ignored, void* stream);

/* The "raytracing_ris" pass (lib.rs:1662-1705): one ray_gen_ris invocation per pixel
 * (ray_gen_ris.slang:12-440): G-buffer + ReSTIR-DI reservoir + ReSTIR-GI initial reservoir. */
int sr_trace_ris(const SrRtParams* params, void* stream);

/* The "raytracing_final" pass (lib.rs:1714-1755): one ray_gen_final invocation per pixel
 * (ray_gen_final.slang:11-436): writes raw_color. Must be enqueued after sr_trace_ris on the same
 * stream when config.enable_restir != 0 (the reservoir hand-off edge, lib.rs:1688-1690). */
int sr_trace_final(const SrRtParams* params, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Post-RT compute chain: what turns raw_color into the presented RGBA8 image    */
/* ------------------------------------------------------------------------------------------ */

/* The compute passes Renderer::build_unified_graph appends after the two ray-tracing passes
 * (src/lib.rs:1576-1615). Images keep the reference's formats (lib.rs:452-461,1492-1516): the
 * accumulation and denoise ping-pong images are B10G11R11_UFLOAT_PACK32, the output R8G8B8A8_UNORM.
 * raw_color is this library's fp32 RGBA radiance; it is rounded to B10G11R11 when read, which is
 * where the reference quantises it (its raw_color image has that format). All device pointers. */
typedef struct SrPostParams {
    const float* raw_color;          /* 4*W*H floats, written by sr_trace_final                     */
    const uint32_t* motion_vec_img;  /* R16G16_SFLOAT, written by sr_trace_ris                      */
    const uint16_t* depth_img;       /* R16_SFLOAT                                                   */
    const uint32_t* normal_img;      /* R8G8B8A8_SNORM (roughness in .a)                             */
    const uint32_t* diffuse_img;     /* B10G11R11                                                    */
    uint32_t* accum[2];              /* temporal ping-pong: target = frame_count % 2 (lib.rs:1360-1361) */
    uint32_t* denoise[2];            /* a-trous ping-pong a / b (lib.rs:1817-1826)                    */
    uint32_t* output_rgba8;          /* W*H, R8G8B8A8_UNORM                                          */
    uint32_t frame_count;            /* relative_frame_count                                         */
    uint32_t width, height;
    float exposure;                  /* EXPOSURE = 1.0 (lib.rs:44)                                   */
    uint32_t denoise_passes;         /* DENOISE_PASSES = 4 (lib.rs:42); step width 1 << pass          */
    uint32_t _pad;
} SrPostParams;

/* "temporal_accumulation" (shaders/temporal_accumulation.slang:60-132, lib.rs:1761-1798):
 * 3x3 luma-gated neighbourhood clamp of the bilinearly reprojected history, lerp 0.14. */
int sr_post_temporal(const SrPostParams* params, void* stream);
/* "denoise_0..N-1" (shaders/denoise.slang:29-116, lib.rs:1800-1870): 5x5 a-trous B-spline with
 * depth / normal / albedo / luma edge stopping on albedo-demodulated illumination. Pass 0 reads
 * accum[frame_count % 2]; the result of the last pass is in denoise[(denoise_passes - 1) % 2]. */
int sr_post_denoise(const SrPostParams* params, void* stream);
/* "postprocess" (shaders/postprocess.slang:22-42, lib.rs:1872-1906): NaN/Inf scrub, exposure,
 * ACES (Narkowicz), gamma 1/2.2, RGBA8 store. Reads denoise[(denoise_passes - 1) % 2]. */
int sr_post_tonemap(const SrPostParams* params, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Renderer facade: the reference's Renderer<K> method surface for the built path */
/* ------------------------------------------------------------------------------------------ */
typedef struct SrRenderer SrRenderer;

/* Renderer::new((w, h), RGBA8_UNORM) (src/lib.rs:212-446): scene + G-buffer + temporal resources
 * (reservoir, accumulation and denoise ping-pongs, lib.rs:320-331) + the noise texture. */
int sr_renderer_create(int device, uint32_t width, uint32_t height, SrRenderer** out);
int sr_renderer_destroy(SrRenderer* renderer);
/* Renderer::resize (lib.rs:586-639): waits for the device, recreates every image at the new extent,
 * relative_frame_count = 0. Same extent: no-op (lib.rs:598-600). */
int sr_renderer_resize(SrRenderer* renderer, uint32_t width, uint32_t height);

/* Frame / resize callbacks (src/lib.rs:537-554). A start-of-frame callback runs once, on the caller's thread, at the
 * start of the next sr_renderer_render (before any per-frame work); an end-of-frame callback runs once the next
 * rendered frame has COMPLETED ON THE GPU — checked at the start of later sr_renderer_render calls, as the reference
 * drains them there (lib.rs:1004-1010) — the deferred-deallocation hook; a resize callback is persistent and runs on
 * every sr_renderer_resize call with the new extent (lib.rs:586-594). Callbacks run in registration order. */
typedef void (*SrFrameCallback)(void* user);
typedef void (*SrResizeCallback)(void* user, uint32_t width, uint32_t height);
int sr_renderer_add_start_of_frame_callback(SrRenderer* renderer, SrFrameCallback callback, void* user);
int sr_renderer_add_end_of_frame_callback(SrRenderer* renderer, SrFrameCallback callback, void* user);
int sr_renderer_add_resize_callback(SrRenderer* renderer, SrResizeCallback callback, void* user);
/* Renderer::load_mesh (lib.rs:873-954); host pointers. */
int sr_renderer_load_mesh(SrRenderer* renderer, uint64_t key, const SrVertex* vertices, uint32_t n_vertices,
                          const uint32_t* indices, uint32_t n_indices, const SrMaterial* material);
/* Knobs for the ray-tracing passes (defaults = the reference's constants). */
int sr_renderer_set_config(SrRenderer* renderer, const SrTraceConfig* config);
/* Renderer::render(camera, instances) (lib.rs:984-1232): enqueues one whole frame (acceleration-structure update /
 * rebuild if the instance list changed, raytracing_ris, raytracing_final, temporal_accumulation, denoise_0..3,
 * postprocess) on the renderer's OWN two streams, ordered after whatever is already enqueued on `stream`, and returns
 * its frame number. Two frames may be in flight (MAX_FRAMES_IN_FLIGHT, lib.rs:71): per-frame images are double-buffered
 * and the RIS pass of frame f+1 overlaps the final pass and post chain of frame f when the caller submits f+1 before
 * waiting for f. Frames complete in order; results equal back-to-back execution. */
int sr_renderer_render(SrRenderer* renderer, const float cam_pos[3], const float cam_target[3], float fov_y_degrees,
                       const uint64_t* keys, const uint32_t* counts, uint32_t n_keys, const SrTransform* transforms,
                       void* stream, uint64_t* out_frame);
/* Renderer::wait_frame (lib.rs:1234-1238). */
int sr_renderer_wait_frame(SrRenderer* renderer, uint64_t frame);
/* Renderer::render_to_host_memory (lib.rs:1908-1934): 16 x (render + wait_frame), then the RGBA8 image
 * (width*height*4 bytes, no padding) copied to out_rgba8 (host). */
int sr_renderer_render_to_host_memory(SrRenderer* renderer, const float cam_pos[3], const float cam_target[3],
                                      float fov_y_degrees, const uint64_t* keys, const uint32_t* counts, uint32_t n_keys,
                                      const SrTransform* transforms, uint8_t* out_rgba8);
/* ------------------------------------------------------------------------------------------ */
/* glTF ingest                                                                  */
/* ------------------------------------------------------------------------------------------ */
/* Host-side parse: Gltf::new + create_default_scene (gltf/mod.rs:57-373) and the CPU side of
 * Scene::load_into_gpu (scene.rs:52-176). `.glb` or `.gltf` (+ external / data: buffers); images: 8-bit PNG and JPEG
 * (baseline, extended sequential, progressive; what `gltf::import` hands on as R8 / RG8 / RGB8 / RGBA8; JPEG texels are
 * decoder-defined within a few units: parity unpinned, csrc/jpeg_decode.cpp). 16-bit PNG (the reference stops there too,
 * image/mod.rs:98-104), CMYK / arithmetic-coded JPEG, camera/light nodes -> SR_ERR_UNSUPPORTED. Sparse accessors are resolved
 * (zeros or the bufferView, with `count` elements substituted), as the gltf crate's readers do. No device needed. */
typedef struct SrGltf SrGltf;
/* The loader's image decoder on its own: extent + channels, and the pixels when `pixels` != NULL (cap >= w*h*channels). */
int sr_decode_image(const uint8_t* data, size_t n, uint32_t* width, uint32_t* height, uint32_t* channels, uint8_t* pixels, size_t cap);
/* image::load_from_memory(bytes).to_rgba8() — lib.rs:281-283 (the embedded blue-noise texture, a 16-bit greyscale PNG) and the png
 * example's host side: any PNG / JPEG above plus 16-bit PNG, widened to RGBA8 (grey -> r = g = b, no alpha -> 255, 16-bit sample v ->
 * (v + 128) / 257 as image-rs 0.25 narrows it). `pixels` may be NULL to query the extent; cap >= w * h * 4. */
int sr_decode_image_rgba8(const uint8_t* data, size_t n, uint32_t* width, uint32_t* height, uint8_t* pixels, size_t cap);
int sr_gltf_open(const char* path, SrGltf** out);
int sr_gltf_close(SrGltf* gltf);
int sr_gltf_counts(const SrGltf* gltf, uint32_t* n_blases, uint32_t* n_instances, uint32_t* n_images,
                   uint32_t* n_samplers, uint32_t* n_textures);
/* One unique BLAS (scene.rs:16-24). `material` is UNRESOLVED like the reference's gltf::Material: its *_image
 * fields hold glTF texture indices (or SR_NULL_TEXTURE), its *_sampler fields SR_NULL_TEXTURE. */
int sr_gltf_blas(const SrGltf* gltf, uint32_t i, const SrVertex** vertices, uint32_t* n_vertices,
                 const uint32_t** indices, uint32_t* n_indices, SrMaterial* material,
                 const SrEmissiveTriangle** emissive, uint32_t* n_emissive);
/* i-th (blas index, world transform) of LoadedScene::instances (scene.rs:31-33), node-traversal order. */
int sr_gltf_instance(const SrGltf* gltf, uint32_t i, uint32_t* blas_index, SrTransform* transform);
int sr_gltf_image(const SrGltf* gltf, uint32_t i, const uint8_t** pixels, uint32_t* width, uint32_t* height,
                  uint32_t* channels);
int sr_gltf_sampler(const SrGltf* gltf, uint32_t i, SrSamplerDesc* out);
/* textures[i] = {sampler: Option<usize> (-1 = none -> the default LINEAR / CLAMP_TO_EDGE sampler,
 * resource_manager.rs:128-136,393), source image}. */
int sr_gltf_texture(const SrGltf* gltf, uint32_t i, int32_t* sampler, uint32_t* source);

/* What Renderer::load_gltf / load_scene return (lib.rs:779-846): the asset group and the scene's instances
 * grouped per BLAS key in BLAS order. Keys are ResourceKey{group, index} packed as group << 32 | index
 * (lib.rs:54-58); BLASes take indices 0..n-1, images the following ones (resource_manager.rs:400-410). */
typedef struct SrLoadedScene SrLoadedScene;
int sr_renderer_load_gltf(SrRenderer* renderer, const char* path, SrLoadedScene** out);
int sr_renderer_load_scene(SrRenderer* renderer, const SrGltf* gltf, SrLoadedScene** out);
int sr_loaded_scene_get(const SrLoadedScene* loaded, uint64_t* group, const uint64_t** keys, const uint32_t** counts,
                        uint32_t* n_keys, const SrTransform** transforms, uint32_t* n_transforms);
int sr_loaded_scene_destroy(SrLoadedScene* loaded);
/* Renderer::unload_scene (lib.rs:849-857) / unload_mesh (lib.rs:965-973). */
int sr_renderer_unload_scene(SrRenderer* renderer, uint64_t group);
int sr_renderer_unload_mesh(SrRenderer* renderer, uint64_t key);

/* Harness access: inner scene (counters, stats), device pointers of the RGBA8 output and the fp32 radiance OF THE LAST
 * SUBMITTED FRAME (valid after sr_renderer_wait_frame of that frame), and relative_frame_count. Any out pointer may be NULL. */
int sr_renderer_get(SrRenderer* renderer, SrScene** scene, const uint32_t** output_rgba8_device,
                    const float** raw_color_device, uint32_t* relative_frame_count);
/* Stand-in for the reference's embedded 128x128 blue-noise PNG (lib.rs:281-309; an input asset, not
 * copied): hashed white noise, RGBA8, grey in rgb, alpha 255. Host pointer, w*h*4 bytes. */
int sr_default_noise_texture(uint32_t w, uint32_t h, uint32_t seed, uint8_t* out_rgba8);
/* Replaces the noise texture of the passes (lib.rs:281-309 builds it once, in Renderer::new, from the PNG embedded in the crate: a host
 * that owns that asset decodes it with sr_decode_image_rgba8 and hands it over here). RGBA8 texels in host memory, w * h * 4 bytes.
 * Waits for the frames in flight; the temporal history is kept. */
int sr_renderer_set_blue_noise(SrRenderer* r, const uint8_t* rgba8, uint32_t w, uint32_t h);

/* Measured cost of the last launch of pass `which` (0 = raytracing_ris, 1 = raytracing_final) with this launch
 * geometry, summed per tile row (8 pixel rows), in shader cycles: the data the library's own tile schedule uses.
 * A tile-parallel host cuts its strips of equal cost from it (sunray_amd/distributed.py). out: cap doubles. */
int sr_scene_read_tile_row_costs(SrScene* scene, int which, uint32_t width, uint32_t y0, uint32_t rows, double* out,
                                 uint32_t cap, uint32_t* n_tile_rows);

/* The same measurement per tile (8x8 pixels), row-major (tile row * tiles per row + tile column): tuning diagnostics. */
int sr_scene_read_tile_costs(SrScene* scene, int which, uint32_t width, uint32_t y0, uint32_t rows, uint32_t* out,
                             uint32_t cap, uint32_t* n_tiles);

/* ------------------------------------------------------------------------------------------ */
/* Tile-parallel rendering across the GPUs of a node                               */
/* ------------------------------------------------------------------------------------------ */
/* The reference renders on one device (src/lib.rs:1166). A multi-GPU host creates one scene / renderer context per GPU
 * (scene replicated), cuts the frame into `world` contiguous strips and has every context trace its strip into full-size
 * buffers; pixels are keyed by global coordinates (tile_* of SrRtParams), so the strips of N contexts are the single-GPU
 * frame bit for bit. ReSTIR's spatial reuse reads a 30-pixel neighbourhood (ray_gen_final.slang:160-188,228-247): the RIS
 * pass of a strip also covers a 30-pixel halo on either side (recomputed, rays counted for the strip only). Temporal reuse
 * under camera motion needs the reservoir bands sr_history_exchange_plan lists, from the ranks that own them. The only
 * data-path collective is the gather of the radiance strips, which stays with the caller (RCCL: ncclAllGather over the
 * per-rank raw_color strips). Everything here is deterministic host arithmetic: every rank derives the
 * same partition and plans from the same inputs. */
#define SR_AXIS_COLS 0u      /* column strips (default: image cost varies mostly with the row, columns hand every GPU the same mix) */
#define SR_AXIS_ROWS 1u
#define SR_SPATIAL_HALO 30u  /* SPATIAL_RADIUS (ray_gen_final.slang:161) >= GI_SPATIAL_RADIUS (:229) */
typedef struct SrPartition SrPartition;
/* `bounds`: world + 1 increasing cut positions along the axis from 0 to its length (e.g. from sr_balanced_bounds), or NULL for
 * equal strips. A partition must stay the same for a whole frame sequence: a rank owns the temporal history of its strip + halo. */
int sr_partition_create(uint32_t width, uint32_t height, uint32_t world, uint32_t axis, const uint32_t* bounds, SrPartition** out);
int sr_partition_destroy(SrPartition* partition);
int sr_partition_get(const SrPartition* partition, uint32_t* width, uint32_t* height, uint32_t* world, uint32_t* axis, const uint32_t** bounds);
/* (start, size) along the axis of rank's strip grown by `grow` positions on both sides, clipped to the image. */
int sr_partition_span(const SrPartition* partition, uint32_t rank, uint32_t grow, uint32_t* start, uint32_t* size);
/* Cuts a per-position cost profile into `world` strips of (nearly) equal summed cost, each at least min_size positions and at
 * most max_share * length / world (the gather pads strips to the largest): bounds_out receives world + 1 cuts. */
int sr_balanced_bounds(const double* cost, uint32_t length, uint32_t world, uint32_t min_size, double max_share, uint32_t* bounds_out);
/* Per-column / per-row cost profile (length entries) from per-tile costs (sr_scene_read_tile_costs, tiles_y x tiles_x, 8x8 pixels). */
int sr_axis_cost_from_tiles(const double* tile_costs, uint32_t tiles_x, uint32_t tiles_y, uint32_t axis, uint32_t length, double* out);
/* One point-to-point transfer of the temporal-history exchange: positions [start, start + size) along the axis, full extent
 * across it, of BOTH current reservoir buffers (reservoirs[frame_count & 1], reservoirs_gi[frame_count & 1]), src -> dst. */
typedef struct SrStripTransfer {
    uint32_t src, dst, start, size;
} SrStripTransfer;
/* The transfers needed after every RIS pass when temporal reprojection can move a pixel by up to motion_halo positions along
 * the axis between frames (0 = static camera: none). Writes at most cap entries, returns the total through *count. */
int sr_history_exchange_plan(const SrPartition* partition, uint32_t motion_halo, SrStripTransfer* out, uint32_t cap, uint32_t* count);
/* Launch rectangles of rank's share of a frame: what sr_strip_trace_ris / sr_strip_trace_final put into SrRtParams.tile_* and
 * SrTraceConfig.count_* (count_window == 0: no counting window, every traced pixel counts). */
typedef struct SrStripRects {
    uint32_t ris_y0, ris_h, ris_x0, ris_w;          /* raytracing_ris: strip + spatial halo (world > 1) */
    uint32_t final_y0, final_h, final_x0, final_w;  /* raytracing_final: the strip */
    uint32_t count_y0, count_rows, count_x0, count_cols;
    uint32_t count_window;
    uint32_t empty;                                 /* the rank's strip has no pixels: nothing to launch */
} SrStripRects;
int sr_strip_rects(const SrPartition* partition, uint32_t rank, SrStripRects* out);
/* sr_trace_ris / sr_trace_final of rank's share: `params` describes the whole frame on this rank's device (full-size buffers);
 * its tile_* and config.count_* fields are replaced. Same stream rules as sr_trace_*; between the two calls the host performs
 * the transfers of sr_history_exchange_plan (if any) on the same stream. */
int sr_strip_trace_ris(const SrRtParams* params, const SrPartition* partition, uint32_t rank, void* stream);
int sr_strip_trace_final(const SrRtParams* params, const SrPartition* partition, uint32_t rank, void* stream);

/* Ray counters since the last reset (device-side atomics, read back synchronously). */
int sr_scene_reset_counters(SrScene* scene, void* stream);
int sr_scene_read_counters(SrScene* scene, void* stream, SrRayCounters* out);

/* Instrumented kernels (count child boxes / triangle records tested, SURVEY §8d B_ray accounting).
 * Off by default: the counting costs registers and time. */
int sr_scene_set_instrumented(SrScene* scene, int on);

/* Per-launch device timing: when enabled every sr_trace_* launch is bracketed by a HIP event pair
 * recorded on the launch's own stream. sr_scene_read_timing waits for the recorded launches of one
 * kind, returns their summed elapsed time and count, and clears that kind's list.
 * kind: 0 = sr_trace_ris, 1 = sr_trace_final, 2 = sr_trace_closest, 3 = sr_trace_any. */
int sr_scene_enable_timing(SrScene* scene, int enable);
int sr_scene_read_timing(SrScene* scene, int kind, double* total_ms, uint32_t* n_launches);

#ifdef __cplusplus
}
#endif

#if defined(__cplusplus)
static_assert(sizeof(SrVertex) == 96, "T1");
static_assert(sizeof(SrMaterial) == 112, "Material");
static_assert(sizeof(SrMeshInfo) == 128, "T2");
static_assert(sizeof(SrEmissiveTriangle) == 64, "T3");
static_assert(sizeof(SrEmissiveIndirectionEntry) == 8, "T4");
static_assert(sizeof(SrTransform) == 48, "T5");
static_assert(sizeof(SrMatrices) == 256, "T6");
static_assert(sizeof(SrReservoir) == 48 && sizeof(SrReservoirGI) == 48, "T7");
static_assert(sizeof(SrRayPayload) == 32, "T8");
static_assert(sizeof(SrRay) == 32 && sizeof(SrHit) == 16, "ray/hit");
static_assert(sizeof(SrTraceConfig) == 40 && sizeof(SrRtParams) == 184, "T9");
static_assert(sizeof(SrPostParams) == 104, "post params");
static_assert(sizeof(SrStripTransfer) == 16 && sizeof(SrStripRects) == 56, "strip plans");
#endif

#endif /* SUNRAY_HIP_H */
