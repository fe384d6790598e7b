"""Python mirror of include/sunray_hip.h: numpy dtypes and ctypes structures of the C ABI.

Pure data definitions (no library is loaded here). Layouts follow the reference's GPU structs
(shaders/rt_types.slang) byte for byte; see the header for the file:line of each.
"""
import ctypes as C

import numpy as np

NULL_TEXTURE = 0xFFFFFFFF
TRACE_FLAG_UNCOUNTED = 1
TRACE_FLAG_TRACE_EVERY_QUERY = 2
MAX_BOUNCES = 8192          # SR_MAX_BOUNCES

# T1 VertexAttributes (rt_types.slang:24-36) — 96 B
VERTEX = np.dtype([
    ("position", "<f4", 3), ("_pad0", "<f4"), ("normal", "<f4", 3), ("_pad1", "<f4"),
    ("tangent", "<f4", 4), ("base_color_tex_coord", "<f4", 2), ("metallic_roughness_tex_coord", "<f4", 2),
    ("normal_tex_coord", "<f4", 2), ("occlusion_tex_coord", "<f4", 2), ("emissive_tex_coord", "<f4", 2),
    ("_pad3", "<f4", 2)])
# Material (resources/material.rs:15-44) — 112 B
MATERIAL = np.dtype([
    ("base_color_value", "<f4", 4), ("metallic_factor", "<f4"), ("roughness_factor", "<f4"),
    ("_pad_mid", "<f4", 2), ("emissive_factor", "<f4", 4), ("alpha_mode", "<u4"), ("alpha_cutoff", "<f4"),
    ("transmission_factor", "<f4"), ("ior", "<f4"),
    ("base_color_image", "<u4"), ("base_color_sampler", "<u4"),
    ("metallic_roughness_image", "<u4"), ("metallic_roughness_sampler", "<u4"),
    ("normal_image", "<u4"), ("normal_sampler", "<u4"),
    ("occlusion_image", "<u4"), ("occlusion_sampler", "<u4"),
    ("emissive_image", "<u4"), ("emissive_sampler", "<u4"), ("_pad_end", "<u4", 2)])
MESH_INFO = np.dtype([("vertices", "<u8"), ("indices", "<u8"), ("material", MATERIAL)])  # T2, 128 B
EMISSIVE_TRIANGLE = np.dtype([("v0", "<f4", 4), ("v1", "<f4", 4), ("v2", "<f4", 4), ("emission", "<f4", 4)])  # T3
EMISSIVE_INDIRECTION = np.dtype([("blas_tri_index", "<u4"), ("entity_id", "<u4")])  # T4
TRANSFORM = np.dtype([("m", "<f4", 12)])  # T5 row-major 3x4
RESERVOIR = np.dtype([("light_pos", "<f4", 3), ("w_sum", "<f4"), ("light_normal", "<f4", 3), ("M", "<f4"),
                      ("light_idx", "<u4"), ("W", "<f4"), ("hit_normal_packed", "<u4"), ("depth", "<f4")])  # T7
RESERVOIR_GI = np.dtype([("sample_pos", "<f4", 3), ("w_sum", "<f4"), ("sample_radiance", "<f4", 3), ("M", "<f4"),
                         ("sample_normal_packed", "<u4"), ("W", "<f4"), ("hit_normal_packed", "<u4"), ("depth", "<f4")])
RAY_PAYLOAD = np.dtype([("emission", "<f4", 3), ("dist", "<f4"), ("albedo_packed", "<u4"), ("normal_packed", "<u4"),
                        ("material_info", "<u4"), ("transmission_ior_packed", "<u4")])  # T8
RAY = np.dtype([("origin", "<f4", 3), ("tmin", "<f4"), ("dir", "<f4", 3), ("tmax", "<f4")])
HIT = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("tri", "<u4")])

assert VERTEX.itemsize == 96 and MATERIAL.itemsize == 112 and MESH_INFO.itemsize == 128
assert EMISSIVE_TRIANGLE.itemsize == 64 and EMISSIVE_INDIRECTION.itemsize == 8 and TRANSFORM.itemsize == 48
assert RESERVOIR.itemsize == 48 and RESERVOIR_GI.itemsize == 48 and RAY_PAYLOAD.itemsize == 32
assert RAY.itemsize == 32 and HIT.itemsize == 16


class SrMatrices(C.Structure):  # T6, 256 B; every float[16] = 4 rows
    _fields_ = [("view_inverse", C.c_float * 16), ("proj_inverse", C.c_float * 16),
                ("view_proj", C.c_float * 16), ("prev_view_proj", C.c_float * 16)]


class SrTraceConfig(C.Structure):
    _fields_ = [("max_bounces", C.c_uint32), ("shadow_bounces", C.c_uint32), ("ris_candidates", C.c_uint32),
                ("virtual_bounces", C.c_uint32), ("enable_restir", C.c_uint32), ("flags", C.c_uint32),
                ("count_y0", C.c_uint32), ("count_rows", C.c_uint32), ("count_x0", C.c_uint32), ("count_cols", C.c_uint32)]

    @staticmethod
    def reference():
        """The reference's compile-time constants (ray_gen_final.slang:40-42, ray_gen_ris.slang:69,187)."""
        return SrTraceConfig(10, 5, 16, 20, 1, 0, 0, 0, 0, 0)


class SrRtParams(C.Structure):  # T9
    _fields_ = [("scene", C.c_void_p), ("raw_color", C.c_void_p), ("depth_img", C.c_void_p),
                ("normal_img", C.c_void_p), ("diffuse_img", C.c_void_p), ("motion_vec_img", C.c_void_p),
                ("matrices", C.POINTER(SrMatrices)), ("blue_noise_tex", C.c_void_p),
                ("blue_noise_w", C.c_uint32), ("blue_noise_h", C.c_uint32),
                ("reservoirs", C.c_void_p * 2), ("reservoirs_gi", C.c_void_p * 2), ("primary_payload", C.c_void_p),
                ("frame_count", C.c_uint32), ("use_srgb", C.c_uint32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("tile_y0", C.c_uint32), ("tile_h", C.c_uint32),
                ("tile_x0", C.c_uint32), ("tile_w", C.c_uint32), ("config", SrTraceConfig)]


class SrPostParams(C.Structure):  # post-RT compute chain (temporal accumulation, a-trous denoise, tonemap)
    _fields_ = [("raw_color", C.c_void_p), ("motion_vec_img", C.c_void_p), ("depth_img", C.c_void_p),
                ("normal_img", C.c_void_p), ("diffuse_img", C.c_void_p), ("accum", C.c_void_p * 2),
                ("denoise", C.c_void_p * 2), ("output_rgba8", C.c_void_p), ("frame_count", C.c_uint32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("exposure", C.c_float),
                ("denoise_passes", C.c_uint32), ("_pad", C.c_uint32)]


def post_params(frame, frame_count, ptr, exposure=1.0, denoise_passes=4):
    """SrPostParams over a HostFrame / DeviceFrame; `ptr(buffer)` gives the address of a buffer.
    Defaults are the reference's EXPOSURE and DENOISE_PASSES (src/lib.rs:42,44)."""
    p = SrPostParams()
    p.raw_color, p.motion_vec_img = ptr(frame.raw_color), ptr(frame.motion)
    p.depth_img, p.normal_img, p.diffuse_img = ptr(frame.depth), ptr(frame.normal), ptr(frame.diffuse)
    p.accum[0], p.accum[1] = ptr(frame.accum[0]), ptr(frame.accum[1])
    p.denoise[0], p.denoise[1] = ptr(frame.denoise[0]), ptr(frame.denoise[1])
    p.output_rgba8 = ptr(frame.output)
    p.frame_count, p.width, p.height = frame_count, frame.width, frame.height
    p.exposure, p.denoise_passes = exposure, denoise_passes
    return p


class SrRayCounters(C.Structure):
    _fields_ = [("closest_queries", C.c_uint64), ("any_queries", C.c_uint64),
                ("boxes_tested", C.c_uint64), ("tris_tested", C.c_uint64), ("reused_primary_hits", C.c_uint64), ("reused_visibility_queries", C.c_uint64)]


class SrBvhStats(C.Structure):
    _fields_ = [("n_triangles", C.c_uint64), ("n_nodes", C.c_uint64), ("node_bytes", C.c_uint64),
                ("tri_bytes", C.c_uint64), ("max_depth", C.c_uint32), ("sah_cost", C.c_float),
                ("max_stack", C.c_uint32), ("_pad", C.c_uint32), ("build_ms", C.c_double)]


def material(base_color=(0.8, 0.8, 0.8, 1.0), metallic=0.0, roughness=0.5, emissive_factor=(0.0, 0.0, 0.0),
             emissive_strength=0.0, transmission=0.0, ior=1.5, textures=None):
    """Material::new (resources/material.rs:52-92). Runtime meshes have all textures NULL (lib.rs:937-943);
    `textures` = {"base_color"|"metallic_roughness"|"normal"|"occlusion"|"emissive": (image slot, sampler slot)}
    is what the glTF path's `resolve` closure fills in."""
    m = np.zeros((), dtype=MATERIAL)
    m["base_color_value"] = base_color
    m["metallic_factor"] = metallic
    m["roughness_factor"] = roughness
    m["emissive_factor"] = tuple(emissive_factor) + (emissive_strength,)
    m["transmission_factor"] = transmission
    m["ior"] = ior
    for k in ("base_color", "metallic_roughness", "normal", "occlusion", "emissive"):
        m[k + "_image"] = NULL_TEXTURE
        m[k + "_sampler"] = NULL_TEXTURE
    for k, (img, smp) in (textures or {}).items():
        m[k + "_image"] = img
        m[k + "_sampler"] = smp
    return m


FILTER_NEAREST, FILTER_LINEAR = 0, 1
ADDRESS_REPEAT, ADDRESS_MIRRORED_REPEAT, ADDRESS_CLAMP_TO_EDGE = 0, 1, 2


class SrSamplerDesc(C.Structure):
    _fields_ = [("min_filter", C.c_uint32), ("mag_filter", C.c_uint32), ("address_mode_u", C.c_uint32), ("address_mode_v", C.c_uint32)]


IDENTITY_TRANSFORM = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float32)


class SrAsState(C.Structure):
    _fields_ = [("changing", C.c_uint32), ("frames_without_changes", C.c_uint32), ("number_of_updates_since_last_rebuild", C.c_uint32),
                ("_pad", C.c_uint32)]


BUILD_RAPIDLY_CHANGING, BUILD_SOMETIMES_CHANGES, BUILD_STATIC = 0, 1, 2
OP_NONE, OP_SLOW_BUILD, OP_FAST_BUILD, OP_UPDATE = 0, 1, 2, 3


class SrStripTransfer(C.Structure):  # one point-to-point transfer of the temporal-history exchange
    _fields_ = [("src", C.c_uint32), ("dst", C.c_uint32), ("start", C.c_uint32), ("size", C.c_uint32)]


class SrStripRects(C.Structure):  # launch rectangles + counting window of one rank's share of a frame
    _fields_ = [("ris_y0", C.c_uint32), ("ris_h", C.c_uint32), ("ris_x0", C.c_uint32), ("ris_w", C.c_uint32),
                ("final_y0", C.c_uint32), ("final_h", C.c_uint32), ("final_x0", C.c_uint32), ("final_w", C.c_uint32),
                ("count_y0", C.c_uint32), ("count_rows", C.c_uint32), ("count_x0", C.c_uint32), ("count_cols", C.c_uint32),
                ("count_window", C.c_uint32), ("empty", C.c_uint32)]


AXIS_COLS, AXIS_ROWS, SPATIAL_HALO = 0, 1, 30
