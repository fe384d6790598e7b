"""Loader of the in-tree HIP library (sunray_amd/libsunray_hip.so). There is NO fallback path: if
the library is missing or fails to load, every product entry point raises."""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SUNRAY_HIP_LIB") or os.path.join(_HERE, "libsunray_hip.so")   # override: kernel-tuning variants only
_lib = None

# every symbol include/sunray_hip.h declares
SYMBOLS = [
    "sr_last_error", "sr_version", "sr_camera_matrices", "sr_material_new", "sr_emissive_triangles_from_mesh",
    "sr_trace_config_default", "sr_scene_create", "sr_scene_destroy", "sr_as_state_initial", "sr_as_state_next_op", "sr_as_state_mark_built", "sr_scene_as_state", "sr_scene_end_frame",
    "sr_bvh_layout", "sr_scene_read_bvh", "sr_scene_force_next_op", "sr_scene_add_mesh", "sr_scene_add_blas", "sr_scene_remove", "sr_scene_add_image", "sr_scene_remove_image", "sr_scene_add_sampler", "sr_scene_set_instances",
    "sr_scene_get_tables", "sr_scene_bvh_stats", "sr_scene_resolve_triangle", "sr_host_bvh_build", "sr_host_bvh_get",
    "sr_host_bvh_destroy", "sr_trace_closest", "sr_trace_any", "sr_shade_closest_hit", "sr_any_hit_ignores", "sr_trace_ris", "sr_trace_final", "sr_post_temporal", "sr_post_denoise", "sr_post_tonemap", "sr_renderer_create", "sr_renderer_destroy",
    "sr_renderer_resize", "sr_renderer_add_start_of_frame_callback", "sr_renderer_add_end_of_frame_callback", "sr_renderer_add_resize_callback", "sr_renderer_load_mesh", "sr_renderer_set_config", "sr_renderer_render", "sr_renderer_wait_frame",
    "sr_renderer_render_to_host_memory", "sr_renderer_get", "sr_decode_image", "sr_decode_image_rgba8", "sr_renderer_set_blue_noise", "sr_gltf_open", "sr_gltf_close", "sr_gltf_counts", "sr_gltf_blas", "sr_gltf_instance", "sr_gltf_image",
    "sr_gltf_sampler", "sr_gltf_texture", "sr_renderer_load_gltf", "sr_renderer_load_scene", "sr_loaded_scene_get", "sr_loaded_scene_destroy",
    "sr_renderer_unload_scene", "sr_renderer_unload_mesh", "sr_default_noise_texture",
    "sr_scene_read_tile_row_costs", "sr_scene_read_tile_costs", "sr_scene_reset_counters", "sr_scene_read_counters", "sr_scene_set_instrumented", "sr_scene_enable_timing",
    "sr_scene_read_timing",
    "sr_partition_create", "sr_partition_destroy", "sr_partition_get", "sr_partition_span", "sr_balanced_bounds", "sr_axis_cost_from_tiles",
    "sr_history_exchange_plan", "sr_strip_rects", "sr_strip_trace_ris", "sr_strip_trace_final", "sr_scene_set_instancing", "sr_scene_instancing",
]


class SunrayError(RuntimeError):
    """SrError (src/error.rs:6-46): status code + description."""

    def __init__(self, code, description):
        super().__init__("sunray_hip error %d: %s" % (code, description))
        self.code = code
        self.description = description


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "sunray_amd: %s is missing — build it with `python -m sunray_amd.build` (or "
                "__graft_entry__.build()). There is no CPU fallback for the product path." % LIB_PATH)
        # The harnesses hand torch tensors to the library, so both must talk to ONE HIP runtime. torch ships its own copy of
        # libamdhip64; loading ours first would bring in /opt/rocm's as a second runtime (and the later one finds "no
        # ROCm-capable device"). Importing torch first makes the dynamic linker reuse its runtime for this library.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.sr_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SunrayError(rc, lib().sr_last_error().decode("utf-8", "replace"))
