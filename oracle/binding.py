"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module. Parity status: UNPINNED (the reference holds no golden vectors
for this path; the oracle is pinned by the hand-derived KATs in tests/golden/ only).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from sunray_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with its Makefile (gcc only; no GPU, no reference sources needed)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "_build/liboracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_exp.restype = C.c_float
        L.orc_exp.argtypes = [C.c_float]
        L.orc_log.restype = C.c_float
        L.orc_log.argtypes = [C.c_float]
        L.orc_pow.restype = C.c_float
        L.orc_pow.argtypes = [C.c_float, C.c_float]
        L.orc_smith_v_ggx.restype = C.c_float
        L.orc_smith_g1_ggx.restype = C.c_float
        L.orc_smoothstep.restype = C.c_float
        for name in ("orc_smith_v_ggx", "orc_smoothstep"):
            getattr(L, name).argtypes = [C.c_float] * 3
        L.orc_smith_g1_ggx.argtypes = [C.c_float] * 2
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f16_to_f32.argtypes = [C.c_uint32]
        L.orc_pack_normal.argtypes = [C.c_float] * 3
        L.orc_pack_unorm_4x8.argtypes = [C.c_float] * 4
        L.orc_pack_rgba8_snorm.argtypes = [C.c_float] * 4
        L.orc_pack_half_2x16.argtypes = [C.c_float] * 2
        L.orc_pack_snorm_2x16.argtypes = [C.c_float] * 2
        L.orc_pack_b10g11r11.argtypes = [C.c_float] * 3
        for name in ("orc_pack_normal", "orc_pack_unorm_4x8", "orc_pack_rgba8_snorm", "orc_pack_half_2x16",
                     "orc_pack_snorm_2x16", "orc_pack_b10g11r11", "orc_f32_to_f16", "orc_pcg_hash",
                     "orc_init_rng", "orc_rnd_stream"):
            getattr(L, name).restype = C.c_uint32
        _lib = L
    return _lib


def usable_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def set_threads(n):
    lib().orc_set_threads(C.c_int(n))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def unpack_normal(p):
    out = (C.c_float * 3)()
    lib().orc_unpack_normal(C.c_uint32(p), out)
    return np.array(list(out), dtype=np.float32)


class OracleScene:
    """Oracle counterpart of sunray_amd.Scene: same methods, numpy (host) buffers."""

    def __init__(self):
        self._h = C.c_void_p(lib().orc_scene_create())
        self._keep = []

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_mesh(self, key, vertices, indices, material):
        v = np.ascontiguousarray(vertices, dtype=abi.VERTEX)
        i = np.ascontiguousarray(indices, dtype=np.uint32)
        m = np.ascontiguousarray(material, dtype=abi.MATERIAL)
        slot = lib().orc_scene_add_mesh(self._h, C.c_uint64(key), _p(v), C.c_uint32(len(v)), _p(i), C.c_uint32(len(i)), _p(m))
        if slot < 0:
            raise ValueError("oracle add_mesh rejected the mesh (same checks as Renderer::load_mesh, lib.rs:880-899)")
        return slot

    def add_blas(self, key, vertices, indices, material, emissive):
        v = np.ascontiguousarray(vertices, dtype=abi.VERTEX)
        i = np.ascontiguousarray(indices, dtype=np.uint32)
        m = np.ascontiguousarray(material, dtype=abi.MATERIAL)
        e = np.ascontiguousarray(emissive, dtype=abi.EMISSIVE_TRIANGLE)
        slot = lib().orc_scene_add_blas(self._h, C.c_uint64(key), _p(v), C.c_uint32(len(v)), _p(i), C.c_uint32(len(i)), _p(m),
                                        _p(e) if len(e) else None, C.c_uint32(len(e)))
        if slot < 0:
            raise ValueError("oracle add_blas rejected the mesh")
        return slot

    def remove(self, key):
        lib().orc_scene_remove(self._h, C.c_uint64(key))

    def add_image(self, pixels):
        a = np.ascontiguousarray(pixels, dtype=np.uint8)
        ch = 1 if a.ndim == 2 else a.shape[2]
        slot = lib().orc_scene_add_image(self._h, _p(a), C.c_uint32(a.shape[1]), C.c_uint32(a.shape[0]), C.c_uint32(ch))
        if slot < 0:
            raise ValueError("oracle add_image rejected the image")
        return slot

    def add_sampler(self, min_filter, mag_filter, address_mode_u, address_mode_v):
        d = abi.SrSamplerDesc(min_filter, mag_filter, address_mode_u, address_mode_v)
        slot = lib().orc_scene_add_sampler(self._h, C.byref(d))
        if slot < 0:
            raise ValueError("oracle add_sampler rejected the sampler")
        return slot

    def sample_texture(self, image, sampler, u, v, fallback=(0.0, 0.0, 0.0, 0.0)):
        out = (C.c_float * 4)()
        lib().orc_sample_texture(self._h, C.c_uint32(image), C.c_uint32(sampler), C.c_float(u), C.c_float(v), (C.c_float * 4)(*fallback), out)
        return np.array(list(out), dtype=np.float32)

    def set_instances(self, instances):
        keys = np.array([k for k, _ in instances], dtype=np.uint64)
        counts = np.array([len(t) for _, t in instances], dtype=np.uint32)
        xf = np.array([np.asarray(t, dtype=np.float32).reshape(12) for _, ts in instances for t in ts], dtype=np.float32).reshape(-1, 12)
        if len(xf) == 0:
            xf = np.zeros((1, 12), dtype=np.float32)
        rc = lib().orc_scene_set_instances(self._h, _p(keys), _p(counts), C.c_uint32(len(keys)), _p(np.ascontiguousarray(xf)))
        if rc != 0:
            raise ValueError("frame_instance_data: instance references a BLAS key that was never loaded")

    def load(self, desc):
        for img in desc.images:
            self.add_image(img)
        for smp in desc.samplers:
            self.add_sampler(*smp)
        for m in desc.meshes:
            self.add_mesh(m.key, m.vertices, m.indices, m.material)
        self.set_instances(desc.instances)
        return self

    def set_brute_force(self, on):
        lib().orc_scene_set_brute_force(self._h, C.c_int(1 if on else 0))

    def tables(self):
        tp, ip, ep = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nt, nl, ne, ntri = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().orc_scene_get_tables(self._h, C.byref(tp), C.byref(nt), C.byref(ip), C.byref(nl), C.byref(ep), C.byref(ne), C.byref(ntri))
        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dtype=dt)
            buf = (C.c_char * (n * dt.itemsize)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dt).copy()
        return {"transforms": arr(tp, nt.value, abi.TRANSFORM), "indirection": arr(ip, nl.value, abi.EMISSIVE_INDIRECTION),
                "emissive_triangles": arr(ep, ne.value, abi.EMISSIVE_TRIANGLE), "num_lights": nl.value,
                "n_triangles": ntri.value}

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY)
        hits = np.zeros(len(rays), dtype=abi.HIT)
        lib().orc_trace_closest(self._h, _p(rays), C.c_uint32(len(rays)), _p(hits))
        return hits

    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY)
        occ = np.zeros(len(rays), dtype=np.uint32)
        lib().orc_trace_any(self._h, _p(rays), C.c_uint32(len(rays)), _p(occ))
        return occ

    def shade_closest_hit(self, hits):
        hits = np.ascontiguousarray(hits, dtype=abi.HIT)
        out = np.zeros(len(hits), dtype=abi.RAY_PAYLOAD)
        lib().orc_shade_closest_hit(self._h, _p(hits), C.c_uint32(len(hits)), _p(out))
        return out

    def any_hit_ignores(self, hits):
        out = np.zeros(len(hits), dtype=np.uint32)
        lib().orc_any_hit(self._h, _p(hits), C.c_uint32(len(hits)), _p(out))
        return out

    def _params(self, frame, matrices, frame_count, config, tile=None):
        p = abi.SrRtParams()
        p.scene = None
        p.raw_color = frame.raw_color.ctypes.data
        p.depth_img = frame.depth.ctypes.data
        p.normal_img = frame.normal.ctypes.data
        p.diffuse_img = frame.diffuse.ctypes.data
        p.motion_vec_img = frame.motion.ctypes.data
        self._m = matrices
        p.matrices = C.pointer(matrices)
        p.blue_noise_tex = frame.blue_noise.ctypes.data
        p.blue_noise_h, p.blue_noise_w = frame.blue_noise.shape[:2]
        p.reservoirs[0], p.reservoirs[1] = frame.reservoirs[0].ctypes.data, frame.reservoirs[1].ctypes.data
        p.reservoirs_gi[0], p.reservoirs_gi[1] = frame.reservoirs_gi[0].ctypes.data, frame.reservoirs_gi[1].ctypes.data
        p.primary_payload = frame.primary.ctypes.data     # written by trace_ris for the tests, never read by trace_final
        p.frame_count = frame_count
        p.use_srgb = 0
        p.width, p.height = frame.width, frame.height
        if tile:                      # (y0, h) rows, or (y0, h, x0, w) rows x columns; h == 0 / w == 0: all of them
            p.tile_y0, p.tile_h = tile[0], tile[1]
            if len(tile) == 4:
                p.tile_x0, p.tile_w = tile[2], tile[3]
        p.config = config
        return p

    def trace_ris(self, frame, matrices, frame_count, config=None, tile=None):
        p = self._params(frame, matrices, frame_count, config or abi.SrTraceConfig.reference(), tile)
        lib().orc_trace_ris(self._h, C.byref(p))

    def trace_final(self, frame, matrices, frame_count, config=None, tile=None):
        p = self._params(frame, matrices, frame_count, config or abi.SrTraceConfig.reference(), tile)
        lib().orc_trace_final(self._h, C.byref(p))

    def reset_counters(self):
        lib().orc_reset_counters(self._h)

    def counters(self):
        c = abi.SrRayCounters()
        lib().orc_read_counters(self._h, C.byref(c))
        return c


class HostFrame:
    """Frame buffers in host memory with the layouts of SrRtParams (numpy)."""

    def __init__(self, width, height, blue_noise):
        self.width, self.height = width, height
        n = width * height
        self.raw_color = np.zeros((n, 4), dtype=np.float32)
        self.depth = np.zeros(n, dtype=np.uint16)
        self.normal = np.zeros(n, dtype=np.uint32)
        self.diffuse = np.zeros(n, dtype=np.uint32)
        self.motion = np.zeros(n, dtype=np.uint32)
        self.reservoirs = [np.zeros(n, dtype=abi.RESERVOIR), np.zeros(n, dtype=abi.RESERVOIR)]
        self.reservoirs_gi = [np.zeros(n, dtype=abi.RESERVOIR_GI), np.zeros(n, dtype=abi.RESERVOIR_GI)]
        self.blue_noise = np.ascontiguousarray(blue_noise, dtype=np.uint8)
        self.accum = [np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)]
        self.denoise = [np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)]
        self.output = np.zeros(n, dtype=np.uint32)
        self.primary = np.zeros(n, dtype=abi.RAY_PAYLOAD)     # camera-ray payloads as trace_ris sees them (test aid)


def post_chain(frame, frame_count, exposure=1.0, denoise_passes=4, stages=("temporal", "denoise", "tonemap")):
    """The oracle's temporal_accumulation -> a-trous denoise -> postprocess on a HostFrame."""
    p = abi.post_params(frame, frame_count, lambda a: a.ctypes.data, exposure, denoise_passes)
    if "temporal" in stages:
        lib().orc_post_temporal(C.byref(p))
    if "denoise" in stages:
        lib().orc_post_denoise(C.byref(p))
    if "tonemap" in stages:
        lib().orc_post_tonemap(C.byref(p))


def camera_matrices(pos, target, fov_y, width, height, prev_view_proj=None):
    m = abi.SrMatrices()
    prev = None
    if prev_view_proj is not None:
        prev = (C.c_float * 16)(*[float(x) for x in prev_view_proj])
    lib().orc_camera_matrices(_f3(pos), _f3(target), C.c_float(fov_y), C.c_uint32(width), C.c_uint32(height), prev, C.byref(m))
    return m
