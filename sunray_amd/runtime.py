"""Thin Python harness over the C ABI (include/sunray_hip.h) for tests and bench.py.

PyTorch is used only as plumbing: device memory (torch tensors as frame buffers), the current HIP
stream, and torch.distributed for the multi-GPU gather. Every compute call goes through
libsunray_hip.so; there is no CPU or torch fallback.
"""
import ctypes as C

import numpy as np

from . import abi
from ._lib import SunrayError, check, lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def camera_matrices(pos, target, fov_y, width, height, prev_view_proj=None):
    """Camera::as_matrices + transposed upload (camera.rs:33-63, lib.rs:1017-1048). Host only."""
    m = abi.SrMatrices()
    prev = None
    if prev_view_proj is not None:
        prev = (C.c_float * 16)(*[float(x) for x in prev_view_proj])
    check(lib().sr_camera_matrices(_f3(pos), _f3(target), C.c_float(fov_y), C.c_uint32(width), C.c_uint32(height), prev, C.byref(m)))
    return m


def material_new(base_color, metallic, roughness, emissive_factor, emissive_strength, transmission, ior):
    m = np.zeros((), dtype=abi.MATERIAL)
    check(lib().sr_material_new((C.c_float * 4)(*base_color), C.c_float(metallic), C.c_float(roughness), _f3(emissive_factor),
                                C.c_float(emissive_strength), C.c_float(transmission), C.c_float(ior), _p(m)))
    return m


def decode_image(data):
    """The glTF loader's image decoder on its own (8-bit PNG, JPEG) -> (h, w, channels) uint8."""
    buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    w, h, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    check(lib().sr_decode_image(buf, C.c_size_t(len(data)), C.byref(w), C.byref(h), C.byref(c), None, C.c_size_t(0)))
    out = np.zeros((h.value, w.value, c.value), dtype=np.uint8)
    check(lib().sr_decode_image(buf, C.c_size_t(len(data)), C.byref(w), C.byref(h), C.byref(c), _p(out), C.c_size_t(out.size)))
    return out


def decode_image_rgba8(data):
    """image::load_from_memory(..).to_rgba8() (lib.rs:281-283): PNG (8- and 16-bit) / JPEG -> (h, w, 4) uint8."""
    buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    w, h = C.c_uint32(), C.c_uint32()
    check(lib().sr_decode_image_rgba8(buf, C.c_size_t(len(data)), C.byref(w), C.byref(h), None, C.c_size_t(0)))
    out = np.zeros((h.value, w.value, 4), dtype=np.uint8)
    check(lib().sr_decode_image_rgba8(buf, C.c_size_t(len(data)), C.byref(w), C.byref(h), _p(out), C.c_size_t(out.size)))
    return out


def emissive_triangles_from_mesh(vertices, indices, material):
    v = np.ascontiguousarray(vertices, dtype=abi.VERTEX)
    i = np.ascontiguousarray(indices, dtype=np.uint32)
    m = np.ascontiguousarray(material, dtype=abi.MATERIAL)
    out = np.zeros(max(len(i) // 3, 1), dtype=abi.EMISSIVE_TRIANGLE)
    n = C.c_uint32()
    check(lib().sr_emissive_triangles_from_mesh(_p(v), C.c_uint32(len(v)), _p(i), C.c_uint32(len(i)), _p(m), _p(out), C.c_uint32(len(out)), C.byref(n)))
    return out[: n.value].copy()


def bvh_layout():
    """(children per node, dwords per node, first plane dword, first child dword) of this build's BVH (csrc/bvh_layout.h)."""
    w, nd, po, co = (C.c_uint32() for _ in range(4))
    check(lib().sr_bvh_layout(C.byref(w), C.byref(nd), C.byref(po), C.byref(co)))
    return w.value, nd.value, po.value, co.value


def host_bvh(v0_e1_e2):
    """Host-only BVH build (no GPU): returns (nodes[n, node_dwords] u32, tris[n,12] f32, max_depth, max_stack)."""
    v = np.ascontiguousarray(v0_e1_e2, dtype=np.float32).reshape(-1, 9)
    h = C.c_void_p()
    check(lib().sr_host_bvh_build(_p(v), C.c_uint32(len(v)), C.byref(h)))
    try:
        np_, tp = C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)()
        nn, nt, md, ms = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().sr_host_bvh_get(h, C.byref(np_), C.byref(nn), C.byref(tp), C.byref(nt), C.byref(md), C.byref(ms)))
        nodes = np.ctypeslib.as_array(np_, shape=(nn.value, bvh_layout()[1])).copy()
        tris = np.ctypeslib.as_array(tp, shape=(nt.value, 12)).copy() if nt.value else np.zeros((0, 12), np.float32)
    finally:
        lib().sr_host_bvh_destroy(h)
    return nodes, tris, md.value, ms.value


def decode_node(node):
    """Child boxes of one quantised node as the kernel decodes them: plane = fmaf(q, 2^e, origin).
    Returns (lo[W,3], hi[W,3], child[W]) in float32 / int32."""
    W, _, po, co = bvh_layout()
    n = np.asarray(node, dtype=np.uint32)
    origin = n[0:3].view(np.float32)
    ex = int(n[3])
    scale = np.array([np.uint32(((ex >> (8 * a)) & 0xFF) << 23) for a in range(3)], dtype=np.uint32).view(np.float32)
    pd = W // 4                      # dwords per plane; planes LX LY LZ HX HY HZ
    lo = np.zeros((W, 3), np.float32)
    hi = np.zeros((W, 3), np.float32)
    for c in range(W):
        for a in range(3):
            ql = np.float32((int(n[po + a * pd + c // 4]) >> (8 * (c % 4))) & 0xFF)
            qh = np.float32((int(n[po + (3 + a) * pd + c // 4]) >> (8 * (c % 4))) & 0xFF)
            # one rounding, like v_fma_f32: the product q * 2^e is exact in float64
            lo[c, a] = np.float32(np.float64(ql) * np.float64(scale[a]) + np.float64(origin[a]))
            hi[c, a] = np.float32(np.float64(qh) * np.float64(scale[a]) + np.float64(origin[a]))
    return lo, hi, n[co:co + W].view(np.int32)


class DeviceFrame:
    """Frame buffers in HBM with the layouts of SrRtParams (torch tensors on one device)."""

    def __init__(self, width, height, blue_noise, device="cuda:0", primary=None):
        """`primary`: allocate the primary-hit hand-off buffer (SrRtParams.primary_payload). Default: yes, unless
        SUNRAY_PRIMARY_REUSE=0 is set in the environment (A/B switch of tests and scripts)."""
        import os
        import torch
        self.width, self.height = width, height
        n = width * height
        self.device = torch.device(device)
        z = lambda *shape, dtype: torch.zeros(*shape, dtype=dtype, device=self.device)
        self.raw_color = z(n, 4, dtype=torch.float32)
        self.depth = z(n, dtype=torch.int16)       # R16_SFLOAT bits
        self.normal = z(n, dtype=torch.int32)      # R8G8B8A8_SNORM
        self.diffuse = z(n, dtype=torch.int32)     # B10G11R11_UFLOAT
        self.motion = z(n, dtype=torch.int32)      # R16G16_SFLOAT
        self.reservoirs = [z(n, 12, dtype=torch.int32), z(n, 12, dtype=torch.int32)]      # 48 B records
        self.reservoirs_gi = [z(n, 12, dtype=torch.int32), z(n, 12, dtype=torch.int32)]
        # post-RT chain images (B10G11R11 ping-pongs, RGBA8 output)
        self.accum = [z(n, dtype=torch.int32), z(n, dtype=torch.int32)]
        self.denoise = [z(n, dtype=torch.int32), z(n, dtype=torch.int32)]
        self.output = z(n, dtype=torch.int32)
        if primary is None:
            primary = os.environ.get("SUNRAY_PRIMARY_REUSE", "1") != "0"
        self.primary = z(n, 8, dtype=torch.int32) if primary else None      # 32-B RayPayload of the camera ray (RIS -> final hand-off)
        bn = np.ascontiguousarray(blue_noise, dtype=np.uint8)
        self.blue_noise_shape = bn.shape[:2]
        self.blue_noise = torch.from_numpy(bn.copy()).to(self.device)

    # host copies in the oracle's dtypes
    def host(self):
        out = {
            "raw_color": self.raw_color.cpu().numpy(),
            "depth": self.depth.cpu().numpy().view(np.uint16),
            "normal": self.normal.cpu().numpy().view(np.uint32),
            "diffuse": self.diffuse.cpu().numpy().view(np.uint32),
            "motion": self.motion.cpu().numpy().view(np.uint32),
            "reservoirs": [r.cpu().numpy().view(np.uint32).reshape(-1).view(abi.RESERVOIR) for r in self.reservoirs],
            "reservoirs_gi": [r.cpu().numpy().view(np.uint32).reshape(-1).view(abi.RESERVOIR_GI) for r in self.reservoirs_gi],
            "accum": [a.cpu().numpy().view(np.uint32) for a in self.accum],
            "denoise": [a.cpu().numpy().view(np.uint32) for a in self.denoise],
            "output": self.output.cpu().numpy().view(np.uint32),
        }
        return out


class Scene:
    """ResourceManager + acceleration structures of one GPU (sr_scene_*)."""

    def __init__(self, device_index=0, instancing=None):
        self._h = C.c_void_p()
        check(lib().sr_scene_create(C.c_int(device_index), C.byref(self._h)))
        self.device_index = device_index
        if instancing is not None:
            self.set_instancing(instancing)

    def close(self):
        if getattr(self, "_h", None):
            lib().sr_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # Renderer::load_mesh (lib.rs:873-954)
    def add_mesh(self, key, vertices, indices, material):
        v = np.ascontiguousarray(vertices, dtype=abi.VERTEX)
        i = np.ascontiguousarray(indices, dtype=np.uint32)
        m = np.ascontiguousarray(material, dtype=abi.MATERIAL)
        slot = C.c_uint32()
        check(lib().sr_scene_add_mesh(self._h, C.c_uint64(key), _p(v), C.c_uint32(len(v)), _p(i), C.c_uint32(len(i)), _p(m), C.byref(slot)))
        return slot.value

    def add_blas(self, key, vertices, indices, material, emissive):
        v = np.ascontiguousarray(vertices, dtype=abi.VERTEX)
        i = np.ascontiguousarray(indices, dtype=np.uint32)
        m = np.ascontiguousarray(material, dtype=abi.MATERIAL)
        e = np.ascontiguousarray(emissive, dtype=abi.EMISSIVE_TRIANGLE)
        slot = C.c_uint32()
        check(lib().sr_scene_add_blas(self._h, C.c_uint64(key), _p(v), C.c_uint32(len(v)), _p(i), C.c_uint32(len(i)), _p(m),
                                      _p(e) if len(e) else None, C.c_uint32(len(e)), C.byref(slot)))
        return slot.value

    def remove(self, key):
        check(lib().sr_scene_remove(self._h, C.c_uint64(key)))

    # Image::new_from_data (image/mod.rs:82-111) / Sampler::new (image/sampler.rs:44-67)
    def add_image(self, pixels):
        a = np.ascontiguousarray(pixels, dtype=np.uint8)
        h, w = a.shape[0], a.shape[1]
        ch = 1 if a.ndim == 2 else a.shape[2]
        slot = C.c_uint32()
        check(lib().sr_scene_add_image(self._h, _p(a), C.c_uint32(w), C.c_uint32(h), C.c_uint32(ch), C.byref(slot)))
        return slot.value

    def add_sampler(self, min_filter, mag_filter, address_mode_u, address_mode_v):
        d = abi.SrSamplerDesc(min_filter, mag_filter, address_mode_u, address_mode_v)
        slot = C.c_uint32()
        check(lib().sr_scene_add_sampler(self._h, C.byref(d), C.byref(slot)))
        return slot.value

    # frame_instance_data (resource_manager.rs:216-267) + TLAS build
    def set_instances(self, instances):
        keys = np.array([k for k, _ in instances], dtype=np.uint64)
        counts = np.array([len(t) for _, t in instances], dtype=np.uint32)
        xf = np.array([np.asarray(t, dtype=np.float32).reshape(12) for _, ts in instances for t in ts], dtype=np.float32).reshape(-1, 12)
        if len(xf) == 0:
            xf = np.zeros((1, 12), dtype=np.float32)
        check(lib().sr_scene_set_instances(self._h, _p(keys), _p(counts), C.c_uint32(len(keys)), _p(np.ascontiguousarray(xf))))

    def tile_row_costs(self, which, width, y0, rows):
        """Measured cycles per 8-pixel tile row of the last launch of pass `which` (0 ris, 1 final) with this geometry."""
        out = np.zeros((rows + 7) // 8 + 1, dtype=np.float64)
        n = C.c_uint32()
        check(lib().sr_scene_read_tile_row_costs(self._h, C.c_int(which), C.c_uint32(width), C.c_uint32(y0), C.c_uint32(rows), _p(out),
                                                 C.c_uint32(len(out)), C.byref(n)))
        return out[:n.value]

    def tile_costs(self, which, width, y0, rows):
        """Measured cycles of every 8x8 tile (row-major, ty * tiles_x + tx) of the last full-width launch of pass `which`."""
        out = np.zeros(((width + 7) // 8) * ((rows + 7) // 8), dtype=np.uint32)
        n = C.c_uint32()
        check(lib().sr_scene_read_tile_costs(self._h, C.c_int(which), C.c_uint32(width), C.c_uint32(y0), C.c_uint32(rows), _p(out),
                                             C.c_uint32(len(out)), C.byref(n)))
        return out[:n.value]

    def set_instancing(self, mode):
        """Form of the acceleration structure at the next set_instances: "auto" | "flat" | "two_level" (sr_scene_set_instancing)."""
        check(lib().sr_scene_set_instancing(self._h, C.c_uint32({"auto": 0, "flat": 1, "two_level": 2}[mode])))
        return self

    def two_level(self):
        """True while the structure is built in the two-level form."""
        now = C.c_uint32()
        check(lib().sr_scene_instancing(self._h, None, C.byref(now)))
        return bool(now.value)

    def force_next_op(self, op):
        check(lib().sr_scene_force_next_op(self._h, C.c_uint32(op)))

    def end_frame(self):
        check(lib().sr_scene_end_frame(self._h))

    def as_state(self):
        """-> (SrAsState, last op) of the scene's acceleration structure."""
        st, op = abi.SrAsState(), C.c_uint32()
        check(lib().sr_scene_as_state(self._h, C.byref(st), C.byref(op)))
        return st, op.value

    def read_bvh(self):
        st = self.bvh_stats()
        nodes = np.zeros((st.n_nodes, bvh_layout()[1]), dtype=np.uint32)
        tris = np.zeros((max(st.n_triangles, 1), 12), dtype=np.float32)
        check(lib().sr_scene_read_bvh(self._h, _p(nodes), _p(tris)))
        return nodes, tris[:st.n_triangles]

    def load(self, desc):
        for img in desc.images:
            self.add_image(img)
        for smp in desc.samplers:
            self.add_sampler(*smp)
        for m in desc.meshes:
            self.add_mesh(m.key, m.vertices, m.indices, m.material)
        self.set_instances(desc.instances)
        return self

    def tables(self):
        tp, ip, ep, mp = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nt, nl, ne, nm = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().sr_scene_get_tables(self._h, C.byref(tp), C.byref(nt), C.byref(ip), C.byref(nl), C.byref(ep), C.byref(ne), C.byref(mp), C.byref(nm)))

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dtype=dt)
            buf = (C.c_char * (n * dt.itemsize)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dt).copy()
        return {"transforms": arr(tp, nt.value, abi.TRANSFORM), "indirection": arr(ip, nl.value, abi.EMISSIVE_INDIRECTION),
                "emissive_triangles": arr(ep, ne.value, abi.EMISSIVE_TRIANGLE), "num_lights": nl.value,
                "meshes_info": arr(mp, nm.value, abi.MESH_INFO)}

    def bvh_stats(self):
        s = abi.SrBvhStats()
        check(lib().sr_scene_bvh_stats(self._h, C.byref(s)))
        return s

    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # TraceRay test hooks: torch uint8/any tensors holding abi.RAY records
    def trace_closest(self, rays_t, n):
        import torch
        hits = torch.empty(n, 4, dtype=torch.float32, device=rays_t.device)
        check(lib().sr_trace_closest(self._h, C.c_void_p(rays_t.data_ptr()), C.c_uint32(n), C.c_void_p(hits.data_ptr()), self._stream()))
        return hits

    def trace_any(self, rays_t, n):
        import torch
        occ = torch.empty(n, dtype=torch.int32, device=rays_t.device)
        check(lib().sr_trace_any(self._h, C.c_void_p(rays_t.data_ptr()), C.c_uint32(n), C.c_void_p(occ.data_ptr()), self._stream()))
        return occ

    def shade_closest_hit(self, hits_t, n):
        import torch
        out = torch.empty(n, 8, dtype=torch.int32, device=hits_t.device)
        check(lib().sr_shade_closest_hit(self._h, C.c_void_p(hits_t.data_ptr()), C.c_uint32(n), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def any_hit_ignores(self, hits_t, n):
        import torch
        out = torch.empty(n, dtype=torch.int32, device=hits_t.device)
        check(lib().sr_any_hit_ignores(self._h, C.c_void_p(hits_t.data_ptr()), C.c_uint32(n), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def params(self, frame, matrices, frame_count, config=None, tile=None):
        p = abi.SrRtParams()
        p.scene = self._h
        p.raw_color = frame.raw_color.data_ptr()
        p.depth_img = frame.depth.data_ptr()
        p.normal_img = frame.normal.data_ptr()
        p.diffuse_img = frame.diffuse.data_ptr()
        p.motion_vec_img = frame.motion.data_ptr()
        self._m = matrices
        p.matrices = C.pointer(matrices)
        p.blue_noise_tex = frame.blue_noise.data_ptr()
        p.blue_noise_h, p.blue_noise_w = frame.blue_noise_shape
        p.reservoirs[0], p.reservoirs[1] = frame.reservoirs[0].data_ptr(), frame.reservoirs[1].data_ptr()
        p.reservoirs_gi[0], p.reservoirs_gi[1] = frame.reservoirs_gi[0].data_ptr(), frame.reservoirs_gi[1].data_ptr()
        prim = getattr(frame, "primary", None)
        p.primary_payload = prim.data_ptr() if prim is not None else None
        p.frame_count = frame_count
        p.use_srgb = 0
        p.width, p.height = frame.width, frame.height
        if tile:                      # (y0, h) rows, or (y0, h, x0, w) rows x columns; h == 0 / w == 0: all of them
            p.tile_y0, p.tile_h = tile[0], tile[1]
            if len(tile) == 4:
                p.tile_x0, p.tile_w = tile[2], tile[3]
        p.config = config or abi.SrTraceConfig.reference()
        return p

    def trace_ris(self, frame, matrices, frame_count, config=None, tile=None):
        p = self.params(frame, matrices, frame_count, config, tile)
        check(lib().sr_trace_ris(C.byref(p), self._stream()))

    def trace_final(self, frame, matrices, frame_count, config=None, tile=None):
        p = self.params(frame, matrices, frame_count, config, tile)
        check(lib().sr_trace_final(C.byref(p), self._stream()))

    def reset_counters(self):
        check(lib().sr_scene_reset_counters(self._h, self._stream()))

    def counters(self):
        c = abi.SrRayCounters()
        check(lib().sr_scene_read_counters(self._h, self._stream(), C.byref(c)))
        return c

    def set_instrumented(self, on):
        check(lib().sr_scene_set_instrumented(self._h, C.c_int(1 if on else 0)))

    def enable_timing(self, on):
        check(lib().sr_scene_enable_timing(self._h, C.c_int(1 if on else 0)))

    def read_timing(self, kind):
        ms, n = C.c_double(), C.c_uint32()
        check(lib().sr_scene_read_timing(self._h, C.c_int(kind), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def post_chain(frame, frame_count, exposure=1.0, denoise_passes=4):
    """temporal_accumulation -> denoise_0..N-1 -> postprocess on the current stream (lib.rs:1576-1615)."""
    p = abi.post_params(frame, frame_count, lambda t: t.data_ptr(), exposure, denoise_passes)
    check(lib().sr_post_temporal(C.byref(p), _stream()))
    check(lib().sr_post_denoise(C.byref(p), _stream()))
    check(lib().sr_post_tonemap(C.byref(p), _stream()))


def _instance_arrays(instances):
    keys = np.array([k for k, _ in instances], dtype=np.uint64)
    counts = np.array([len(t) for _, t in instances], dtype=np.uint32)
    xf = np.array([np.asarray(t, dtype=np.float32).reshape(12) for _, ts in instances for t in ts], dtype=np.float32).reshape(-1, 12)
    if len(xf) == 0:
        xf = np.zeros((1, 12), dtype=np.float32)
    return keys, counts, np.ascontiguousarray(xf)


def default_noise_texture(w=128, h=128, seed=7):
    out = np.zeros((h, w, 4), dtype=np.uint8)
    check(lib().sr_default_noise_texture(C.c_uint32(w), C.c_uint32(h), C.c_uint32(seed), _p(out)))
    return out


def _np_from(ptr, n, dt):
    if n == 0:
        return np.zeros(0, dtype=dt)
    buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr.value)
    return np.frombuffer(buf, dtype=dt).copy()


def gltf_parse(path):
    """sr_gltf_open + accessors -> dict(blases=[dict(vertices, indices, material (unresolved), emissive)], instances=[(blas, 3x4)],
    images=[(h,w,c) uint8], samplers=[(min, mag, u, v)], textures=[(sampler or -1, source)])."""
    g = C.c_void_p()
    check(lib().sr_gltf_open(path.encode(), C.byref(g)))
    try:
        nb, ni, nim, ns, nt = (C.c_uint32() for _ in range(5))
        check(lib().sr_gltf_counts(g, C.byref(nb), C.byref(ni), C.byref(nim), C.byref(ns), C.byref(nt)))
        out = dict(blases=[], instances=[], images=[], samplers=[], textures=[])
        for i in range(nb.value):
            vp, ip, ep = C.c_void_p(), C.c_void_p(), C.c_void_p()
            nv, nx, ne = C.c_uint32(), C.c_uint32(), C.c_uint32()
            m = np.zeros((), dtype=abi.MATERIAL)
            check(lib().sr_gltf_blas(g, C.c_uint32(i), C.byref(vp), C.byref(nv), C.byref(ip), C.byref(nx), _p(m), C.byref(ep), C.byref(ne)))
            out["blases"].append(dict(vertices=_np_from(vp, nv.value, abi.VERTEX), indices=_np_from(ip, nx.value, np.uint32), material=m,
                                      emissive=_np_from(ep, ne.value, abi.EMISSIVE_TRIANGLE)))
        for i in range(ni.value):
            b = C.c_uint32()
            t = np.zeros(12, dtype=np.float32)
            check(lib().sr_gltf_instance(g, C.c_uint32(i), C.byref(b), _p(t)))
            out["instances"].append((b.value, t))
        for i in range(nim.value):
            pp = C.c_void_p()
            w, h, ch = C.c_uint32(), C.c_uint32(), C.c_uint32()
            check(lib().sr_gltf_image(g, C.c_uint32(i), C.byref(pp), C.byref(w), C.byref(h), C.byref(ch)))
            out["images"].append(_np_from(pp, w.value * h.value * ch.value, np.uint8).reshape(h.value, w.value, ch.value))
        for i in range(ns.value):
            d = abi.SrSamplerDesc()
            check(lib().sr_gltf_sampler(g, C.c_uint32(i), C.byref(d)))
            out["samplers"].append((d.min_filter, d.mag_filter, d.address_mode_u, d.address_mode_v))
        for i in range(nt.value):
            sm, src = C.c_int32(), C.c_uint32()
            check(lib().sr_gltf_texture(g, C.c_uint32(i), C.byref(sm), C.byref(src)))
            out["textures"].append((sm.value, src.value))
        return out
    finally:
        lib().sr_gltf_close(g)


class Renderer:
    """The reference's `Renderer<K>` surface for the built path (src/lib.rs:212-446, 586-639, 873-954,
    984-1238, 1908-1934), over sr_renderer_*. `camera` = (position, target, fov_y_degrees) — the
    reference's Camera (camera.rs:11-44); `instances` = [(mesh key, [3x4 row-major transform, ...]), ...]."""

    def __init__(self, size, device_index=0):
        self._h = C.c_void_p()
        self.size = (int(size[0]), int(size[1]))
        check(lib().sr_renderer_create(C.c_int(device_index), C.c_uint32(self.size[0]), C.c_uint32(self.size[1]), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().sr_renderer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def resize(self, size):
        check(lib().sr_renderer_resize(self._h, C.c_uint32(int(size[0])), C.c_uint32(int(size[1]))))
        self.size = (int(size[0]), int(size[1]))

    # Frame / resize callbacks (lib.rs:537-554). The ctypes thunks are kept alive for the renderer's lifetime.
    _FRAME_CB = C.CFUNCTYPE(None, C.c_void_p)
    _RESIZE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32)

    def _keep(self, thunk):
        self.__dict__.setdefault("_thunks", []).append(thunk)
        return thunk

    def add_start_of_frame_callback(self, fn):
        check(lib().sr_renderer_add_start_of_frame_callback(self._h, self._keep(self._FRAME_CB(lambda _u: fn())), None))

    def add_end_of_frame_callback(self, fn):
        check(lib().sr_renderer_add_end_of_frame_callback(self._h, self._keep(self._FRAME_CB(lambda _u: fn())), None))

    def add_resize_callback(self, fn):
        check(lib().sr_renderer_add_resize_callback(self._h, self._keep(self._RESIZE_CB(lambda _u, w, h: fn((w, h)))), None))

    def set_blue_noise(self, rgba8):
        """Replaces the built-in noise texture (lib.rs:281-309): (h, w, 4) uint8, e.g. decode_image_rgba8 of the crate's PNG."""
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 4:
            raise ValueError("set_blue_noise: expected (h, w, 4) uint8")
        check(lib().sr_renderer_set_blue_noise(self._h, _p(a), C.c_uint32(a.shape[1]), C.c_uint32(a.shape[0])))

    def load_mesh(self, key, vertices, indices, material):
        v = np.ascontiguousarray(vertices, dtype=abi.VERTEX)
        i = np.ascontiguousarray(indices, dtype=np.uint32)
        m = np.ascontiguousarray(material, dtype=abi.MATERIAL)
        check(lib().sr_renderer_load_mesh(self._h, C.c_uint64(key), _p(v), C.c_uint32(len(v)), _p(i), C.c_uint32(len(i)), _p(m)))

    def set_config(self, config):
        check(lib().sr_renderer_set_config(self._h, C.byref(config)))

    def load_gltf(self, path):
        """Renderer::load_gltf (lib.rs:779-786) -> (group, [(key, [3x4 transforms])])."""
        ls = C.c_void_p()
        check(lib().sr_renderer_load_gltf(self._h, path.encode(), C.byref(ls)))
        try:
            group, nk, nt = C.c_uint64(), C.c_uint32(), C.c_uint32()
            kp, cp, tp = C.c_void_p(), C.c_void_p(), C.c_void_p()
            check(lib().sr_loaded_scene_get(ls, C.byref(group), C.byref(kp), C.byref(cp), C.byref(nk), C.byref(tp), C.byref(nt)))
            keys, counts = _np_from(kp, nk.value, np.uint64), _np_from(cp, nk.value, np.uint32)
            xf = _np_from(tp, nt.value * 12, np.float32).reshape(-1, 12)
            inst, o = [], 0
            for k, c in zip(keys, counts):
                inst.append((int(k), [xf[o + j].copy() for j in range(int(c))]))
                o += int(c)
            return group.value, inst
        finally:
            lib().sr_loaded_scene_destroy(ls)

    def unload_scene(self, group):
        check(lib().sr_renderer_unload_scene(self._h, C.c_uint64(group)))

    def unload_mesh(self, key):
        check(lib().sr_renderer_unload_mesh(self._h, C.c_uint64(key)))

    def render(self, camera, instances):
        pos, tgt, fov = camera
        keys, counts, xf = _instance_arrays(instances)
        frame = C.c_uint64()
        check(lib().sr_renderer_render(self._h, _f3(pos), _f3(tgt), C.c_float(fov), _p(keys), _p(counts), C.c_uint32(len(keys)), _p(xf),
                                       None, C.byref(frame)))
        return frame.value

    def wait_frame(self, frame):
        check(lib().sr_renderer_wait_frame(self._h, C.c_uint64(frame)))

    def render_to_host_memory(self, camera, instances):
        pos, tgt, fov = camera
        keys, counts, xf = _instance_arrays(instances)
        out = np.zeros((self.size[1], self.size[0], 4), dtype=np.uint8)
        check(lib().sr_renderer_render_to_host_memory(self._h, _f3(pos), _f3(tgt), C.c_float(fov), _p(keys), _p(counts), C.c_uint32(len(keys)),
                                                      _p(xf), _p(out)))
        return out

    @property
    def relative_frame_count(self):
        n = C.c_uint32()
        check(lib().sr_renderer_get(self._h, None, None, None, C.byref(n)))
        return n.value


def rays_to_device(rays_np, device="cuda:0"):
    import torch
    a = np.ascontiguousarray(rays_np, dtype=abi.RAY)
    return torch.from_numpy(a.view(np.float32).reshape(-1, 8).copy()).to(device)


def hits_from_device(hits_t):
    return hits_t.cpu().numpy().reshape(-1).view(abi.HIT)
