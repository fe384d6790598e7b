"""ORACLE — test infrastructure only. numpy restatement of the reference's glTF ingest for the ray-tracing
path: `Gltf::new` + `create_default_scene` (src/vulkan_abstraction/gltf/mod.rs:57-373), the CPU side of
`Scene::load_into_gpu` (src/scene.rs:52-176), `add_scene_assets` (resource_manager.rs:372-413) and the
grouping of `Renderer::load_scene` (src/lib.rs:794-846).

The reference parses with the `gltf` crate 1.4.1 and decodes images with `image` 0.25.10 (Cargo.lock; neither
is vendored), so container / accessor / node-transform / material-default rules follow the glTF 2.0
specification. Parity unpinned: the reference holds no loader tests; its five example .glb files are used as
inputs where /root/reference is present (tests/test_gltf.py) and checked against hand-read values.
"""
import base64
import json
import os
import struct
import zlib

import numpy as np

from sunray_amd import abi

NULL = abi.NULL_TEXTURE
_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def decode_png(data, keep_16bit=False):
    """8-bit PNG -> (h, w, channels) uint8 the way image-rs hands it to the gltf crate. keep_16bit: also accept 16-bit
    samples (big-endian pairs) and return them as (h, w, channels) uint16 — the reference's blue-noise asset is one."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, plte, trns = 8, b"", None, None
    while pos < len(data):
        ln, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + ln]
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            plte = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif typ == b"tRNS":
            trns = np.frombuffer(body, dtype=np.uint8)
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            break
        pos += 12 + ln
    assert interlace == 0 and (depth in (1, 2, 4, 8) or (depth == 16 and keep_16bit))
    samples = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    stride = (w * samples * depth + 7) // 8
    bpp = max(samples * depth // 8, 1)
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, stride + 1)
    img = np.zeros((h, stride), dtype=np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        up = img[y - 1] if y else np.zeros(stride, dtype=np.int32)
        cur = img[y]
        if ft == 0:
            cur[:] = line
        elif ft == 2:
            cur[:] = (line + up) & 255
        else:
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = up[x]
                c = up[x - bpp] if x >= bpp else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pp = a + b - c
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
    img = img.astype(np.uint8)
    if depth == 16:
        assert ctype in (0, 2, 4, 6)
        v = img.reshape(h, w * samples, 2).astype(np.uint16)
        return np.ascontiguousarray(((v[..., 0] << 8) | v[..., 1]).reshape(h, w, samples))
    if depth < 8:
        bits = np.unpackbits(img, axis=1)[:, :w * samples * depth].reshape(h, w * samples, depth)
        vals = (bits * (1 << np.arange(depth - 1, -1, -1))).sum(axis=2)
    else:
        vals = img.astype(np.int64)
    if ctype == 3:
        out = plte[vals.reshape(h, w)]
        if trns is not None:
            alpha = np.full(256, 255, dtype=np.uint8)
            alpha[:len(trns)] = trns
            out = np.concatenate([out, alpha[vals.reshape(h, w)][..., None]], axis=-1)
        return np.ascontiguousarray(out.astype(np.uint8))
    if depth < 8:
        vals = vals * (255 // ((1 << depth) - 1))
    out = vals.reshape(h, w, samples).astype(np.uint8)
    return np.ascontiguousarray(out)


class GltfRef:
    def __init__(self, path):
        self.dir = os.path.dirname(os.path.abspath(path))
        raw = open(path, "rb").read()
        self.bin = None
        if raw[:4] == b"glTF":
            _, version, length = struct.unpack("<4sII", raw[:12])
            assert version == 2
            pos, js = 12, None
            while pos + 8 <= length:
                clen, ctype = struct.unpack("<II", raw[pos:pos + 8])
                body = raw[pos + 8:pos + 8 + clen]
                if ctype == 0x4E4F534A and js is None:
                    js = body
                elif ctype == 0x004E4942 and self.bin is None:
                    self.bin = body
                pos += 8 + clen
            self.doc = json.loads(js)
        else:
            self.doc = json.loads(raw)
        self.buffers = []
        for i, b in enumerate(self.doc.get("buffers", [])):
            self.buffers.append(self._uri(b["uri"]) if "uri" in b else self.bin)
        self._build()

    def _uri(self, uri):
        if uri.startswith("data:"):
            return base64.b64decode(uri.split(",", 1)[1])
        from urllib.parse import unquote
        return open(os.path.join(self.dir, unquote(uri)), "rb").read()

    # accessor -> (count, ncomp) array; integer components are normalised like `into_f32()` when as_float
    def accessor(self, index, as_float=True):
        a = self.doc["accessors"][index]
        dt, nc = np.dtype(_COMP[a["componentType"]]), _NCOMP[a["type"]]

        def rows_of(view, extra, count, itemsize, packed=False):
            bv = self.doc["bufferViews"][view]
            off = bv.get("byteOffset", 0) + extra
            stride = itemsize if packed else (bv.get("byteStride", 0) or itemsize)
            buf = np.frombuffer(self.buffers[bv["buffer"]], dtype=np.uint8)
            return np.ascontiguousarray(np.lib.stride_tricks.as_strided(buf[off:], shape=(count, itemsize), strides=(stride, 1)))
        if "bufferView" in a:
            arr = rows_of(a["bufferView"], a.get("byteOffset", 0), a["count"], dt.itemsize * nc).view(dt).reshape(a["count"], nc)
        else:
            arr = np.zeros((a["count"], nc), dtype=dt)
        if "sparse" in a:       # glTF 2.0 sparse accessor: `count` elements replaced at `indices` by tightly packed `values`
            sp = a["sparse"]
            idt = np.dtype(_COMP[sp["indices"]["componentType"]])
            at = rows_of(sp["indices"]["bufferView"], sp["indices"].get("byteOffset", 0), sp["count"], idt.itemsize, True).view(idt).reshape(-1)
            vals = rows_of(sp["values"]["bufferView"], sp["values"].get("byteOffset", 0), sp["count"], dt.itemsize * nc, True).view(dt).reshape(-1, nc)
            arr = arr.copy()
            arr[at.astype(np.int64)] = vals
        if not as_float:
            return arr.astype(np.uint32)
        if dt == np.float32:
            return arr
        if dt == np.uint8:
            return arr.astype(np.float32) / np.float32(255.0)
        if dt == np.uint16:
            return arr.astype(np.float32) / np.float32(65535.0)
        if dt == np.int8:
            return np.maximum(arr.astype(np.float32) / np.float32(127.0), np.float32(-1.0))
        if dt == np.int16:
            return np.maximum(arr.astype(np.float32) / np.float32(32767.0), np.float32(-1.0))
        return arr.astype(np.float32)

    @staticmethod
    def _tex(owner, name):
        t = (owner or {}).get(name)
        return (t["index"], t.get("texCoord", 0)) if t is not None else (NULL, 0)

    def material(self, prim):
        mat = self.doc["materials"][prim["material"]] if "material" in prim else {}
        pbr = mat.get("pbrMetallicRoughness", {})
        ext = mat.get("extensions", {})
        strength = ext["KHR_materials_emissive_strength"].get("emissiveStrength", 1.0) if "KHR_materials_emissive_strength" in ext else 0.0
        transmission = ext.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0)
        ior = ext.get("KHR_materials_ior", {}).get("ior", 1.5) if "KHR_materials_ior" in ext else 1.5
        m = abi.material(base_color=tuple(pbr.get("baseColorFactor", [1, 1, 1, 1])), metallic=pbr.get("metallicFactor", 1.0),
                         roughness=pbr.get("roughnessFactor", 1.0), emissive_factor=tuple(mat.get("emissiveFactor", [0, 0, 0])),
                         emissive_strength=strength, transmission=transmission, ior=ior)
        sets = []
        for field, (owner, name) in (("base_color", (pbr, "baseColorTexture")), ("metallic_roughness", (pbr, "metallicRoughnessTexture")),
                                     ("normal", (mat, "normalTexture")), ("occlusion", (mat, "occlusionTexture")), ("emissive", (mat, "emissiveTexture"))):
            idx, s = self._tex(owner, name)
            m[field + "_image"] = idx           # unresolved: glTF texture index
            sets.append(s)
        ef = m["emissive_factor"]
        return m, sets, bool(ef[3] > 0 or ef[0] != 0 or ef[1] != 0 or ef[2] != 0)

    @staticmethod
    def node_matrix(node):
        if "matrix" in node:
            return np.array(node["matrix"], dtype=np.float32).reshape(4, 4).T.copy()
        t = np.array(node.get("translation", [0, 0, 0]), dtype=np.float32)
        x, y, z, w = np.array(node.get("rotation", [0, 0, 0, 1]), dtype=np.float32)
        s = np.array(node.get("scale", [1, 1, 1]), dtype=np.float32)
        x2, y2, z2 = x + x, y + y, z + z
        xx2, xy2, xz2, yy2, yz2, zz2 = x2 * x, x2 * y, x2 * z, y2 * y, y2 * z, z2 * z
        sy2, sz2, sx2 = y2 * w, z2 * w, x2 * w
        one = np.float32(1.0)
        R = np.array([[one - yy2 - zz2, xy2 - sz2, xz2 + sy2], [xy2 + sz2, one - xx2 - zz2, yz2 - sx2], [xz2 - sy2, yz2 + sx2, one - xx2 - yy2]], dtype=np.float32)
        M = np.eye(4, dtype=np.float32)
        M[:3, :3] = R * s[None, :]
        M[:3, 3] = t
        return M

    @staticmethod
    def matmul(a, b):
        r = np.zeros((4, 4), dtype=np.float32)
        for i in range(4):
            for j in range(4):
                r[i, j] = ((a[i, 0] * b[0, j] + a[i, 1] * b[1, j]) + a[i, 2] * b[2, j]) + a[i, 3] * b[3, j]
        return r

    def _build(self):
        d = self.doc
        filt = lambda v: abi.FILTER_LINEAR if v is None else (abi.FILTER_NEAREST if v in (9728, 9984, 9986) else abi.FILTER_LINEAR)
        wrap = lambda v: {33071: abi.ADDRESS_CLAMP_TO_EDGE, 33648: abi.ADDRESS_MIRRORED_REPEAT}.get(v, abi.ADDRESS_REPEAT)
        self.samplers = [(filt(s.get("minFilter")), filt(s.get("magFilter")), wrap(s.get("wrapS")), wrap(s.get("wrapT"))) for s in d.get("samplers", [])]
        self.textures = [(t.get("sampler", -1), t["source"]) for t in d.get("textures", [])]
        self.images = []
        for im in d.get("images", []):
            if "uri" in im:
                data = self._uri(im["uri"])
            else:
                bv = d["bufferViews"][im["bufferView"]]
                data = self.buffers[bv["buffer"]][bv.get("byteOffset", 0):bv.get("byteOffset", 0) + bv["byteLength"]]
            self.images.append(decode_png(bytes(data)))
        self.blases, self.instances = [], []          # blas = dict(vertices, indices, material, emissive)
        self._data, self._blas_of = {}, {}
        scene = d["scenes"][d.get("scene", 0)]
        for n in scene.get("nodes", []):
            self._explore(n, np.eye(4, dtype=np.float32))

    def _explore(self, ni, parent):
        node = self.doc["nodes"][ni]
        xf = self.matmul(parent, self.node_matrix(node))
        if "mesh" in node:
            i = 0
            for prim in self.doc["meshes"][node["mesh"]]["primitives"]:
                if prim.get("mode", 4) != 4:
                    continue
                at = prim["attributes"]
                key = (at["POSITION"], prim["indices"] if "indices" in prim else i)
                mat, sets, emissive = self.material(prim)
                pos = self.accessor(at["POSITION"])
                et = np.zeros(0, dtype=abi.EMISSIVE_TRIANGLE)
                if emissive:
                    idx = self.accessor(prim["indices"], False).ravel() if "indices" in prim else np.arange(len(pos), dtype=np.uint32)
                    tri = idx[:len(idx) // 3 * 3].reshape(-1, 3)
                    et = np.zeros(len(tri), dtype=abi.EMISSIVE_TRIANGLE)
                    for k, name in enumerate(("v0", "v1", "v2")):
                        et[name][:, :3] = pos[tri[:, k]]
                    ef = mat["emissive_factor"]
                    et["emission"][:, :3] = np.array([ef[0] * ef[3], ef[1] * ef[3], ef[2] * ef[3]], dtype=np.float32)
                if key not in self._data:
                    v = np.zeros(len(pos), dtype=abi.VERTEX)
                    v["position"] = pos
                    v["normal"] = self.accessor(at["NORMAL"])[:len(pos)]
                    if "TANGENT" in at:
                        v["tangent"] = self.accessor(at["TANGENT"])[:len(pos)]
                    idx = self.accessor(prim["indices"], False).ravel() if "indices" in prim else np.arange(len(pos) // 3, dtype=np.uint32)
                    for field, s in zip(("base_color", "metallic_roughness", "normal", "occlusion", "emissive"), sets):
                        uv = self.accessor(at["TEXCOORD_%d" % s])
                        n = min(len(uv), len(pos))
                        v[field + "_tex_coord"][:n] = uv[:n]
                    self._data[key] = (v, idx.astype(np.uint32))
                if key not in self._blas_of:
                    v, idx = self._data[key]
                    self._blas_of[key] = len(self.blases)
                    self.blases.append(dict(vertices=v, indices=idx, material=mat, emissive=et))
                self.instances.append((self._blas_of[key], xf[:3, :].reshape(12).copy()))
                i += 1
        for c in node.get("children", []):
            self._explore(c, xf)

    # Renderer::load_scene on top of the parse: slot resolution + keys + grouped instances
    def loaded(self, group=0, first_image_slot=0, sampler_slots=None):
        """Returns (meshes [(key, vertices, indices, resolved material, emissive)], grouped instances, images,
        samplers-to-add). `sampler_slots` maps a sampler tuple to its slot (dedup state carried across loads)."""
        sampler_slots = {} if sampler_slots is None else sampler_slots
        new_samplers = []

        def slot_of(desc):
            if desc not in sampler_slots:
                sampler_slots[desc] = len(sampler_slots)
                new_samplers.append(desc)
            return sampler_slots[desc]
        smp = [slot_of(s) for s in self.samplers]
        default = slot_of((abi.FILTER_LINEAR, abi.FILTER_LINEAR, abi.ADDRESS_CLAMP_TO_EDGE, abi.ADDRESS_CLAMP_TO_EDGE))
        meshes = []
        for b, blas in enumerate(self.blases):
            m = blas["material"].copy()
            for field in ("base_color", "metallic_roughness", "normal", "occlusion", "emissive"):
                t = int(m[field + "_image"])
                if t != NULL:
                    s, src = self.textures[t]
                    m[field + "_image"] = first_image_slot + src
                    m[field + "_sampler"] = smp[s] if s >= 0 else default
            meshes.append(((group << 32) | b, blas["vertices"], blas["indices"], m, blas["emissive"]))
        grouped = [(k, []) for k, *_ in meshes]
        for b, xf in self.instances:
            grouped[b][1].append(xf)
        return meshes, grouped, self.images, new_samplers
