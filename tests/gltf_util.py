"""Test helper: writes glTF 2.0 files (.glb, or .gltf + .bin / data: URIs) from numpy data, with PNG-encoded
images, so the loader tests have inputs on machines without /root/reference (the GPU box)."""
import base64
import json
import struct
import zlib

import numpy as np


def encode_png(img, filter_type=None):
    """(h, w[, c]) uint8 -> PNG bytes; colour type from the channel count; per-row filter cycles 0..4 unless fixed."""
    a = np.ascontiguousarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[..., None]
    h, w, c = a.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    rows = a.reshape(h, w * c).astype(np.int32)
    raw = bytearray()
    for y in range(h):
        ft = (y % 5) if filter_type is None else filter_type
        cur = rows[y]
        up = rows[y - 1] if y else np.zeros_like(cur)
        left = np.concatenate([np.zeros(c, dtype=np.int32), cur[:-c]])
        upleft = np.concatenate([np.zeros(c, dtype=np.int32), up[:-c]])
        if ft == 0:
            out = cur
        elif ft == 1:
            out = cur - left
        elif ft == 2:
            out = cur - up
        elif ft == 3:
            out = cur - ((left + up) >> 1)
        else:
            p = left + up - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
            out = cur - pred
        raw.append(ft)
        raw += (out & 255).astype(np.uint8).tobytes()

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)
    comp = zlib.compress(bytes(raw), 6)
    half = len(comp) // 2          # two IDAT chunks: the decoder must concatenate them
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
            chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b""))


class GltfBuilder:
    def __init__(self):
        self.bin = bytearray()
        self.doc = {"asset": {"version": "2.0"}, "buffers": [], "bufferViews": [], "accessors": [], "meshes": [], "nodes": [],
                    "scenes": [], "materials": [], "textures": [], "samplers": [], "images": []}

    def view(self, data, stride=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        bv = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride:
            bv["byteStride"] = stride
        self.bin += data
        self.doc["bufferViews"].append(bv)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, arr, type_, normalized=False, interleave_pad=0):
        a = np.ascontiguousarray(arr)
        ct = {np.dtype(np.float32): 5126, np.dtype(np.uint8): 5121, np.dtype(np.uint16): 5123, np.dtype(np.uint32): 5125,
              np.dtype(np.int8): 5120, np.dtype(np.int16): 5122}[a.dtype]
        rows = a.reshape(len(a), -1)
        if interleave_pad:          # strided bufferView: each element followed by `interleave_pad` junk bytes
            body = b"".join(r.tobytes() + b"\xAB" * interleave_pad for r in rows)
            bv = self.view(body, stride=rows.shape[1] * a.dtype.itemsize + interleave_pad)
        else:
            bv = self.view(rows.tobytes())
        acc = {"bufferView": bv, "componentType": ct, "count": len(a), "type": type_}
        if normalized:
            acc["normalized"] = True
        if type_ == "VEC3" and ct == 5126:
            acc["min"], acc["max"] = rows.min(0).tolist(), rows.max(0).tolist()
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    def image(self, pixels, embed=True):
        png = encode_png(pixels)
        if embed:
            self.doc["images"].append({"bufferView": self.view(png), "mimeType": "image/png"})
        else:
            self.doc["images"].append({"uri": "data:image/png;base64," + base64.b64encode(png).decode()})
        return len(self.doc["images"]) - 1

    def add(self, key, obj):
        self.doc[key].append(obj)
        return len(self.doc[key]) - 1

    def finish(self):
        doc = {k: v for k, v in self.doc.items() if not (isinstance(v, list) and len(v) == 0)}
        doc["buffers"] = [{"byteLength": len(self.bin)}]
        return doc

    def write_glb(self, path):
        doc = self.finish()
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        binc = bytes(self.bin) + b"\0" * (-len(self.bin) % 4)
        total = 12 + 8 + len(js) + 8 + len(binc)
        with open(path, "wb") as f:
            f.write(struct.pack("<4sII", b"glTF", 2, total))
            f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
            f.write(struct.pack("<II", len(binc), 0x004E4942) + binc)

    def write_gltf(self, path, external_bin=True):
        doc = self.finish()
        if external_bin:
            name = path.rsplit("/", 1)[-1].rsplit(".", 1)[0] + " data.bin"      # space: exercises percent-decoding
            open(path.rsplit("/", 1)[0] + "/" + name, "wb").write(bytes(self.bin))
            doc["buffers"][0]["uri"] = name.replace(" ", "%20")
        else:
            doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bytes(self.bin)).decode()
        open(path, "w").write(json.dumps(doc, indent=1))


def scene_to_gltf(desc, path, texcoord_u16=False):
    """Writes a scenes.SceneDesc (meshes, instances, images, samplers, resolved materials) as .glb: one glTF mesh per
    MeshDesc, one node per instance, one glTF texture per distinct (image, sampler) pair. Returns nothing."""
    from sunray_amd import abi
    b = GltfBuilder()
    for img in desc.images:
        b.image(img)
    for mn, mg, wu, wv in desc.samplers:
        b.add("samplers", {"minFilter": 9729 if mn else 9728, "magFilter": 9729 if mg else 9728,
                           "wrapS": {0: 10497, 1: 33648, 2: 33071}[wu], "wrapT": {0: 10497, 1: 33648, 2: 33071}[wv]})
    tex_of = {}

    def texture(img, smp):
        if (img, smp) not in tex_of:
            tex_of[(img, smp)] = b.add("textures", {"source": int(img), "sampler": int(smp)})
        return tex_of[(img, smp)]
    mesh_index = {}
    for m in desc.meshes:
        mat = m.material
        pbr = {"baseColorFactor": [float(x) for x in mat["base_color_value"]], "metallicFactor": float(mat["metallic_factor"]),
               "roughnessFactor": float(mat["roughness_factor"])}
        gm = {"pbrMetallicRoughness": pbr, "emissiveFactor": [float(x) for x in mat["emissive_factor"][:3]], "extensions": {}}
        if float(mat["emissive_factor"][3]) != 0.0:
            gm["extensions"]["KHR_materials_emissive_strength"] = {"emissiveStrength": float(mat["emissive_factor"][3])}
        if float(mat["transmission_factor"]) != 0.0:
            gm["extensions"]["KHR_materials_transmission"] = {"transmissionFactor": float(mat["transmission_factor"])}
        gm["extensions"]["KHR_materials_ior"] = {"ior": float(mat["ior"])}
        sets = {}
        for field, owner, name, setidx in (("base_color", pbr, "baseColorTexture", 0), ("metallic_roughness", pbr, "metallicRoughnessTexture", 0),
                                           ("normal", gm, "normalTexture", 1), ("emissive", gm, "emissiveTexture", 0)):
            if int(mat[field + "_image"]) != abi.NULL_TEXTURE:
                owner[name] = {"index": texture(int(mat[field + "_image"]), int(mat[field + "_sampler"])), "texCoord": setidx}
        mi = b.add("materials", gm)
        v = m.vertices
        attrs = {"POSITION": b.accessor(v["position"].astype(np.float32), "VEC3"), "NORMAL": b.accessor(v["normal"].astype(np.float32), "VEC3"),
                 "TANGENT": b.accessor(v["tangent"].astype(np.float32), "VEC4")}
        if texcoord_u16:
            raise NotImplementedError
        attrs["TEXCOORD_0"] = b.accessor(v["base_color_tex_coord"].astype(np.float32), "VEC2")
        attrs["TEXCOORD_1"] = b.accessor(v["normal_tex_coord"].astype(np.float32), "VEC2")
        mesh_index[m.key] = b.add("meshes", {"primitives": [{"attributes": attrs, "indices": b.accessor(m.indices.astype(np.uint32), "SCALAR"), "material": mi}]})
    roots = []
    for key, xforms in desc.instances:
        for x in xforms:
            M = np.eye(4, dtype=np.float32)
            M[:3, :] = np.asarray(x, dtype=np.float32).reshape(3, 4)
            roots.append(b.add("nodes", {"mesh": mesh_index[key], "matrix": [float(t) for t in M.T.reshape(-1)]}))
    b.add("scenes", {"nodes": roots})
    b.doc["scene"] = 0
    b.write_glb(path)
