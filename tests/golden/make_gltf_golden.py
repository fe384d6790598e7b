"""Writes tests/golden/mini_scene.glb (a textured two-mesh scene with a node hierarchy, written by tests/gltf_util.py) and
mini_scene_expected.npz (what oracle/gltf_ref.py parses from it). The committed pair freezes one input / output of the
loader independently of the writer's future changes.  python tests/golden/make_gltf_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import gltf_util  # noqa: E402
from oracle import gltf_ref  # noqa: E402


def build(path):
    rng = np.random.default_rng(2024)
    b = gltf_util.GltfBuilder()
    b.image(rng.integers(0, 256, size=(6, 5, 3), dtype=np.uint8))
    b.image(rng.integers(0, 256, size=(4, 4), dtype=np.uint8))
    b.add("samplers", {"magFilter": 9728, "wrapS": 33648})
    b.add("textures", {"source": 0, "sampler": 0})
    b.add("textures", {"source": 1})
    b.add("materials", {"pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.8, 0.7, 1.0], "roughnessFactor": 0.6, "metallicFactor": 0.1,
                                                 "baseColorTexture": {"index": 0}}, "normalTexture": {"index": 1},
                        "emissiveFactor": [1.0, 0.5, 0.25], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 4.0}}})
    b.add("materials", {"pbrMetallicRoughness": {"roughnessFactor": 0.3}, "extensions": {"KHR_materials_transmission": {"transmissionFactor": 1.0},
                                                                                           "KHR_materials_ior": {"ior": 1.45}}})
    for k, idx_dtype in enumerate((np.uint16, np.uint8)):
        n = 4 + k
        pos = rng.normal(size=(3 * n, 3)).astype(np.float32)
        nrm = rng.normal(size=(3 * n, 3)).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        attrs = {"POSITION": b.accessor(pos, "VEC3"), "NORMAL": b.accessor(nrm, "VEC3"),
                 "TANGENT": b.accessor(rng.normal(size=(3 * n, 4)).astype(np.float32), "VEC4"),
                 "TEXCOORD_0": b.accessor(rng.random((3 * n, 2)).astype(np.float32), "VEC2")}
        b.add("meshes", {"primitives": [{"attributes": attrs, "indices": b.accessor(rng.permutation(3 * n).astype(idx_dtype), "SCALAR"), "material": k}]})
    child = b.add("nodes", {"mesh": 1, "translation": [0.5, -1.0, 2.0], "rotation": [0.0, 0.3826834, 0.0, 0.9238795], "scale": [1.0, 2.0, 0.5]})
    root = b.add("nodes", {"mesh": 0, "children": [child], "translation": [1.0, 2.0, 3.0]})
    again = b.add("nodes", {"mesh": 1, "matrix": [0, 0, 1, 0, 0, 1, 0, 0, -1, 0, 0, 0, 7, 8, 9, 1]})
    b.add("scenes", {"nodes": [root, again]})
    b.doc["scene"] = 0
    b.write_glb(path)


if __name__ == "__main__":
    glb = os.path.join(HERE, "mini_scene.glb")
    build(glb)
    ref = gltf_ref.GltfRef(glb)
    out = {"n_blases": len(ref.blases), "instance_blas": np.array([b for b, _ in ref.instances], dtype=np.uint32),
           "instance_xf": np.stack([x for _, x in ref.instances]), "samplers": np.array(ref.samplers, dtype=np.uint32),
           "textures": np.array(ref.textures, dtype=np.int64)}
    for i, bl in enumerate(ref.blases):
        out["b%d_vertices" % i] = bl["vertices"].view(np.uint8)
        out["b%d_indices" % i] = bl["indices"]
        out["b%d_material" % i] = np.frombuffer(bl["material"].tobytes(), dtype=np.uint8)
        out["b%d_emissive" % i] = bl["emissive"].view(np.uint8) if len(bl["emissive"]) else np.zeros(0, np.uint8)
    for i, im in enumerate(ref.images):
        out["image%d" % i] = im
    np.savez_compressed(os.path.join(HERE, "mini_scene_expected.npz"), **out)
    print("wrote", glb)
