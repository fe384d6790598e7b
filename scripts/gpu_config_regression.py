"""ms per frame of every BASELINE.json configuration's stand-in, the reference's own png-example workload and a moving-camera
sequence of the bench scene — the table every kept heuristic change is held against (profiles/r03_config_regression.txt), not
the bench frame alone. Uses only harness calls that exist since round 1, so the same file runs in an older tree:

    python scripts/gpu_config_regression.py [label]                         (this tree)
    scripts/gpu_regression_vs_round1.sh                                     (this tree and a checkout of round 1's, same box)

Per row: undisturbed device time of the two passes (HIP events around every launch, passes back to back), mean of frames
3..9; rays are the traversals the kernels executed."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sunray_amd import abi, scenes, runtime as rt

ASSETS = os.environ.get("SUNRAY_REF_ASSETS", os.path.join(ROOT, "tests", "golden", "ref_assets"))
label = sys.argv[1] if len(sys.argv) > 1 else "HEAD"


def run(name, W, H, scene, camera_of_frame, cfg, frames=10):
    fr = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
    prev = None
    scene.enable_timing(True)
    rows = []
    for f in range(frames):
        pos, tgt, fov = camera_of_frame(f)
        m = rt.camera_matrices(pos, tgt, fov, W, H, prev)
        prev = list(m.view_proj)
        scene.reset_counters()
        if cfg.enable_restir:
            scene.trace_ris(fr, m, f, cfg)
        scene.trace_final(fr, m, f, cfg)
        c = scene.counters()
        a = scene.read_timing(0)[0] if cfg.enable_restir else 0.0
        b = scene.read_timing(1)[0]
        rows.append((a, b, c.closest_queries + c.any_queries))
    scene.enable_timing(False)
    r = np.array(rows[3:], dtype=np.float64)
    ms = r[:, 0].mean() + r[:, 1].mean()
    crc = int(torch.sum(fr.raw_color.view(torch.int32).to(torch.int64) & 0xFFFF).item()) & 0xFFFFFFFF     # cheap content check, tree-independent
    print("%-7s %-52s %4dx%-4d %8d tris | ris %6.3f + final %6.3f = %6.3f ms | %5.1f Mray/frame | sum16 %08x" % (
        label, name, W, H, scene.bvh_stats().n_triangles, r[:, 0].mean(), r[:, 1].mean(), ms, r[:, 2].mean() / 1e6, crc), flush=True)
    del fr
    return ms


def static(desc):
    return lambda f: (desc.camera_pos, desc.camera_target, desc.fov_y)


def gltf_scene(path):
    """Scene of a .glb through gltf_parse + add_blas (untextured assets only: materials carry NULL textures)."""
    g = rt.gltf_parse(path)
    sc = rt.Scene(0)
    for i, b in enumerate(g["blases"]):
        m = b["material"].copy()
        for k in ("base_color", "metallic_roughness", "normal", "occlusion", "emissive"):
            assert int(m[k + "_image"]) == abi.NULL_TEXTURE
        sc.add_blas(i, b["vertices"], b["indices"], m, b["emissive"])
    grouped = [(i, [x for bi, x in g["instances"] if bi == i]) for i in range(len(g["blases"]))]
    sc.set_instances([(k, xs) for k, xs in grouped if xs])
    return sc


ref = abi.SrTraceConfig.reference()
c2 = abi.SrTraceConfig.reference(); c2.enable_restir, c2.max_bounces, c2.shadow_bounces = 0, 1, 1
c3 = abi.SrTraceConfig.reference(); c3.enable_restir, c3.max_bounces, c3.shadow_bounces = 0, 2, 2

d = scenes.cornell_box()
run("config 1 Cornell box, reference constants", 1920, 1080, rt.Scene(0).load(d), static(d), ref)
d = scenes.torus_knot()
knot = rt.Scene(0).load(d)
run("config 2 torus knot 70k, diffuse only (1 bounce + NEE)", 1920, 1080, knot, static(d), c2)
run("config 2 scene, reference constants", 1920, 1080, knot, static(d), ref)
del knot
d = scenes.heightfield(708)
hf = rt.Scene(0).load(d)
run("config 3 heightfield 1M, 2 bounces + NEE", 1920, 1080, hf, static(d), c3)
run("bench: heightfield 1M, reference constants", 1920, 1080, hf, static(d), ref)
run("bench scene, camera sliding 0.15 / frame + rising", 1920, 1080, hf,
    lambda f: ((d.camera_pos[0] + 0.15 * f, d.camera_pos[1] + 0.04 * f, d.camera_pos[2] - 0.1 * f), d.camera_target, d.fov_y), ref)
run("config 5 extent: heightfield 1M at 3840x2160", 3840, 2160, hf, static(d), ref, frames=7)
del hf
d = scenes.atrium()
run("config 4 textured atrium 250k, RIS + final", 1920, 1080, rt.Scene(0).load(d), static(d), ref)
room = os.path.join(ASSETS, "ReflectionRoom.glb")
if os.path.exists(room) and hasattr(rt, "gltf_parse"):
    cam = ((13.0, 30.0, 25.0), (0.0, 13.0, 0.0), 45.0)      # examples/png/main.rs:52-55
    run("reference png example: ReflectionRoom.glb", 1600, 1200, gltf_scene(room), lambda f: cam, ref)
