"""Randomised parity fuzz of the two passes + post chain against the oracle: boxes of random triangles / spheres with
random materials (diffuse, rough and smooth metals, glass with random ior, lights of random strength, textured), several
frames with a slowly moving camera. Exit code 1 on any differing bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from sunray_amd import abi, scenes, runtime as rt

def rand_material(rng, tex):
    kind = rng.integers(0, 6)
    base = tuple(rng.uniform(0.1, 0.95, 3)) + (1.0,)
    if kind == 0: return abi.material(base_color=base, roughness=float(rng.uniform(0.25, 1.0)))
    if kind == 1: return abi.material(base_color=base, metallic=1.0, roughness=float(rng.uniform(0.0, 0.3)))
    if kind == 2: return abi.material(base_color=base, roughness=float(rng.uniform(0.0, 0.15)), transmission=1.0, ior=float(rng.uniform(1.0, 2.2)))
    if kind == 3: return abi.material(base_color=base, roughness=0.5, emissive_factor=tuple(rng.uniform(0.2, 1.0, 3)), emissive_strength=float(rng.uniform(0.5, 30.0)))
    if kind == 4: return abi.material(base_color=base, metallic=float(rng.uniform(0, 1)), roughness=float(rng.uniform(0.05, 0.5)))
    return abi.material(base_color=(1, 1, 1, 1), roughness=1.0, metallic=1.0, textures=tex)

bad = 0
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    desc = scenes.cornell_box()
    desc.name = "fuzz"
    img = rng.integers(0, 256, size=(16, 16, 4), dtype=np.uint8)
    nrm = np.concatenate([rng.integers(96, 160, size=(8, 8, 2), dtype=np.uint8), np.full((8, 8, 1), 255, np.uint8), np.full((8, 8, 1), 255, np.uint8)], axis=2)
    desc.images = [img, nrm]
    desc.samplers = [(1, int(rng.integers(0, 2)), int(rng.integers(0, 3)), int(rng.integers(0, 3)))]
    tex = {"base_color": (0, 0), "metallic_roughness": (0, 0), "normal": (1, 0), "emissive": (0, 0)} if rng.integers(0, 2) else {"base_color": (0, 0)}
    for m in desc.meshes:
        m.material = rand_material(rng, None if True else tex) if m.key != 6 else m.material     # walls: untextured random
    for k in range(int(rng.integers(2, 6))):
        if rng.integers(0, 2):
            v, i = scenes.uv_sphere(float(rng.uniform(0.1, 0.4)), 16, 8)
        else:
            v, i = scenes.grid_patch(rng.uniform(-0.5, 0.0, 3), rng.uniform(-0.6, 0.6, 3), rng.uniform(-0.6, 0.6, 3), 2, 2, (0, 1, 0), (1, 0, 0))
        desc.meshes.append(scenes.MeshDesc(100 + k, v, i, rand_material(rng, tex)))
        xs = [scenes.scale_rotate_y(float(rng.uniform(0, 6)), *rng.uniform(0.5, 1.5, 3), float(rng.uniform(-0.6, 0.6)), float(rng.uniform(0.3, 1.6)), float(rng.uniform(-0.6, 0.6)))
              for _ in range(int(rng.integers(1, 3)))]
        desc.instances.append((100 + k, xs))
    W, H = int(rng.integers(40, 160)), int(rng.integers(40, 120))
    noise = scenes.white_noise_rgba8()
    osc, gsc = ob.OracleScene().load(desc), rt.Scene(0).load(desc)
    of, gf = ob.HostFrame(W, H, noise), rt.DeviceFrame(W, H, noise)
    prev, diff = None, 0
    for f in range(4):
        pos = (desc.camera_pos[0] + 0.03 * f, desc.camera_pos[1], desc.camera_pos[2] - 0.02 * f)
        om = ob.camera_matrices(pos, desc.camera_target, desc.fov_y, W, H, prev)
        gm = rt.camera_matrices(pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(om.view_proj)
        osc.trace_ris(of, om, f); osc.trace_final(of, om, f); ob.post_chain(of, f)
        gsc.trace_ris(gf, gm, f); gsc.trace_final(gf, gm, f); rt.post_chain(gf, f)
        h = gf.host()
        for name, a, b in (("raw_color", of.raw_color, h["raw_color"]), ("reservoirs", of.reservoirs[f & 1], h["reservoirs"][f & 1]),
                           ("reservoirs_gi", of.reservoirs_gi[f & 1], h["reservoirs_gi"][f & 1]), ("normal", of.normal, h["normal"]),
                           ("depth", of.depth, h["depth"]), ("output", of.output, h["output"])):
            d = int((np.ascontiguousarray(a).view(np.uint8) != np.ascontiguousarray(b).view(np.uint8)).sum())
            if d:
                print("   frame %d %s: %d differing bytes" % (f, name, d))
            diff += d
    nan = int(np.isnan(h["raw_color"]).sum())
    print("scene %2d %dx%d meshes %d lights %d: differing bytes %d (NaN in radiance: %d)" % (it, W, H, len(desc.meshes), osc.tables()["num_lights"], diff, nan), flush=True)
    bad += diff
print("TOTAL differing bytes", bad)
sys.exit(1 if bad else 0)
