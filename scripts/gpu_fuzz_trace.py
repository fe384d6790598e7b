"""Randomised parity fuzz of the tracers against the oracle (brute force for small soups, oracle BVH for big ones):
triangle soups over many scales, long thin triangles, rotated / non-uniformly scaled instances, rays that start on
geometry, axis-parallel rays, rays aimed at vertices and edges. Prints mismatch counts; exit code 1 on any mismatch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import binding as ob
from sunray_amd import abi, scenes, runtime as rt

def soup(rng, n, scale, thin):
    c = rng.normal(size=(n, 3)) * scale
    e = rng.normal(size=(n, 2, 3)) * scale * 0.05
    if thin:
        e[:, 1] *= 1e-3
    pos = np.concatenate([c[:, None], c[:, None] + e], axis=1).reshape(-1, 3).astype(np.float32)
    nrm = np.tile(np.array([[0, 1, 0]], np.float32), (3 * n, 1))
    return scenes.make_vertices(pos, nrm), np.arange(3 * n, dtype=np.uint32)

def xform(rng, scale):
    a = rng.uniform(0, 2 * np.pi, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
    Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
    S = np.diag(rng.uniform(0.3, 2.5, 3))
    if rng.random() < 0.15:                      # squashed nearly flat: ill-conditioned, the two-level form walks a baked world-space copy
        S[rng.integers(0, 3), :] *= 1e-4         # (exactly flat sheets are left to tests/test_gpu_two_level.py: with them the rays of this
                                                 # fuzzer that START on the sheet get hits at t ~ 0 whose order is pure rounding noise — the
                                                 # oracle's own tree and its brute force then differ in ~1 ray of 10 000)
    M = np.zeros((3, 4)); M[:, :3] = Rx @ Ry @ S; M[:, 3] = rng.normal(size=3) * scale
    return M.astype(np.float32).reshape(12)

def rays_for(rng, desc, osc, n, scale):
    o = (rng.normal(size=(n, 3)) * scale * 1.5).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = n // 5
    d[:k] = np.eye(3)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1))          # axis-parallel
    # aimed exactly at vertices of the first mesh's world-space triangles
    m = desc.meshes[0]; M = np.asarray(desc.instances[0][1][0]).reshape(3, 4)
    wp = m.vertices["position"] @ M[:, :3].T + M[:, 3]
    tgt = wp[rng.integers(0, len(wp), k)]
    dd = tgt - o[k:2 * k]; d[k:2 * k] = dd / np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-20)
    # aimed at edge midpoints
    tri = wp.reshape(-1, 3, 3)[rng.integers(0, len(wp) // 3, k)]
    tgt = 0.5 * (tri[:, 0] + tri[:, 1])
    dd = tgt - o[2 * k:3 * k]; d[2 * k:3 * k] = dd / np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-20)
    # starting exactly on geometry (secondary-ray style)
    o[3 * k:4 * k] = (tri[:, 0] * 0.3 + tri[:, 1] * 0.3 + tri[:, 2] * 0.4).astype(np.float32)
    r = np.zeros(n, dtype=abi.RAY)
    r["origin"] = o; r["dir"] = d.astype(np.float32); r["tmin"] = 0.001 * min(scale, 1.0); r["tmax"] = 1e4 * scale
    r["tmin"][::7] = -2.0 * scale * rng.random(len(r["tmin"][::7])).astype(np.float32)       # hits behind the origin are legal queries
    r["tmax"][3::5] = (scale * 3.0 * rng.random(len(r["tmax"][3::5]))).astype(np.float32)    # short segments
    return r

bad = 0
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    scale = float(10.0 ** rng.uniform(-2, 3))
    n = int(rng.choice([50, 400, 3000, 20000]))
    desc = scenes.SceneDesc("fuzz")
    for k in range(int(rng.integers(1, 4))):
        v, i = soup(rng, n, scale, thin=bool(rng.integers(0, 2)))
        desc.meshes.append(scenes.MeshDesc(k + 1, v, i, abi.material()))
        desc.instances.append((k + 1, [xform(rng, scale) for _ in range(int(rng.integers(1, 4)))]))
    osc = ob.OracleScene().load(desc)
    gsc = rt.Scene(0).load(desc)
    brute = desc.n_triangles() <= 5000
    osc.set_brute_force(brute)
    rays = rays_for(rng, desc, osc, 40000 if brute else 200000, scale)
    for mode in ("sah", "lbvh", "update", "two_level"):
        if mode == "two_level":              # a tree per mesh in object space + a top-level tree over the instances
            gsc.set_instancing("two_level"); gsc.set_instances(desc.instances)
            assert gsc.two_level()
        if mode == "lbvh":
            gsc.force_next_op(abi.OP_FAST_BUILD); gsc.set_instances(desc.instances)
        if mode == "update":
            gsc.force_next_op(abi.OP_UPDATE); gsc.set_instances(desc.instances)
        rt_ = rt.rays_to_device(rays)
        got = rt.hits_from_device(gsc.trace_closest(rt_, len(rays)))
        want = osc.trace_closest(rays)
        occ_g = gsc.trace_any(rt_, len(rays)).cpu().numpy().view(np.uint32)
        occ_w = osc.trace_any(rays)
        m1 = int((got.view(np.uint32).reshape(-1, 4) != want.view(np.uint32).reshape(-1, 4)).any(axis=1).sum())
        m2 = int((occ_g != occ_w).sum())
        bad += m1 + m2
        for r in np.nonzero((got.view(np.uint32).reshape(-1, 4) != want.view(np.uint32).reshape(-1, 4)).any(axis=1))[0][:4]:
            print("   ray %d %s\n      gpu %s oracle %s" % (r, rays[r], got[r], want[r]), flush=True)
        print("scene %2d scale %9.3g tris %6d (%s, %s): closest mismatches %d, any mismatches %d" % (it, scale, desc.n_triangles(), "brute" if brute else "oracle bvh", mode, m1, m2), flush=True)
print("TOTAL mismatches", bad)
sys.exit(1 if bad else 0)
