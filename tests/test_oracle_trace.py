"""Ray/triangle queries of the oracle: the BVH path must equal the brute-force
ground truth exactly, bounds are exclusive, ties go to the lowest triangle index, and the analytic
scene KATs of SURVEY.md §8c hold."""
import ctypes as C

import numpy as np
import pytest

from sunray_amd import abi, scenes


def make_rays(origins, dirs, tmin=0.001, tmax=10000.0):
    r = np.zeros(len(origins), dtype=abi.RAY)
    r["origin"] = origins
    r["dir"] = dirs
    r["tmin"] = tmin
    r["tmax"] = tmax
    return r


def random_rays(n, seed, box=((-1.2, -0.2, -1.2), (1.2, 2.2, 3.5))):
    rng = np.random.default_rng(seed)
    lo, hi = np.array(box[0]), np.array(box[1])
    o = (rng.random((n, 3)) * (hi - lo) + lo).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return make_rays(o, d.astype(np.float32))


def camera_rays(oracle, desc, W, H):
    """Primary rays exactly as K1 generates them (ray_gen_ris.slang:44-53), via numpy fp32."""
    m = oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
    vi = np.array(list(m.view_inverse), dtype=np.float32).reshape(4, 4)
    pi = np.array(list(m.proj_inverse), dtype=np.float32).reshape(4, 4)
    px, py = np.meshgrid(np.arange(W), np.arange(H))
    dx = ((px + 0.5) / W * 2 - 1).astype(np.float32).ravel()
    dy = ((py + 0.5) / H * 2 - 1).astype(np.float32).ravel()
    tgt = np.stack([dx, dy, np.ones_like(dx), np.ones_like(dx)], 1) @ pi.T
    t3 = tgt[:, :3] / np.linalg.norm(tgt[:, :3], axis=1, keepdims=True)
    d = (np.concatenate([t3, np.zeros((len(t3), 1), np.float32)], 1) @ vi.T)[:, :3]
    o = np.tile(vi[:3, 3], (len(d), 1))
    return make_rays(o.astype(np.float32), d.astype(np.float32))


@pytest.mark.parametrize("scene_fn", [scenes.cornell_box, scenes.cornell_glass_mirror])
def test_bvh_equals_brute_force(oracle, scene_fn):
    desc = scene_fn()
    s = oracle.OracleScene().load(desc)
    rays = np.concatenate([random_rays(20000, 3), camera_rays(oracle, desc, 96, 96)])
    # shadow-like short segments too
    short = random_rays(5000, 4)
    short["tmax"] = np.random.default_rng(5).random(5000).astype(np.float32) * 2 + 0.01
    rays = np.concatenate([rays, short])
    s.set_brute_force(False)
    hb, ob_ = s.trace_closest(rays), s.trace_any(rays)
    s.set_brute_force(True)
    hg, og = s.trace_closest(rays), s.trace_any(rays)
    assert np.array_equal(hb.view(np.uint32), hg.view(np.uint32))
    assert np.array_equal(ob_, og)
    assert (hg["t"][hg["tri"] != 0xFFFFFFFF] > 0).all()
    # any-hit must agree with closest-hit's existence
    assert np.array_equal(og != 0, hg["tri"] != 0xFFFFFFFF)


def test_bvh_equals_brute_force_axis_aligned_and_degenerate_rays(oracle):
    """Rays with zero direction components and origins ON wall planes (0*inf NaNs in the slab test)."""
    desc = scenes.cornell_box()
    s = oracle.OracleScene().load(desc)
    o, d = [], []
    for ax in range(3):
        for sign in (-1.0, 1.0):
            for k in range(60):
                p = np.array([0.05 * (k % 7) - 0.2, 0.3 + 0.02 * k, 0.1 * (k % 5) - 0.3], dtype=np.float32)
                v = np.zeros(3, np.float32); v[ax] = sign
                o.append(p); d.append(v)
    # origins exactly on the floor / back wall / left wall planes, sliding along them
    for k in range(50):
        o.append(np.array([-1.0, 0.5 + 0.01 * k, 0.2], np.float32)); d.append(np.array([0.0, 0.6, -0.8], np.float32))
        o.append(np.array([0.1, 0.0, 0.2], np.float32)); d.append(np.array([0.6, 0.0, -0.8], np.float32))
    rays = make_rays(np.array(o), np.array(d))
    s.set_brute_force(False); hb = s.trace_closest(rays); ab = s.trace_any(rays)
    s.set_brute_force(True); hg = s.trace_closest(rays); ag = s.trace_any(rays)
    assert np.array_equal(hb.view(np.uint32), hg.view(np.uint32)) and np.array_equal(ab, ag)


def test_small_heightfield_bvh_equals_brute_force(oracle):
    desc = scenes.heightfield(n=40, n_lights=2)  # 3042 + 4 triangles
    s = oracle.OracleScene().load(desc)
    rays = np.concatenate([camera_rays(oracle, desc, 80, 45), random_rays(4000, 9, box=((-15, 0, -15), (15, 8, 15)))])
    s.set_brute_force(False); hb = s.trace_closest(rays)
    s.set_brute_force(True); hg = s.trace_closest(rays)
    assert np.array_equal(hb.view(np.uint32), hg.view(np.uint32))


def test_analytic_cornell_kats(oracle):
    """centre-pixel primary t = camera-to-back-wall distance; light area; num_lights = 2."""
    desc = scenes.cornell_box()
    s = oracle.OracleScene().load(desc)
    t = s.tables()
    assert t["num_lights"] == 2 and t["n_triangles"] == 12 + 960
    # light quad 0.6 x 0.6 -> two triangles of area 0.18
    for e in t["emissive_triangles"]:
        a, b, c = e["v0"][:3], e["v1"][:3], e["v2"][:3]
        assert abs(0.5 * np.linalg.norm(np.cross(b - a, c - a)) - 0.18) < 1e-6
        assert np.allclose(e["emission"], [10, 10, 10, 0])
    # camera at z=3.4 looking down -z at the back wall z=-1: t = 4.4 for the exact centre ray
    rays = make_rays(np.array([[0, 1, 3.4]], np.float32), np.array([[0, 0, -1]], np.float32))
    h = s.trace_closest(rays)[0]
    assert h["t"] == np.float32(4.4) and h["tri"] in (4, 5)  # back wall = mesh key 3 -> triangles 4,5
    # 255x255: pixel (127,127) has d = 0 -> direction exactly (0,0,-1)
    cr = camera_rays(oracle, desc, 255, 255)
    assert np.allclose(cr["dir"][127 * 255 + 127], [0, 0, -1], atol=1e-7)
    # row 0 is the TOP of the image, +x is right
    assert cr["dir"][0][0] < 0 and cr["dir"][0][1] > 0


def test_interval_is_exclusive_and_two_sided(oracle):
    L = oracle.lib()
    tuv = (C.c_float * 3)()
    v = lambda *a: (C.c_float * 3)(*a)
    tri = (v(0, 0, 0), v(1, 0, 0), v(0, 1, 0))
    # hit at t = 1 from either side (cull disabled: resource_manager.rs:249)
    assert L.orc_intersect_tri(v(.25, .25, 1), v(0, 0, -1), *tri, C.c_float(0.001), C.c_float(10000), tuv) == 1 and tuv[0] == 1.0
    assert (tuv[1], tuv[2]) == (0.25, 0.25)  # barycentrics = weights of v1, v2 (closest_hit.slang:15-17)
    assert L.orc_intersect_tri(v(.25, .25, -1), v(0, 0, 1), *tri, C.c_float(0.001), C.c_float(10000), tuv) == 1
    # exclusive bounds
    assert L.orc_intersect_tri(v(.25, .25, 1), v(0, 0, -1), *tri, C.c_float(1.0), C.c_float(10000), tuv) == 0
    assert L.orc_intersect_tri(v(.25, .25, 1), v(0, 0, -1), *tri, C.c_float(0.001), C.c_float(1.0), tuv) == 0
    # edges and vertices are inside (u >= -eps, v >= -eps, u + v <= 1 + eps; eps = 1e-6)
    assert L.orc_intersect_tri(v(0, 0, 1), v(0, 0, -1), *tri, C.c_float(0.001), C.c_float(10000), tuv) == 1
    assert L.orc_intersect_tri(v(.5, .5, 1), v(0, 0, -1), *tri, C.c_float(0.001), C.c_float(10000), tuv) == 1
    assert L.orc_intersect_tri(v(.6, .5, 1), v(0, 0, -1), *tri, C.c_float(0.001), C.c_float(10000), tuv) == 0
    # parallel ray and degenerate triangle never hit (det = 0 -> inf/NaN fail every comparison)
    assert L.orc_intersect_tri(v(.25, .25, 1), v(1, 0, 0), *tri, C.c_float(0.001), C.c_float(10000), tuv) == 0
    assert L.orc_intersect_tri(v(.25, .25, 1), v(0, 0, -1), v(0, 0, 0), v(1, 0, 0), v(2, 0, 0), C.c_float(0.001), C.c_float(10000), tuv) == 0


def test_tie_goes_to_lowest_triangle_index(oracle):
    """Two coincident quads: equal t -> the hit reports the lower global triangle index, in both
    the brute-force and the BVH path (the oracle's definition of the driver's unspecified tie-break)."""
    s = oracle.OracleScene()
    v, i = scenes.quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1), (0, 1, 0))
    s.add_mesh(10, v, i, abi.material())
    s.add_mesh(11, v, i, abi.material(base_color=(0.1, 0.2, 0.3, 1)))
    s.set_instances([(11, [abi.IDENTITY_TRANSFORM]), (10, [abi.IDENTITY_TRANSFORM])])
    rays = make_rays(np.array([[0.3, 1, 0.2], [-0.3, 1, -0.2]], np.float32), np.array([[0, -1, 0], [0, -1, 0]], np.float32))
    for brute in (True, False):
        s.set_brute_force(brute)
        h = s.trace_closest(rays)
        assert list(h["tri"]) == [0, 1] and (h["t"] == 1.0).all()
    # instance order (not key order) defines the index: instance 0 is key 11
    pl = s.shade_closest_hit(s.trace_closest(rays))
    assert pl["albedo_packed"][0] == oracle.lib().orc_pack_unorm_4x8(*[C.c_float(x) for x in (0.1, 0.2, 0.3, 1.0)])


def test_empty_scene_and_miss(oracle):
    s = oracle.OracleScene()
    s.set_instances([])
    t = s.tables()
    # dummy padding of Renderer::render (lib.rs:1058-1081)
    assert t["num_lights"] == 1 and len(t["transforms"]) == 1 and t["n_triangles"] == 0
    assert np.allclose(t["transforms"][0]["m"], abi.IDENTITY_TRANSFORM)
    rays = random_rays(16, 1)
    h = s.trace_closest(rays)
    assert (h["t"] == -1.0).all() and (h["tri"] == 0xFFFFFFFF).all() and not s.trace_any(rays).any()
    pl = s.shade_closest_hit(h)
    assert (pl["dist"] == -1.0).all() and not pl["emission"].any()  # ray_miss.slang:11-12


def test_mesh_validation_matches_load_mesh(oracle):
    s = oracle.OracleScene()
    v, i = scenes.quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1), (0, 1, 0))
    s.add_mesh(1, v, i, abi.material())
    with pytest.raises(ValueError):
        s.add_mesh(1, v, i, abi.material())            # key already registered (lib.rs:880-884)
    with pytest.raises(ValueError):
        s.add_mesh(2, v, i[:5], abi.material())        # not a triangle list (lib.rs:885-891)
    with pytest.raises(ValueError):
        s.add_mesh(3, v, np.array([0, 1, 7], np.uint32), abi.material())  # index out of range (:892-899)
    with pytest.raises(ValueError):
        s.set_instances([(99, [abi.IDENTITY_TRANSFORM])])  # unknown key (resource_manager.rs:227-231)


def test_watertightness_of_shared_edges(oracle):
    """Plain Möller–Trumbore leaves cracks at shared edges (~11 % of rays aimed exactly AT an edge miss
    both triangles). With the barycentric bounds widened by 1e-6 the crack rate on a finely tessellated,
    rotated plane must be negligible."""
    n = 33
    g = np.linspace(-1, 1, n, dtype=np.float32)
    xx, zz = np.meshgrid(g, g, indexing="ij")
    pos = np.stack([xx, np.zeros_like(xx), zz], -1).reshape(-1, 3)
    vid = lambda a, b: a * n + b
    ii, jj = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    ii, jj = ii.ravel(), jj.ravel()
    idx = np.stack([vid(ii, jj), vid(ii, jj + 1), vid(ii + 1, jj + 1), vid(ii, jj), vid(ii + 1, jj + 1), vid(ii + 1, jj)], 1).ravel().astype(np.uint32)
    s = oracle.OracleScene()
    s.add_mesh(1, scenes.make_vertices(pos, np.tile([0, 1, 0], (len(pos), 1))), idx, abi.material())
    s.set_instances([(1, [scenes.rotate_y(0.3, 0.1, 0.0, -0.2, 1.3)])])
    rng = np.random.default_rng(2)
    # targets: points on grid edges (one coordinate on a grid line) in the instance's local frame
    k = 40000
    a = g[rng.integers(1, n - 1, k)]
    b = (rng.random(k) * 1.8 - 0.9).astype(np.float32)
    swap = rng.random(k) < 0.5
    local = np.stack([np.where(swap, a, b), np.zeros(k, np.float32), np.where(swap, b, a)], 1)
    M = scenes.rotate_y(0.3, 0.1, 0.0, -0.2, 1.3).reshape(3, 4)
    world = local @ M[:, :3].T + M[:, 3]
    o = world + np.stack([rng.normal(size=k) * 0.5, np.full(k, 2.0) + rng.random(k), rng.normal(size=k) * 0.5], 1)
    d = world - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    h = s.trace_closest(make_rays(o.astype(np.float32), d.astype(np.float32)))
    miss = (h["tri"] == 0xFFFFFFFF).mean()
    assert miss < 2e-3, miss
