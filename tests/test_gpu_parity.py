"""GPU parity tests proper: every result of the HIP path (through the C ABI, libsunray_hip.so) is
compared with the CPU oracle on the same seeded inputs. The bar:
  * integer / packed / index outputs (hit records, G-buffer, reservoirs' packed fields): bit-exact;
  * fp32 outputs: the numerics contract makes them bit-exact too, which is stricter than the
    north-star's stated tolerance — image RMSE < 1e-3 on the fp32 radiance buffer. Both are asserted.
Run on an MI355X with:  python -m pytest tests -m gpu -x -q
"""
import os
import sys

import numpy as np
import pytest

from sunray_amd import abi, scenes

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)
sys.path.insert(0, os.path.dirname(__file__))
import make_golden  # noqa: E402
from test_oracle_trace import camera_rays, make_rays, random_rays  # noqa: E402

RMSE_BOUND = 1e-3  # BASELINE.json north_star: "pixel RMSE <1e-3 vs reference"


@pytest.fixture(scope="module")
def rt():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: run them through gpurun (-m gpu)")
    from sunray_amd import runtime
    return runtime


def rmse(a, b):
    d = a[:, :3].astype(np.float64) - b[:, :3].astype(np.float64)
    ok = np.isfinite(d).all(axis=1)
    return float(np.sqrt(np.mean(d[ok] ** 2))), int((~ok).sum())


def assert_bits_equal(a, b, what):
    a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    b = np.ascontiguousarray(b).view(np.uint8).reshape(-1)
    nd = int((a != b).sum())
    assert nd == 0, "%s: %d of %d bytes differ" % (what, nd, a.size)


def ref_closest(c):
    """Closest-hit TraceRay calls the REFERENCE issues for what was rendered: the traversals executed plus the camera-ray
    queries of the final pass that were answered from the RIS pass's hand-off (SrRtParams.primary_payload)."""
    return c.closest_queries + c.reused_primary_hits


def ref_any(c):
    """Existence queries the REFERENCE issues: the traversals executed plus the final GI visibility queries whose answer was known
    from the identical query of a spatial neighbour (SrRayCounters.reused_visibility_queries)."""
    return c.any_queries + c.reused_visibility_queries


def run_both(rt, oracle, desc, W, H, frames, blue_noise, cfg=None, check_counters=True, primary=None):
    """Renders `frames` consecutive frames on GPU and oracle; asserts bitwise parity every frame. `primary`: with / without
    the primary-hit hand-off buffer (None: the harness default, i.e. with)."""
    osc = oracle.OracleScene().load(desc)
    gsc = rt.Scene(0).load(desc)
    of, gf = oracle.HostFrame(W, H, blue_noise), rt.DeviceFrame(W, H, blue_noise, primary=primary)
    cfg = cfg or abi.SrTraceConfig.reference()
    prev = None
    for f in range(frames):
        om = oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        gm = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        assert bytes(om) == bytes(gm)
        prev = list(om.view_proj)
        osc.reset_counters(); gsc.reset_counters()
        if cfg.enable_restir:
            osc.trace_ris(of, om, f, cfg); gsc.trace_ris(gf, gm, f, cfg)
        osc.trace_final(of, om, f, cfg); gsc.trace_final(gf, gm, f, cfg)
        h = gf.host()
        cur = f & 1
        if cfg.enable_restir:
            assert_bits_equal(of.depth, h["depth"], "depth_img f%d" % f)
            assert_bits_equal(of.normal, h["normal"], "normal_img f%d" % f)
            assert_bits_equal(of.diffuse, h["diffuse"], "diffuse_img f%d" % f)
            assert_bits_equal(of.motion, h["motion"], "motion_vec_img f%d" % f)
            assert_bits_equal(of.reservoirs[cur], h["reservoirs"][cur], "reservoirs f%d" % f)
            assert_bits_equal(of.reservoirs_gi[cur], h["reservoirs_gi"][cur], "reservoirs_gi f%d" % f)
        e, n_nan = rmse(of.raw_color, h["raw_color"])
        assert e < RMSE_BOUND and n_nan == 0, (e, n_nan)
        assert_bits_equal(of.raw_color, h["raw_color"], "raw_color f%d" % f)
        if check_counters:
            oc, gc = osc.counters(), gsc.counters()
            assert (oc.closest_queries, oc.any_queries) == (ref_closest(gc), ref_any(gc))
            if cfg.flags & abi.TRACE_FLAG_TRACE_EVERY_QUERY:
                assert gc.reused_visibility_queries == 0
            # with the hand-off every pixel's camera ray is traversed once per frame (by the RIS pass), without it twice
            assert gc.reused_primary_hits == (W * H if (cfg.enable_restir and cfg.virtual_bounces and gf.primary is not None) else 0)
    return osc, gsc, of, gf


# ---- K2 / K3: TraceRay ------------------------------------------------------------------------
@pytest.mark.parametrize("scene_fn", [scenes.cornell_box, scenes.cornell_glass_mirror])
def test_trace_closest_and_any_equal_brute_force(rt, oracle, scene_fn):
    desc = scene_fn()
    osc = oracle.OracleScene().load(desc)
    osc.set_brute_force(True)
    gsc = rt.Scene(0).load(desc)
    short = random_rays(20000, 4)
    short["tmax"] = np.random.default_rng(5).random(20000).astype(np.float32) * 2 + 0.01
    rays = np.concatenate([random_rays(60000, 3), camera_rays(oracle, desc, 128, 128), short])
    rd = rt.rays_to_device(rays)
    hits = rt.hits_from_device(gsc.trace_closest(rd, len(rays)))
    occ = gsc.trace_any(rd, len(rays)).cpu().numpy().view(np.uint32)
    assert_bits_equal(osc.trace_closest(rays), hits, "closest hits")
    assert np.array_equal(osc.trace_any(rays), occ)
    c = gsc.counters()
    assert c.closest_queries == len(rays) and c.any_queries == len(rays)


def test_trace_axis_aligned_rays_on_box_faces(rt, oracle):
    """Zero direction components + origins exactly on wall planes / vertex coordinates."""
    desc = scenes.cornell_box()
    osc = oracle.OracleScene().load(desc); osc.set_brute_force(True)
    gsc = rt.Scene(0).load(desc)
    o, d = [], []
    for ax in range(3):
        for sign in (-1.0, 1.0):
            for k in range(60):
                o.append(np.array([0.05 * (k % 7) - 0.2, 0.3 + 0.02 * k, 0.1 * (k % 5) - 0.3], np.float32))
                v = np.zeros(3, np.float32); v[ax] = sign; d.append(v)
    for k in range(50):
        o.append(np.array([-1.0, 0.5 + 0.01 * k, 0.2], np.float32)); d.append(np.array([0.0, 0.6, -0.8], np.float32))
        o.append(np.array([0.1, 0.0, 0.2], np.float32)); d.append(np.array([0.6, 0.0, -0.8], np.float32))
    rays = make_rays(np.array(o), np.array(d))
    rd = rt.rays_to_device(rays)
    assert_bits_equal(osc.trace_closest(rays), rt.hits_from_device(gsc.trace_closest(rd, len(rays))), "axis-aligned closest")
    assert np.array_equal(osc.trace_any(rays), gsc.trace_any(rd, len(rays)).cpu().numpy().view(np.uint32))


def test_trace_ragged_sizes_and_empty_scene(rt, oracle):
    desc = scenes.cornell_box()
    osc = oracle.OracleScene().load(desc)
    gsc = rt.Scene(0).load(desc)
    for n in (1, 63, 64, 65, 257, 1000):
        rays = random_rays(n, 100 + n)
        rd = rt.rays_to_device(rays)
        assert_bits_equal(osc.trace_closest(rays), rt.hits_from_device(gsc.trace_closest(rd, n)), "n=%d" % n)
    import torch
    empty = torch.zeros(0, 8, dtype=torch.float32, device="cuda:0")
    assert gsc.trace_closest(empty, 0).shape[0] == 0
    g0 = rt.Scene(0); g0.set_instances([])
    rays = random_rays(100, 7)
    h = rt.hits_from_device(g0.trace_closest(rt.rays_to_device(rays), 100))
    assert (h["t"] == -1.0).all() and (h["tri"] == 0xFFFFFFFF).all()
    assert not g0.trace_any(rt.rays_to_device(rays), 100).cpu().numpy().any()
    assert g0.tables()["num_lights"] == 1  # dummy entry (lib.rs:1075-1081)


def test_trace_medium_scenes_equal_oracle_bvh(rt, oracle):
    for desc, box in ((scenes.torus_knot(), ((-4, 0, -4), (4, 5, 4))), (scenes.heightfield(n=160), ((-15, 0, -15), (15, 9, 15)))):
        osc = oracle.OracleScene().load(desc)
        gsc = rt.Scene(0).load(desc)
        rays = np.concatenate([camera_rays(oracle, desc, 320, 180), random_rays(50000, 11, box=box)])
        rd = rt.rays_to_device(rays)
        assert_bits_equal(osc.trace_closest(rays), rt.hits_from_device(gsc.trace_closest(rd, len(rays))), desc.name)
        assert np.array_equal(osc.trace_any(rays), gsc.trace_any(rd, len(rays)).cpu().numpy().view(np.uint32))


def test_trace_lone_long_rays_and_coincident_twins(rt, oracle):
    """The traversal hands subtrees of busy lanes to idle lanes of the same wave and merges the partial results by
    (t, triangle id): waves with ONE long ray (grazing the whole terrain) among 63 trivial ones force that path, a second
    instance of the same mesh at the same place gives every hit an equal-t twin with a higher id in another subtree, and
    negative tmin / short tmax exercise the ordered-key encoding of t."""
    desc = scenes.heightfield(n=160)
    key, xf = desc.instances[0]
    desc.instances[0] = (key, list(xf) + [xf[0].copy()])                   # coincident twin instance
    osc = oracle.OracleScene().load(desc)
    gsc = rt.Scene(0).load(desc)
    n_tris_one = len(desc.meshes[0].indices) // 3
    rng = np.random.default_rng(77)
    n = 64 * 300
    o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
    o[:] = (rng.random((n, 3)) * [30, 0, 30] + [-15, 12, -15]); d[:] = (0, 1, 0)        # trivial: above the terrain, pointing up
    long_ = np.arange(0, n, 64) + rng.integers(0, 64, n // 64)
    ang = rng.random(len(long_)) * 2 * np.pi
    o[long_] = np.stack([-16 * np.cos(ang), 2.0 + 6 * rng.random(len(long_)), -16 * np.sin(ang)], 1)
    d[long_] = np.stack([np.cos(ang), -0.02 * rng.random(len(long_)), np.sin(ang)], 1)   # grazing, across the whole field
    rays = make_rays(o, d / np.linalg.norm(d, axis=1, keepdims=True))
    mixed = random_rays(64 * 200, 78, box=((-15, 0, -15), (15, 9, 15)))
    mixed["tmin"][::3] = -5.0                                                            # hits behind the origin are legal
    mixed["tmax"][1::4] = rng.random(len(mixed["tmax"][1::4])).astype(np.float32) * 3
    rays = np.concatenate([rays, mixed])
    rd = rt.rays_to_device(rays)
    hits = rt.hits_from_device(gsc.trace_closest(rd, len(rays)))
    assert_bits_equal(osc.trace_closest(rays), hits, "closest hits, lone long rays")
    assert np.array_equal(osc.trace_any(rays), gsc.trace_any(rd, len(rays)).cpu().numpy().view(np.uint32))
    hit = hits["tri"] != 0xFFFFFFFF
    tri = hits["tri"][hit]
    assert hit[long_].any() and (tri < n_tris_one).any() and not ((tri >= n_tris_one) & (tri < 2 * n_tris_one)).any()   # the first copy, never its twin
    assert (hits["t"][hit] < 0).any()


# ---- K4 / K6: closest_hit / miss ------------------------------------------------------------------
def test_shade_closest_hit_payloads(rt, oracle):
    desc = scenes.cornell_glass_mirror()  # includes a rotated + scaled instance: non-trivial WorldToObject
    osc = oracle.OracleScene().load(desc)
    gsc = rt.Scene(0).load(desc)
    rays = np.concatenate([camera_rays(oracle, desc, 160, 160), random_rays(20000, 21)])
    hits_t = gsc.trace_closest(rt.rays_to_device(rays), len(rays))
    hits = rt.hits_from_device(hits_t)
    pl = gsc.shade_closest_hit(hits_t, len(rays)).cpu().numpy().view(np.uint32).reshape(-1).view(abi.RAY_PAYLOAD)
    assert_bits_equal(osc.shade_closest_hit(hits), pl, "RayPayload")
    assert (pl["dist"][hits["tri"] == 0xFFFFFFFF] == -1.0).all()


def small_atrium():
    return scenes.atrium(columns_per_side=4, col_segments=16, col_rings=4, floor_div=8, tex=64, n_lamps=6)


def test_textured_closest_hit_payloads(rt, oracle):
    """closest_hit with every texture path (base colour / metallic-roughness / normal map / emissive; LINEAR and
    NEAREST; REPEAT, MIRRORED_REPEAT, CLAMP_TO_EDGE; 1-, 3- and 4-channel images; scaled + rotated instances)."""
    desc = small_atrium()
    osc = oracle.OracleScene().load(desc)
    gsc = rt.Scene(0).load(desc)
    rays = np.concatenate([camera_rays(oracle, desc, 200, 120), random_rays(30000, 33, box=((-8, 0.1, -17), (8, 6.5, 17)))])
    hits_t = gsc.trace_closest(rt.rays_to_device(rays), len(rays))
    hits = rt.hits_from_device(hits_t)
    assert_bits_equal(osc.trace_closest(rays), hits, "SrHit")
    pl = gsc.shade_closest_hit(hits_t, len(rays)).cpu().numpy().view(np.uint32).reshape(-1).view(abi.RAY_PAYLOAD)
    want = osc.shade_closest_hit(hits)
    assert_bits_equal(want, pl, "textured RayPayload")
    hit = hits["tri"] != 0xFFFFFFFF
    assert hit.mean() > 0.9 and len(np.unique(pl["albedo_packed"][hit])) > 2000   # textures really vary the payload
    assert len(np.unique(pl["material_info"][hit])) > 500


@pytest.mark.parametrize("instancing", ["flat", "two_level"])
def test_any_hit_alpha_test_equals_oracle(rt, oracle, instancing):
    """K5, any_hit.slang:11-43, on both sides (and in both forms of the structure: the two-level one finds a hit's shade record
    through the instance's mesh). The reference never runs the shader (OPAQUE geometry, blas.rs:276; alpha_mode
    forced 0, material.rs:74), so the materials are given MASK modes by hand: textured with a varying alpha channel, an RGB
    image (alpha widened to 0, utils.rs:27-43), untextured (the base-colour factor's alpha is the fallback), and
    alpha_mode == 0 (returns before sampling)."""
    desc = small_atrium()
    img = desc.images[0].copy()                                  # marble, RGBA: alpha becomes a diagonal ramp with noise
    hh, ww = img.shape[:2]
    jj, ii = np.meshgrid(np.arange(ww), np.arange(hh))
    img[..., 3] = ((jj * 3 + ii * 5 + (img[..., 0].astype(np.int64) % 7) * 9) % 256).astype(np.uint8)
    desc.images[0] = img
    for k, m in enumerate(desc.meshes):
        m.material = m.material.copy()
        m.material["alpha_mode"] = [1, 2, 0, 1][k % 4]           # MASK / BLEND are both "!= 0" to the shader
        m.material["alpha_cutoff"] = [0.5, 0.25, 0.9, 1.5, 0.0][k % 5]
        if k % 3 == 1:
            bc = m.material["base_color_value"].copy(); bc[..., 3] = 0.3; m.material["base_color_value"] = bc
    osc = oracle.OracleScene().load(desc)
    gsc = rt.Scene(0, instancing=instancing).load(desc)
    assert gsc.two_level() == (instancing == "two_level")
    rays = np.concatenate([camera_rays(oracle, desc, 200, 120), random_rays(30000, 35, box=((-8, 0.1, -17), (8, 6.5, 17)))])
    hits_t = gsc.trace_closest(rt.rays_to_device(rays), len(rays))
    hits = rt.hits_from_device(hits_t)
    assert_bits_equal(osc.trace_closest(rays), hits, "SrHit")
    got = gsc.any_hit_ignores(hits_t, len(rays)).cpu().numpy().view(np.uint32)
    want = osc.any_hit_ignores(hits)
    assert np.array_equal(want, got)
    hit = hits["tri"] != 0xFFFFFFFF
    assert 0.1 < got[hit].mean() < 0.9 and not got[~hit].any()      # both outcomes occur; misses are never "ignored"
    # the traversal itself stays opaque: any-hit queries report the same occlusion as before the materials changed
    plain = rt.Scene(0).load(small_atrium())
    rd = rt.rays_to_device(rays)
    assert np.array_equal(plain.trace_any(rd, len(rays)).cpu().numpy(), gsc.trace_any(rd, len(rays)).cpu().numpy())


def test_texture_error_behaviour(rt):
    from sunray_amd._lib import SunrayError
    g = rt.Scene(0)
    v, i = scenes.grid_patch((-1, 0, 1), (2, 0, 0), (0, 0, -2), 1, 1, (0, 1, 0), (1, 0, 0))
    with pytest.raises(SunrayError) as e:    # slot never added
        g.add_mesh(1, v, i, abi.material(textures={"base_color": (0, 0)}))
    assert e.value.code == -1 and "slot" in e.value.description
    img = g.add_image(np.zeros((4, 4, 3), dtype=np.uint8))
    with pytest.raises(SunrayError):         # image exists, sampler does not
        g.add_mesh(1, v, i, abi.material(textures={"base_color": (img, 0)}))
    with pytest.raises(SunrayError):
        g.add_sampler(0, 5, 0, 0)
    with pytest.raises(SunrayError):
        g.add_sampler(0, 1, 3, 0)            # CLAMP_TO_BORDER and beyond are not produced by scene.rs:255-262
    with pytest.raises(SunrayError):
        g.add_image(np.zeros((4, 4, 5), dtype=np.uint8))
    smp = g.add_sampler(1, 1, 0, 2)
    assert g.add_mesh(1, v, i, abi.material(textures={"base_color": (img, smp), "occlusion": (img, smp)})) == 0


def test_degenerate_and_coincident_geometry(rt, oracle, blue_noise):
    """Zero-area and collinear triangles never hit (det = 0 -> inf/NaN -> all comparisons false), exactly coincident
    triangles resolve to the lowest global index, needle triangles and far-away / huge ones stay consistent with the
    oracle's brute force — and a frame over such a scene is still bit-exact."""
    desc = scenes.cornell_box()
    v = scenes.make_vertices(np.array([[0, 1, 0], [0, 1, 0], [0, 1, 0],                 # a point
                                       [-0.5, 0.5, 0.2], [0.0, 0.5, 0.2], [0.5, 0.5, 0.2],   # collinear
                                       [-0.3, 0.8, 0.5], [0.3, 0.8, 0.5], [0.0, 1.3, 0.5],   # a real triangle ...
                                       [-0.3, 0.8, 0.5], [0.3, 0.8, 0.5], [0.0, 1.3, 0.5],   # ... and its exact copy
                                       [0.0, 0.2, 0.6], [1e-7, 0.2, 0.6], [0.0, 1.5, 0.6],   # needle
                                       [-4e5, -1, -4e5], [4e5, -1, -4e5], [0, -1, 5e5]], dtype=np.float32),   # huge ground triangle
                             np.tile(np.array([[0, 0, 1]], dtype=np.float32), (18, 1)))
    desc.meshes.append(scenes.MeshDesc(50, v, np.arange(18, dtype=np.uint32), abi.material(base_color=(0.3, 0.7, 0.3, 1.0), roughness=0.6)))
    desc.instances.append((50, [abi.IDENTITY_TRANSFORM.copy(), scenes.translate(0.0, 0.0, -0.2)]))
    osc = oracle.OracleScene().load(desc)
    osc.set_brute_force(True)
    gsc = rt.Scene(0).load(desc)
    rays = np.concatenate([camera_rays(oracle, desc, 200, 200), random_rays(20000, 91),
                           make_rays([(0.0, 1.0, 2.0)] * 3, [(0, 0, -1), (0.0, 0.05, -1.0), (1e-4, 0.0, -1.0)])])
    rays_t = rt.rays_to_device(rays)
    hits = rt.hits_from_device(gsc.trace_closest(rays_t, len(rays)))
    want = osc.trace_closest(rays)
    assert_bits_equal(want, hits, "SrHit with degenerate / coincident triangles")
    # the coincident pair: hits report the first copy (lower global index) of instance 0, never the second
    n_before = sum(len(m.indices) // 3 * len(x) for m, (k, x) in zip(desc.meshes[:-1], desc.instances[:-1]))
    tri_hit = hits["tri"][hits["tri"] != 0xFFFFFFFF]
    assert (tri_hit == n_before + 2).any() and not (tri_hit == n_before + 3).any()
    assert not np.isin(tri_hit, [n_before + 0, n_before + 1]).any()          # point and collinear triangles are never hit
    assert np.array_equal(osc.trace_any(rays), gsc.trace_any(rays_t, len(rays)).cpu().numpy().view(np.uint32))
    osc.set_brute_force(False)
    run_both(rt, oracle, desc, 96, 96, 2, blue_noise)


# ---- the two passes -----------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(make_golden.CASES))
def test_passes_match_committed_golden(rt, name, blue_noise):
    fn, W, H, frames, over = make_golden.CASES[name]
    want = np.load(os.path.join(GOLDEN, "pass_%s.npz" % name))
    desc = fn()
    cfg = make_golden.make_config(over)
    gsc = rt.Scene(0).load(desc)
    gf = rt.DeviceFrame(W, H, blue_noise)
    prev = None
    for f in range(frames):
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        if cfg.enable_restir:
            gsc.trace_ris(gf, m, f, cfg)
        gsc.trace_final(gf, m, f, cfg)
        h = gf.host()
        cur = f & 1
        assert_bits_equal(want["f%d_raw_color" % f], h["raw_color"], "raw_color")
        if cfg.enable_restir:
            for key, arr in (("depth", h["depth"]), ("normal", h["normal"]), ("diffuse", h["diffuse"]), ("motion", h["motion"]),
                             ("reservoir", h["reservoirs"][cur]), ("reservoir_gi", h["reservoirs_gi"][cur])):
                assert_bits_equal(want["f%d_%s" % (f, key)], arr, key)


@pytest.mark.parametrize("scene_fn,W,H,frames", [
    (scenes.cornell_box, 256, 256, 4),          # BASELINE.json config 1 geometry
    (scenes.cornell_glass_mirror, 200, 152, 3), # ragged extent: not a multiple of the 16x16 tile
    (scenes.torus_knot, 320, 180, 2),           # config 2 stand-in, reduced extent
    (small_atrium, 240, 136, 3),                # config 4 stand-in (textured), reduced
    (lambda: scenes.heightfield(n=300), 320, 180, 2),
])
def test_passes_equal_oracle(rt, oracle, blue_noise, scene_fn, W, H, frames):
    run_both(rt, oracle, scene_fn(), W, H, frames, blue_noise)


@pytest.mark.parametrize("scene_fn,W,H,frames", [
    (scenes.cornell_glass_mirror, 200, 152, 3),
    (small_atrium, 240, 136, 2),
    (lambda: scenes.heightfield(n=300), 320, 180, 2),
])
def test_passes_without_the_primary_hit_hand_off(rt, oracle, blue_noise, scene_fn, W, H, frames):
    """SrRtParams.primary_payload == NULL: the final pass traces its camera ray itself, as the reference does
    (ray_gen_final.slang:80). Same bits as with the hand-off; the ray counters then hold every query of the reference."""
    cfg = abi.SrTraceConfig.reference()
    cfg.flags = abi.TRACE_FLAG_TRACE_EVERY_QUERY          # ... and every visibility query, also the repeated ones
    osc, gsc, of, gf = run_both(rt, oracle, scene_fn(), W, H, frames, blue_noise, cfg=cfg, primary=False)
    c, oc = gsc.counters(), osc.counters()
    assert c.reused_primary_hits == 0 and c.closest_queries >= 2 * W * H
    assert (c.closest_queries, c.any_queries, c.reused_visibility_queries) == (oc.closest_queries, oc.any_queries, 0)


def test_primary_hit_hand_off_holds_the_camera_ray_payload(rt, oracle, blue_noise):
    """What sr_trace_ris leaves in SrRtParams.primary_payload is the RayPayload of the camera ray — closest_hit / miss
    applied to ray_gen_ris.slang:75 at virtual bounce 0 — for every pixel of the launch: compared with the payload the
    oracle's RIS pass holds at that point, on a scene with glass, mirror and sky pixels, a textured one, over two frames."""
    for desc, W, H in ((scenes.cornell_glass_mirror(), 96, 64), (small_atrium(), 120, 72)):
        for frames in (1, 2):
            osc, gsc, of, gf = run_both(rt, oracle, desc, W, H, frames, blue_noise)
            got = gf.primary.cpu().numpy().view(np.uint32).reshape(-1).view(abi.RAY_PAYLOAD)
            assert_bits_equal(of.primary, got, "primary payload")


def test_repeated_gi_visibility_query_is_answered_once(rt, oracle, blue_noise):
    """ray_gen_final.slang:304-316 traces hitPos -> combined.sample_pos again although :276-286 traced exactly that segment when
    the sample came from a neighbour. The final pass answers the repeat without a traversal (it is counted apart); with
    SR_TRACE_FLAG_TRACE_EVERY_QUERY it traverses. Both equal the oracle, which always traverses — so the two are the same query."""
    desc = small_atrium()
    osc, gsc, of, gf = run_both(rt, oracle, desc, 120, 72, 3, blue_noise)
    c = gsc.counters()
    assert c.reused_visibility_queries > 0 and ref_any(c) == osc.counters().any_queries
    every = abi.SrTraceConfig.reference(); every.flags = abi.TRACE_FLAG_TRACE_EVERY_QUERY
    osc2, gsc2, of2, gf2 = run_both(rt, oracle, desc, 120, 72, 3, blue_noise, cfg=every)
    c2 = gsc2.counters()
    assert c2.reused_visibility_queries == 0 and c2.any_queries == c.any_queries + c.reused_visibility_queries
    assert_bits_equal(gf.host()["raw_color"], gf2.host()["raw_color"], "raw_color with / without the repeated query")


@pytest.mark.parametrize("W,H", [(5, 3), (8, 8), (130, 17), (1000, 9), (24, 300), (129, 129)])
def test_passes_odd_extents_through_tile_schedule_updates(rt, oracle, blue_noise, W, H):
    """Fewer tile columns than XCD bands, one tile row, tall and narrow: the launch geometry and the cost-derived tile
    schedule (bands of equal cost, sweep direction; re-derived after each of the first launches) must cover every pixel
    exactly once — five consecutive frames, every buffer bit-exact."""
    run_both(rt, oracle, scenes.cornell_box(), W, H, 5, blue_noise)


def test_passes_without_restir_and_bounce_knobs(rt, oracle, blue_noise):
    cfg = abi.SrTraceConfig.reference()
    cfg.enable_restir, cfg.max_bounces, cfg.shadow_bounces = 0, 2, 2   # BASELINE.json config 3 settings
    run_both(rt, oracle, scenes.torus_knot(), 256, 144, 2, blue_noise, cfg)
    cfg.max_bounces = cfg.shadow_bounces = 1                           # config 2: primary + one NEE shadow ray
    run_both(rt, oracle, scenes.cornell_glass_mirror(), 128, 128, 2, blue_noise, cfg)


def test_moving_camera_temporal_reprojection(rt, oracle, blue_noise):
    desc = scenes.cornell_box()
    W = H = 128
    osc, gsc = oracle.OracleScene().load(desc), rt.Scene(0).load(desc)
    of, gf = oracle.HostFrame(W, H, blue_noise), rt.DeviceFrame(W, H, blue_noise)
    prev = None
    for f in range(4):
        pos = (0.15 * f, 1.0 + 0.05 * f, 3.4 - 0.1 * f)
        om = oracle.camera_matrices(pos, desc.camera_target, desc.fov_y, W, H, prev)
        gm = rt.camera_matrices(pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(om.view_proj)
        osc.trace_ris(of, om, f); osc.trace_final(of, om, f)
        gsc.trace_ris(gf, gm, f); gsc.trace_final(gf, gm, f)
        h = gf.host()
        assert_bits_equal(of.motion, h["motion"], "motion f%d" % f)
        assert_bits_equal(of.reservoirs[f & 1], h["reservoirs"][f & 1], "reservoir f%d" % f)
        assert_bits_equal(of.raw_color, h["raw_color"], "raw_color f%d" % f)


def test_row_tiles_compose_to_full_frame(rt, blue_noise):
    desc = scenes.cornell_box()
    W, H = 200, 150
    gsc = rt.Scene(0).load(desc)
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
    full, tiled = rt.DeviceFrame(W, H, blue_noise), rt.DeviceFrame(W, H, blue_noise)
    gsc.trace_ris(full, m, 0); gsc.trace_final(full, m, 0)
    bands = ((0, 40), (40, 37), (77, 73))
    for b in bands:
        gsc.trace_ris(tiled, m, 0, tile=b)
    for b in bands[::-1]:
        gsc.trace_final(tiled, m, 0, tile=b)
    a, b = full.host(), tiled.host()
    assert_bits_equal(a["raw_color"], b["raw_color"], "tiled raw_color")
    assert_bits_equal(a["reservoirs"][0], b["reservoirs"][0], "tiled reservoirs")


def test_frames_in_flight_equal_sequential_frames(rt, blue_noise):
    """distributed.FramePipeline: RIS of frame f+1 on its own stream while the final pass of frame f runs, two G-buffer sets
    sharing the reservoir ping-pong — the frames are those of sequential execution, bit for bit (moving camera over sky and
    geometry, so temporal reuse reads real history, including the stale GI reservoirs sky pixels leave behind)."""
    import torch
    from sunray_amd import distributed as sd
    desc = scenes.cornell_glass_mirror()
    W, H, frames = 200, 152, 7
    cfg = abi.SrTraceConfig.reference()
    part = sd.Partition(W, H, 1)
    gsc = rt.Scene(0).load(desc)
    mats, prev = [], None
    for f in range(frames):
        m = rt.camera_matrices((desc.camera_pos[0] + 0.03 * f, desc.camera_pos[1], desc.camera_pos[2]), desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        mats.append(m)
    seq = rt.DeviceFrame(W, H, blue_noise)
    want = []
    for f in range(frames):
        gsc.trace_ris(seq, mats[f], f, cfg); gsc.trace_final(seq, mats[f], f, cfg)
        want.append(seq.raw_color.cpu().numpy().copy())
    fp = sd.FramePipeline(rt.DeviceFrame(W, H, blue_noise), rt.DeviceFrame(W, H, blue_noise))
    got = []
    for f in range(frames):
        fp.step(gsc, mats[f], f, cfg, part, 0, after_final=lambda fr: got.append(fr.raw_color.clone()))   # cloned on the final stream, in order
    torch.cuda.synchronize()
    for f in range(frames):
        assert_bits_equal(want[f], got[f].cpu().numpy(), "frame %d with two frames in flight" % f)


def test_empty_scene_renders_sky(rt, blue_noise):
    g = rt.Scene(0); g.set_instances([])
    W, H = 64, 48
    gf = rt.DeviceFrame(W, H, blue_noise)
    m = rt.camera_matrices((0, 0, 1), (0, 0, 0), 45.0, W, H)
    g.trace_ris(gf, m, 0); g.trace_final(gf, m, 0)
    h = gf.host()
    assert (h["depth"] == 0x7C00).all() and not h["raw_color"][:, :3].any() and (h["raw_color"][:, 3] == 1).all()


def test_error_behaviour(rt, blue_noise):
    from sunray_amd._lib import SunrayError
    g = rt.Scene(0)
    v, i = scenes.quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1), (0, 1, 0))
    g.add_mesh(1, v, i, abi.material())
    gf = rt.DeviceFrame(16, 16, blue_noise)
    m = rt.camera_matrices((0, 1, 3), (0, 0, 0), 45.0, 16, 16)
    with pytest.raises(SunrayError) as e:      # trace before the TLAS exists
        g.trace_ris(gf, m, 0)
    assert e.value.code == -4
    with pytest.raises(SunrayError) as e:      # duplicate key (lib.rs:880-884)
        g.add_mesh(1, v, i, abi.material())
    assert e.value.code == -1 and "already registered" in e.value.description
    with pytest.raises(SunrayError) as e:      # not a triangle list (lib.rs:885-891)
        g.add_mesh(2, v, i[:4], abi.material())
    assert "invalid mesh" in e.value.description
    with pytest.raises(SunrayError) as e:      # index out of range (lib.rs:892-899)
        g.add_mesh(3, v, np.array([0, 1, 9], np.uint32), abi.material())
    assert "out of range" in e.value.description
    with pytest.raises(SunrayError) as e:      # unknown key (resource_manager.rs:227-231)
        g.set_instances([(42, [abi.IDENTITY_TRANSFORM])])
    assert "never loaded" in e.value.description
    tex = abi.material(); tex["base_color_image"] = 3
    with pytest.raises(SunrayError) as e:
        g.add_mesh(4, v, i, tex)
    assert e.value.code == -1 and "never added" in e.value.description   # dangling texture slot
    # non-finite input never reaches the builders (their quantiser would be undefined on it)
    bad = v.copy(); bad["position"][2, 1] = np.nan
    with pytest.raises(SunrayError) as e:
        g.add_mesh(5, bad, i, abi.material())
    assert "non-finite position" in e.value.description
    nan_xf = np.array(abi.IDENTITY_TRANSFORM, dtype=np.float32).copy(); nan_xf.reshape(-1)[7] = np.inf
    with pytest.raises(SunrayError) as e:
        g.set_instances([(1, [nan_xf])])
    assert "non-finite" in e.value.description
    import ctypes as C
    from sunray_amd._lib import lib, check
    keys, counts = (C.c_uint64 * 1)(1), (C.c_uint32 * 1)(2)
    with pytest.raises(SunrayError) as e:      # instances announced, no transforms
        check(lib().sr_scene_set_instances(g._h, keys, counts, C.c_uint32(1), None))
    assert "transforms is null" in e.value.description
    counts[0] = 0xFFFFFFFF                     # 2^32 - 1 instances of a 2-triangle mesh: the 32-bit triangle offset would wrap
    with pytest.raises(SunrayError) as e:
        check(lib().sr_scene_set_instances(g._h, keys, counts, C.c_uint32(1), C.c_void_p(8)))
    assert e.value.code == -5 and "2^32 - 1 triangles" in e.value.description
    g.set_instancing("flat")                   # the one-level form stops at 2^28 triangles (leaf reference encoding); the two-level form takes over in auto mode
    counts[0] = 1 << 27
    with pytest.raises(SunrayError) as e:
        check(lib().sr_scene_set_instances(g._h, keys, counts, C.c_uint32(1), C.c_void_p(8)))
    assert e.value.code == -5 and "2^28 triangles" in e.value.description
    g.set_instancing("auto")
    g.set_instances([(1, [abi.IDENTITY_TRANSFORM])])          # the scene is still usable
    rays = rt.rays_to_device(random_rays(64, 1))
    with pytest.raises(SunrayError) as e:
        check(lib().sr_trace_closest(g._h, C.c_void_p(rays.data_ptr()), C.c_uint32(0x90000000), C.c_void_p(rays.data_ptr()), None))
    assert "2^31 rays" in e.value.description
    # the passes count a pixel's queries in packed fields: bounce limits above SR_MAX_BOUNCES are refused, not wrapped
    frame = rt.DeviceFrame(16, 16, scenes.white_noise_rgba8())
    m = rt.camera_matrices((0, 0, 3), (0, 0, 0), 60.0, 16, 16)
    cfg = abi.SrTraceConfig.reference(); cfg.max_bounces = abi.MAX_BOUNCES + 1
    with pytest.raises(SunrayError) as e:
        g.trace_final(frame, m, 0, cfg)
    assert e.value.code == -1 and "SR_MAX_BOUNCES" in e.value.description
    cfg = abi.SrTraceConfig.reference(); cfg.virtual_bounces = abi.MAX_BOUNCES + 1
    with pytest.raises(SunrayError) as e:
        g.trace_ris(frame, m, 0, cfg)
    assert "SR_MAX_BOUNCES" in e.value.description


# ---- post-RT compute chain --------------------------------------------------------
@pytest.mark.parametrize("name", list(make_golden.POST_CASES))
def test_post_chain_matches_committed_golden(rt, name, blue_noise):
    fn, W, H, frames = make_golden.POST_CASES[name]
    want = np.load(os.path.join(GOLDEN, "pass_%s.npz" % name))
    desc = fn()
    gsc = rt.Scene(0).load(desc)
    gf = rt.DeviceFrame(W, H, blue_noise)
    prev = None
    for f in range(frames):
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        gsc.trace_ris(gf, m, f); gsc.trace_final(gf, m, f)
        rt.post_chain(gf, f)
        if f >= frames - 3:
            h = gf.host()
            assert_bits_equal(want["f%d_accum" % f], h["accum"][f % 2], "accum f%d" % f)
            assert_bits_equal(want["f%d_denoise" % f], h["denoise"][1], "denoise f%d" % f)
            assert_bits_equal(want["f%d_output" % f], h["output"], "RGBA8 output f%d" % f)


@pytest.mark.parametrize("scene_fn,W,H,frames", [
    (scenes.cornell_box, 200, 152, 6),                      # ragged extent; temporal history active from frame 3
    (lambda: scenes.heightfield(n=300), 320, 180, 5),
])
def test_full_frames_with_post_chain_equal_oracle(rt, oracle, blue_noise, scene_fn, W, H, frames):
    desc = scene_fn()
    osc, gsc = oracle.OracleScene().load(desc), rt.Scene(0).load(desc)
    of, gf = oracle.HostFrame(W, H, blue_noise), rt.DeviceFrame(W, H, blue_noise)
    prev = None
    for f in range(frames):
        pos = (desc.camera_pos[0] + 0.02 * f, desc.camera_pos[1], desc.camera_pos[2])   # slow dolly: real reprojection
        om = oracle.camera_matrices(pos, desc.camera_target, desc.fov_y, W, H, prev)
        gm = rt.camera_matrices(pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(om.view_proj)
        osc.trace_ris(of, om, f); osc.trace_final(of, om, f); oracle.post_chain(of, f)
        gsc.trace_ris(gf, gm, f); gsc.trace_final(gf, gm, f); rt.post_chain(gf, f)
        h = gf.host()
        assert_bits_equal(of.accum[f % 2], h["accum"][f % 2], "accum f%d" % f)
        assert_bits_equal(of.denoise[0], h["denoise"][0], "denoise a f%d" % f)
        assert_bits_equal(of.denoise[1], h["denoise"][1], "denoise b f%d" % f)
        assert_bits_equal(of.output, h["output"], "RGBA8 output f%d" % f)
    rgba = h["output"].view(np.uint8).reshape(-1, 4)
    assert (rgba[:, 3] == 255).all() and rgba[:, :3].any()


def test_post_chain_error_behaviour(rt, blue_noise):
    import ctypes as C
    from sunray_amd._lib import lib
    gf = rt.DeviceFrame(16, 16, blue_noise)
    p = abi.post_params(gf, 0, lambda t: t.data_ptr())
    p.accum[1] = None
    assert lib().sr_post_temporal(C.byref(p), None) == -1 and b"ping-pong" in lib().sr_last_error()
    p = abi.post_params(gf, 0, lambda t: t.data_ptr())
    p.denoise_passes = 0
    assert lib().sr_post_denoise(C.byref(p), None) == -1
    p = abi.post_params(gf, 0, lambda t: t.data_ptr())
    p.output_rgba8 = None
    assert lib().sr_post_tonemap(C.byref(p), None) == -1


# ---- Renderer facade --------------------------------------------------------------
def _oracle_render_to_host_memory(oracle, desc, W, H, instances_per_frame, noise, first_frame=0, of=None, prev=None):
    """The reference's render loop restated with the oracle: per frame matrices (prev_view_proj injected),
    ris, final, temporal, denoise x4, tonemap."""
    osc = oracle.OracleScene()
    for m in desc.meshes:
        osc.add_mesh(m.key, m.vertices, m.indices, m.material)
    of = of or oracle.HostFrame(W, H, noise)
    last = None
    for i, inst in enumerate(instances_per_frame):
        f = first_frame + i
        if inst is not last:
            osc.set_instances(inst); last = inst
        om = oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(om.view_proj)
        osc.trace_ris(of, om, f); osc.trace_final(of, om, f); oracle.post_chain(of, f)
    return of, prev


def test_renderer_render_to_host_memory_equals_oracle(rt, oracle):
    desc = scenes.cornell_box()
    W, H = 96, 80
    noise = rt.default_noise_texture()
    r = rt.Renderer((W, H))
    for m in desc.meshes:
        r.load_mesh(m.key, m.vertices, m.indices, m.material)
    cam = (desc.camera_pos, desc.camera_target, desc.fov_y)
    img = r.render_to_host_memory(cam, desc.instances)
    assert img.shape == (H, W, 4) and r.relative_frame_count == 16          # WARMUP_FRAMES (lib.rs:1909)
    of, prev = _oracle_render_to_host_memory(oracle, desc, W, H, [desc.instances] * 16, noise)
    assert_bits_equal(of.output, img.view(np.uint32).reshape(-1), "render_to_host_memory")
    assert (img[..., 3] == 255).all() and img[..., :3].any()
    # render + wait_frame continue the same history: frame 17 equals the oracle's 17th frame
    fr = r.render(cam, desc.instances)
    assert fr == 17
    r.wait_frame(fr)
    with pytest.raises(rt.SunrayError):
        r.wait_frame(fr + 5)
    r.close()


def test_renderer_with_a_host_supplied_noise_texture(rt, oracle):
    """lib.rs:281-309: the reference embeds a 16-bit greyscale PNG and uploads its to_rgba8() form. A host that owns that asset
    decodes it (sr_decode_image_rgba8) and hands it to the renderer; here a synthetic 16-bit PNG of another extent stands in.
    The texture feeds the first BRDF bounce (ray_gen_final.slang:44-50,393-396). With ReSTIR on, every first rough hit ends the
    walk (:327), so that branch is reached after specular chains only; with enable_restir = 0 every rough first hit bounces
    with the texture's numbers — there two different textures MUST give different images, and each must equal the oracle's."""
    import struct
    import zlib
    rng = np.random.default_rng(77)
    v = rng.integers(0, 65536, size=(48, 64), dtype=np.uint16)

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 64, 48, 16, 0, 0, 0, 0)) +
           chunk(b"IDAT", zlib.compress(b"".join(b"\x00" + v[y].astype(">u2").tobytes() for y in range(48)))) + chunk(b"IEND", b""))
    noise = rt.decode_image_rgba8(png)
    assert noise.shape == (48, 64, 4) and (noise[..., 0] == ((v.astype(np.uint32) + 128) // 257)).all() and (noise[..., 3] == 255).all()
    desc = scenes.cornell_glass_mirror()
    W, H = 72, 56
    cam = (desc.camera_pos, desc.camera_target, desc.fov_y)
    images = []
    for restir in (1, 0):
        cfg = abi.SrTraceConfig.reference()
        cfg.enable_restir = restir
        for tex in (noise, rt.default_noise_texture()):
            r = rt.Renderer((W, H))
            r.set_config(cfg)
            r.set_blue_noise(tex)
            for m in desc.meshes:
                r.load_mesh(m.key, m.vertices, m.indices, m.material)
            img = r.render_to_host_memory(cam, desc.instances)
            osc = oracle.OracleScene().load(desc)
            of, prev = oracle.HostFrame(W, H, tex), None
            for f in range(16):
                om = oracle.camera_matrices(cam[0], cam[1], cam[2], W, H, prev)
                prev = list(om.view_proj)
                if restir:
                    osc.trace_ris(of, om, f, cfg)
                osc.trace_final(of, om, f, cfg)
                oracle.post_chain(of, f)
            assert_bits_equal(of.output, img.view(np.uint32).reshape(-1), "render_to_host_memory with a host-supplied noise texture (restir %d)" % restir)
            images.append(img)
            if restir == 0:
                with pytest.raises(rt.SunrayError):
                    r.set_blue_noise(np.zeros((0, 4, 4), dtype=np.uint8))
            r.close()
    assert (images[2] != images[3]).any(), "the noise texture does not reach the image"


REF_ASSET_DIR = os.path.join(GOLDEN, "ref_assets")     # the reference's example rooms and its blue-noise PNG (data files)
PNG_EXAMPLE_CAMERA = ((13.0, 30.0, 25.0), (0.0, 13.0, 0.0), 45.0)      # examples/png/main.rs:52-55


def _render_ref_asset(rt, oracle, name, W, H, frames, noise):
    """Renderer::load_gltf + `frames` x (render, wait_frame) of one of the reference's example rooms against the oracle's
    restatement of the same loop; returns (gpu RGBA8 image, seconds per frame on the GPU for a second, pipelined run)."""
    import time
    import torch
    path = os.path.join(REF_ASSET_DIR, name)
    r = rt.Renderer((W, H))
    r.set_blue_noise(noise)
    group, inst = r.load_gltf(path)
    osc, grouped, keys, n_img = _oracle_scene_from_gltf(oracle, path, 0, 0, {})
    assert [k for k, _ in inst] == keys and n_img == 0
    if frames == 16:
        img = r.render_to_host_memory(PNG_EXAMPLE_CAMERA, inst)
    else:
        for _ in range(frames):
            fr = r.render(PNG_EXAMPLE_CAMERA, inst)
        r.wait_frame(fr)
        import ctypes as C
        from sunray_amd._lib import lib
        outp = C.c_void_p()
        assert lib().sr_renderer_get(r._h, None, C.byref(outp), None, None) == 0
        img = np.zeros((H, W, 4), dtype=np.uint8)
        assert C.CDLL("libamdhip64.so").hipMemcpy(img.ctypes.data_as(C.c_void_p), outp, C.c_size_t(img.nbytes), C.c_int(2)) == 0
    osc.set_instances(grouped)
    of, _ = _oracle_frames(oracle, osc, PNG_EXAMPLE_CAMERA, W, H, frames, noise)
    assert_bits_equal(of.output, img.view(np.uint32).reshape(-1), "%s %dx%d, %d frames" % (name, W, H, frames))
    # frame time of the same workload with two frames in flight (not part of the parity claim)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(32):
        fr = r.render(PNG_EXAMPLE_CAMERA, inst)
    r.wait_frame(fr)
    dt = (time.perf_counter() - t0) / 32
    r.close()
    return img, dt


def test_reference_png_example_workload(rt, oracle):
    """The reference's own offline workload (examples/png/main.rs:43-61, src/lib.rs:1908-1934): ReflectionRoom.glb (1 986
    triangles, one emissive material of strength 61.6, one transmissive), 1600x1200, the crate's blue-noise texture
    (lib.rs:281-284: noise.png, 128x128 16-bit grey, through to_rgba8), camera (13, 30, 25) -> (0, 13, 0), fov 45,
    render_to_host_memory = 16 frames. Every byte of the RGBA8 image against the oracle's restatement of that loop."""
    noise = rt.decode_image_rgba8(open(os.path.join(REF_ASSET_DIR, "noise.png"), "rb").read())
    assert noise.shape == (128, 128, 4)
    img, dt = _render_ref_asset(rt, oracle, "ReflectionRoom.glb", 1600, 1200, 16, noise)
    assert (img[..., 3] == 255).all() and len(np.unique(img.reshape(-1, 4), axis=0)) > 2000
    print("ReflectionRoom.glb 1600x1200: %.3f ms per frame (whole frame incl. post chain, two in flight)" % (dt * 1e3))


@pytest.mark.parametrize("name", ["ReflectionRoom3.glb", "Room.glb", "Room2.glb", "Room3.glb"])
def test_reference_example_rooms(rt, oracle, name):
    """The other four example rooms of the reference at 400x300, two frames (temporal reuse active), same camera and noise."""
    noise = rt.decode_image_rgba8(open(os.path.join(REF_ASSET_DIR, "noise.png"), "rb").read())
    img, dt = _render_ref_asset(rt, oracle, name, 400, 300, 2, noise)
    assert img[..., :3].any()


def test_renderer_resize_and_instance_change(rt, oracle):
    desc = scenes.cornell_box()
    noise = rt.default_noise_texture()
    r = rt.Renderer((64, 48))
    for m in desc.meshes:
        r.load_mesh(m.key, m.vertices, m.indices, m.material)
    cam = (desc.camera_pos, desc.camera_target, desc.fov_y)
    for _ in range(3):
        r.wait_frame(r.render(cam, desc.instances))
    assert r.relative_frame_count == 3
    r.resize((64, 48))
    assert r.relative_frame_count == 3                                      # same extent: no-op (lib.rs:598-600)
    r.resize((80, 56))
    assert r.relative_frame_count == 0 and r.size == (80, 56)
    # moved instances from frame 2 on: the acceleration structure must follow the caller's list
    moved = [(k, [np.asarray(t, dtype=np.float32).reshape(3, 4) + np.float32(0.05) * np.eye(3, 4, 3, dtype=np.float32) for t in ts])
             for k, ts in desc.instances]
    per_frame = [desc.instances, desc.instances, moved, moved]
    for inst in per_frame:
        last = r.render(cam, inst)
    r.wait_frame(last)
    # oracle: history buffers are fresh after the resize, but prev_view_proj survives it (lib.rs:586-639
    # does not touch it), so frame 0 after the resize sees the old extent's view_proj
    prev = None
    for _ in range(3):
        prev = list(oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, 64, 48, prev).view_proj)
    of, _ = _oracle_render_to_host_memory(oracle, desc, 80, 56, per_frame, noise, prev=prev)
    import ctypes as C
    from sunray_amd._lib import lib
    import torch
    outp = C.c_void_p()
    assert lib().sr_renderer_get(r._h, None, C.byref(outp), None, None) == 0
    got = np.zeros(80 * 56, dtype=np.uint32)
    assert torch.cuda.current_device() == 0
    torch.cuda.synchronize()
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), outp, C.c_size_t(got.nbytes), C.c_int(2)) == 0
    assert_bits_equal(of.output, got, "output after resize + moved instances")
    r.close()


def test_renderer_frame_and_resize_callbacks(rt):
    """add_start_of_frame_callback / add_end_of_frame_callback / add_resize_callback (lib.rs:537-594): a start-of-frame
    callback runs once at the start of the next render; an end-of-frame callback only once that frame has completed on
    the GPU, drained at the start of a later render; resize callbacks are persistent and see every resize call."""
    desc = scenes.cornell_box()
    r = rt.Renderer((64, 48))
    for m in desc.meshes:
        r.load_mesh(m.key, m.vertices, m.indices, m.material)
    cam = (desc.camera_pos, desc.camera_target, desc.fov_y)
    log = []
    r.add_start_of_frame_callback(lambda: log.append(("start", r.relative_frame_count)))
    r.add_end_of_frame_callback(lambda: log.append(("end", r.relative_frame_count)))
    r.add_resize_callback(lambda size: log.append(("resize", size)))
    assert log == []
    f1 = r.render(cam, desc.instances)                       # frame 1: the start callback fires before any frame work
    assert log == [("start", 0)]
    r.wait_frame(f1)
    assert log == [("start", 0)]                             # completion alone does not run it: the reference drains in render()
    f2 = r.render(cam, desc.instances)
    assert log == [("start", 0), ("end", 1)]                 # frame 1 is complete -> drained at the start of frame 2, once
    r.wait_frame(f2)
    r.wait_frame(r.render(cam, desc.instances))
    assert log == [("start", 0), ("end", 1)]
    # a callback registered from inside a callback is tagged absolute_frame_count + 1 like any other (lib.rs:541) — the frame
    # being started — so the same drain loop picks it up (lib.rs:558-568 re-reads the list's length every iteration)
    r.add_start_of_frame_callback(lambda: r.add_start_of_frame_callback(lambda: log.append(("chained", r.relative_frame_count))))
    r.wait_frame(r.render(cam, desc.instances)); assert log[-1] == ("chained", 3)
    r.wait_frame(r.render(cam, desc.instances)); assert log.count(("chained", 3)) == 1 and len(log) == 3
    r.resize((64, 48)); r.resize((80, 56))
    assert log[-2:] == [("resize", (64, 48)), ("resize", (80, 56))] and r.relative_frame_count == 0
    with pytest.raises(rt.SunrayError):
        from sunray_amd._lib import lib, check
        check(lib().sr_renderer_add_resize_callback(r._h, None, None))
    r.close()


def test_load_unload_cycles_do_not_leak_hbm(rt, tmp_path):
    """Renderer::unload_scene frees every asset of the group — BLASes AND images (ResourceManager::remove,
    resource_manager.rs:459-472): loading and unloading a textured scene repeatedly leaves the free HBM where it was."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gltf_util
    desc = scenes.atrium(columns_per_side=4, col_segments=16, col_rings=4, floor_div=8, tex=512, n_lamps=6)   # ~6 MB of texels
    path = str(tmp_path / "atrium512.glb")
    gltf_util.scene_to_gltf(desc, path)
    r = rt.Renderer((64, 48))
    cam = (desc.camera_pos, desc.camera_target, desc.fov_y)
    free = []
    for cycle in range(6):
        group, inst = r.load_gltf(path)
        r.wait_frame(r.render(cam, inst))
        r.unload_scene(group)
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
    assert group == 5
    assert abs(free[-1] - free[1]) < (1 << 20), free          # after the first cycle (allocator warm-up) nothing accumulates
    # a freed image slot is not usable until re-added
    from sunray_amd._lib import lib, check
    import ctypes as C
    sc = C.c_void_p()
    check(lib().sr_renderer_get(r._h, C.byref(sc), None, None, None))
    with pytest.raises(rt.SunrayError):
        check(lib().sr_scene_remove_image(sc, C.c_uint32(0)))
    r.close()


def _oracle_scene_from_gltf(oracle, path, group, first_image_slot=0, sampler_slots=None, osc=None):
    from oracle import gltf_ref
    meshes, grouped, images, samplers = gltf_ref.GltfRef(path).loaded(group, first_image_slot, sampler_slots)
    osc = osc or oracle.OracleScene()
    for im in images:
        osc.add_image(im)
    for smp in samplers:
        osc.add_sampler(*smp)
    for key, v, i, m, et in meshes:
        osc.add_blas(key, v, i, m, et)
    return osc, grouped, [k for k, *_ in meshes], len(images)


def _oracle_frames(oracle, osc, cam, W, H, n, noise, of=None, prev=None, first_frame=0):
    of = of or oracle.HostFrame(W, H, noise)
    for f in range(first_frame, first_frame + n):
        om = oracle.camera_matrices(cam[0], cam[1], cam[2], W, H, prev)
        prev = list(om.view_proj)
        osc.trace_ris(of, om, f); osc.trace_final(of, om, f); oracle.post_chain(of, f)
    return of, prev


def test_renderer_load_gltf_unload_reload(rt, oracle, tmp_path):
    """Renderer::load_gltf -> render_to_host_memory equals the oracle fed by the numpy glTF restatement; after
    unload_scene + a second load (mesh-info and emissive slots reused LIFO, new image slots, keys of group 1) the
    renderer still equals the oracle that replays the same add/remove sequence."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gltf_util
    desc = small_atrium()
    path = str(tmp_path / "atrium.glb")
    gltf_util.scene_to_gltf(desc, path)
    W, H = 72, 48
    cam = (desc.camera_pos, desc.camera_target, desc.fov_y)
    noise = rt.default_noise_texture()
    r = rt.Renderer((W, H))
    group, inst = r.load_gltf(path)
    sampler_slots = {}
    osc, grouped, keys, n_img = _oracle_scene_from_gltf(oracle, path, 0, 0, sampler_slots)
    assert group == 0 and [k for k, _ in inst] == keys and keys[1] == 1
    for (k1, x1), (k2, x2) in zip(inst, grouped):
        assert len(x1) == len(x2) and all(a.tobytes() == b.tobytes() for a, b in zip(x1, x2))
    img = r.render_to_host_memory(cam, inst)
    osc.set_instances(grouped)
    of, prev = _oracle_frames(oracle, osc, cam, W, H, 16, noise)
    assert_bits_equal(of.output, img.view(np.uint32).reshape(-1), "glTF scene, 16 frames")
    assert len(np.unique(img.reshape(-1, 4), axis=0)) > 500
    # unload, reload: group 1
    r.unload_scene(group)
    with pytest.raises(rt.SunrayError):
        r.render(cam, inst)                                   # the old keys are gone: "never loaded"
    for k in keys:
        osc.remove(k)
    group2, inst2 = r.load_gltf(path)
    osc, grouped2, keys2, _ = _oracle_scene_from_gltf(oracle, path, 1, n_img, sampler_slots, osc)
    assert group2 == 1 and [k for k, _ in inst2] == keys2 and keys2[0] == 1 << 32
    osc.set_instances(grouped2)
    for _ in range(2):
        fr = r.render(cam, inst2)
    r.wait_frame(fr)
    of, prev = _oracle_frames(oracle, osc, cam, W, H, 2, noise, of, prev, first_frame=16)
    import ctypes as C
    from sunray_amd._lib import lib
    outp = C.c_void_p()
    assert lib().sr_renderer_get(r._h, None, C.byref(outp), None, None) == 0
    got = np.zeros(W * H, dtype=np.uint32)
    assert C.CDLL("libamdhip64.so").hipMemcpy(got.ctypes.data_as(C.c_void_p), outp, C.c_size_t(got.nbytes), C.c_int(2)) == 0
    assert_bits_equal(of.output, got, "after unload + reload")
    # unload_mesh of one key; rendering the rest still works
    r.unload_mesh(keys2[-1])
    r.wait_frame(r.render(cam, inst2[:-1]))
    r.close()


def test_png_example_end_to_end(rt, tmp_path):
    """examples/png.py (the reference's examples/png): glTF in, PNG out; the PNG decodes back to the renderer's bytes."""
    import subprocess
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gltf_util
    from oracle import gltf_ref
    desc = small_atrium()
    glb, png = str(tmp_path / "a.glb"), str(tmp_path / "a.png")
    gltf_util.scene_to_gltf(desc, glb)
    cam = ",".join(str(v) for v in (*desc.camera_pos, *desc.camera_target, desc.fov_y))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "examples", "png.py"), glb, png, "--size", "160x96", "--camera", cam])
    img = gltf_ref.decode_png(open(png, "rb").read())
    r = rt.Renderer((160, 96))
    _, inst = r.load_gltf(glb)
    want = r.render_to_host_memory((desc.camera_pos, desc.camera_target, desc.fov_y), inst)
    assert img.shape == (96, 160, 4) and (img == want).all() and (img[..., 3] == 255).all() and img[..., :3].std() > 5


def test_randomised_soups_all_tree_kinds(rt, oracle):
    """scripts/gpu_fuzz_trace.py in small: random triangle soups over five decades of scale, thin triangles, rotated and
    non-uniformly scaled instances; axis-parallel rays, rays aimed at vertices and edge midpoints, rays starting on
    geometry — against the oracle, for the host SAH tree, the device LBVH and an in-place update of it."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_fuzz_trace.py"), "3", "6"], capture_output=True, text=True)
    assert out.returncode == 0 and "TOTAL mismatches 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_randomised_materials_full_frames(rt, oracle):
    """scripts/gpu_fuzz_passes.py in small: Cornell boxes filled with random spheres / patches of random materials (diffuse,
    rough and smooth metal, glass of random ior, lights of random strength, textured with normal maps), four frames with a
    moving camera; radiance, reservoirs, G-buffer and the tonemapped output against the oracle, bit for bit."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_fuzz_passes.py"), "5", "4"], capture_output=True, text=True)
    assert out.returncode == 0 and "TOTAL differing bytes 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_non_finite_rays_terminate_as_misses(rt, oracle):
    """NaN / infinite origins, directions and intervals: every comparison of the triangle test is false -> miss, on both
    sides, and the traversal terminates."""
    desc = scenes.cornell_box()
    osc, gsc = oracle.OracleScene().load(desc), rt.Scene(0).load(desc)
    nan, inf = float("nan"), float("inf")
    o = [(nan, 1, 2), (0, 1, 2), (inf, 1, 2), (0, 1, 2), (0, 1, 2), (0, 1, 2), (0, -inf, 2), (0, 1, 2)]
    d = [(0, 0, -1), (nan, 0, -1), (0, 0, -1), (inf, 0, -1), (0, 0, 0), (0, 0, -1), (0, 0, -1), (0, 0, -1)]
    rays = make_rays(o, d)
    rays["tmin"][5] = nan
    rays["tmax"][7] = nan
    rays_t = rt.rays_to_device(rays)
    hits = rt.hits_from_device(gsc.trace_closest(rays_t, len(rays)))
    assert_bits_equal(osc.trace_closest(rays), hits, "non-finite rays")
    assert (hits["tri"] == 0xFFFFFFFF).all() and (hits["t"] == -1.0).all()
    assert np.array_equal(osc.trace_any(rays), gsc.trace_any(rays_t, len(rays)).cpu().numpy().view(np.uint32))


# ---- acceleration-structure maintenance -------------------------------------------
def _moving_instances(desc, f):
    """Every instance orbits / spins / breathes a little differently each frame; scales stay non-uniform."""
    out = []
    for k, (key, xs) in enumerate(desc.instances):
        moved = []
        for j, x in enumerate(xs):
            M = np.eye(4, dtype=np.float64); M[:3, :] = np.asarray(x, dtype=np.float64).reshape(3, 4)
            a = 0.11 * f * (1 + (k + j) % 3)
            R = np.array([[np.cos(a), 0, np.sin(a), 0.02 * f * ((k % 3) - 1)], [0, 1 + 0.01 * f * (k % 2), 0, 0.01 * f], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]])
            moved.append((R @ M)[:3, :].astype(np.float32).reshape(12) if key >= 7 else np.asarray(x, dtype=np.float32))
        out.append((key, moved))
    return out


def test_dynamic_instances_update_in_place_and_rebuild_cycle(rt, oracle, blue_noise):
    """The instance transforms change every frame: the scene follows AsState (8 in-place updates = device re-flatten +
    refit, then a fast rebuild), and every frame's queries and passes still equal the oracle, which rebuilds its own
    BVH from scratch each time. The refitted tree read back from the device is a valid conservative BVH."""
    from test_host_abi import _check_bvh
    desc = scenes.cornell_glass_mirror()
    W, H = 96, 72
    osc, gsc = oracle.OracleScene().load(desc), rt.Scene(0).load(desc)
    of, gf = oracle.HostFrame(W, H, blue_noise), rt.DeviceFrame(W, H, blue_noise)
    rays = np.concatenate([camera_rays(oracle, desc, 64, 64), random_rays(6000, 77)])
    rays_t = rt.rays_to_device(rays)
    ops, prev = [gsc.as_state()[1]], None
    stats0 = gsc.bvh_stats()
    for f in range(1, 12):
        inst = _moving_instances(desc, f)
        osc.set_instances(inst); gsc.set_instances(inst)
        ops.append(gsc.as_state()[1])
        assert_bits_equal(osc.trace_closest(rays), rt.hits_from_device(gsc.trace_closest(rays_t, len(rays))), "SrHit f%d" % f)
        assert np.array_equal(osc.trace_any(rays), gsc.trace_any(rays_t, len(rays)).cpu().numpy().view(np.uint32)), "occluded f%d" % f
        om = oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        gm = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(om.view_proj)
        osc.trace_ris(of, om, f - 1); osc.trace_final(of, om, f - 1)
        gsc.trace_ris(gf, gm, f - 1); gsc.trace_final(gf, gm, f - 1)
        h = gf.host()
        assert_bits_equal(of.raw_color, h["raw_color"], "raw_color f%d" % f)
        assert_bits_equal(of.reservoirs[(f - 1) & 1], h["reservoirs"][(f - 1) & 1], "reservoir f%d" % f)
        if f in (3, 8):
            nodes, tris = gsc.read_bvh()
            st = gsc.bvh_stats()
            assert st.n_nodes == stats0.n_nodes                                   # topology kept by updates
            _check_bvh(nodes, tris, st.max_depth, st.max_stack)
    U, F, S = abi.OP_UPDATE, abi.OP_FAST_BUILD, abi.OP_SLOW_BUILD
    assert ops == [S] + [U] * 8 + [F] + [U] * 2
    # a different instance LAYOUT cannot be updated in place -> rebuild
    gsc.set_instances(_moving_instances(desc, 12)[:-1])
    assert gsc.as_state()[1] == F
    # quiet frames: 15 idle, then the settle rebuild (SlowBuild), then Optimal
    quiet = []
    for _ in range(17):
        gsc.end_frame(); quiet.append(gsc.as_state()[1])
    assert quiet == [abi.OP_NONE] * 15 + [S] + [abi.OP_NONE] and gsc.as_state()[0].changing == 0


@pytest.mark.parametrize("scene_fn,box", [
    (scenes.torus_knot, ((-4, 0, -4), (4, 5, 4))),
    (lambda: scenes.heightfield(n=300), ((-15, 0, -15), (15, 9, 15))),
    (lambda: scenes.atrium(columns_per_side=6, col_segments=24, col_rings=6, floor_div=16, tex=32, n_lamps=8), ((-8, 0.1, -17), (8, 6.5, 17))),
])
def test_device_lbvh_fast_build(rt, oracle, blue_noise, scene_fn, box):
    """OpType::FastBuild = linear BVH built on the device: a valid conservative tree, the same query results as the
    oracle (results do not depend on the tree), and it can be updated in place afterwards."""
    from test_host_abi import _check_bvh
    desc = scene_fn()
    osc = oracle.OracleScene().load(desc)
    gsc = rt.Scene(0).load(desc)
    sah = gsc.bvh_stats()
    gsc.force_next_op(abi.OP_FAST_BUILD)
    gsc.set_instances(desc.instances)
    st = gsc.bvh_stats()
    assert gsc.as_state()[1] == abi.OP_FAST_BUILD and st.n_triangles == sah.n_triangles and st.sah_cost == 0.0   # built on the device
    print("%s: %d tris, host SAH %.1f ms (%d nodes, stack %d) | device LBVH %.2f ms (%d nodes, depth %d, stack %d)" % (
        desc.name, st.n_triangles, sah.build_ms, sah.n_nodes, sah.max_stack, st.build_ms, st.n_nodes, st.max_depth, st.max_stack))
    nodes, tris = gsc.read_bvh()
    _check_bvh(nodes, tris, st.max_depth, st.max_stack, stack_limit=47)
    rays = np.concatenate([camera_rays(oracle, desc, 160, 90), random_rays(20000, 5, box=box)])
    rays_t = rt.rays_to_device(rays)
    hits_t = gsc.trace_closest(rays_t, len(rays))
    hits = rt.hits_from_device(hits_t)
    assert_bits_equal(osc.trace_closest(rays), hits, "SrHit (LBVH)")
    assert np.array_equal(osc.trace_any(rays), gsc.trace_any(rays_t, len(rays)).cpu().numpy().view(np.uint32))
    pl = gsc.shade_closest_hit(hits_t, len(rays)).cpu().numpy().view(np.uint32).reshape(-1).view(abi.RAY_PAYLOAD)
    assert_bits_equal(osc.shade_closest_hit(hits), pl, "RayPayload (device-built shade records)")
    W, H = 160, 96
    of, gf = oracle.HostFrame(W, H, blue_noise), rt.DeviceFrame(W, H, blue_noise)
    om = oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
    gm = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
    osc.trace_ris(of, om, 0); osc.trace_final(of, om, 0)
    gsc.trace_ris(gf, gm, 0); gsc.trace_final(gf, gm, 0)
    assert_bits_equal(of.raw_color, gf.host()["raw_color"], "raw_color (LBVH)")
    # an in-place update of the device-built tree
    inst = [(k, [np.asarray(x, dtype=np.float32) + np.float32(0.05) * np.eye(3, 4, 3, dtype=np.float32).reshape(12) for x in xs]) for k, xs in desc.instances]
    gsc.set_instances(inst); osc.set_instances(inst)
    assert gsc.as_state()[1] == abi.OP_UPDATE
    assert_bits_equal(osc.trace_closest(rays), rt.hits_from_device(gsc.trace_closest(rays_t, len(rays))), "SrHit (LBVH + update)")
    nodes, tris = gsc.read_bvh()
    _check_bvh(nodes, tris, st.max_depth, st.max_stack, stack_limit=47)


def test_device_radix_tree_fast_build_switch(rt, oracle, monkeypatch):
    """SR_FAST_BUILD=lbvh (read at sr_scene_create) selects Karras' binary radix tree instead of PLOC for the device fast
    build: a different, equally valid tree — same structural checks, same hits as the oracle."""
    from test_host_abi import _check_bvh
    desc = scenes.torus_knot()
    osc = oracle.OracleScene().load(desc)
    plain = rt.Scene(0).load(desc)
    plain.force_next_op(abi.OP_FAST_BUILD); plain.set_instances(desc.instances)
    monkeypatch.setenv("SR_FAST_BUILD", "lbvh")
    gsc = rt.Scene(0).load(desc)
    monkeypatch.delenv("SR_FAST_BUILD")
    gsc.force_next_op(abi.OP_FAST_BUILD); gsc.set_instances(desc.instances)
    st = gsc.bvh_stats()
    assert gsc.as_state()[1] == abi.OP_FAST_BUILD and st.sah_cost == 0.0
    nodes, tris = gsc.read_bvh()
    _check_bvh(nodes, tris, st.max_depth, st.max_stack, stack_limit=47)
    assert (st.n_nodes, st.max_depth) != (plain.bvh_stats().n_nodes, plain.bvh_stats().max_depth)     # really another topology
    rays = np.concatenate([camera_rays(oracle, desc, 160, 90), random_rays(20000, 6, box=((-4, 0, -4), (4, 5, 4)))])
    rays_t = rt.rays_to_device(rays)
    assert_bits_equal(osc.trace_closest(rays), rt.hits_from_device(gsc.trace_closest(rays_t, len(rays))), "SrHit (radix tree)")
    assert np.array_equal(osc.trace_any(rays), gsc.trace_any(rays_t, len(rays)).cpu().numpy().view(np.uint32))


def test_4k_frame_1m_triangles(rt, oracle, blue_noise):
    """BASELINE.json config 5's extent (3840x2160) on the 1M-triangle scene: one full frame against the oracle, bit for
    bit, plus the strip composition a tile-parallel run relies on (strip + halo launches == one full launch), for the
    column strips bench.py cuts across GPUs and for row strips."""
    desc = scenes.heightfield(708)
    W, H = 3840, 2160
    osc, gsc, of, gf = run_both(rt, oracle, desc, W, H, 1, blue_noise)
    full = gf.host()["raw_color"].copy()
    assert np.isfinite(full).all() and full[:, :3].any()
    single = gsc.counters()
    from sunray_amd import distributed as sd
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
    cfg = abi.SrTraceConfig.reference()
    for axis, world in (("cols", 8), ("rows", 4)):
        gf2 = rt.DeviceFrame(W, H, blue_noise)
        part = sd.Partition(W, H, world, axis)
        gsc.reset_counters()
        for rank in range(world):
            sd.render_strip(gsc, gf2, m, 0, cfg, part, rank)
        assert_bits_equal(full, gf2.host()["raw_color"], "%d %s strips (+halo) compose to the 4K frame" % (world, axis))
        c = gsc.counters()      # halo pixels are traced but not counted: the strips' rays add up to the single launch's
        assert (c.closest_queries, c.any_queries, c.reused_primary_hits, c.reused_visibility_queries) == (single.closest_queries, single.any_queries, single.reused_primary_hits, single.reused_visibility_queries)
        del gf2


def test_column_tiles_compose_and_count(rt, oracle, blue_noise):
    """Launch rectangles (tile_y0, tile_h, tile_x0, tile_w) with ragged column cuts on a ragged extent: any tiling gives the
    full launch's bits; count_x0 / count_cols restrict the ray counters to a column window; a tile outside the image is an error."""
    desc = scenes.cornell_box()
    W, H = 203, 77
    osc, gsc, of, gf = run_both(rt, oracle, desc, W, H, 2, blue_noise)         # frames 0, 1 (temporal reuse on frame 1)
    ref = gf.host()
    gf2 = rt.DeviceFrame(W, H, blue_noise)
    prev = None
    for f in range(2):
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        cuts = [(0, 0, 0, 61), (0, 0, 61, 3), (0, 40, 64, 139), (40, 37, 64, 139)]      # h == 0: all rows
        for tile in cuts:
            gsc.trace_ris(gf2, m, f, tile=tile)
        for tile in cuts[::-1]:
            gsc.trace_final(gf2, m, f, tile=tile)
    got = gf2.host()
    assert_bits_equal(ref["raw_color"], got["raw_color"], "column tiles raw_color")
    assert_bits_equal(ref["reservoirs"][1], got["reservoirs"][1], "column tiles reservoirs")
    assert_bits_equal(ref["reservoirs_gi"][1], got["reservoirs_gi"][1], "column tiles GI reservoirs")
    # counting window == oracle's
    cfg = abi.SrTraceConfig.reference()
    cfg.count_x0, cfg.count_cols, cfg.count_y0, cfg.count_rows = 50, 70, 10, 30
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
    om = oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
    gsc.reset_counters(); osc.reset_counters()
    gsc.trace_ris(gf2, m, 0, cfg, tile=(0, 60, 20, 150)); osc.trace_ris(of, om, 0, cfg, tile=(0, 60, 20, 150))
    gsc.trace_final(gf2, m, 0, cfg, tile=(0, 60, 20, 150)); osc.trace_final(of, om, 0, cfg, tile=(0, 60, 20, 150))
    gc, oc = gsc.counters(), osc.counters()
    assert (ref_closest(gc), ref_any(gc)) == (oc.closest_queries, oc.any_queries) and 0 < gc.closest_queries < single_frame_closest(osc, of, om)
    assert gc.reused_primary_hits == 70 * 30
    with pytest.raises(Exception):
        gsc.trace_ris(gf2, m, 0, tile=(0, 0, W, 8))


def single_frame_closest(osc, of, om):
    osc.reset_counters()
    osc.trace_ris(of, om, 0); osc.trace_final(of, om, 0)
    return osc.counters().closest_queries


def test_device_lbvh_1m_triangles(rt, oracle):
    desc = scenes.heightfield(708)
    gsc = rt.Scene(0).load(desc)
    sah = gsc.bvh_stats()
    for rep in range(2):           # second build: scratch already allocated, kernels loaded
        gsc.force_next_op(abi.OP_FAST_BUILD)
        gsc.set_instances(desc.instances)
    st = gsc.bvh_stats()
    print("1M tris: host SAH %.1f ms (%d nodes, stack %d) | device LBVH %.2f ms (%d nodes, depth %d, stack %d, on device: %s)" % (
        sah.build_ms, sah.n_nodes, sah.max_stack, st.build_ms, st.n_nodes, st.max_depth, st.max_stack, st.sah_cost == 0.0))
    osc = oracle.OracleScene().load(desc)
    rays = camera_rays(oracle, desc, 320, 180)
    assert_bits_equal(osc.trace_closest(rays), rt.hits_from_device(gsc.trace_closest(rt.rays_to_device(rays), len(rays))), "SrHit 1M (LBVH)")


def test_update_in_place_large_scene_timing(rt, oracle):
    """1M-triangle scene + a moving light rig: an in-place update must be far cheaper than the host rebuild and give
    the same hits as the oracle."""
    desc = scenes.heightfield(708)
    osc, gsc = oracle.OracleScene().load(desc), rt.Scene(0).load(desc)
    build_ms = gsc.bvh_stats().build_ms
    inst = [(k, [np.asarray(x, dtype=np.float32) + (np.float32(0.3) * np.eye(3, 4, 3, dtype=np.float32).reshape(12) if k == 2 else 0) for x in xs])
            for k, xs in desc.instances]
    gsc.set_instances(inst); osc.set_instances(inst)
    st, op = gsc.as_state()
    assert op == abi.OP_UPDATE
    update_ms = gsc.bvh_stats().build_ms
    print("host build %.1f ms, in-place update %.2f ms" % (build_ms, update_ms))
    assert update_ms < 0.25 * build_ms
    rays = camera_rays(oracle, desc, 256, 144)
    assert_bits_equal(osc.trace_closest(rays), rt.hits_from_device(gsc.trace_closest(rt.rays_to_device(rays), len(rays))), "SrHit after update")


# ---- BASELINE.json full sizes ---------------------------------------------------------------------
def test_full_size_1m_triangles_1080p(rt, oracle, blue_noise):
    """The bench workload itself (1920x1080, 999 714 triangles, reference constants): the oracle is fast
    enough on the GPU box's host cores to check the FULL frame bit for bit, plus size-independent
    properties: determinism, finiteness, the radiance cap, and counters equal to the oracle's."""
    desc = scenes.heightfield(708)
    W, H = 1920, 1080
    osc, gsc, of, gf = run_both(rt, oracle, desc, W, H, 2, blue_noise)
    h1 = gf.host()["raw_color"].copy()
    assert np.isfinite(h1).all() and (h1[:, :3] <= 10.0).all() and (h1[:, 3] == 1.0).all()
    # determinism: re-render frame 1 from the same history -> identical bits
    m0 = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
    m1 = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, list(m0.view_proj))
    gsc.trace_ris(gf, m1, 1); gsc.trace_final(gf, m1, 1)
    assert_bits_equal(h1, gf.host()["raw_color"], "re-render")
    st = gsc.bvh_stats()
    assert st.n_triangles == 999714 and st.max_depth <= 32


def _knobs(enable_restir, bounces):
    cfg = abi.SrTraceConfig.reference()
    cfg.enable_restir, cfg.max_bounces, cfg.shadow_bounces = enable_restir, bounces, bounces
    return cfg


def test_config2_torus_knot_1080p_diffuse_only(rt, oracle, blue_noise):
    """BASELINE.json configs[1] at its stated workload: the 70 004-triangle torus knot (the Bunny stand-in),
    1920x1080, 1 spp, enable_restir = 0, one bounce (primary + one NEE shadow ray; ray_gen_final.slang:40-42,135-136
    with BOUNCES = SHADOW_BOUNCES = 1). Every pixel against the oracle, bit for bit, ray counters included."""
    desc = scenes.torus_knot()
    osc, gsc, of, gf = run_both(rt, oracle, desc, 1920, 1080, 1, blue_noise, _knobs(0, 1))
    assert gsc.bvh_stats().n_triangles == 70004
    c = gsc.counters()
    assert c.closest_queries == 1920 * 1080 and 0 < c.any_queries <= c.closest_queries   # one primary per pixel, at most one NEE ray
    h = gf.host()["raw_color"]
    assert np.isfinite(h).all() and h[:, :3].any()


def test_config3_heightfield_1080p_two_bounces_four_frames(rt, oracle, blue_noise):
    """BASELINE.json configs[2] at its stated workload: 999 714-triangle heightfield, 1920x1080, 4 spp as frames 0..3
    (frame_count is the per-sample seed, rt_utils.slang:47-52), enable_restir = 0, two bounces + NEE shadow rays."""
    desc = scenes.heightfield(708)
    osc, gsc, of, gf = run_both(rt, oracle, desc, 1920, 1080, 4, blue_noise, _knobs(0, 2))
    assert gsc.bvh_stats().n_triangles == 999714
    c = gsc.counters()
    assert 1920 * 1080 < c.closest_queries <= 2 * 1920 * 1080


def test_config4_textured_atrium_1080p_ris_and_final(rt, oracle, blue_noise):
    """BASELINE.json configs[3] at its stated extent: the full textured atrium (248 384 triangles, 512x512 base-colour /
    metallic-roughness / normal textures, 64 emissive triangles, 5 % mirrors), 1920x1080, reference constants
    (raytracing_ris + raytracing_final, temporal reuse active from frame 1), three consecutive frames."""
    desc = scenes.atrium()
    osc, gsc, of, gf = run_both(rt, oracle, desc, 1920, 1080, 3, blue_noise)
    assert gsc.bvh_stats().n_triangles == 248384
