"""BASELINE.json configurations 2-5 at their FULL stated workloads (scene, extent, knobs and frame / sample count), GPU vs oracle,
every buffer of every frame bit for bit — the long form of the config tests in tests/test_gpu_parity.py (which stop after 3-4
frames to keep the suite short). Config 5 (3840x2160, 16 samples) runs on one GPU, and every frame is also rendered as 8 column
strips (+ halo) whose composition must equal the single launch.   usage: python scripts/gpu_full_configs_parity.py [2 3 4 5]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from sunray_amd import abi, scenes, runtime as rt, distributed as sd

ob.set_threads(min(ob.usable_cores(), 16))
bn = scenes.white_noise_rgba8()
which = [int(x) for x in sys.argv[1:]] or [2, 3, 4, 5]


def knobs(restir, bounces):
    c = abi.SrTraceConfig.reference()
    if not restir:
        c.enable_restir, c.max_bounces, c.shadow_bounces = 0, bounces, bounces
    return c


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8).reshape(-1), np.ascontiguousarray(b).view(np.uint8).reshape(-1))


def run(name, desc, W, H, frames, cfg, strips=0):
    t0 = time.time()
    osc, gsc = ob.OracleScene().load(desc), rt.Scene(0).load(desc)
    of, gf = ob.HostFrame(W, H, bn), rt.DeviceFrame(W, H, bn)
    gs = rt.DeviceFrame(W, H, bn) if strips else None
    part = sd.Partition(W, H, strips) if strips else None
    prev, bad, rays = None, 0, 0
    for f in range(frames):
        om = ob.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        gm = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(om.view_proj)
        osc.reset_counters(); gsc.reset_counters()
        if cfg.enable_restir:
            osc.trace_ris(of, om, f, cfg); gsc.trace_ris(gf, gm, f, cfg)
        osc.trace_final(of, om, f, cfg); gsc.trace_final(gf, gm, f, cfg)
        h = gf.host()
        oc, gc = osc.counters(), gsc.counters()
        # the oracle traverses every TraceRay of the reference; the GPU answers the final pass's camera-ray query from the RIS pass's payload and
        # its repeated GI visibility query from the neighbour's identical one — counted apart (SrRayCounters.reused_*)
        ok = same(of.raw_color, h["raw_color"]) and (oc.closest_queries, oc.any_queries) == (gc.closest_queries + gc.reused_primary_hits, gc.any_queries + gc.reused_visibility_queries)
        if cfg.enable_restir:
            cur = f & 1
            ok = ok and same(of.depth, h["depth"]) and same(of.normal, h["normal"]) and same(of.diffuse, h["diffuse"]) and same(of.motion, h["motion"]) \
                and same(of.reservoirs[cur], h["reservoirs"][cur]) and same(of.reservoirs_gi[cur], h["reservoirs_gi"][cur])
        if strips:
            for r in range(strips):
                sd.render_strip(gsc, gs, gm, f, cfg, part, r)
            ok = ok and same(h["raw_color"], gs.raw_color.cpu().numpy())
        rays += gc.closest_queries + gc.any_queries
        bad += 0 if ok else 1
        print("  %s frame %2d: %s (%d closest + %d any-hit queries)" % (name, f, "bit-exact" if ok else "MISMATCH", gc.closest_queries, gc.any_queries), flush=True)
    st = gsc.bvh_stats()
    print("%s: %d triangles, %dx%d, %d frames, %.1f M rays, %d frames differ  [%.0f s]" % (name, st.n_triangles, W, H, frames, rays / 1e6, bad, time.time() - t0), flush=True)
    return bad


bad = 0
if 2 in which:
    bad += run("config 2 (torus knot, diffuse only: restir off, 1 bounce, 1 spp)", scenes.torus_knot(), 1920, 1080, 1, knobs(False, 1))
if 3 in which:
    bad += run("config 3 (1M-triangle heightfield, restir off, 2 bounces + NEE, 4 spp)", scenes.heightfield(708), 1920, 1080, 4, knobs(False, 2))
if 4 in which:
    bad += run("config 4 (textured atrium 250k, RIS + final, 8 spp)", scenes.atrium(), 1920, 1080, 8, knobs(True, 0))
if 5 in which:
    bad += run("config 5 (1M-triangle heightfield, 3840x2160, 16 spp, + 8 column strips per frame)", scenes.heightfield(708), 3840, 2160, 16, knobs(True, 0), strips=8)
print("TOTAL frames differing: %d" % bad)
sys.exit(1 if bad else 0)
