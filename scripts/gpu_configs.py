"""Throughput of the other BASELINE.json configurations' stand-in scenes (not the bench line): ms per pass and Mray/s, 1080p."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt
W, H = 1920, 1080
def run(name, desc, cfg, frames=10):
    sc = rt.Scene(0).load(desc)
    fr = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
    prev = None
    sc.enable_timing(True)
    rows = []
    for f in range(frames):
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
        sc.reset_counters()
        if cfg.enable_restir: sc.trace_ris(fr, m, f, cfg)
        sc.trace_final(fr, m, f, cfg)
        c = sc.counters()
        a = sc.read_timing(0)[0] if cfg.enable_restir else 0.0
        b = sc.read_timing(1)[0]
        rows.append((a, b, c.closest_queries + c.any_queries))
    r = np.array(rows[3:], dtype=np.float64)
    st = sc.bvh_stats()
    ms = r[:, 0].mean() + r[:, 1].mean()
    print("%-46s %8d tris | ris %.3f ms final %.3f ms | %.1f Mray/frame -> %.0f Mray/s | build %.0f ms" % (
        name, st.n_triangles, r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean() / 1e6, r[:, 2].mean() / ms / 1e3, st.build_ms), flush=True)
ref = abi.SrTraceConfig.reference()
c2 = abi.SrTraceConfig.reference(); c2.enable_restir, c2.max_bounces, c2.shadow_bounces = 0, 1, 1
c3 = abi.SrTraceConfig.reference(); c3.enable_restir, c3.max_bounces, c3.shadow_bounces = 0, 2, 2
run("config 1 Cornell box (reference constants)", scenes.cornell_box(), ref)
run("config 2 torus knot 70k, diffuse only", scenes.torus_knot(), c2)
run("config 2 scene, reference constants", scenes.torus_knot(), ref)
run("config 3 heightfield 1M, 2 bounces + NEE", scenes.heightfield(708), c3)
run("config 4 textured atrium 250k, RIS + final", scenes.atrium(), ref)
run("bench: heightfield 1M, reference constants", scenes.heightfield(708), ref)
