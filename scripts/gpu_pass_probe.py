"""Where a scene's pass time goes, by switching parts of the passes off through SrTraceConfig (results then differ — this is a
probe, not a parity run): python scripts/gpu_pass_probe.py [atrium|heightfield|knot|cornell]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sunray_amd import abi, scenes, runtime as rt

which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
desc = {"heightfield": lambda: scenes.heightfield(708), "atrium": scenes.atrium, "knot": scenes.torus_knot, "cornell": scenes.cornell_box}[which]()
W, H = 1920, 1080
sc = rt.Scene(0).load(desc)
st = sc.bvh_stats()
print(which, "triangles", st.n_triangles, "nodes", st.n_nodes, "lights", len(sc.tables()["emissive_triangles"]) if hasattr(sc, "tables") else "?")


def run(label, cfg, instrumented=False, frames=8):
    fr = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
    sc.set_instrumented(instrumented)
    prev = None
    sc.enable_timing(True)
    rows = []
    for f in range(frames):
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        sc.reset_counters()
        sc.trace_ris(fr, m, f, cfg); c1 = sc.counters()
        sc.reset_counters()
        sc.trace_final(fr, m, f, cfg); c2 = sc.counters()
        rows.append((sc.read_timing(0)[0], sc.read_timing(1)[0], c1.closest_queries, c1.any_queries, c1.boxes_tested, c1.tris_tested,
                     c2.closest_queries, c2.any_queries, c2.boxes_tested, c2.tris_tested))
    sc.enable_timing(False); sc.set_instrumented(False)
    r = np.array(rows[3:], dtype=np.float64).mean(axis=0)
    s = "%-44s ris %.3f ms (%.2f M closest %.2f M any" % (label, r[0], r[2] / 1e6, r[3] / 1e6)
    if instrumented: s += ", %.1f boxes %.2f tris per ray" % (r[4] / max(r[2] + r[3], 1), r[5] / max(r[2] + r[3], 1))
    s += ") final %.3f ms (%.2f M closest %.2f M any" % (r[1], r[6] / 1e6, r[7] / 1e6)
    if instrumented: s += ", %.1f boxes %.2f tris per ray" % (r[8] / max(r[6] + r[7], 1), r[9] / max(r[6] + r[7], 1))
    print(s + ")", flush=True)
    del fr


ref = abi.SrTraceConfig.reference()
run("reference constants", ref)
run("reference constants, instrumented build", ref, instrumented=True)
c = abi.SrTraceConfig.reference(); c.ris_candidates = 1
run("1 RIS candidate instead of 16", c)
c = abi.SrTraceConfig.reference(); c.virtual_bounces = 1
run("1 virtual bounce instead of 20", c)
c = abi.SrTraceConfig.reference(); c.max_bounces = 1
run("final pass: 1 bounce", c)
