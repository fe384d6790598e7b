"""Lane utilisation inside the traversal (tuning diagnostics): run with SUNRAY_HIP_LIB pointing at a -DSR_DIAG_UTIL=k variant
(1 = node steps, 2 = query entry, 3 = triangle tests); the instrumented counters then hold active lanes and 64 x wave-steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt
W, H = 1920, 1080
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
fr = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
cfg = abi.SrTraceConfig.reference()
sc.set_instrumented(True)
prev = None
for f in range(5):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    sc.reset_counters(); sc.trace_ris(fr, m, f, cfg); c1 = sc.counters()
    sc.reset_counters(); sc.trace_final(fr, m, f, cfg); c2 = sc.counters()
    if f >= 3:
        print("frame %d ris: lanes %d / slots %d = %.3f | final: lanes %d / slots %d = %.3f" % (
            f, c1.boxes_tested, c1.tris_tested, c1.boxes_tested / max(c1.tris_tested, 1), c2.boxes_tested, c2.tris_tested, c2.boxes_tested / max(c2.tris_tested, 1)))
