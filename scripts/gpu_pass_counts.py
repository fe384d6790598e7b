"""Per-pass ray counts and times on the bench frame."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt
W, H = 1920, 1080
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
fr = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
cfg = abi.SrTraceConfig.reference()
prev = None
sc.enable_timing(True)
for f in range(6):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    sc.reset_counters(); sc.trace_ris(fr, m, f, cfg); c1 = sc.counters()
    sc.reset_counters(); sc.trace_final(fr, m, f, cfg); c2 = sc.counters()
    a, _ = sc.read_timing(0); b, _ = sc.read_timing(1)
    print("frame %d ris: closest %d any %d (%.3f ms) | final: closest %d any %d (%.3f ms)" % (f, c1.closest_queries, c1.any_queries, a, c2.closest_queries, c2.any_queries, b))
d = fr.host()["depth"]
print("sky pixels: %.1f %%" % (100.0 * (d == 0x7c00).mean()))
