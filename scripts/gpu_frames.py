"""Per-frame device times of both passes (HIP events from the library) for N consecutive frames."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt
W, H = 1920, 1080
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
fr = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
cfg = abi.SrTraceConfig.reference()
prev = None
sc.enable_timing(True)
if len(sys.argv) > 2: sc.set_instrumented(True)
for f in range(n):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    sc.reset_counters()
    sc.trace_ris(fr, m, f, cfg); sc.trace_final(fr, m, f, cfg)
    c = sc.counters()
    a, _ = sc.read_timing(0); b, _ = sc.read_timing(1)
    rc = fr.raw_color
    print("frame %2d ris %7.3f ms final %7.3f ms closest %d any %d nan %d inf %d" % (f, a, b, c.closest_queries, c.any_queries, int(torch.isnan(rc).sum()), int(torch.isinf(rc).sum())), flush=True)
