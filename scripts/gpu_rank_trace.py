"""Runs one rank's pipelined steps (for a rocprofv3 --kernel-trace look at the overlap of the two streams)."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt, distributed as sd
W, H = 1920, 1080
world, rank = int(sys.argv[1]), int(sys.argv[2])
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
cfg = abi.SrTraceConfig.reference(); cfg.flags |= abi.TRACE_FLAG_UNCOUNTED
cal = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
prev = None
for f in range(4):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    sc.trace_ris(cal, m, f, cfg); sc.trace_final(cal, m, f, cfg)
torch.cuda.synchronize()
rows = np.repeat((sc.tile_row_costs(0, W, 0, H) + sc.tile_row_costs(1, W, 0, H)) / 8.0, 8)[:H]
bounds = sd.balanced_bounds(rows, world)
fr = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
fp = sd.FramePipeline(fr, rt.DeviceFrame(W, H, scenes.white_noise_rgba8()))
prev = None
for f in range(24):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    fp.step(sc, m, f, cfg, world, rank, bounds=bounds)
torch.cuda.synchronize()
# period from events on the final stream (no profiler needed)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(40)]
import time
t0 = time.perf_counter()
for i, f in enumerate(range(24, 64)):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    fp.step(sc, m, f, cfg, world, rank, bounds=bounds, after_final=lambda g, i=i: evs[i].record(fp.s_final))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
d = [evs[i].elapsed_time(evs[i + 1]) for i in range(39)]
print("event periods (ms): min %.3f median %.3f mean %.3f max %.3f | wall %.3f ms/step, issue %.3f ms/step" % (min(d), sorted(d)[19], sum(d) / 39, max(d), (t2 - t0) / 40 * 1e3, (t1 - t0) / 40 * 1e3))
print(" ".join("%.2f" % x for x in d))
