"""Device time of every rank's share of the bench frame at N ranks (no collective), with the cost-balanced cut bench.py
uses: how even is the split, and what strong scaling can the tracing alone reach?"""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt, distributed as sd
W, H = 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
cfg = abi.SrTraceConfig.reference()
cal = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
ccfg = copy.copy(cfg); ccfg.flags |= abi.TRACE_FLAG_UNCOUNTED
prev = None
for f in range(4):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    sc.trace_ris(cal, m, f, ccfg); sc.trace_final(cal, m, f, ccfg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for f in range(4, 24):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    sc.trace_ris(cal, m, f, ccfg); sc.trace_final(cal, m, f, ccfg)
torch.cuda.synchronize()
full = (time.perf_counter() - t0) / 20 * 1e3
rows = np.repeat((sc.tile_row_costs(0, W, 0, H) + sc.tile_row_costs(1, W, 0, H)) / 8.0, 8)[:H]
cases = [("equal rows, sequential", [min(r * ((H + world - 1) // world), H) for r in range(world)] + [H], False),
         ("measured-cost strips, 2 frames in flight", sd.balanced_bounds(rows, world), True)]
for name, bounds, piped in cases:
    times = []
    for rank in range(world):
        fr = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
        fp = sd.FramePipeline(fr, rt.DeviceFrame(W, H, scenes.white_noise_rgba8())) if piped else None
        send = torch.zeros((bounds[rank + 1] - bounds[rank]) * W, 4, device="cuda:0")
        prev = None
        def step(f):
            global prev
            m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
            if fp is not None:
                fp.step(sc, m, f, cfg, world, rank, bounds=bounds, after_final=lambda g: send.copy_(g.raw_color[bounds[rank] * W:bounds[rank + 1] * W]))
                return
            sd.render_strip(sc, fr, m, f, cfg, world, rank, abi.TRACE_FLAG_UNCOUNTED, bounds=bounds)
            send.copy_(fr.raw_color[bounds[rank] * W:bounds[rank + 1] * W])
        for f in range(6): step(f)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(6, 46): step(f)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 40 * 1e3)
        del fr, fp
    print("%-14s N=%d rows %s" % (name, world, [bounds[i + 1] - bounds[i] for i in range(world)]))
    print("   ms/step per rank %s  -> max %.3f ms, 1-GPU frame %.3f ms, tracing-only speed-up %.2fx" % (["%.2f" % t for t in times], max(times), full, full / max(times)))
