"""Emulates bench.py's feedback balancer on ONE GPU: each rank's pipelined step time is measured in turn (same two streams for
every rank), rank 0's rule re-weights the rows and cuts again. Prints the per-rank periods of every round."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt, distributed as sd
W, H = 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
cfg = abi.SrTraceConfig.reference(); cfg.flags |= abi.TRACE_FLAG_UNCOUNTED
cal = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
prev = None
for f in range(4):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    sc.trace_ris(cal, m, f, cfg); sc.trace_final(cal, m, f, cfg)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for f in range(4, 24):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    sc.trace_ris(cal, m, f, cfg); sc.trace_final(cal, m, f, cfg)
ev1.record(); torch.cuda.synchronize()
full = ev0.elapsed_time(ev1) / 20
row_cost = np.repeat((sc.tile_row_costs(0, W, 0, H) + sc.tile_row_costs(1, W, 0, H)) / 8.0, 8)[:H]
bounds = sd.balanced_bounds(row_cost, world)
fp = sd.FramePipeline(cal, rt.DeviceFrame(W, H, scenes.white_noise_rgba8()))
state = {"f": 24, "prev": prev}
def steps(n, rank, b, evs=None):
    for i in range(n):
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, state["prev"]); state["prev"] = list(m.view_proj)
        fp.step(sc, m, state["f"], cfg, world, rank, bounds=b, after_final=(lambda g, i=i: evs[i].record(fp.s_final)) if evs is not None else None)
        state["f"] += 1
best = (1e9, bounds)
for it in range(rounds):
    periods = []
    for rank in range(world):
        steps(4, rank, bounds)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
        steps(13, rank, bounds, evs)
        torch.cuda.synchronize()
        periods.append(sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(12))[6])
    print("round %d rows %s" % (it, [bounds[i + 1] - bounds[i] for i in range(world)]))
    print("   periods %s -> max %.3f ms; 1-GPU frame %.3f ms; tracing-only speed-up %.2fx" % (["%.2f" % p for p in periods], max(periods), full, full / max(periods)))
    if max(periods) < best[0]: best = (max(periods), bounds)
    row_cost, bounds = sd.refine_bounds(row_cost, bounds, periods)
print("kept: max %.3f ms, rows %s" % (best[0], [best[1][i + 1] - best[1][i] for i in range(world)]))
