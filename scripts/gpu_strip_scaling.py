"""Strong scaling of one frame over N GPUs, emulated on ONE GPU: it plays every rank of an N-rank run in turn (its own frame
buffers and temporal history, RIS over strip + 30-pixel halo, final over the strip, two frames in flight as bench.py runs
N > 1) and reports the slowest rank's steady-state time per frame — what an N-GPU run is bound by, the gather being
asynchronous. Row strips vs column strips, equal cuts vs cuts of equal measured cost, 1920x1080 and 3840x2160
(BASELINE.json config 5's extent).   usage: python scripts/gpu_strip_scaling.py [1080|2160] [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt, distributed as sd

H = int(sys.argv[1]) if len(sys.argv) > 1 else 1080
W = H * 16 // 9
worlds = [int(x) for x in sys.argv[2:]] or [2, 4, 8]
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
bn = scenes.white_noise_rgba8()
cfg = abi.SrTraceConfig.reference()
cfg.flags |= abi.TRACE_FLAG_UNCOUNTED
FRAMES, WARM = 14, 6


def mats(n):
    out, prev = [], None
    for _ in range(n):
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        out.append(m)
    return out


M = mats(WARM + FRAMES)


def rank_time(part, rank):
    """Median step period (ms) of `rank` in steady state, two frames in flight."""
    fa, fb = rt.DeviceFrame(W, H, bn), rt.DeviceFrame(W, H, bn)
    fp = sd.FramePipeline(fa, fb)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(FRAMES)]
    for f in range(WARM + FRAMES):
        cb = (lambda fr, i=f - WARM: evs[i].record(fp.s_final)) if f >= WARM else None
        fp.step(sc, M[f], f, cfg, part, rank, after_final=cb)
    torch.cuda.synchronize()
    gaps = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(FRAMES - 1))
    del fp, fa, fb
    return gaps[len(gaps) // 2]


one = rank_time(sd.Partition(W, H, 1), 0)
print("%dx%d, 999 714 triangles, reference constants, two frames in flight: 1 GPU %.3f ms per frame" % (W, H, one), flush=True)
# measured per-tile cost of whole-frame launches (what bench.py cuts its strips from)
tiles_x = (W + 7) // 8
tc = sc.tile_costs(0, W, 0, H).astype(np.float64) + sc.tile_costs(1, W, 0, H)
for world in worlds:
    for axis in os.environ.get("STRIP_AXES", "rows,cols").split(","):
        L = W if axis == "cols" else H
        cost = sd.axis_cost_from_tiles(tc, tiles_x, axis, L)
        for name, bounds in (("equal", None), ("equal-cost", sd.balanced_bounds(cost, world, min_size=sd.SPATIAL_HALO + 2))):
            part = sd.Partition(W, H, world, axis, bounds)
            ts = [rank_time(part, r) for r in range(world)]
            halo = sum(part.grown(r, sd.SPATIAL_HALO)[1] for r in range(world)) / float(L) - 1.0
            print("N=%d %-4s %-10s slowest rank %.3f ms -> %.2fx   (ranks %s; RIS halo overhead %.0f %%)" % (
                world, axis, name, max(ts), one / max(ts), " ".join("%.2f" % t for t in ts), 100 * halo), flush=True)
