"""Where does a rank's step time go at N ranks? Per rank of the cost-balanced cut: device time of its RIS launch and of its final
launch alone (HIP events), the period with two frames in flight, and the host's issue time per step."""
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
rows = np.repeat((sc.tile_row_costs(0, W, 0, H) + sc.tile_row_costs(1, W, 0, H)) / 8.0, 8)[:H]
bounds = sd.balanced_bounds(rows, world)
print("rows", [bounds[i + 1] - bounds[i] for i in range(world)])
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1
for rank in range(world):
    if only >= 0 and rank != only: continue
    fr = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
    fp = sd.FramePipeline(fr, rt.DeviceFrame(W, H, scenes.white_noise_rgba8()))
    prev = None
    mats = []
    for f in range(60):
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj); mats.append(m)
    # sequential, timed per pass
    sc.enable_timing(True)
    for f in range(6): sd.render_strip(sc, fr, mats[f], f, ccfg, world, rank, abi.TRACE_FLAG_UNCOUNTED, bounds=bounds)
    torch.cuda.synchronize(); sc.read_timing(0); sc.read_timing(1)
    for f in range(6, 26): sd.render_strip(sc, fr, mats[f], f, ccfg, world, rank, abi.TRACE_FLAG_UNCOUNTED, bounds=bounds)
    torch.cuda.synchronize()
    a, na = sc.read_timing(0); b, nb = sc.read_timing(1)
    sc.enable_timing(False)
    # pipelined period and host issue time
    for f in range(26, 30): fp.step(sc, mats[f], f, ccfg, world, rank, bounds=bounds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(30, 60): fp.step(sc, mats[f], f, ccfg, world, rank, bounds=bounds)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rank %d: ris %.3f ms  final %.3f ms  | 2 in flight: period %.3f ms, host issue %.3f ms/step" % (rank, a / max(na, 1), b / max(nb, 1), (t2 - t0) / 30 * 1e3, (t1 - t0) / 30 * 1e3))
    del fr, fp
