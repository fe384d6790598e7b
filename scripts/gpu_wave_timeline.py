"""How many waves run at any moment of a pass launch? Needs a -DSR_DIAG_TIMELINE=9 build (SUNRAY_HIP_LIB): in frame 9 every wave
records its end time (device-wide 100 MHz counter) and duration; prints the number of resident waves over the launch."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt
from sunray_amd._lib import lib, check
W, H = 1920, 1080
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
fr = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
cfg = abi.SrTraceConfig.reference()
prev = None
for f in range(10):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    sc.trace_ris(fr, m, f, cfg); sc.trace_final(fr, m, f, cfg)
torch.cuda.synchronize()
for which, name in ((0, "ris"), (1, "final")):
    out = np.zeros(40000, dtype=np.uint32); n = C.c_uint32()
    check(lib().sr_scene_read_tile_costs(sc._h, which, W, 0, H, out.ctypes.data_as(C.c_void_p), len(out), C.byref(n)))
    v = out[:n.value].astype(np.int64)
    end = ((v >> 12) & 0xFFFFF) * 0.01                  # us on the device-wide 100 MHz counter
    dur_cyc = (v & 0xFFF) * 256.0
    wrap = 0x100000 * 0.01                                # the 20 bits kept wrap every 10.5 ms: unwrap around the median
    end = (end - np.median(end) + wrap / 2) % wrap
    end -= end.min()
    # shader clock from the data: at most 4096 waves are resident at any time
    def resident(ghz, n=200):
        st = end - dur_cyc / (ghz * 1e3)
        ts = np.linspace(st.min(), end.max(), n)
        return max(int(((st <= t) & (end > t)).sum()) for t in ts)
    lo, hi = 1.0, 3.0
    for _ in range(12):
        mid = 0.5 * (lo + hi)
        if resident(mid) > 4096: lo = mid
        else: hi = mid
    ghz = hi
    dur = dur_cyc / (ghz * 1e3)
    start = end - dur
    t0, T = start.min(), end.max()
    print("%s: launch %.0f us at ~%.1f GHz; waves %d; mean duration %.0f us, max %.0f us" % (name, T - t0, ghz, len(v), dur.mean(), dur.max()))
    edges = np.linspace(t0, T, 26)
    res = [int(((start <= 0.5 * (edges[i] + edges[i + 1])) & (end > 0.5 * (edges[i] + edges[i + 1]))).sum()) for i in range(25)]
    print("   resident waves per 4 %% slice (of 4096 slots): " + " ".join("%d" % x for x in res))
    print("   mean %.0f of 4096" % ((dur.sum()) / (T - t0)))
    # which waves make the tail? (tile rows and durations of the waves that end in the last 15 % of the launch)
    tiles_x = W // 8
    row = np.arange(len(v)) // tiles_x
    late = end > t0 + 0.85 * (T - t0)
    print("   waves ending in the last 15 %% of the launch: %d; started (as %% of the launch) p10 %.0f p50 %.0f p90 %.0f; duration p10 %.0f p50 %.0f p90 %.0f us" % (
        late.sum(), *(100 * (np.percentile(start[late], q) - t0) / (T - t0) for q in (10, 50, 90)), *(np.percentile(dur[late], q) for q in (10, 50, 90))))
    hist = np.bincount(row[late] // 9, minlength=15)
    print("   their tile rows, in bands of 72 pixel rows from the top: " + " ".join("%d" % x for x in hist))
    stt = (start - t0) / (T - t0)
    print("   start time (%% of launch) of each band of rows, median: " + " ".join("%.0f" % (100 * np.median(stt[(row // 9) == b])) for b in range(15)))
