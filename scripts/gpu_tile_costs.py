"""Distribution of the per-wave (8x8 tile) cycle counts of both passes on the bench frame: how long are the slowest waves?"""
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
sc.enable_timing(True)
for f in range(8):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
    sc.trace_ris(fr, m, f, cfg); sc.trace_final(fr, m, f, cfg)
    a, _ = sc.read_timing(0); b, _ = sc.read_timing(1)
for which, name, ms in ((0, "ris", a), (1, "final", b)):
    out = np.zeros(40000, dtype=np.uint32); n = C.c_uint32()
    check(lib().sr_scene_read_tile_costs(sc._h, which, W, 0, H, out.ctypes.data_as(C.c_void_p), len(out), C.byref(n)))
    c = out[:n.value].astype(np.float64) / 2100.0    # s_memtime ticks are shader cycles (~2.1 GHz under this load) -> microseconds
    q = np.percentile(c, [50, 90, 99, 99.9, 100])
    print("%-5s kernel %.3f ms | wave duration us: mean %.0f median %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f | waves longer than half the kernel: %d of %d" % (
        name, ms, c.mean(), q[0], q[1], q[2], q[3], q[4], int((c > ms * 500).sum()), len(c)))
# spatial profile: mean and max wave duration per band of 8 tile rows (64 pixel rows), final pass
out = np.zeros(40000, dtype=np.uint32); n = C.c_uint32()
check(lib().sr_scene_read_tile_costs(sc._h, 1, W, 0, H, out.ctypes.data_as(C.c_void_p), len(out), C.byref(n)))
tiles_x, tiles_y = W // 8, H // 8
grid = out[:tiles_x * tiles_y].astype(np.float64).reshape(tiles_y, tiles_x) / 2100.0
for r0 in range(0, tiles_y, 9):
    g = grid[r0:r0 + 9]
    print("pixel rows %4d-%4d: mean %.0f us  max %.0f us" % (r0 * 8, min((r0 + 9) * 8, H) - 1, g.mean(), g.max()))
# per tile column: summed wave time of both passes; equal-width XCD bands vs what the library's cost-balanced bands even out
col = np.zeros(tiles_x)
for which in (0, 1):
    out = np.zeros(40000, dtype=np.uint32); n = C.c_uint32()
    check(lib().sr_scene_read_tile_costs(sc._h, which, W, 0, H, out.ctypes.data_as(C.c_void_p), len(out), C.byref(n)))
    col += out[:tiles_x * tiles_y].astype(np.float64).reshape(tiles_y, tiles_x).sum(axis=0) / 2100.0
eq = np.array([col[(tiles_x * i) >> 3:(tiles_x * (i + 1)) >> 3].sum() for i in range(8)])
print("equal-width bands, wave time per band (ms of one wave slot): %s | max / mean = %.3f" % (" ".join("%.0f" % (t / 1e3) for t in eq), eq.max() / eq.mean()))
