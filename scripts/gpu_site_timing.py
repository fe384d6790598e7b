"""Wave-time per trace call site of the two passes on the bench frame (needs the throw-away `sites` variant:
profiles/experiments/r03_site_timing.patch applied, scripts/build_variant.sh sites; SUNRAY_HIP_LIB selects it)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sunray_amd import abi, runtime as rt, scenes
from sunray_amd._lib import lib

W, H = 1920, 1080
which = sys.argv[1] if len(sys.argv) > 1 else "heightfield"
desc = {"heightfield": lambda: scenes.heightfield(708), "atrium": scenes.atrium, "knot": scenes.torus_knot, "cornell": scenes.cornell_box}[which]()
bn = scenes.white_noise_rgba8()
sc = rt.Scene(0).load(desc)
fr = rt.DeviceFrame(W, H, bn)
prev = None
RIS = ["primary", "later virtual bounces", "DI visibility", "GI bounce (closest)", "NEE at GI hit", "-", "-", "-", "-", "-", "-", "TOTAL wave time"]
FIN = ["primary", "later closest", "DI shadow", "GI neighbour 0", "GI neighbour 1", "GI neighbour 2", "GI final visibility", "NEE", "-", "-", "-", "TOTAL wave time"]


def misc():
    out = (C.c_ulonglong * 32)()
    lib().sr_debug_read_misc(sc._h, out, 32)
    return np.array(list(out), dtype=np.float64)


for f in range(12):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
    prev = list(m.view_proj)
    sc.reset_counters()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); sc.trace_ris(fr, m, f); e[1].record(); sc.trace_final(fr, m, f); e[2].record()
    torch.cuda.synchronize()
    if f < 9:
        continue
    v = misc()
    print("frame %d: ris %.3f ms final %.3f ms; closest %d any %d reused %d" % (f, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), v[0], v[1], v[4]))
    for name, base, labels in (("ris_kernel", 8, RIS), ("final_kernel", 20, FIN)):
        t = v[base:base + 12]
        tot = t[11]
        print("  %s: share of summed wave time per call site" % name)
        for i in range(11):
            if t[i] > 0:
                print("    %-24s %5.1f %%" % (labels[i], 100 * t[i] / tot))
        print("    %-24s %5.1f %%" % ("outside traversal", 100 * (tot - t[:11].sum()) / tot))
