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
RIS = ["setup + shading between virtual bounces", "TRAVERSAL primary", "TRAVERSAL later virtual bounces", "G-buffer / motion after the walk", "16 RIS candidates",
       "reservoir W + temporal DI reuse", "DI visibility ray setup", "TRAVERSAL DI visibility", "store DI, GI sample direction", "TRAVERSAL GI bounce (closest)",
       "NEE sample at the GI hit", "TRAVERSAL NEE at GI hit", "GI reservoir initial weights", "GI temporal reuse + stores"]
FIN = ["setup (rng, noise, camera ray)", "TRAVERSAL closest", "payload load + material decode", "DI: centre reservoir", "DI: 5 spatial neighbours", "DI: winner, shadow ray setup",
       "TRAVERSAL DI shadow", "GI: 3 neighbour candidates + merges", "TRAVERSAL GI neighbours", "GI: last merge", "GI: final weight, ray setup", "TRAVERSAL GI final visibility",
       "NEE setup (later bounces)", "TRAVERSAL NEE", "contributions, BRDF bounce, store"]


def misc():
    out = (C.c_ulonglong * 32)()
    lib().sr_debug_read_misc(sc._h, out, 32)
    return np.array(list(out), dtype=np.float64)


def show(name, labels, t):
    tot = t.sum()
    print("  %s: share of summed wave time" % name)
    for i, l in enumerate(labels):
        if t[i] > 0:
            print("    %-44s %5.1f %%" % (l, 100 * t[i] / tot))
    trav = sum(t[i] for i, l in enumerate(labels) if l.startswith("TRAVERSAL"))
    print("    %-44s %5.1f %%" % ("= traversal in all", 100 * trav / tot))


for f in range(12):
    m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
    prev = list(m.view_proj)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    sc.reset_counters()
    e[0].record(); sc.trace_ris(fr, m, f); e[1].record()
    torch.cuda.synchronize()
    v1 = misc()
    sc.reset_counters()
    e[1].record(); sc.trace_final(fr, m, f); e[2].record()
    torch.cuda.synchronize()
    v2 = misc()
    if f < 10:
        continue
    print("frame %d: ris %.3f ms final %.3f ms" % (f, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
    show("ris_kernel", RIS, v1[8:8 + len(RIS)])
    show("final_kernel", FIN, v2[8:8 + len(FIN)])
