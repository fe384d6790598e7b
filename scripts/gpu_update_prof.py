"""In-place update timings on a host-built and a device-built tree (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
for tag in ("SAH", "LBVH"):
    if tag == "LBVH":
        sc.force_next_op(abi.OP_FAST_BUILD); sc.set_instances(desc.instances)
        print("fast build %.2f ms" % sc.bvh_stats().build_ms)
    for rep in range(4):
        sc.force_next_op(abi.OP_UPDATE)
        t0 = time.time(); sc.set_instances(desc.instances); t1 = time.time()
        print("%s update %d: %.2f ms inside, %.2f ms call" % (tag, rep, sc.bvh_stats().build_ms, (t1 - t0) * 1e3), flush=True)
