"""Trace cost of the same frames on a host binned-SAH tree, a device LBVH and an LBVH after in-place updates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt
W, H = 1920, 1080
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
fr = rt.DeviceFrame(W, H, scenes.white_noise_rgba8())
cfg = abi.SrTraceConfig.reference()


def frames(tag, n=8):
    prev = None
    sc.enable_timing(True)
    tot = []
    for f in range(n):
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev); prev = list(m.view_proj)
        sc.trace_ris(fr, m, f, cfg); sc.trace_final(fr, m, f, cfg)
        a, _ = sc.read_timing(0); b, _ = sc.read_timing(1)
        tot.append((a, b))
    t = np.array(tot[2:])
    st = sc.bvh_stats()
    print("%-28s ris %.3f ms final %.3f ms | nodes %d depth %d stack %d build %.2f ms" % (tag, t[:, 0].mean(), t[:, 1].mean(), st.n_nodes, st.max_depth, st.max_stack, st.build_ms), flush=True)


frames("host binned SAH")
for rep in range(2):
    sc.force_next_op(abi.OP_FAST_BUILD); sc.set_instances(desc.instances)
frames("device fast build (PLOC)")
sc.force_next_op(abi.OP_UPDATE); sc.set_instances(desc.instances)
frames("fast build + refit (same)")
print("update %.2f ms" % sc.bvh_stats().build_ms)
sc.force_next_op(abi.OP_SLOW_BUILD); sc.set_instances(desc.instances)
sc.force_next_op(abi.OP_UPDATE); sc.set_instances(desc.instances)
print("update of SAH tree %.2f ms" % sc.bvh_stats().build_ms)
frames("host SAH + refit (same)")
