"""Quick GPU-vs-oracle comparison used during bring-up (not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import binding as ob
from sunray_amd import abi, scenes, runtime as rt

def compare(desc, W, H, frames=2, cfg=None):
    bn = scenes.white_noise_rgba8()
    osc = ob.OracleScene().load(desc)
    gsc = rt.Scene(0).load(desc)
    st = gsc.bvh_stats()
    print(desc.name, "tris", st.n_triangles, "nodes", st.n_nodes, "depth", st.max_depth, "build_ms %.1f" % st.build_ms)
    of = ob.HostFrame(W, H, bn)
    gf = rt.DeviceFrame(W, H, bn)
    prev = None
    for f in range(frames):
        om = ob.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        gm = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        assert bytes(om) == bytes(gm), "camera matrices differ"
        prev = list(om.view_proj)
        t0 = time.time(); osc.trace_ris(of, om, f, cfg); osc.trace_final(of, om, f, cfg); t_or = time.time() - t0
        gsc.reset_counters()
        torch.cuda.synchronize(); t0 = time.time()
        gsc.trace_ris(gf, gm, f, cfg); gsc.trace_final(gf, gm, f, cfg)
        torch.cuda.synchronize(); t_gpu = time.time() - t0
        c = gsc.counters()
        h = gf.host()
        cur = f & 1
        def cmp(name, a, b):
            a = np.ascontiguousarray(a).view(np.uint8).reshape(-1); b = np.ascontiguousarray(b).view(np.uint8).reshape(-1)
            nd = int((a != b).sum())
            print("   %-14s bytes differing: %d / %d" % (name, nd, a.size))
            return nd
        print(" frame", f, "oracle %.2fs gpu %.4fs rays %d+%d -> %.1f Mray/s" % (t_or, t_gpu, c.closest_queries, c.any_queries, (c.closest_queries + c.any_queries) / t_gpu / 1e6))
        cmp("depth", of.depth, h["depth"]); cmp("normal", of.normal, h["normal"]); cmp("diffuse", of.diffuse, h["diffuse"]); cmp("motion", of.motion, h["motion"])
        cmp("reservoir", of.reservoirs[cur], h["reservoirs"][cur]); cmp("reservoir_gi", of.reservoirs_gi[cur], h["reservoirs_gi"][cur])
        for nm, a, b in (("reservoir", of.reservoirs[cur], h["reservoirs"][cur]), ("reservoir_gi", of.reservoirs_gi[cur], h["reservoirs_gi"][cur])):
            av = a.view(np.uint32).reshape(-1, 12); bv = b.view(np.uint32).reshape(-1, 12)
            rows = np.nonzero((av != bv).any(axis=1))[0][:4]
            for r in rows:
                print("   DIFF", nm, "pixel", r % W, r // W, "oracle", a[r], "gpu", b[r], "hex", [hex(x) for x in av[r]], [hex(x) for x in bv[r]])
        nd = cmp("raw_color", of.raw_color, h["raw_color"])
        d = of.raw_color[:, :3].astype(np.float64) - h["raw_color"][:, :3].astype(np.float64)
        bad = np.isnan(d).any(axis=1)
        print("   rmse %.3e  maxabs %.3e  nan-pixels %d  differing pixels %d" % (np.sqrt(np.mean(d[~bad] ** 2)), np.abs(d[~bad]).max(), bad.sum(), (np.abs(d).max(axis=1) > 0).sum()))
    return osc, gsc, of, gf

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "cornell"
    if which == "cornell":
        compare(scenes.cornell_box(), 256, 256, 3)
        compare(scenes.cornell_glass_mirror(), 256, 256, 3)
    elif which == "knot":
        compare(scenes.torus_knot(), 480, 270, 2)
    elif which == "height":
        compare(scenes.heightfield(708), 480, 270, 2)
    elif which == "heightfull":
        compare(scenes.heightfield(708), 1920, 1080, 5)
