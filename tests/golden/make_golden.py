"""Generates tests/golden/pass_*.npz with the CPU oracle (oracle/), our restatement of the reference
shaders. The reference itself cannot run here (Rust + Slang + Vulkan RT) and holds no
golden images, so these fixtures pin oracle and HIP kernels to EACH OTHER and guard regressions;
they do not pin either to the reference ("parity unpinned").

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import binding as ob  # noqa: E402
from sunray_amd import abi, scenes  # noqa: E402

CASES = {
    "cornell_box": (scenes.cornell_box, 48, 48, 3, None),
    "cornell_glass_mirror": (scenes.cornell_glass_mirror, 48, 48, 2, None),
    "cornell_box_norestir": (scenes.cornell_box, 48, 48, 2, dict(enable_restir=0, max_bounces=3, shadow_bounces=3)),
    "atrium_textured": (lambda: scenes.atrium(columns_per_side=3, col_segments=12, col_rings=3, floor_div=6, tex=32, n_lamps=4), 56, 40, 2, None),
}
# full frames incl. the post-RT compute chain (temporal accumulation -> 4x a-trous -> tonemap), RGBA8 output
POST_CASES = {
    "cornell_glass_mirror_post": (scenes.cornell_glass_mirror, 40, 40, 6),
}


def make_config(over):
    cfg = abi.SrTraceConfig.reference()
    for k, v in (over or {}).items():
        setattr(cfg, k, v)
    return cfg


def render(name):
    fn, W, H, frames, over = CASES[name]
    desc = fn()
    cfg = make_config(over)
    s = ob.OracleScene().load(desc)
    fr = ob.HostFrame(W, H, scenes.white_noise_rgba8())
    prev, out = None, {}
    for f in range(frames):
        m = ob.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        if cfg.enable_restir:
            s.trace_ris(fr, m, f, cfg)
        s.trace_final(fr, m, f, cfg)
        cur = f & 1
        out["f%d_raw_color" % f] = fr.raw_color.copy()
        if cfg.enable_restir:
            out["f%d_depth" % f] = fr.depth.copy()
            out["f%d_normal" % f] = fr.normal.copy()
            out["f%d_diffuse" % f] = fr.diffuse.copy()
            out["f%d_motion" % f] = fr.motion.copy()
            out["f%d_reservoir" % f] = fr.reservoirs[cur].view(np.uint32).copy()
            out["f%d_reservoir_gi" % f] = fr.reservoirs_gi[cur].view(np.uint32).copy()
    return out


def render_post(name):
    fn, W, H, frames = POST_CASES[name]
    desc = fn()
    s = ob.OracleScene().load(desc)
    fr = ob.HostFrame(W, H, scenes.white_noise_rgba8())
    prev, out = None, {}
    for f in range(frames):
        m = ob.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        s.trace_ris(fr, m, f)
        s.trace_final(fr, m, f)
        ob.post_chain(fr, f)
        if f >= frames - 3:   # temporal accumulation only blends history from frame_count > 2 on
            out["f%d_accum" % f] = fr.accum[f % 2].copy()
            out["f%d_denoise" % f] = fr.denoise[1].copy()
            out["f%d_output" % f] = fr.output.copy()
    return out


if __name__ == "__main__":
    for name in POST_CASES:
        np.savez_compressed(os.path.join(HERE, "pass_%s.npz" % name), **render_post(name))
        print("wrote", name)
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, "pass_%s.npz" % name), **render(name))
        print("wrote", name)
