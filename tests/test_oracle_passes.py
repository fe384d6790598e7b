"""The oracle's two passes against the committed golden fixtures (regression pin) and against the
semantic facts SURVEY.md §8a lists for ray_gen_ris / ray_gen_final."""
import os
import sys

import numpy as np
import pytest

from sunray_amd import abi, scenes

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)
import make_golden  # noqa: E402


@pytest.mark.parametrize("name", list(make_golden.CASES))
def test_oracle_reproduces_golden(oracle, name):
    want = np.load(os.path.join(GOLDEN, "pass_%s.npz" % name))
    got = make_golden.render(name)
    assert set(got) == set(want.files)
    for k in want.files:
        assert np.array_equal(np.asarray(got[k]).view(np.uint8), want[k].view(np.uint8)), k


def test_frame0_semantics(oracle, blue_noise):
    """Frame 0: prev_view_proj = 0 -> motion = inUV + 2 on geometry; sky pixels get the
    sentinel G-buffer, an empty DI reservoir and an untouched GI reservoir (ray_gen_ris.slang:143-172)."""
    desc = scenes.cornell_box()
    desc.camera_pos = (0.0, 1.0, 6.5)  # far enough that the sky is visible around the open box
    W = H = 64
    s = oracle.OracleScene().load(desc)
    fr = oracle.HostFrame(W, H, blue_noise)
    fr.reservoirs_gi[0]["M"] = 123.0  # canary
    m = oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
    s.trace_ris(fr, m, 0)
    s.trace_final(fr, m, 0)
    depth = fr.depth.view(np.float16).astype(np.float32)
    sky = np.isinf(depth)
    assert sky.any() and (~sky).any()
    assert (fr.normal[sky] == 0).all() and (fr.diffuse[sky] == 0).all() and (fr.motion[sky] == 0).all()
    assert (fr.reservoirs[0]["W"][sky] == 0).all() and (fr.reservoirs[0]["M"][sky] == 0).all()
    assert (fr.reservoirs_gi[0]["M"][sky] == 123.0).all()          # stale by design
    assert (fr.reservoirs_gi[0]["M"][~sky] != 123.0).all()
    assert not fr.raw_color[sky][:, :3].any() and (fr.raw_color[:, 3] == 1.0).all()
    mv = fr.motion[~sky].view(np.float16).reshape(-1, 2).astype(np.float32)
    ys, xs = np.nonzero((~sky).reshape(H, W))
    assert np.allclose(mv[:, 0], (xs + 0.5) / W + 2, atol=2e-3) and np.allclose(mv[:, 1], (ys + 0.5) / H + 2, atol=2e-3)
    # geometry pixels: RIS ran 16 candidates (M = 16) wherever roughness > 0.2
    assert (fr.reservoirs[0]["M"][~sky] == 16.0).all()
    assert (fr.raw_color[:, :3] <= 10.0).all() and np.isfinite(fr.raw_color).all()  # radiance cap :431


def test_temporal_reuse_kicks_in_after_frame0(oracle, blue_noise):
    desc = scenes.cornell_box()
    W = H = 48
    s = oracle.OracleScene().load(desc)
    fr = oracle.HostFrame(W, H, blue_noise)
    prev = None
    Ms = []
    for f in range(3):
        m = oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        s.trace_ris(fr, m, f)
        s.trace_final(fr, m, f)
        Ms.append(fr.reservoirs[f & 1]["M"].copy())
    assert Ms[0].max() == 16.0 and Ms[1].max() > 16.0 and Ms[1].max() <= 26.0  # M <= 16 + min(M_hist, 10)
    # static camera: motion vector ~ 0 from frame 1 on
    mv = fr.motion.view(np.float16).astype(np.float32)
    assert np.abs(mv).max() < 2e-3


def test_ray_counts_are_as_survey_estimates(oracle, blue_noise):
    """~10 ray queries / pixel / frame in a diffuse scene."""
    desc = scenes.cornell_box()
    W = H = 64
    s = oracle.OracleScene().load(desc)
    fr = oracle.HostFrame(W, H, blue_noise)
    m = oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
    s.reset_counters()
    s.trace_ris(fr, m, 0)
    s.trace_final(fr, m, 0)
    c = s.counters()
    per_px = (c.closest_queries + c.any_queries) / (W * H)
    assert 4.0 < per_px < 14.0, per_px
    assert c.boxes_tested > 0 and c.tris_tested > 0


def test_tiles_compose_to_the_full_frame(oracle, blue_noise):
    """Pixels are keyed by global coordinates: rendering row bands separately equals one full launch
    (the property multi-GPU tiling relies on). With ReSTIR the RIS pass must cover the
    whole image before any final tile runs (spatial reuse reads neighbours)."""
    desc = scenes.cornell_box()
    W, H = 40, 36
    s = oracle.OracleScene().load(desc)
    m = oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
    full = oracle.HostFrame(W, H, blue_noise)
    s.trace_ris(full, m, 0); s.trace_final(full, m, 0)
    tiled = oracle.HostFrame(W, H, blue_noise)
    for y0, h in ((0, 10), (10, 17), (27, 9)):
        s.trace_ris(tiled, m, 0, tile=(y0, h))
    for y0, h in ((27, 9), (0, 10), (10, 17)):
        s.trace_final(tiled, m, 0, tile=(y0, h))
    assert np.array_equal(full.raw_color.view(np.uint32), tiled.raw_color.view(np.uint32))
    assert np.array_equal(full.reservoirs[0].view(np.uint32), tiled.reservoirs[0].view(np.uint32))
