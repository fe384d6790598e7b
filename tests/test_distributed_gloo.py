"""world_size-2 `gloo` test of the tile-parallel path on CPU: the same strip / halo /
gather code bench.py runs on N GPUs, with the oracle standing in for the HIP kernels. The 2-rank
result must equal the single-process frame bit for bit over several frames (static camera)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _as_tensor(buf):
    """torch view of a numpy host-frame buffer (48-byte reservoir records -> [n, 12] int32), sharing its memory."""
    import torch
    if buf.dtype.fields is not None:
        return torch.from_numpy(buf.view(np.int32).reshape(-1, 12))
    return torch.from_numpy(buf)


def _camera(desc, f, moving):
    """(position, target): a static camera, or one that slides sideways a third of a unit per frame (position and target
    alike, so reprojection moves last frame's history by several pixels)."""
    if not moving:
        return desc.camera_pos, desc.camera_target
    dx = -0.6 + 0.3 * f
    return (desc.camera_pos[0] + dx, desc.camera_pos[1], desc.camera_pos[2]), (desc.camera_target[0] + dx, desc.camera_target[1], desc.camera_target[2])


def _worker(rank, world, port, W, H, frames, out_dir, axis, moving, motion_halo):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import binding as ob
    from sunray_amd import abi, distributed as sd, scenes
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    desc = scenes.cornell_box()
    bn = scenes.white_noise_rgba8()
    s = ob.OracleScene().load(desc)
    fr = ob.HostFrame(W, H, bn)
    cfg = abi.SrTraceConfig.reference()
    prev = None
    images = []
    # uneven, cost-balanced strips (equal ones for the moving camera) + the double-buffered asynchronous gather bench.py uses (gloo works on host tensors)
    L = W if axis == "cols" else H
    bounds = None
    if not moving:
        bounds = sd.balanced_bounds(np.concatenate([np.ones(L // 2), np.full(L - L // 2, 3.0)]), world, min_size=4)
        assert bounds[0] == 0 and bounds[-1] == L and bounds[1] > L // 2  # the cheap first half makes the first strip larger
    part = sd.Partition(W, H, world, axis, bounds)
    pipe = sd.GatherPipeline(part, rank, "cpu")
    pending = []
    counted = 0
    for f in range(frames):
        cam_pos, cam_target = _camera(desc, f, moving)
        m = ob.camera_matrices(cam_pos, cam_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        s.reset_counters()
        sd.render_strip(s, fr, m, f, cfg, part, rank, motion_halo=motion_halo, as_tensor=_as_tensor)
        c = s.counters()
        counted += c.closest_queries + c.any_queries
        pending.append(pipe.submit(torch.from_numpy(fr.raw_color)))        # frame f's gather is in flight while f+1 is traced
        if len(pending) == 2:
            images.append(pipe.image(pending.pop(0)).numpy().copy())
        # max-over-ranks reduction used for timing in bench.py
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
    while pending:
        images.append(pipe.image(pending.pop(0)).numpy().copy())
    total = torch.tensor([float(counted)], dtype=torch.float64)
    dist.all_reduce(total, op=dist.ReduceOp.SUM)                           # halo pixels are uncounted: the sum is the 1-GPU count
    dist.barrier()
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.stack(images))
    np.save(os.path.join(out_dir, "rays%d.npy" % rank), total.numpy())
    dist.destroy_process_group()


def test_strip_helpers():
    from sunray_amd import distributed as sd
    p = sd.Partition(1920, 1080, 8)                                   # column strips by default
    assert p.axis == "cols" and [p.span(r) for r in range(8)] == [(240 * r, 240) for r in range(8)]
    assert p.tile(240, 240) == (0, 1080, 240, 240) and p.grown(0, 30) == (0, 270) and p.grown(3, 30) == (690, 300) and p.grown(7, 30) == (1650, 270)
    r = sd.Partition(1920, 1080, 8, "rows")
    assert [r.span(k) for k in range(8)] == [(135 * k, 135) for k in range(8)] and r.tile(135, 135) == (135, 135, 0, 1920)
    assert r.grown(0, 30) == (0, 165) and r.grown(1, 30) == (105, 195) and r.grown(7, 30) == (915, 165)
    q = sd.Partition(7, 10, 4, "rows")
    assert [q.span(k) for k in range(4)] == [(0, 3), (3, 3), (6, 3), (9, 1)] and sd.Partition(7, 2, 4, "rows").span(3) == (2, 0)
    with pytest.raises(ValueError):
        sd.Partition(100, 100, 2, "cols", [0, 60])
    with pytest.raises(ValueError):
        sd.Partition(100, 100, 2, "diag")
    # cost-balanced cuts: equal cost -> equal strips; a cheap first third -> a larger first strip, capped at 2.5x the equal share
    assert sd.balanced_bounds(np.ones(1080), 8) == [0, 136, 271, 406, 541, 676, 811, 946, 1080]
    cost = np.concatenate([np.full(300, 1.0), np.full(780, 4.0)])
    b = sd.balanced_bounds(cost, 8)
    sizes = [b[i + 1] - b[i] for i in range(8)]
    assert b[0] == 0 and b[-1] == 1080 and all(h >= 8 for h in sizes) and max(sizes) <= 338
    shares = [cost[b[i]:b[i + 1]].sum() for i in range(8)]
    assert max(shares) / np.mean(shares) < 1.02 < (4.0 * 135) / (cost.sum() / 8)      # equal strips would be 26 % off
    assert max(sd.balanced_bounds(cost, 8, max_share=1.5)[i + 1] - sd.balanced_bounds(cost, 8, max_share=1.5)[i] for i in range(8)) <= 203
    assert sd.balanced_bounds(np.zeros(10), 4) == [0, 3, 6, 8, 10] and sd.balanced_bounds(np.ones(5), 8)[-1] == 5
    # per-column / per-row cost from the library's per-tile cycle counts
    tiles = np.arange(6, dtype=np.float64).reshape(2, 3)                  # 2 tile rows x 3 tile columns
    assert list(sd.axis_cost_from_tiles(tiles.reshape(-1), 3, "cols", 20)) == [3 / 8.0] * 8 + [5 / 8.0] * 8 + [7 / 8.0] * 4
    assert list(sd.axis_cost_from_tiles(tiles.reshape(-1), 3, "rows", 12)) == [3 / 8.0] * 8 + [12 / 8.0] * 4
    # temporal-history exchange plan: who owns the band beyond a rank's traced region
    assert sd.history_exchange_plan(p, 0) == [] and sd.history_exchange_plan(sd.Partition(64, 64, 1), 8) == []
    plan = sd.history_exchange_plan(p, 10)
    assert (1, 0, 270, 10) in plan and (0, 1, 200, 10) in plan and (2, 1, 510, 10) in plan and (6, 7, 1640, 10) in plan
    assert len(plan) == 14 and all(src != dst for src, dst, _, _ in plan)
    narrow = sd.history_exchange_plan(sd.Partition(100, 50, 4, "cols", [0, 40, 45, 60, 100]), 20)   # bands that span two owners
    assert (1, 0, 70 - 0, 0) not in narrow and (2, 0, 70, 0) not in narrow
    got = sorted((src, x0, n) for src, dst, x0, n in narrow if dst == 0)
    assert got == [(3, 70, 20)]                                            # rank 0 traced [0, 70); [70, 90) belongs to rank 3


def _single_process_frames(oracle, W, H, frames, moving):
    from sunray_amd import scenes
    desc = scenes.cornell_box()
    s = oracle.OracleScene().load(desc)
    fr = oracle.HostFrame(W, H, scenes.white_noise_rgba8())
    prev, out, rays = None, [], 0
    for f in range(frames):
        cam_pos, cam_target = _camera(desc, f, moving)
        m = oracle.camera_matrices(cam_pos, cam_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        s.reset_counters()
        s.trace_ris(fr, m, f); s.trace_final(fr, m, f)
        c = s.counters()
        rays += c.closest_queries + c.any_queries
        out.append(fr.raw_color.copy())
    return out, rays


def _run_ranks(tmp_path, world, W, H, frames, axis, moving, motion_halo):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, W, H, frames, str(tmp_path), axis, moving, motion_halo), nprocs=world, join=True)
    imgs = [np.load(tmp_path / ("rank%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(imgs[0].view(np.uint32), imgs[r].view(np.uint32))   # every rank holds the full image
    return imgs[0], float(np.load(tmp_path / "rays0.npy")[0])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("axis,W,H", [("cols", 72, 40), ("rows", 40, 72)])   # strips of ~36: the 30-pixel halo does NOT cover the image
def test_two_rank_gloo_equals_single_process(oracle, tmp_path, axis, W, H):
    frames = 3
    got, rays = _run_ranks(tmp_path, 2, W, H, frames, axis, False, 0)
    want, want_rays = _single_process_frames(oracle, W, H, frames, False)
    for f in range(frames):
        assert np.array_equal(want[f].view(np.uint32), got[f].view(np.uint32)), "frame %d" % f
    assert rays == want_rays                                                 # halo pixels do not count their rays


@pytest.mark.timeout(300)
def test_moving_camera_needs_and_gets_history_exchange(oracle, tmp_path):
    """Under camera motion temporal reuse reads last frame's reservoirs at the reprojected pixel: a halo pixel's history can
    lie in pixels the rank never traced. With the owners' bands exchanged after every RIS pass (motion_halo) three ranks
    reproduce the single-process frames bit for bit over 5 frames; without the exchange they do not (so the test does
    exercise the hazard)."""
    W, H, frames = 120, 48, 5           # three column strips of ~40: strip + 30-pixel halo leaves pixels nobody but the owner traces
    want, _ = _single_process_frames(oracle, W, H, frames, True)
    a = tmp_path / "with"; a.mkdir()
    got, _ = _run_ranks(a, 3, W, H, frames, "cols", True, 16)
    for f in range(frames):
        assert np.array_equal(want[f].view(np.uint32), got[f].view(np.uint32)), "frame %d (history exchanged)" % f
    b = tmp_path / "without"; b.mkdir()
    stale, _ = _run_ranks(b, 3, W, H, frames, "cols", True, 0)
    assert any(not np.array_equal(want[f].view(np.uint32), stale[f].view(np.uint32)) for f in range(frames))
