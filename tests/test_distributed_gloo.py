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


def _worker(rank, world, port, W, H, frames, out_dir):
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
    # uneven, cost-balanced strips + the double-buffered asynchronous gather bench.py uses (gloo works on host tensors)
    bounds = sd.balanced_bounds(np.concatenate([np.ones(H // 2), np.full(H - H // 2, 3.0)]), world, min_rows=4)
    assert bounds[0] == 0 and bounds[-1] == H and bounds[1] > H // 2      # the cheap top half makes the first strip taller
    pipe = sd.GatherPipeline(W, H, world, rank, "cpu", bounds=bounds)
    pending = []
    for f in range(frames):
        m = ob.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        s.reset_counters()
        sd.render_strip(s, fr, m, f, cfg, world, rank, bounds=bounds)
        pending.append(pipe.submit(torch.from_numpy(fr.raw_color)))        # frame f's gather is in flight while f+1 is traced
        if len(pending) == 2:
            images.append(pipe.image(pending.pop(0)).numpy().copy())
        # the synchronous equal-split helper still gathers the same strips when asked to
        if f == 0:
            assert sd.gather_strips(torch.from_numpy(fr.raw_color), W, H, world, rank).shape == (H * W, 4)
        # max-over-ranks reduction used for timing in bench.py
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
    while pending:
        images.append(pipe.image(pending.pop(0)).numpy().copy())
    dist.barrier()
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.stack(images))
    dist.destroy_process_group()


def test_strip_helpers():
    from sunray_amd import distributed as sd
    assert [sd.strip_rows(1080, 8, r) for r in range(8)] == [(135 * r, 135) for r in range(8)]
    assert [sd.strip_rows(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 3), (9, 1)]
    assert sd.strip_rows(2, 4, 3) == (2, 0)
    assert sd.halo_bands(1080, 0, 135) == [(135, 30)]
    assert sd.halo_bands(1080, 135, 135) == [(105, 30), (270, 30)]
    assert sd.halo_bands(1080, 945, 135) == [(915, 30)]
    assert sd.halo_bands(40, 10, 10) == [(0, 10), (20, 20)]
    assert sd.halo_bands(40, 40, 0) == []
    # cost-balanced cuts: equal cost -> equal strips; a cheap top third -> a taller first strip, capped at 2.5x the equal share
    assert sd.balanced_bounds(np.ones(1080), 8) == [0, 136, 271, 406, 541, 676, 811, 946, 1080]
    cost = np.concatenate([np.full(300, 1.0), np.full(780, 4.0)])
    b = sd.balanced_bounds(cost, 8)
    heights = [b[i + 1] - b[i] for i in range(8)]
    assert b[0] == 0 and b[-1] == 1080 and all(h >= 8 for h in heights) and max(heights) <= 338
    shares = [cost[b[i]:b[i + 1]].sum() for i in range(8)]
    assert max(shares) / np.mean(shares) < 1.02 < (4.0 * 135) / (cost.sum() / 8)      # equal rows would be 26 % off
    assert max(sd.balanced_bounds(cost, 8, max_share=1.5)[i + 1] - sd.balanced_bounds(cost, 8, max_share=1.5)[i] for i in range(8)) <= 203
    assert sd.balanced_bounds(np.zeros(10), 4) == [0, 3, 6, 8, 10] and sd.balanced_bounds(np.ones(5), 8)[-1] == 5
    # feedback balancer: a synthetic machine whose ranks cost `sum(rows) + a floor for the strip holding rows 300..400`
    def machine(b):
        return [cost[b[i]:b[i + 1]].sum() / 400.0 + (0.4 if b[i] < 400 and b[i + 1] > 300 else 0.0) for i in range(8)]
    rc, bb, best = cost.copy(), sd.balanced_bounds(cost, 8), None
    first = max(machine(bb))
    for _ in range(4):
        p = machine(bb)
        best = min(best or 1e9, max(p))
        rc, bb = sd.refine_bounds(rc, bb, p)
        assert bb[0] == 0 and bb[-1] == 1080 and all(bb[i + 1] - bb[i] >= 32 for i in range(8))
    assert best < first * 0.95                      # the slow strip was shrunk
    depth = np.full((4, 6), 0x7C00, dtype=np.uint16); depth[2:] = 0x4000
    assert list(sd.row_cost_from_depth(depth, 6, 4)) == [6.0, 6.0, 24.0, 24.0]


@pytest.mark.timeout(300)
def test_two_rank_gloo_equals_single_process(oracle, tmp_path):
    import torch.multiprocessing as mp
    W, H, frames = 40, 72, 3   # strips of 36 rows: the 30-row halo does NOT cover the whole image
    port = _free_port()
    mp.spawn(_worker, args=(2, port, W, H, frames, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0.view(np.uint32), r1.view(np.uint32))   # every rank holds the full image
    from sunray_amd import abi, scenes
    desc = scenes.cornell_box()
    s = oracle.OracleScene().load(desc)
    fr = oracle.HostFrame(W, H, scenes.white_noise_rgba8())
    prev = None
    for f in range(frames):
        m = oracle.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        s.trace_ris(fr, m, f); s.trace_final(fr, m, f)
        assert np.array_equal(fr.raw_color.view(np.uint32), r0[f].view(np.uint32)), "frame %d" % f
