#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X ray-tracing hot path.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`: "Mray/s + ms/frame at 1920x1080 1spp, 1M-tri scene"): one step =
one frame = the two ray-tracing passes of the reference (raytracing_ris + raytracing_final,
src/lib.rs:1549-1574) with the reference's constants (1 sample per pixel, 10 bounces, 16 RIS
candidates) at 1920x1080 on the 999 714-triangle procedural heightfield (SURVEY.md §8d config 3
scene; no 1M-triangle asset exists offline). Scene, BVH, camera matrices and every frame buffer are
resident in HBM before the timed region. Rays are counted by the kernels (closest-hit + any-hit
queries actually issued). With N GPUs the frame's rows are split into N strips (strong scaling of one
frame; sunray_amd/distributed.py) and the radiance strips are all-gathered over RCCL every step.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured-achievable)
KIND_RIS, KIND_FINAL = 0, 1


def algorithmic_bytes(c, n_pixels, which):
    """B_ray = 32*N_boxes + 48*N_tris + 32 (ray) + 16 (hit) per query, +460 B per
    closest-hit surface fetch (128 MeshInfo + 12 indices + 288 vertices + 32 payload), plus the
    per-pixel frame-buffer traffic of the pass (reference formats)."""
    rays = c.closest_queries + c.any_queries
    b = 32 * c.boxes_tested + 48 * c.tris_tested + 48 * rays + 460 * c.closest_queries
    if which == KIND_RIS:
        b += n_pixels * (14 + 96)            # G-buffer 2+4+4+4 B written + DI and GI reservoirs written
    else:
        b += n_pixels * (16 + 9 * 48 + 9 * 6)  # radiance written + up to 9 reservoir gathers + normal/depth texels
    return b


def pmc_traffic(dom, world, W, H, grid):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/
    r01_pmc_hbm_traffic.json: FETCH_SIZE / WRITE_SIZE collected in separate rocprofv3 --pmc runs of this
    very script and corrected as MI355X_MICROARCH.md §HBM prescribes). Counters cannot be read from inside
    a normal run, so this is the profiled value for the default single-GPU workload, else null."""
    if world != 1 or (W, H, grid) != (1920, 1080, 708):
        return None
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
    try:
        with open(path) as fh:
            k = json.load(fh)["kernels"]["final_kernel" if dom == KIND_FINAL else "ris_kernel"]
        return k["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(desc, W, H, blue_noise, budget_s=12.0, max_frames=8):
    """The oracle (our CPU restatement: the reference has no CPU path) timed on this host's cores on
    the same scene / extent / constants; whole frames until ~budget_s of work."""
    from oracle import binding as ob
    threads = ob.usable_cores()   # affinity capped by the cgroup CPU quota (16 on a 1-GPU box)
    ob.set_threads(threads)
    s = ob.OracleScene().load(desc)
    fr = ob.HostFrame(W, H, blue_noise)
    prev, rays, t_total, frames = None, 0, 0.0, 0
    while frames < max_frames and (t_total < budget_s or frames < 2):
        m = ob.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        s.reset_counters()
        t0 = time.perf_counter()
        s.trace_ris(fr, m, frames)
        s.trace_final(fr, m, frames)
        t_total += time.perf_counter() - t0
        c = s.counters()
        rays += c.closest_queries + c.any_queries
        frames += 1
    return {"value": rays / t_total / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port",
            "sample": "%d full frames at %dx%d of the same scene and constants, OpenMP over scanlines, %.1f s"
                      % (frames, W, H, t_total)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--grid", type=int, default=708, help="heightfield grid: 2*(grid-1)^2 triangles")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from sunray_amd import abi, distributed as sd, runtime as rt, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch --gpus %d through torch.distributed.run (one process per GPU)" % args.gpus)
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # Rehearsal switch for a box with ONE GPU: every rank renders on cuda:0 and the gather goes through
    # gloo (host staging). Never used by the driver's runs.
    rehearsal = os.environ.get("SUNRAY_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = "cuda:%d" % dev_index
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(device))

    W, H = args.width, args.height
    desc = scenes.heightfield(args.grid)
    blue_noise = scenes.white_noise_rgba8()
    scene = rt.Scene(dev_index).load(desc)
    st = scene.bvh_stats()
    frame = rt.DeviceFrame(W, H, blue_noise, device=device)
    cfg = abi.SrTraceConfig.reference()
    import copy
    bounds = None
    if world > 1:
        # Strip boundaries of equal MEASURED cost (sky rows are far cheaper than surface rows, rows near the horizon the most
        # expensive), from a few uncounted whole-frame frames on a throwaway frame buffer. The cut is frozen before frame 0
        # (a rank owns the temporal history of its rows + halo); all ranks must agree on it, so rank 0's cut is broadcast.
        cal = rt.DeviceFrame(W, H, blue_noise, device=device)
        ccfg = copy.copy(cfg)
        ccfg.flags = cfg.flags | abi.TRACE_FLAG_UNCOUNTED
        cprev = None
        for f in range(4):      # a few frames: the final pass's visibility rays appear as the reservoirs fill up
            cm = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, cprev)
            cprev = list(cm.view_proj)
            scene.trace_ris(cal, cm, f, ccfg)
            scene.trace_final(cal, cm, f, ccfg)
        # measured cycles per tile row of both passes (the data behind the library's tile schedule), spread over pixel rows
        tile_rows = scene.tile_row_costs(0, W, 0, H) + scene.tile_row_costs(1, W, 0, H)
        row_cost = np.repeat(tile_rows / 8.0, 8)[:H]
        bounds = sd.balanced_bounds(row_cost, world)
        blist = [bounds]
        dist.broadcast_object_list(blist, src=0)      # cycle counts differ a little from GPU to GPU: take rank 0's cut
        bounds = [int(v) for v in blist[0]]
        # Feedback rounds: a rank's share is about one round of waves, so its step time is not proportional to the cycle sum
        # of its rows. Every rank measures its real step time with the current cut (two frames in flight, as in the timed
        # region), rank 0 re-weights the rows of the slow ranks and cuts again; the cut with the smallest maximum is kept.
        if os.environ.get("SUNRAY_BENCH_FEEDBACK", "1") == "1":
            cpipe = sd.FramePipeline(cal, rt.DeviceFrame(W, H, blue_noise, device=device))
            cstate = {"f": 4, "prev": cprev}

            def cal_steps(n, b, evs=None):
                for i in range(n):
                    cm = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, cstate["prev"])
                    cstate["prev"] = list(cm.view_proj)
                    cpipe.step(scene, cm, cstate["f"], ccfg, world, rank, bounds=b,
                               after_final=(lambda g, i=i: evs[i].record(cpipe.s_final)) if evs is not None else None)
                    cstate["f"] += 1

            best = (float("inf"), bounds)
            for it in range(4):
                cal_steps(4, bounds)                       # refill the temporal history of rows that changed hands
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
                cal_steps(13, bounds, evs)
                torch.cuda.synchronize()
                gaps = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(12))
                mine = torch.tensor([gaps[6]], dtype=torch.float64, device="cpu" if rehearsal else device)   # median step time
                allp = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allp, mine)
                periods = [float(x.item()) for x in allp]
                if max(periods) < best[0]:
                    best = (max(periods), bounds)
                if it == 3:
                    break
                blist = [None]
                if rank == 0:
                    row_cost, nb = sd.refine_bounds(row_cost, bounds, periods)
                    blist = [nb]
                dist.broadcast_object_list(blist, src=0)
                bounds = [int(v) for v in blist[0]]
            bounds = best[1]
            del cpipe
        del cal
    # the gather of frame f overlaps the tracing of frame f+1 (RCCL runs on its own stream); rehearsal: gloo on host copies
    pipe = sd.GatherPipeline(W, H, world, rank, "cpu" if rehearsal else device, bounds=bounds) if world > 1 else None

    state = {"prev": None, "frame": 0, "last": frame}
    # Two frames in flight per GPU (RIS of frame f+1 overlaps the draining final pass of frame f): a rank's share of a frame
    # is about one round of waves, i.e. latency-bound on its own. On by default for N > 1; N = 1 runs the passes back to
    # back so that the per-launch durations behind `roofline` are those of undisturbed kernels.
    pipelined = os.environ.get("SUNRAY_BENCH_PIPELINE", "1" if world > 1 else "0") == "1"
    fpipe = sd.FramePipeline(frame, rt.DeviceFrame(W, H, blue_noise, device=device)) if pipelined else None

    def submit_gather(fr):
        if world > 1:
            if rehearsal:
                torch.cuda.current_stream().synchronize()
                pipe.submit(fr.raw_color.cpu())
            else:
                pipe.submit(fr.raw_color)

    def step():
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, state["prev"])
        state["prev"] = list(m.view_proj)
        if fpipe is not None:
            state["last"] = fpipe.step(scene, m, state["frame"], cfg, world, rank, bounds=bounds, after_final=submit_gather)
        else:
            sd.render_strip(scene, frame, m, state["frame"], cfg, world, rank, abi.TRACE_FLAG_UNCOUNTED, bounds=bounds)
            submit_gather(frame)
        state["frame"] += 1

    def fence():
        if world > 1:
            pipe.wait()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()

    # instrumented replay of ONE frame (untimed): boxes / triangles tested per launch of each pass
    scene.set_instrumented(True)
    per_kind = {}
    m_i = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, state["prev"])
    y0, h = (bounds[rank], bounds[rank + 1] - bounds[rank]) if bounds is not None else sd.strip_rows(H, world, rank)
    halo_cfg = copy.copy(cfg)
    halo_cfg.flags = cfg.flags | abi.TRACE_FLAG_UNCOUNTED
    iframe = fpipe.frames[state["frame"] & 1] if fpipe is not None else frame
    scene.reset_counters()
    scene.trace_ris(iframe, m_i, state["frame"], cfg, tile=(y0, h))
    per_kind[KIND_RIS] = scene.counters()
    if world > 1:   # keep the halo rows' reservoirs current: they are next frame's temporal history
        for band in sd.halo_bands(H, y0, h):
            scene.trace_ris(iframe, m_i, state["frame"], halo_cfg, tile=band)
    scene.reset_counters()
    scene.trace_final(iframe, m_i, state["frame"], cfg, tile=(y0, h))
    per_kind[KIND_FINAL] = scene.counters()
    state["prev"] = list(m_i.view_proj)
    state["frame"] += 1
    scene.set_instrumented(False)
    fence()

    scene.reset_counters()
    scene.enable_timing(os.environ.get("SUNRAY_BENCH_TIMING", "1") == "1")
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    scene.enable_timing(False)
    c = scene.counters()
    ris_ms, ris_n = scene.read_timing(KIND_RIS)
    fin_ms, fin_n = scene.read_timing(KIND_FINAL)

    red_dev = "cpu" if rehearsal else device
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    rays = torch.tensor([float(c.closest_queries + c.any_queries), float(c.closest_queries), float(c.any_queries)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(t_el.item())
    total_rays, total_closest, total_any = [float(x) for x in rays.tolist()]

    # CRC of the last frame's full fp32 radiance image (outside the timed region): N-GPU runs of the same
    # --steps/--warmup must print the same value as the 1-GPU run (tiling is bit-exact).
    import zlib
    final_image = pipe.image() if world > 1 else state["last"].raw_color
    frame_crc = "%08x" % (zlib.crc32(final_image.cpu().numpy().tobytes()) & 0xFFFFFFFF)

    if rank == 0:
        # dominant kernel of this rank: the pass with the larger summed device time
        dom = KIND_FINAL if fin_ms >= ris_ms else KIND_RIS
        dom_ms, dom_n = (fin_ms, fin_n) if dom == KIND_FINAL else (ris_ms, ris_n)
        dom_name = "final_kernel (raytracing_final)" if dom == KIND_FINAL else "ris_kernel (raytracing_ris)"
        # with N > 1 the RIS pass is 1 strip launch + up to 2 halo launches per step: price it per step
        avg_ms = dom_ms / max(dom_n, 1) if (dom == KIND_FINAL or world == 1) else dom_ms / args.steps
        bytes_per_launch = algorithmic_bytes(per_kind[dom], W * h, dom)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0   # SUNRAY_BENCH_TIMING=0 (tuning): no per-launch events
        ck = per_kind[dom]
        nq = max(ck.closest_queries + ck.any_queries, 1)
        out = {
            "metric": "Mray/s (closest-hit + any-hit queries issued per second), 1920x1080, 1 spp, 1M-triangle scene",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "procedural heightfield %d triangles (%d BVH nodes), %dx%d, 1 spp, full reference frame per step: "
                            "raytracing_ris + raytracing_final, 10 bounces, 16 RIS candidates, ReSTIR DI+GI on"
                            % (st.n_triangles, st.n_nodes, W, H),
                "rays_per_frame": total_rays / args.steps,
                "closest_per_frame": total_closest / args.steps,
                "any_per_frame": total_any / args.steps,
                "parallelism": ("rows split into %d cost-balanced strips %s, RIS halo %d rows recomputed, radiance strips all-gathered over RCCL "
                                "asynchronously (frame f's gather overlaps frame f+1)" % (world, [bounds[i + 1] - bounds[i] for i in range(world)], sd.SPATIAL_HALO))
                               if world > 1 else "single GPU",
                "frames_in_flight": 2 if pipelined else 1,
                "bvh_build_ms_host": st.build_ms,
                "last_frame_crc32": frame_crc,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(dom, world, W, H, args.grid),
                "avg_launch_ms": avg_ms,
                "launches_timed": dom_n,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "boxes_per_ray": ck.boxes_tested / nq,
                "tris_per_ray": ck.tris_tested / nq,
                "other_pass_avg_ms": (ris_ms / max(ris_n, 1)) if dom == KIND_FINAL else (fin_ms / max(fin_n, 1)),
                "note": "achieved = algorithmic bytes (32 B per box test actually executed, incl. speculative ones) / launch time; the node "
                        "and triangle arrays are served mostly by L2 / Infinity Cache (`traffic` = physical HBM bytes per launch from the "
                        "PMC counters), so frac is not bounded by 1: the pass is latency-bound, not HBM-bound (DESIGN.md section 5)",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, W, H, blue_noise)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
