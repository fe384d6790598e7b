#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X ray-tracing hot path.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...            (no launcher: starts torch.distributed.run with N ranks as a child and relays the line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    SUNRAY_BENCH_ONE_DEVICE=1 python bench.py --gpus 4   (rehearsal on a 1-GPU box: every rank on cuda:0, gather over gloo)

Workload (BASELINE.json `metric`: "Mray/s + ms/frame at 1920x1080 1spp, 1M-tri scene"): one step =
one frame = the two ray-tracing passes of the reference (raytracing_ris + raytracing_final,
src/lib.rs:1549-1574) with the reference's constants (1 sample per pixel, 10 bounces, 16 RIS
candidates) at 1920x1080 on the 999 714-triangle procedural heightfield (SURVEY.md §8d config 3
scene; no 1M-triangle asset exists offline). Scene, BVH, camera matrices and every frame buffer are
resident in HBM before the timed region. Rays are counted by the kernels (closest-hit + any-hit
queries actually issued). With N GPUs the frame's rows are split into N strips (strong scaling of one
frame; sunray_amd/distributed.py) and the radiance strips are all-gathered over RCCL every step.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
`roofline` holds three physical bounds per pass (HBM traffic, L2-resident gather rate, VALU issue), each a fraction <= 1;
the HBM traffic and the VALU / cache counters are measured live at N = 1: before touching the GPU the process runs four
short `rocprofv3 --pmc` passes of this same workload as child processes (about a minute; --no-pmc skips them).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured-achievable)
L2_GATHER_PEAK_GBS = 17800.0  # MI355X_MICROARCH.md "Indexed rows: gather into LDS": rows served from the XCDs' L2, 16.8-18.8 TB/s chip-wide
N_SIMD = 1024                 # 256 CUs x 4 SIMD-32; one wave64 VALU instruction occupies a SIMD for 2 cycles
KIND_RIS, KIND_FINAL = 0, 1
KERNEL_OF = {KIND_RIS: "ris_kernel", KIND_FINAL: "final_kernel"}

# Counter groups of the live PMC passes (one rocprofv3 --pmc run each: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"; GRBM has its own slots).
PMC_GROUPS = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TA_TA_BUSY_sum"],
]


def survey_bytes(c, n_pixels, which):
    """SURVEY.md §8d's formula, kept for reference: B_ray = 32*N_boxes + 48*N_tris + 32 (ray) + 16 (hit) per query, +460 B
    per closest-hit surface fetch of the REFERENCE layout (128 MeshInfo + 12 indices + 288 vertices + 32 payload), plus
    the per-pixel frame-buffer traffic of the pass. It prices a box test at a BVH2 node's 32 B; the shipped node holds
    four boxes in 64 B, so this figure overstates what the kernels request (see requested_bytes)."""
    rays = c.closest_queries + c.any_queries
    b = 32 * c.boxes_tested + 48 * c.tris_tested + 48 * rays + 460 * c.closest_queries
    if which == KIND_RIS:
        b += n_pixels * (14 + 96)            # G-buffer 2+4+4+4 B written + DI and GI reservoirs written
    else:
        b += n_pixels * (16 + 9 * 48 + 9 * 6)  # radiance written + up to 9 reservoir gathers + normal/depth texels
    return b


def requested_bytes(c, n_pixels, which, reuse=False):
    """ALGORITHMIC bytes of one launch at the sizes of the data layout the kernels actually read:
    16 B per box test (one 64-byte quantised node per four boxes, csrc/bvh_layout.h), 48 B per triangle test, per
    closest hit the 48-byte shade record + 36 B WorldToObject + the 32-byte mesh constants = 116 B, and the per-pixel
    frame-buffer traffic of the pass (with the primary-hit hand-off: 32 B more written by the RIS pass and read by the
    final pass). Rays live in registers (no ray / hit records are read or written)."""
    b = 16 * c.boxes_tested + 48 * c.tris_tested + 116 * c.closest_queries
    if which == KIND_RIS:
        b += n_pixels * (14 + 96 + 2 * 48 + (32 if reuse else 0))   # G-buffer + both reservoirs written, both history reservoirs read
    else:
        b += n_pixels * (16 + 9 * 48 + 9 * 6 + (32 if reuse else 0))
    return b


def live_pmc(argv_tail, timeout_s=75):
    """Physical counters of THIS run's workload, measured now: one `rocprofv3 --pmc <group> -- python3 bench.py ...`
    child per counter group (separate passes, the program itself after `--`), started before this process touches the
    GPU. Returns {kernel: {counter: mean per launch}} plus provenance, or {"error": ...}: the bench line then carries
    nulls instead of stale numbers."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {"error": "this process itself runs under a profiler (no nested counter passes)"}
    work = tempfile.mkdtemp(prefix="sunray_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    out = {"kernels": {}, "passes": []}
    t0 = time.time()
    try:
        for gi, group in enumerate(PMC_GROUPS):
            d = os.path.join(work, "g%d" % gi)
            cmd = [exe, "--pmc"] + group + ["--output-format", "csv", "-d", d, "-o", "p", "--",
                                            "python3", os.path.abspath(__file__)] + argv_tail
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return {"error": "pmc pass %s timed out after %d s" % (group, timeout_s)}
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": "pmc pass %s failed (rc %d): %s" % (group, r.returncode, r.stdout.decode("utf-8", "replace")[-300:])}
            acc = {}
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        k = row["Kernel_Name"]
                        for name in ("ris_kernel<0>", "final_kernel<0>"):
                            if "srd::" + name in k:
                                acc.setdefault((name[:-3], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            for (k, cn), v in acc.items():
                out["kernels"].setdefault(k, {})[cn] = sum(v) / len(v)
                out["kernels"][k]["launches_" + cn] = len(v)
            out["passes"].append(" ".join(["rocprofv3", "--pmc"] + group + ["--", "python3", "bench.py"] + argv_tail))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out["seconds"] = time.time() - t0
    return out


def git_head():
    import subprocess
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              timeout=10).stdout.decode().strip() or None
    except Exception:
        return None


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start `torch.distributed.run` with N ranks as a CHILD process
    (before this process imports torch or touches the GPU: a GPU-initialised process must never exec another program),
    relay rank 0's JSON line, exit with the child's code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    text = r.stdout.decode("utf-8", "replace")
    lines = [ln for ln in text.splitlines() if ln.startswith('{"metric"')]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stdout.write(text)
    if r.returncode != 0 or not lines:
        raise SystemExit(r.returncode or 1)
    raise SystemExit(0)


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
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--grid", type=int, default=708, help="heightfield grid: 2*(grid-1)^2 triangles")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (roofline.traffic and the hbm / valu_issue bounds become null)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)      # does not return

    # Physical counters of this very workload, measured before this process touches the GPU (N = 1 only).
    pmc = None
    import torch      # no GPU call yet: importing pages the libraries in (1-2 minutes on a fresh box), so the children below start fast
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_pmc:
        pmc = live_pmc(["--gpus", "1", "--steps", "3", "--warmup", "3", "--no-cpu-baseline", "--no-pmc",
                        "--width", str(args.width), "--height", str(args.height), "--grid", str(args.grid)])

    import numpy as np
    import torch
    import torch.distributed as dist
    from sunray_amd import abi, distributed as sd, runtime as rt, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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
    cfg.flags |= int(os.environ.get("SUNRAY_BENCH_TRACE_FLAGS", "0"), 0)   # experiments only (e.g. 1 = no ray counting); the default run sets none
    import copy
    axis = os.environ.get("SUNRAY_BENCH_AXIS", "cols")          # column strips (default) or "rows"
    bounds = None
    if world > 1:
        # Strip boundaries of equal MEASURED cost, from a few uncounted whole-frame frames on a throwaway frame buffer: the
        # library records every tile's cycle count for its own tile schedule. Column strips give every rank the same mix of
        # rows (sky / horizon / foreground), so the cut only has to even out the centre-to-edge difference. The cut is
        # frozen before frame 0 (a rank owns the temporal history of its strip + halo); rank 0's cut is broadcast.
        cal = rt.DeviceFrame(W, H, blue_noise, device=device)
        ccfg = copy.copy(cfg)
        ccfg.flags = cfg.flags | abi.TRACE_FLAG_UNCOUNTED
        cprev = None
        for f in range(4):      # a few frames: the final pass's visibility rays appear as the reservoirs fill up
            cm = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, cprev)
            cprev = list(cm.view_proj)
            scene.trace_ris(cal, cm, f, ccfg)
            scene.trace_final(cal, cm, f, ccfg)
        tiles_x = (W + 7) // 8
        tile_costs = scene.tile_costs(0, W, 0, H).astype(np.float64) + scene.tile_costs(1, W, 0, H)
        cost = sd.axis_cost_from_tiles(tile_costs, tiles_x, axis, W if axis == "cols" else H)
        blist = [sd.balanced_bounds(cost, world, min_size=sd.SPATIAL_HALO + 2)]
        dist.broadcast_object_list(blist, src=0)      # cycle counts differ a little from GPU to GPU: take rank 0's cut
        bounds = [int(v) for v in blist[0]]
        del cal
    part = sd.Partition(W, H, world, axis, bounds)
    # the gather of frame f overlaps the tracing of frame f+1 (RCCL runs on its own stream); rehearsal: gloo on host copies
    pipe = sd.GatherPipeline(part, rank, "cpu" if rehearsal else device) if world > 1 else None

    state = {"prev": None, "frame": 0, "last": frame}
    # Two frames in flight per GPU, as the reference renders (MAX_FRAMES_IN_FLIGHT = 2, src/lib.rs:71): raytracing_ris of frame
    # f+1 runs on its own stream while raytracing_final of frame f drains, so the tail of one launch is filled by the head of
    # the next (bit-identical to sequential execution: same CRC). Worth 5-6 % at N = 1 and more on a strip (N > 1: a rank's
    # share is about one round of waves). Per-launch durations then include the time a kernel shares the GPU with its
    # neighbour; SUNRAY_BENCH_PIPELINE=0 runs the passes back to back (undisturbed kernel durations).
    pipelined = os.environ.get("SUNRAY_BENCH_PIPELINE", "1") == "1"
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
            state["last"] = fpipe.step(scene, m, state["frame"], cfg, part, rank, after_final=submit_gather)
        else:
            sd.render_strip(scene, frame, m, state["frame"], cfg, part, rank)
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
    a0, an = part.span(rank)
    own = part.tile(a0, an)
    n_own_pixels = an * (H if axis == "cols" else W)
    halo_cfg = copy.copy(cfg)
    halo_cfg.flags = cfg.flags | abi.TRACE_FLAG_UNCOUNTED
    iframe = fpipe.frames[state["frame"] & 1] if fpipe is not None else frame
    scene.reset_counters()
    scene.trace_ris(iframe, m_i, state["frame"], cfg, tile=own)
    per_kind[KIND_RIS] = scene.counters()
    if world > 1:   # keep the halo's reservoirs current: they are next frame's temporal history
        g0, gn = part.grown(rank, sd.SPATIAL_HALO)
        for b0, bn in ((g0, a0 - g0), (a0 + an, g0 + gn - (a0 + an))):
            if bn > 0:
                scene.trace_ris(iframe, m_i, state["frame"], halo_cfg, tile=part.tile(b0, bn))
    scene.reset_counters()
    scene.trace_final(iframe, m_i, state["frame"], cfg, tile=own)
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

    # the last TIMED frame's full fp32 radiance image, kept on the device (the frames below reuse the buffers); its CRC is taken
    # after all measurements: N-GPU runs of the same --steps/--warmup must print the 1-GPU run's value
    final_image = (pipe.image() if world > 1 else state["last"].raw_color).clone()

    # Undisturbed launch durations (outside the timed region): a few more frames of the same sequence with both passes enqueued
    # back to back on ONE stream, so no launch shares the GPU with its neighbour and the GPU never idles in between (a host
    # synchronisation per frame, or host work before this segment, would let the clocks drop). The per-pass roofline fractions are priced with these — what
    # `rocprofv3 --kernel-trace` shows for SUNRAY_BENCH_PIPELINE=0 — the frame-level ones with ms_per_step of the timed region.
    seq_frames = 8
    fence()
    scene.enable_timing(True)
    for _ in range(seq_frames):
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, state["prev"])
        state["prev"] = list(m.view_proj)
        fr = fpipe.frames[state["frame"] & 1] if fpipe is not None else frame
        sd.render_strip(scene, fr, m, state["frame"], cfg, part, rank)
        submit_gather(fr)
        state["frame"] += 1
    fence()
    scene.enable_timing(False)
    ris_seq_ms, ris_seq_n = scene.read_timing(KIND_RIS)
    fin_seq_ms, fin_seq_n = scene.read_timing(KIND_FINAL)
    import zlib
    frame_crc = "%08x" % (zlib.crc32(final_image.cpu().numpy().tobytes()) & 0xFFFFFFFF)

    red_dev = "cpu" if rehearsal else device
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    rays = torch.tensor([float(c.closest_queries + c.any_queries), float(c.closest_queries), float(c.any_queries), float(c.reused_primary_hits + c.reused_visibility_queries)],
                        dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(t_el.item())
    total_rays, total_closest, total_any, total_reused = [float(x) for x in rays.tolist()]

    if rank == 0:
        # per pass: wall time of a launch inside the timed region (two frames in flight: it shares the GPU with its neighbour)
        # and its undisturbed duration (the synchronised frames after the timed region); with N > 1 the RIS pass may be
        # several launches per step, so it is priced per step
        overlapped_ms = {KIND_RIS: ris_ms / max(args.steps if world > 1 else ris_n, 1), KIND_FINAL: fin_ms / max(fin_n, 1)}
        alone_ms = {KIND_RIS: ris_seq_ms / max(seq_frames if world > 1 else ris_seq_n, 1), KIND_FINAL: fin_seq_ms / max(fin_seq_n, 1)}
        dom = KIND_FINAL if alone_ms[KIND_FINAL] >= alone_ms[KIND_RIS] else KIND_RIS       # dominant kernel: the longer pass
        other = KIND_RIS if dom == KIND_FINAL else KIND_FINAL
        names = {KIND_FINAL: "final_kernel (raytracing_final)", KIND_RIS: "ris_kernel (raytracing_ris)"}
        reuse = frame.primary is not None
        req = {k: requested_bytes(per_kind[k], n_own_pixels, k, reuse) for k in (KIND_RIS, KIND_FINAL)}
        ck = per_kind[dom]
        nq = max(ck.closest_queries + ck.any_queries, 1)
        ms_per_step = elapsed / args.steps * 1e3
        pk = (pmc or {}).get("kernels", {}) if pmc and "error" not in pmc else {}

        def bounds_of(kind):
            """Fractions of the three rooflines a pass can be held against, each <= 1 by construction: per-launch counters (PMC
            children: serialised launches) over the UNDISTURBED launch duration of this run."""
            k = pk.get(KERNEL_OF[kind], {})
            t = alone_ms[kind] * 1e-3
            b = {"l2_gather": {"achieved": req[kind] / t / 1e9 if t > 0 else 0.0, "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s",
                               "what": "requested bytes (16 B per box test, 48 B per triangle test, 116 B per closest hit, frame buffers) / undisturbed launch "
                                       "time vs the L2-resident gather rate of MI355X_MICROARCH.md (16.8-18.8 TB/s)"}}
            b["l2_gather"]["frac"] = b["l2_gather"]["achieved"] / L2_GATHER_PEAK_GBS
            if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
                traffic = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0   # KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B: doubled
                b["hbm"] = {"achieved": traffic / t / 1e9 if t > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic_bytes_per_launch": traffic,
                            "what": "physical HBM bytes per launch (live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, read side x2 as the guide prescribes "
                                    "for gfx950) / undisturbed launch time vs 8 TB/s"}
                b["hbm"]["frac"] = b["hbm"]["achieved"] / HBM_PEAK_GBS
            if "SQ_INSTS_VALU" in k and "GRBM_GUI_ACTIVE" in k:
                cycles = k["GRBM_GUI_ACTIVE"] / 8.0                             # summed over the 8 XCDs
                b["valu_issue"] = {"achieved": k["SQ_INSTS_VALU"] * 2.0 / cycles if cycles > 0 else 0.0, "peak": float(N_SIMD), "unit": "SIMD-cycles per cycle",
                                   "what": "SQ_INSTS_VALU x 2 cycles (wave64 on a SIMD-32) / kernel cycles (GRBM_GUI_ACTIVE / 8) vs 1024 SIMDs; counters from "
                                           "the same launch under rocprofv3 --pmc",
                                   "lane_utilisation": k["SQ_THREAD_CYCLES_VALU"] / 64.0 / k["SQ_ACTIVE_INST_VALU"] if k.get("SQ_ACTIVE_INST_VALU") else None,
                                   "wave_time_waiting": k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"] if k.get("SQ_WAVE_CYCLES") else None}
                b["valu_issue"]["frac"] = b["valu_issue"]["achieved"] / N_SIMD
            if "TCC_HIT_sum" in k and "TCC_MISS_sum" in k and (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]) > 0:
                b["l2_gather"]["l2_hit_rate"] = k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"])
            if "TA_TA_BUSY_sum" in k and "GRBM_GUI_ACTIVE" in k and k["GRBM_GUI_ACTIVE"] > 0:
                b["l2_gather"]["ta_busy"] = k["TA_TA_BUSY_sum"] / 256.0 / (k["GRBM_GUI_ACTIVE"] / 8.0)
            return b

        bounds = {k: bounds_of(k) for k in (KIND_RIS, KIND_FINAL)}
        bounds_dom = bounds[dom]
        binding_name = max(bounds_dom, key=lambda n: bounds_dom[n]["frac"])
        binding = bounds_dom[binding_name]

        # Frame level: both passes' per-launch figures summed over the frame time of the timed region (two frames in flight).
        t_step = ms_per_step * 1e-3
        frame_roof = {"l2_gather": {"achieved": (req[KIND_RIS] + req[KIND_FINAL]) / t_step / 1e9, "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s"}}
        frame_roof["l2_gather"]["frac"] = frame_roof["l2_gather"]["achieved"] / L2_GATHER_PEAK_GBS
        if all("hbm" in bounds[k] for k in bounds):
            tb = sum(bounds[k]["hbm"]["traffic_bytes_per_launch"] for k in bounds)
            frame_roof["hbm"] = {"achieved": tb / t_step / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic_bytes_per_frame": tb}
            frame_roof["hbm"]["frac"] = frame_roof["hbm"]["achieved"] / HBM_PEAK_GBS
        if all("SQ_INSTS_VALU" in pk.get(KERNEL_OF[k], {}) and "GRBM_GUI_ACTIVE" in pk.get(KERNEL_OF[k], {}) for k in bounds) and sum(alone_ms.values()) > 0:
            # shader clock of this box: cycles the two kernels took under the counters / their undisturbed durations in this run
            cyc = sum(pk[KERNEL_OF[k]]["GRBM_GUI_ACTIVE"] / 8.0 for k in bounds)
            clock_khz = cyc / sum(alone_ms.values())                      # cycles per ms
            valu = sum(pk[KERNEL_OF[k]]["SQ_INSTS_VALU"] for k in bounds) * 2.0
            frame_roof["valu_issue"] = {"achieved": valu / (ms_per_step * clock_khz), "peak": float(N_SIMD), "unit": "SIMD-cycles per cycle",
                                        "shader_clock_mhz": clock_khz / 1e3}
            frame_roof["valu_issue"]["frac"] = frame_roof["valu_issue"]["achieved"] / N_SIMD
        frame_roof["what"] = ("both passes' per-launch figures (requested bytes; physical HBM bytes; VALU instructions x 2 cycles) summed and divided by "
                              "ms_per_step of the timed region (for VALU: by the cycles of a step at the shader clock = kernel cycles under the counters / "
                              "undisturbed kernel durations)")
        head = git_head()
        pmc_note = ("live: %d rocprofv3 --pmc passes of this workload run by this process before the timed region (%.0f s)%s"
                    % (len(pmc["passes"]), pmc["seconds"], ", HEAD " + head if head else "")) if pmc and "error" not in pmc else \
                   ("not measured: " + (pmc["error"] if pmc else "N > 1 or --no-pmc"))
        out = {
            "metric": "Mray/s (closest-hit + any-hit queries traversed per second), 1920x1080, 1 spp, 1M-triangle scene",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
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
                # the reference issues every pixel's camera-ray TraceRay twice (ray_gen_ris.slang:75, ray_gen_final.slang:80); with the
                # primary-hit hand-off the final pass reads the RIS pass's payload instead, and its final GI visibility query is not traced
                # again where it repeats a neighbour's (SrRayCounters.reused_*). `value` counts traversals executed only.
                "traced_queries_per_frame": total_rays / args.steps,
                "reference_queries_per_frame": (total_rays + total_reused) / args.steps,
                "reference_queries_mray_s": (total_rays + total_reused) / elapsed / 1e6,     # TraceRay calls of the reference answered per second
                "primary_hit_hand_off": bool(reuse),
                "parallelism": ("%s split into %d cost-balanced strips %s, RIS pass over strip + %d-pixel halo (recomputed, uncounted), radiance "
                                "strips all-gathered over RCCL asynchronously (frame f's gather overlaps frame f+1); static camera, so no "
                                "temporal-history exchange (distributed.exchange_history, motion_halo = 0)"
                                % ("columns" if axis == "cols" else "rows", world, part.sizes(), sd.SPATIAL_HALO))
                               if world > 1 else "single GPU",
                "frames_in_flight": 2 if pipelined else 1,
                "bvh_build_ms_host": st.build_ms,
                "last_frame_crc32": frame_crc,
            },
            "roofline": {
                "bound": binding_name,
                "kernel": names[dom],
                "achieved": binding["achieved"],
                "peak": binding["peak"],
                "unit": binding["unit"],
                "frac": binding["frac"],
                "traffic": bounds_dom.get("hbm", {}).get("traffic_bytes_per_launch"),
                "traffic_source": pmc_note,
                "bounds": bounds_dom,
                "avg_launch_ms": alone_ms[dom],
                "avg_launch_ms_overlapped": overlapped_ms[dom],
                "launches_timed": fin_n if dom == KIND_FINAL else ris_n,
                "algorithmic_bytes_per_launch": req[dom],
                "algorithmic_bytes_per_launch_survey_formula": survey_bytes(ck, n_own_pixels, dom),
                "boxes_per_ray": ck.boxes_tested / nq,
                "tris_per_ray": ck.tris_tested / nq,
                "other_pass": {"kernel": KERNEL_OF[other], "avg_launch_ms": alone_ms[other], "avg_launch_ms_overlapped": overlapped_ms[other],
                               "algorithmic_bytes_per_launch": req[other], "bounds": bounds[other]},
                "frame": frame_roof,
                "launches_overlap": bool(pipelined),
                "note": "three physical rooflines per pass, `bound` = the one with the largest fraction. avg_launch_ms is the UNDISTURBED duration of a "
                        "launch (HIP events around the launches of %d frames enqueued back to back on one stream after the timed region) and prices the "
                        "per-pass fractions; "
                        "avg_launch_ms_overlapped is its wall time inside the timed region, where two frames are in flight and a launch shares the GPU "
                        "with its neighbour; `frame` prices both passes together against ms_per_step. The passes gather 64-byte BVH nodes and "
                        "48-byte triangles that live in L2 / Infinity Cache (16.5 MB + 48 MB), so HBM is not the binding roof; none of the three "
                        "is saturated: a node step waits for the slowest of its lanes' dependent fetches (DESIGN.md section 5)" % seq_frames,
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
