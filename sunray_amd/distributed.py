"""Tile-parallel rendering of one frame across the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm). The scene is
replicated; the screen is cut into `world` contiguous row strips. Pixels are keyed by their GLOBAL
coordinates (RNG seed, camera ray), so a strip traced alone carries exactly the values it has in a
single-GPU frame. The only data-path exchange is the gather of the fp32 radiance strips.

ReSTIR makes pixels depend on neighbours: the final pass reads reservoirs / normal / depth within a
30-pixel radius (ray_gen_final.slang:160-188,228-247). Each rank therefore re-traces the RIS pass on
a 30-row halo above and below its strip (recompute instead of exchange; those launches carry
SR_TRACE_FLAG_UNCOUNTED so counted rays stay those of the single-GPU frame). Temporal reuse reads the
previous frame's reservoir at the reprojected pixel (ray_gen_ris.slang:234-266): with a static camera
that is the pixel itself, so strip + halo is self-contained and N-GPU output is bit-identical to
1-GPU; under camera motion the outermost halo rows may read history this rank never computed.
"""
import copy

SPATIAL_HALO = 30  # SPATIAL_RADIUS (ray_gen_final.slang:161) >= GI_SPATIAL_RADIUS (:229)


def strip_rows(height, world, rank):
    """Rows [y0, y0+h) of rank's strip; equal ceil(height/world) rows, the last strips may be short/empty."""
    per = (height + world - 1) // world
    y0 = min(rank * per, height)
    return y0, min(per, height - y0)


def balanced_bounds(row_cost, world, min_rows=8, max_share=2.5):
    """Cuts the rows into `world` contiguous strips of (nearly) equal summed cost: returns world+1 increasing row
    boundaries. Strips are cut one after the other, each taking 1/n of the cost that is left for the n ranks that are
    left, with at least `min_rows` rows and at most max_share * height / world rows (the gather pads every strip to the
    tallest one, so a very tall cheap strip would inflate the collective). Deterministic: every rank that feeds the
    same profile gets the same cut, so no communication is needed to agree on it."""
    import numpy as np
    cost = np.maximum(np.asarray(row_cost, dtype=np.float64), 0.0) + 1e-12
    height = len(cost)
    min_rows = max(1, min(min_rows, height // max(world, 1)))
    max_rows = max(int(np.ceil(max_share * height / max(world, 1))), min_rows)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    bounds = [0]
    for k in range(world - 1):
        y, n = bounds[-1], world - k
        target = cum[y] + (cum[-1] - cum[y]) / n
        cut = int(np.searchsorted(cum, target, side="left"))
        cut = min(max(cut, y + min_rows), y + max_rows)          # this strip: [min_rows, max_rows] rows
        cut = max(cut, height - (n - 1) * max_rows)              # the ranks that are left can still cover the rest ...
        cut = min(cut, height - (n - 1) * min_rows)              # ... and each gets its minimum
        bounds.append(max(cut, y))
    bounds.append(height)
    return [int(v) for v in bounds]


def refine_bounds(row_cost, bounds, periods, damping=0.7, min_rows=32, max_share=2.5):
    """One step of the feedback balancer: `periods[r]` is the measured step time of rank r with the cut `bounds`. A rank's
    share of a frame is about one round of waves, so its time is not proportional to the cycle sum of its rows (the strip
    that holds the horizon rows is bound by its slowest waves): the cost model is corrected where it was wrong — the rows
    of rank r are re-weighted by (period_r / mean period) ** damping — and the rows are cut again. Returns
    (new row_cost, new bounds). Deterministic; the caller keeps the cut with the smallest measured maximum."""
    import numpy as np
    cost = np.maximum(np.asarray(row_cost, dtype=np.float64), 0.0).copy()
    world = len(bounds) - 1
    p = np.maximum(np.asarray(periods, dtype=np.float64), 1e-9)
    mean = float(p.mean())
    for r in range(world):
        cost[bounds[r]:bounds[r + 1]] *= (p[r] / mean) ** damping
    return cost, balanced_bounds(cost, world, min_rows=min_rows, max_share=max_share)


def row_cost_from_depth(depth_u16, width, height, sky_weight=1.0, surface_weight=4.0):
    """Per-row cost estimate from a G-buffer depth image (R16F bits): a sky pixel (depth +inf, ray_gen_ris.slang:168)
    costs one primary ray per pass, a surface pixel the whole ReSTIR sequence (~6 rays, mostly incoherent)."""
    import numpy as np
    d = np.asarray(depth_u16).reshape(height, width)
    sky = (d & 0x7FFF) >= 0x7C00
    return sky_weight * sky.sum(axis=1) + surface_weight * (~sky).sum(axis=1)


def halo_bands(height, y0, h, halo=SPATIAL_HALO):
    """Row bands outside [y0, y0+h) within `halo` rows of it, clipped to the image."""
    bands = []
    if h <= 0:
        return bands
    top0 = max(0, y0 - halo)
    if top0 < y0:
        bands.append((top0, y0 - top0))
    bot1 = min(height, y0 + h + halo)
    if bot1 > y0 + h:
        bands.append((y0 + h, bot1 - (y0 + h)))
    return bands


def _strip(height, world, rank, bounds):
    return (bounds[rank], bounds[rank + 1] - bounds[rank]) if bounds is not None else strip_rows(height, world, rank)


def trace_ris_strip(scene, frame, matrices, frame_count, cfg, world, rank, bounds=None):
    """The RIS pass of this rank's strip. With world > 1 it is ONE launch over the strip and its halo rows: a rank's share
    of a frame is small, so every extra launch adds a tail in which the GPU drains; only the strip's own rows count
    their rays (SrTraceConfig.count_y0 / count_rows)."""
    y0, h = _strip(frame.height, world, rank, bounds)
    if h <= 0 or not cfg.enable_restir:
        return
    if world > 1:
        bands = halo_bands(frame.height, y0, h)
        r0 = min([y0] + [b[0] for b in bands])
        r1 = max([y0 + h] + [b[0] + b[1] for b in bands])
        rcfg = copy.copy(cfg)
        rcfg.count_y0, rcfg.count_rows = y0, h
        scene.trace_ris(frame, matrices, frame_count, rcfg, tile=(r0, r1 - r0))
    else:
        scene.trace_ris(frame, matrices, frame_count, cfg, tile=(y0, h))


def trace_final_strip(scene, frame, matrices, frame_count, cfg, world, rank, bounds=None):
    y0, h = _strip(frame.height, world, rank, bounds)
    if h > 0:
        scene.trace_final(frame, matrices, frame_count, cfg, tile=(y0, h))


def render_strip(scene, frame, matrices, frame_count, cfg, world, rank, uncounted_flag=1, bounds=None):
    """Traces this rank's part of one frame into the full-size buffers of `frame`.

    `scene` is a sunray_amd.runtime.Scene (GPU) — or, in the CPU tests, the oracle's scene: both offer
    trace_ris / trace_final(frame, matrices, frame_count, cfg, tile=(y0, h)). `bounds` (balanced_bounds) replaces
    the equal split; it must stay the same for the whole frame sequence (a rank keeps the temporal history of
    exactly its rows + halo)."""
    trace_ris_strip(scene, frame, matrices, frame_count, cfg, world, rank, bounds)
    trace_final_strip(scene, frame, matrices, frame_count, cfg, world, rank, bounds)
    return _strip(frame.height, world, rank, bounds)


class FramePipeline:
    """Two frames in flight on one GPU: raytracing_ris of frame f+1 runs on its own stream while raytracing_final of frame
    f still drains, so the tail of one launch is filled by the head of the next (the reference keeps
    MAX_FRAMES_IN_FLIGHT = 2 frames in flight as well, src/lib.rs:71). What makes it legal: the RIS pass only WRITES the
    G-buffer images and the current reservoir buffers and only READS the previous frame's reservoirs; the final pass
    reads the G-buffer and the current reservoirs. With the G-buffer double-buffered (two frame objects that share
    their reservoir arrays) the only orderings left are RIS(f) -> final(f), RIS(f) -> RIS(f+1) and final(f-2) -> RIS(f).
    Results are those of sequential execution, bit for bit."""

    def __init__(self, frame_a, frame_b):
        import torch
        frame_b.reservoirs, frame_b.reservoirs_gi = frame_a.reservoirs, frame_a.reservoirs_gi
        self.frames = [frame_a, frame_b]
        self.s_ris, self.s_final = torch.cuda.Stream(), torch.cuda.Stream()
        self.ev_ris = [torch.cuda.Event(), torch.cuda.Event()]
        self.ev_final = [torch.cuda.Event(), torch.cuda.Event()]
        self.torch = torch

    def step(self, scene, matrices, frame_count, cfg, world, rank, bounds=None, after_final=None):
        torch = self.torch
        k = frame_count & 1
        fr = self.frames[k]
        with torch.cuda.stream(self.s_ris):
            self.s_ris.wait_event(self.ev_final[k])          # final(f-2) has finished reading this G-buffer
            trace_ris_strip(scene, fr, matrices, frame_count, cfg, world, rank, bounds)
            self.ev_ris[k].record(self.s_ris)
        with torch.cuda.stream(self.s_final):
            self.s_final.wait_event(self.ev_ris[k])
            trace_final_strip(scene, fr, matrices, frame_count, cfg, world, rank, bounds)
            self.ev_final[k].record(self.s_final)
            if after_final is not None:
                after_final(fr)                               # e.g. GatherPipeline.submit(fr.raw_color), on the final stream
        return fr


def gather_strips(raw_color, width, height, world, rank, out=None, scratch=None):
    """All-gathers the radiance strips into a full [H*W, 4] image on every rank (one collective).

    raw_color: this rank's full-size [H*W, 4] float32 torch tensor, valid in its own strip."""
    import torch
    import torch.distributed as dist
    per = (height + world - 1) // world
    y0, h = strip_rows(height, world, rank)
    if scratch is None:
        scratch = torch.zeros(per * width, 4, dtype=raw_color.dtype, device=raw_color.device)
    if h > 0:
        scratch[: h * width].copy_(raw_color[y0 * width:(y0 + h) * width])
    if out is None:
        out = torch.empty(world * per * width, 4, dtype=raw_color.dtype, device=raw_color.device)
    if world > 1:
        dist.all_gather_into_tensor(out, scratch)
    else:
        out.copy_(scratch)
    return out[: height * width]


class GatherPipeline:
    """Asynchronous, double-buffered gather of the radiance strips: the collective of frame f runs on the collective
    library's own stream while the kernels of frame f+1 are already tracing (xGMI transfer hidden behind compute).
    Strips may have different heights (balanced_bounds): every rank contributes a buffer padded to the tallest strip,
    one `all_gather_into_tensor` per frame, and `image()` reassembles the rows.

    submit(raw_color) copies this rank's strip out of the frame buffer (so the next frame may overwrite it) and starts
    the collective; wait(slot) makes the current stream wait for it. At most `depth` gathers are in flight."""

    def __init__(self, width, height, world, rank, device, bounds=None, depth=2, dtype=None):
        import torch
        self.torch = torch
        self.width, self.height, self.world, self.rank = width, height, world, rank
        self.bounds = list(bounds) if bounds is not None else [min(r * ((height + world - 1) // world), height) for r in range(world)] + [height]
        self.rows_max = max(self.bounds[r + 1] - self.bounds[r] for r in range(world))
        dtype = dtype or torch.float32
        n = self.rows_max * width
        self.send = [torch.zeros(n, 4, dtype=dtype, device=device) for _ in range(depth)]
        self.recv = [torch.empty(world * n, 4, dtype=dtype, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.next = 0
        self.last = None

    def submit(self, raw_color, all_gather=None):
        import torch.distributed as dist
        k = self.next
        self.wait(k)                                   # the slot's previous gather must be done before its buffers are reused
        y0, y1 = self.bounds[self.rank], self.bounds[self.rank + 1]
        if y1 > y0:
            self.send[k][: (y1 - y0) * self.width].copy_(raw_color[y0 * self.width:y1 * self.width])
        if self.world > 1:
            fn = all_gather or (lambda out, inp: dist.all_gather_into_tensor(out, inp, async_op=True))
            self.work[k] = fn(self.recv[k], self.send[k])
        else:
            self.recv[k].copy_(self.send[k])
        self.last = k
        self.next = (k + 1) % len(self.send)
        return k

    def wait(self, k=None):
        ks = range(len(self.work)) if k is None else [k]
        for i in ks:
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None

    def image(self, k=None):
        """Full [H*W, 4] image of slot k (default: the last submitted), rows re-assembled from the padded strips."""
        k = self.last if k is None else k
        self.wait(k)
        n = self.rows_max * self.width
        parts = [self.recv[k][r * n: r * n + (self.bounds[r + 1] - self.bounds[r]) * self.width] for r in range(self.world)]
        return self.torch.cat(parts, dim=0)
