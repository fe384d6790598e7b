"""Tile-parallel rendering of one frame across the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm). The scene is replicated; the screen is
cut into `world` contiguous strips — COLUMN strips by default: image cost varies mostly with the row (sky / horizon /
foreground), so column strips hand every GPU the same mix of rows, exactly as the kernels cut the image into column bands
across the XCDs of one GPU; row strips remain available (`axis="rows"`). Pixels are keyed by their GLOBAL coordinates
(RNG seed, camera ray), so a strip traced alone carries exactly the values it has in a single-GPU frame. The data-path
exchange is the gather of the fp32 radiance strips, one collective per frame.

ReSTIR makes pixels depend on neighbours:
  * spatial reuse — the final pass reads reservoirs / normal / depth within a 30-pixel radius
    (ray_gen_final.slang:160-188,228-247). Each rank therefore traces the RIS pass over its strip plus a 30-pixel halo
    on either side (recompute instead of exchange: 2*30/240 = 25 % extra RIS columns at 1080p on 8 GPUs, 12.5 % at 4K;
    with row strips 44 % / 22 %); only the strip's own pixels count their rays (SrTraceConfig.count_*).
  * temporal reuse — the RIS pass reads the PREVIOUS frame's reservoirs at the reprojected pixel
    (ray_gen_ris.slang:234-266,408-431). With a static camera that is the pixel itself and strip + halo is
    self-contained. Under camera motion the reprojected pixel of a halo pixel can lie up to `motion_halo` pixels further
    out, in pixels this rank never traced: `exchange_history` fetches those reservoir bands from the ranks that own them
    (point-to-point over RCCL / gloo) after every RIS pass, so the history a rank reads is always the single-GPU one and
    N-GPU output stays bit-identical to 1-GPU for a moving camera as well (tests/test_distributed_gloo.py).
"""
import copy

SPATIAL_HALO = 30  # SPATIAL_RADIUS (ray_gen_final.slang:161) >= GI_SPATIAL_RADIUS (:229)


class Partition:
    """`world` contiguous strips of a width x height image along one axis. `bounds` (world + 1 increasing cut positions,
    e.g. from balanced_bounds) replaces the equal split; it must stay the same for a whole frame sequence: a rank owns
    the temporal history of exactly its strip + halo."""

    def __init__(self, width, height, world, axis="cols", bounds=None):
        if axis not in ("cols", "rows"):
            raise ValueError("axis must be 'cols' or 'rows'")
        self.width, self.height, self.world, self.axis = width, height, world, axis
        self.length = width if axis == "cols" else height
        if bounds is None:
            per = (self.length + world - 1) // world
            bounds = [min(r * per, self.length) for r in range(world)] + [self.length]
        bounds = [int(v) for v in bounds]
        if len(bounds) != world + 1 or bounds[0] != 0 or bounds[-1] != self.length or any(b > a for b, a in zip(bounds, bounds[1:])):
            raise ValueError("bounds must be %d increasing cuts from 0 to %d" % (world + 1, self.length))
        self.bounds = bounds

    def span(self, rank):
        """(start, size) of rank's strip along the axis."""
        return self.bounds[rank], self.bounds[rank + 1] - self.bounds[rank]

    def sizes(self):
        return [self.bounds[r + 1] - self.bounds[r] for r in range(self.world)]

    def tile(self, a0, n):
        """The (y0, h, x0, w) launch rectangle of positions [a0, a0 + n) along the axis, full extent across it."""
        return (0, self.height, a0, n) if self.axis == "cols" else (a0, n, 0, self.width)

    def grown(self, rank, grow):
        """(start, size) of rank's strip grown by `grow` on both sides, clipped to the image."""
        a0, n = self.span(rank)
        lo, hi = max(0, a0 - grow), min(self.length, a0 + n + grow)
        return lo, hi - lo

    def view(self, flat, channels=None):
        """[H, W(, C)] view of a per-pixel buffer (torch tensor or numpy array of H*W rows)."""
        return flat.reshape(self.height, self.width, -1) if channels is None else flat.reshape(self.height, self.width, channels)

    def cut(self, img, a0, n):
        """Slice [a0, a0 + n) along the axis of an [H, W, C] view."""
        return img[:, a0:a0 + n] if self.axis == "cols" else img[a0:a0 + n]


def balanced_bounds(cost, world, min_size=8, max_share=2.5):
    """Cuts positions 0 .. len(cost) into `world` contiguous strips of (nearly) equal summed cost: returns world + 1
    increasing cut positions. Strips are cut one after the other, each taking 1/n of the cost that is left for the n ranks
    that are left, with at least `min_size` positions and at most max_share * length / world (the gather pads every strip
    to the largest one, so a very large cheap strip would inflate the collective). Deterministic: every rank that feeds
    the same profile gets the same cut."""
    import numpy as np
    cost = np.maximum(np.asarray(cost, dtype=np.float64), 0.0) + 1e-12
    length = len(cost)
    min_size = max(1, min(min_size, length // max(world, 1)))
    max_size = max(int(np.ceil(max_share * length / max(world, 1))), min_size)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    bounds = [0]
    for k in range(world - 1):
        y, n = bounds[-1], world - k
        target = cum[y] + (cum[-1] - cum[y]) / n
        cut = int(np.searchsorted(cum, target, side="left"))
        cut = min(max(cut, y + min_size), y + max_size)          # this strip: [min_size, max_size]
        cut = max(cut, length - (n - 1) * max_size)              # the ranks that are left can still cover the rest ...
        cut = min(cut, length - (n - 1) * min_size)              # ... and each gets its minimum
        bounds.append(max(cut, y))
    bounds.append(length)
    return [int(v) for v in bounds]


def axis_cost_from_tiles(tile_costs, tiles_x, axis, length, tile=8):
    """Per-pixel-column (or per-pixel-row) cost from the per-tile cycle counts the library records for its own tile
    schedule (sr_scene_read_tile_costs, row-major ty * tiles_x + tx): what balanced_bounds cuts."""
    import numpy as np
    t = np.asarray(tile_costs, dtype=np.float64).reshape(-1, tiles_x)
    per_tile = t.sum(axis=0) if axis == "cols" else t.sum(axis=1)
    return np.repeat(per_tile / float(tile), tile)[:length]


def trace_ris_strip(scene, frame, matrices, frame_count, cfg, part, rank):
    """The RIS pass of this rank's strip. With world > 1 it is ONE launch over the strip and its spatial halo: a rank's
    share of a frame is small, so every extra launch adds a tail in which the GPU drains; only the strip's own pixels
    count their rays."""
    a0, n = part.span(rank)
    if n <= 0 or not cfg.enable_restir:
        return
    if part.world > 1:
        g0, gn = part.grown(rank, SPATIAL_HALO)
        rcfg = copy.copy(cfg)
        if part.axis == "cols":
            rcfg.count_x0, rcfg.count_cols = a0, n
        else:
            rcfg.count_y0, rcfg.count_rows = a0, n
        scene.trace_ris(frame, matrices, frame_count, rcfg, tile=part.tile(g0, gn))
    else:
        scene.trace_ris(frame, matrices, frame_count, cfg, tile=part.tile(a0, n))


def trace_final_strip(scene, frame, matrices, frame_count, cfg, part, rank):
    a0, n = part.span(rank)
    if n > 0:
        scene.trace_final(frame, matrices, frame_count, cfg, tile=part.tile(a0, n))


def history_exchange_plan(part, motion_halo):
    """Who sends which reservoir band to whom after a RIS pass: rank r needs the pixels within SPATIAL_HALO + motion_halo
    of its strip that lie outside strip + SPATIAL_HALO (those it traced itself); every such pixel is owned — and was
    traced with exact history — by exactly one other rank. Returns a list of (src, dst, start, size) along the axis, in
    a deterministic order every rank derives alike."""
    plan = []
    if motion_halo <= 0 or part.world <= 1:
        return plan
    for dst in range(part.world):
        a0, n = part.span(dst)
        if n <= 0:
            continue
        g0, gn = part.grown(dst, SPATIAL_HALO)
        h0, hn = part.grown(dst, SPATIAL_HALO + motion_halo)
        for lo, hi in ((h0, g0), (g0 + gn, h0 + hn)):         # the band before and the band after the traced region
            for src in range(part.world):
                if src == dst:
                    continue
                s0, sn = part.span(src)
                x0, x1 = max(lo, s0), min(hi, s0 + sn)
                if x1 > x0:
                    plan.append((src, dst, x0, x1 - x0))
    return plan


def exchange_history(frame, frame_count, part, rank, motion_halo, as_tensor=None):
    """After RIS(frame_count): fetch the DI and GI reservoirs of the bands `history_exchange_plan` assigns to this rank
    from their owners, so that RIS(frame_count + 1) finds exact history wherever temporal reprojection can land
    (|reprojected - pixel| <= motion_halo along the strip axis). Point-to-point (batch_isend_irecv); a no-op for a
    static camera (motion_halo = 0). `as_tensor` turns a frame buffer into a torch tensor sharing its memory (numpy host
    frames in the CPU tests); device frames hold torch tensors already."""
    plan = [p for p in history_exchange_plan(part, motion_halo) if rank in (p[0], p[1])]
    if not plan:
        return
    import torch
    import torch.distributed as dist
    cur = frame_count & 1
    tt = as_tensor or (lambda b: b)
    di = part.view(tt(frame.reservoirs[cur]), 12)
    gi = part.view(tt(frame.reservoirs_gi[cur]), 12)
    ops, landing = [], []
    for src, dst, x0, n in plan:
        if src == rank:
            buf = torch.cat([part.cut(di, x0, n), part.cut(gi, x0, n)], dim=2).contiguous()
            ops.append(dist.P2POp(dist.isend, buf, dst))
        else:
            shape = list(part.cut(di, x0, n).shape)
            shape[2] = 24
            buf = torch.empty(shape, dtype=di.dtype, device=di.device)
            ops.append(dist.P2POp(dist.irecv, buf, src))
            landing.append((buf, x0, n))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    for buf, x0, n in landing:
        part.cut(di, x0, n).copy_(buf[:, :, :12])
        part.cut(gi, x0, n).copy_(buf[:, :, 12:])


def render_strip(scene, frame, matrices, frame_count, cfg, part, rank, motion_halo=0, as_tensor=None):
    """Traces this rank's part of one frame into the full-size buffers of `frame`.

    `scene` is a sunray_amd.runtime.Scene (GPU) — or, in the CPU tests, the oracle's scene: both offer
    trace_ris / trace_final(frame, matrices, frame_count, cfg, tile=(y0, h, x0, w))."""
    trace_ris_strip(scene, frame, matrices, frame_count, cfg, part, rank)
    if cfg.enable_restir:
        exchange_history(frame, frame_count, part, rank, motion_halo, as_tensor)
    trace_final_strip(scene, frame, matrices, frame_count, cfg, part, rank)
    return part.span(rank)


class FramePipeline:
    """Two frames in flight on one GPU: raytracing_ris of frame f+1 runs on its own stream while raytracing_final of frame
    f still drains, so the tail of one launch is filled by the head of the next (the reference keeps
    MAX_FRAMES_IN_FLIGHT = 2 frames in flight as well, src/lib.rs:71). What makes it legal: the RIS pass only WRITES the
    G-buffer images and the current reservoir buffers and only READS the previous frame's reservoirs; the final pass
    reads the G-buffer and the current reservoirs. With the G-buffer double-buffered (two frame objects that share
    their reservoir arrays) the only orderings left are RIS(f) -> final(f), RIS(f) -> RIS(f+1) and final(f-2) -> RIS(f).
    The primary-hit hand-off (RIS(f) writes the camera ray's payload, final(f) reads it) belongs to that double-buffered set.
    Results are those of sequential execution, bit for bit (test_frames_in_flight_equal_sequential_frames).
    Two is also the limit: a third frame in flight needs a third physical reservoir buffer, and the reference's reservoirs are
    a two-buffer ping-pong whose STALE contents are observable — a sky pixel leaves its GI reservoir unwritten
    (ray_gen_ris.slang:171), so a pixel that reprojects onto last frame's sky reads what was stored there three frames ago;
    with three rotating buffers that would be four frames ago (tried: identical for a static camera, different bits for a
    moving one, and no faster — 1.94 vs 1.93 ms per frame)."""

    def __init__(self, frame_a, frame_b):
        import torch
        frame_b.reservoirs, frame_b.reservoirs_gi = frame_a.reservoirs, frame_a.reservoirs_gi
        self.frames = [frame_a, frame_b]
        self.s_ris, self.s_final = torch.cuda.Stream(), torch.cuda.Stream()
        self.ev_ris = [torch.cuda.Event(), torch.cuda.Event()]
        self.ev_final = [torch.cuda.Event(), torch.cuda.Event()]
        self.torch = torch

    def step(self, scene, matrices, frame_count, cfg, part, rank, after_final=None, motion_halo=0):
        torch = self.torch
        k = frame_count & 1
        fr = self.frames[k]
        with torch.cuda.stream(self.s_ris):
            self.s_ris.wait_event(self.ev_final[k])          # final(f-2) has finished reading this G-buffer
            trace_ris_strip(scene, fr, matrices, frame_count, cfg, part, rank)
            if cfg.enable_restir:
                exchange_history(fr, frame_count, part, rank, motion_halo)   # ordered on the RIS stream (RIS(f+1) reads it)
            self.ev_ris[k].record(self.s_ris)
        with torch.cuda.stream(self.s_final):
            self.s_final.wait_event(self.ev_ris[k])
            trace_final_strip(scene, fr, matrices, frame_count, cfg, part, rank)
            self.ev_final[k].record(self.s_final)
            if after_final is not None:
                after_final(fr)                               # e.g. GatherPipeline.submit(fr.raw_color), on the final stream
        return fr


class GatherPipeline:
    """Asynchronous, double-buffered gather of the radiance strips: the collective of frame f runs on the collective
    library's own stream while the kernels of frame f+1 are already tracing (xGMI transfer hidden behind compute).
    Strips may differ in size (balanced_bounds): every rank contributes a buffer padded to the largest strip, one
    `all_gather_into_tensor` per frame, and `image()` reassembles the image.

    submit(raw_color) copies this rank's strip out of the frame buffer (so the next frame may overwrite it) and starts
    the collective; wait(slot) makes the current stream wait for it. At most `depth` gathers are in flight."""

    def __init__(self, part, rank, device, depth=2, dtype=None):
        import torch
        self.torch = torch
        self.part, self.rank = part, rank
        self.size_max = max(part.sizes())
        dtype = dtype or torch.float32
        shape = (part.height, self.size_max, 4) if part.axis == "cols" else (self.size_max, part.width, 4)
        self.send = [torch.zeros(*shape, dtype=dtype, device=device) for _ in range(depth)]
        self.shape = shape
        self.recv = [torch.empty(part.world * shape[0], shape[1], 4, dtype=dtype, device=device) for _ in range(depth)]   # concatenated along dim 0
        self.work = [None] * depth
        self.next = 0
        self.last = None

    def submit(self, raw_color, all_gather=None):
        import torch.distributed as dist
        k = self.next
        self.wait(k)                                   # the slot's previous gather must be done before its buffers are reused
        a0, n = self.part.span(self.rank)
        if n > 0:
            self.part.cut(self.send[k], 0, n).copy_(self.part.cut(self.part.view(raw_color, 4), a0, n))
        # one rank and no process group: nothing to gather. With a process group the collective is issued even for one
        # rank, so that a 1-GPU run exercises the same RCCL initialisation, stream ordering and image() path as N ranks.
        if self.part.world > 1 or all_gather is not None or (dist.is_available() and dist.is_initialized()):
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
        """Full [H*W, 4] image of slot k (default: the last submitted), re-assembled from the padded strips."""
        k = self.last if k is None else k
        self.wait(k)
        stacked = self.recv[k].reshape(self.part.world, *self.shape)
        parts = [self.part.cut(stacked[r], 0, self.part.sizes()[r]) for r in range(self.part.world)]
        return self.torch.cat(parts, dim=1 if self.part.axis == "cols" else 0).reshape(self.part.height * self.part.width, 4)
