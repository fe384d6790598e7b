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
The strip geometry, the cost-balanced cut, the history-exchange plan and the two strip launches live in the library behind the
C ABI (csrc/multi_gpu.cpp: sr_partition_*, sr_balanced_bounds, sr_axis_cost_from_tiles, sr_history_exchange_plan, sr_strip_*):
a Rust host calls the same entry points. This module binds them and adds what needs torch: the point-to-point
history exchange, the frame pipeline (streams + events) and the double-buffered asynchronous gather.
"""
import copy

SPATIAL_HALO = 30  # SPATIAL_RADIUS (ray_gen_final.slang:161) >= GI_SPATIAL_RADIUS (:229)


def _u32(n):
    import ctypes as C
    return (C.c_uint32 * n)()


class Partition:
    """`world` contiguous strips of a width x height image along one axis — a handle on the library's SrPartition
    (csrc/multi_gpu.cpp), which owns the geometry. `bounds` (world + 1 increasing cut positions, e.g. from balanced_bounds)
    replaces the equal split; it must stay the same for a whole frame sequence: a rank owns the temporal history of exactly
    its strip + halo."""

    def __init__(self, width, height, world, axis="cols", bounds=None):
        import ctypes as C
        from ._lib import check, lib
        if axis not in ("cols", "rows"):
            raise ValueError("axis must be 'cols' or 'rows'")
        self.width, self.height, self.world, self.axis = int(width), int(height), int(world), axis
        self.length = self.width if axis == "cols" else self.height
        b = None
        if bounds is not None:
            bounds = [int(v) for v in bounds]
            if len(bounds) != world + 1 or any(v < 0 for v in bounds):
                raise ValueError("bounds must be %d increasing cuts from 0 to %d" % (world + 1, self.length))
            b = (C.c_uint32 * len(bounds))(*bounds)
        self._h = C.c_void_p()
        try:
            check(lib().sr_partition_create(C.c_uint32(self.width), C.c_uint32(self.height), C.c_uint32(self.world),
                                            C.c_uint32(0 if axis == "cols" else 1), b, C.byref(self._h)))
        except Exception as e:
            raise ValueError(str(e))
        bp = C.POINTER(C.c_uint32)()
        check(lib().sr_partition_get(self._h, None, None, None, None, C.byref(bp)))
        self.bounds = [int(bp[i]) for i in range(self.world + 1)]

    def __del__(self):
        try:
            from ._lib import lib
            if getattr(self, "_h", None):
                lib().sr_partition_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def span(self, rank):
        """(start, size) of rank's strip along the axis."""
        return self.grown(rank, 0)

    def sizes(self):
        return [self.bounds[r + 1] - self.bounds[r] for r in range(self.world)]

    def tile(self, a0, n):
        """The (y0, h, x0, w) launch rectangle of positions [a0, a0 + n) along the axis, full extent across it."""
        return (0, self.height, a0, n) if self.axis == "cols" else (a0, n, 0, self.width)

    def grown(self, rank, grow):
        """(start, size) of rank's strip grown by `grow` on both sides, clipped to the image."""
        import ctypes as C
        from ._lib import check, lib
        a, n = C.c_uint32(), C.c_uint32()
        check(lib().sr_partition_span(self._h, C.c_uint32(rank), C.c_uint32(grow), C.byref(a), C.byref(n)))
        return a.value, n.value

    def rects(self, rank):
        """SrStripRects of rank: the launch rectangles and counting window of its share of a frame."""
        import ctypes as C
        from . import abi
        from ._lib import check, lib
        r = abi.SrStripRects()
        check(lib().sr_strip_rects(self._h, C.c_uint32(rank), C.byref(r)))
        return r

    def view(self, flat, channels=None):
        """[H, W(, C)] view of a per-pixel buffer (torch tensor or numpy array of H*W rows)."""
        return flat.reshape(self.height, self.width, -1) if channels is None else flat.reshape(self.height, self.width, channels)

    def cut(self, img, a0, n):
        """Slice [a0, a0 + n) along the axis of an [H, W, C] view."""
        return img[:, a0:a0 + n] if self.axis == "cols" else img[a0:a0 + n]


def balanced_bounds(cost, world, min_size=8, max_share=2.5):
    """Cuts positions 0 .. len(cost) into `world` contiguous strips of (nearly) equal summed cost (sr_balanced_bounds): returns
    world + 1 increasing cut positions; every strip has at least `min_size` positions and at most max_share * length / world
    (the gather pads every strip to the largest one). Deterministic: every rank that feeds the same profile gets the same cut."""
    import ctypes as C
    import numpy as np
    from ._lib import check, lib
    c = np.ascontiguousarray(cost, dtype=np.float64)
    out = _u32(world + 1)
    check(lib().sr_balanced_bounds(c.ctypes.data_as(C.c_void_p), C.c_uint32(len(c)), C.c_uint32(world), C.c_uint32(min_size), C.c_double(max_share), out))
    return [int(v) for v in out]


def axis_cost_from_tiles(tile_costs, tiles_x, axis, length, tile=8):
    """Per-pixel-column (or per-pixel-row) cost from the per-tile cycle counts the library records for its own tile
    schedule (sr_scene_read_tile_costs, row-major ty * tiles_x + tx): what balanced_bounds cuts (sr_axis_cost_from_tiles)."""
    import ctypes as C
    import numpy as np
    from ._lib import check, lib
    assert tile == 8
    t = np.ascontiguousarray(tile_costs, dtype=np.float64).reshape(-1, tiles_x)
    out = np.zeros(length, dtype=np.float64)
    check(lib().sr_axis_cost_from_tiles(t.ctypes.data_as(C.c_void_p), C.c_uint32(tiles_x), C.c_uint32(t.shape[0]), C.c_uint32(0 if axis == "cols" else 1),
                                        C.c_uint32(length), out.ctypes.data_as(C.c_void_p)))
    return out


def _is_device_scene(scene):
    from . import runtime
    return isinstance(scene, runtime.Scene)


def trace_ris_strip(scene, frame, matrices, frame_count, cfg, part, rank):
    """The RIS pass of this rank's strip. With world > 1 it is ONE launch over the strip and its spatial halo: a rank's
    share of a frame is small, so every extra launch adds a tail in which the GPU drains; only the strip's own pixels
    count their rays. A device scene goes through sr_strip_trace_ris; the CPU tests' oracle scene gets the same rectangles
    (sr_strip_rects) through its own trace_ris."""
    import ctypes as C
    from ._lib import check, lib
    if _is_device_scene(scene):
        p = scene.params(frame, matrices, frame_count, cfg)
        check(lib().sr_strip_trace_ris(C.byref(p), part._h, C.c_uint32(rank), scene._stream()))
        return
    r = part.rects(rank)
    if r.empty or not cfg.enable_restir:
        return
    rcfg = copy.copy(cfg)
    if r.count_window:
        rcfg.count_y0, rcfg.count_rows, rcfg.count_x0, rcfg.count_cols = r.count_y0, r.count_rows, r.count_x0, r.count_cols
    scene.trace_ris(frame, matrices, frame_count, rcfg, tile=(r.ris_y0, r.ris_h, r.ris_x0, r.ris_w))


def trace_final_strip(scene, frame, matrices, frame_count, cfg, part, rank):
    import ctypes as C
    from ._lib import check, lib
    if _is_device_scene(scene):
        p = scene.params(frame, matrices, frame_count, cfg)
        check(lib().sr_strip_trace_final(C.byref(p), part._h, C.c_uint32(rank), scene._stream()))
        return
    r = part.rects(rank)
    if not r.empty:
        scene.trace_final(frame, matrices, frame_count, cfg, tile=(r.final_y0, r.final_h, r.final_x0, r.final_w))


def history_exchange_plan(part, motion_halo):
    """Who sends which reservoir band to whom after a RIS pass (sr_history_exchange_plan): rank r needs the pixels within
    SPATIAL_HALO + motion_halo of its strip that lie outside strip + SPATIAL_HALO (those it traced itself); every such pixel is
    owned — and was traced with exact history — by exactly one other rank. Returns a list of (src, dst, start, size) along the
    axis, in a deterministic order every rank derives alike."""
    import ctypes as C
    from . import abi
    from ._lib import check, lib
    n = C.c_uint32()
    check(lib().sr_history_exchange_plan(part._h, C.c_uint32(max(int(motion_halo), 0)), None, C.c_uint32(0), C.byref(n)))
    if n.value == 0:
        return []
    buf = (abi.SrStripTransfer * n.value)()
    check(lib().sr_history_exchange_plan(part._h, C.c_uint32(int(motion_halo)), buf, C.c_uint32(n.value), C.byref(n)))
    return [(t.src, t.dst, t.start, t.size) for t in buf]


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
