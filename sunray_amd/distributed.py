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


def render_strip(scene, frame, matrices, frame_count, cfg, world, rank, uncounted_flag=1):
    """Traces this rank's part of one frame into the full-size buffers of `frame`.

    `scene` is a sunray_amd.runtime.Scene (GPU) — or, in the CPU tests, the oracle's scene: both offer
    trace_ris / trace_final(frame, matrices, frame_count, cfg, tile=(y0, h))."""
    y0, h = strip_rows(frame.height, world, rank)
    if h <= 0:
        return y0, h
    if cfg.enable_restir:
        scene.trace_ris(frame, matrices, frame_count, cfg, tile=(y0, h))
        if world > 1:
            hcfg = copy.copy(cfg)
            hcfg.flags = cfg.flags | uncounted_flag
            for band in halo_bands(frame.height, y0, h):
                scene.trace_ris(frame, matrices, frame_count, hcfg, tile=band)
    scene.trace_final(frame, matrices, frame_count, cfg, tile=(y0, h))
    return y0, h


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
