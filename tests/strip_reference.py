"""The round-1/2 Python implementations of the strip geometry and plans (Partition, balanced_bounds, axis_cost_from_tiles,
history_exchange_plan), kept as the reference the C++ port behind the C ABI (csrc/multi_gpu.cpp, sr_partition_* /
sr_balanced_bounds / sr_axis_cost_from_tiles / sr_history_exchange_plan) is checked against (tests/test_host_abi.py)."""
SPATIAL_HALO = 30


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
