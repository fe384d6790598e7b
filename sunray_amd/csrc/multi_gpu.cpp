// Tile-parallel rendering of one frame across the GPUs of a node, host side behind the C ABI
// (include/sunray_hip.h, sr_partition_* / sr_strip_* / sr_balanced_bounds / sr_history_exchange_plan).
//
// The reference renders on one device (src/lib.rs:1166 submits to one queue); what a multi-GPU host adds is: cut the frame
// into N contiguous strips, trace each strip — the RIS pass over strip + 30-pixel spatial halo, rays counted for the strip
// only — on its own device context, exchange the temporal-history bands a moving camera needs, gather the radiance strips.
// This file owns the geometry and the plans (pure host arithmetic, every rank derives the same answers from the same
// inputs) and the two strip launches; the collective itself stays with the caller (RCCL through torch.distributed in the
// Python harness, N device contexts + one all-gather in a Rust host: INTEGRATION.md).
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "host.h"

struct SrPartition {
    uint32_t width = 0, height = 0, world = 0, axis = SR_AXIS_COLS;
    std::vector<uint32_t> bounds;     // world + 1 increasing cut positions along the axis, 0 .. length
    uint32_t length() const { return axis == SR_AXIS_COLS ? width : height; }
};

namespace {
int mfail(int code, const std::string& msg) { return srh::set_error(code, msg); }

void grown(const SrPartition& p, uint32_t rank, uint32_t grow, uint32_t& start, uint32_t& size) {
    const uint32_t a0 = p.bounds[rank], a1 = p.bounds[rank + 1];
    const uint32_t lo = a0 > grow ? a0 - grow : 0u;
    const uint32_t hi = std::min<uint64_t>(p.length(), (uint64_t)a1 + grow);
    start = lo; size = hi - lo;
}
}  // namespace

extern "C" {

int sr_partition_create(uint32_t width, uint32_t height, uint32_t world, uint32_t axis, const uint32_t* bounds, SrPartition** out) {
    if (!out || width == 0 || height == 0 || world == 0) return mfail(SR_ERR_INVALID_ARG, "sr_partition_create: empty extent, no ranks or null out");
    if (axis != SR_AXIS_COLS && axis != SR_AXIS_ROWS) return mfail(SR_ERR_INVALID_ARG, "sr_partition_create: axis must be SR_AXIS_COLS or SR_AXIS_ROWS");
    SrPartition* p = new SrPartition();
    p->width = width; p->height = height; p->world = world; p->axis = axis;
    const uint32_t length = p->length();
    p->bounds.resize((size_t)world + 1);
    if (bounds) {
        for (uint32_t r = 0; r <= world; r++) p->bounds[r] = bounds[r];
        bool ok = p->bounds[0] == 0 && p->bounds[world] == length;
        for (uint32_t r = 0; ok && r < world; r++) ok = p->bounds[r] <= p->bounds[r + 1];
        if (!ok) { delete p; return mfail(SR_ERR_INVALID_ARG, "sr_partition_create: bounds must be world + 1 increasing cuts from 0 to the axis length"); }
    } else {                                                            // equal strips, the last ones possibly shorter / empty
        const uint32_t per = (uint32_t)(((uint64_t)length + world - 1) / world);
        for (uint32_t r = 0; r < world; r++) p->bounds[r] = (uint32_t)std::min<uint64_t>((uint64_t)r * per, length);
        p->bounds[world] = length;
    }
    *out = p;
    return SR_OK;
}

int sr_partition_destroy(SrPartition* p) { delete p; return SR_OK; }

int sr_partition_get(const SrPartition* p, uint32_t* width, uint32_t* height, uint32_t* world, uint32_t* axis, const uint32_t** bounds) {
    if (!p) return mfail(SR_ERR_INVALID_ARG, "sr_partition_get: partition is null");
    if (width) *width = p->width;
    if (height) *height = p->height;
    if (world) *world = p->world;
    if (axis) *axis = p->axis;
    if (bounds) *bounds = p->bounds.data();
    return SR_OK;
}

int sr_partition_span(const SrPartition* p, uint32_t rank, uint32_t grow, uint32_t* start, uint32_t* size) {
    if (!p || !start || !size) return mfail(SR_ERR_INVALID_ARG, "sr_partition_span: null argument");
    if (rank >= p->world) return mfail(SR_ERR_INVALID_ARG, "sr_partition_span: no such rank");
    grown(*p, rank, grow, *start, *size);
    return SR_OK;
}

// Cuts positions 0 .. length into `world` contiguous strips of (nearly) equal summed cost. Strips are cut one after the
// other, each taking 1/n of the cost that is left for the n ranks that are left, with at least min_size positions and at
// most max_share * length / world (the gather pads every strip to the largest one). Deterministic: every rank that feeds
// the same profile gets the same cut.
int sr_balanced_bounds(const double* cost, uint32_t length, uint32_t world, uint32_t min_size, double max_share, uint32_t* bounds_out) {
    if (!cost || !bounds_out || length == 0 || world == 0) return mfail(SR_ERR_INVALID_ARG, "sr_balanced_bounds: null argument, empty profile or no ranks");
    if (!(max_share > 0.0)) return mfail(SR_ERR_INVALID_ARG, "sr_balanced_bounds: max_share must be positive");
    std::vector<double> cum((size_t)length + 1, 0.0);
    for (uint32_t i = 0; i < length; i++) {
        const double c = cost[i];
        if (std::isnan(c)) return mfail(SR_ERR_INVALID_ARG, "sr_balanced_bounds: cost profile holds a NaN");
        cum[i + 1] = cum[i] + (std::max(c, 0.0) + 1e-12);
    }
    const int64_t L = length, Wd = world;
    int64_t mn = std::max<int64_t>(1, std::min<int64_t>(min_size, L / Wd));
    const int64_t mx = std::max<int64_t>((int64_t)std::ceil(max_share * (double)L / (double)Wd), mn);
    bounds_out[0] = 0;
    int64_t y = 0;
    for (int64_t k = 0; k + 1 < Wd; k++) {
        const int64_t n = Wd - k;
        const double target = cum[y] + (cum[L] - cum[y]) / (double)n;
        int64_t cut = std::lower_bound(cum.begin(), cum.end(), target) - cum.begin();   // first position whose running cost reaches the target
        cut = std::min(std::max(cut, y + mn), y + mx);                   // this strip: [min_size, max_size]
        cut = std::max(cut, L - (n - 1) * mx);                           // the ranks that are left can still cover the rest ...
        cut = std::min(cut, L - (n - 1) * mn);                           // ... and each gets its minimum
        cut = std::max(cut, y);
        bounds_out[k + 1] = (uint32_t)cut;
        y = cut;
    }
    bounds_out[world] = length;
    return SR_OK;
}

// Per-pixel-column (axis = SR_AXIS_COLS) or per-pixel-row cost from the per-tile cycle counts the library records for its
// own tile schedule (sr_scene_read_tile_costs, row-major ty * tiles_x + tx, 8x8-pixel tiles): what sr_balanced_bounds cuts.
int sr_axis_cost_from_tiles(const double* tile_costs, uint32_t tiles_x, uint32_t tiles_y, uint32_t axis, uint32_t length, double* out) {
    if (!tile_costs || !out || tiles_x == 0 || tiles_y == 0) return mfail(SR_ERR_INVALID_ARG, "sr_axis_cost_from_tiles: null argument or no tiles");
    if (axis != SR_AXIS_COLS && axis != SR_AXIS_ROWS) return mfail(SR_ERR_INVALID_ARG, "sr_axis_cost_from_tiles: bad axis");
    const uint32_t n = axis == SR_AXIS_COLS ? tiles_x : tiles_y;
    if ((uint64_t)n * 8u < length) return mfail(SR_ERR_INVALID_ARG, "sr_axis_cost_from_tiles: the tiles do not cover the axis");
    std::vector<double> per(n, 0.0);
    for (uint32_t ty = 0; ty < tiles_y; ty++)
        for (uint32_t tx = 0; tx < tiles_x; tx++) per[axis == SR_AXIS_COLS ? tx : ty] += tile_costs[(size_t)ty * tiles_x + tx];
    for (uint32_t i = 0; i < length; i++) out[i] = per[i / 8u] / 8.0;
    return SR_OK;
}

// Who sends which reservoir band to whom after a RIS pass: rank r needs the pixels within SR_SPATIAL_HALO + motion_halo of
// its strip that lie outside strip + SR_SPATIAL_HALO (those it traced itself); every such pixel is owned — and was traced
// with exact history — by exactly one other rank. Deterministic order (by destination, band before / after, source).
int sr_history_exchange_plan(const SrPartition* p, uint32_t motion_halo, SrStripTransfer* out, uint32_t cap, uint32_t* count) {
    if (!p || !count) return mfail(SR_ERR_INVALID_ARG, "sr_history_exchange_plan: null argument");
    if (!out && cap > 0) return mfail(SR_ERR_INVALID_ARG, "sr_history_exchange_plan: out is null but cap > 0 (pass cap = 0 to query the count)");
    uint32_t n = 0;
    if (motion_halo > 0 && p->world > 1) {
        for (uint32_t dst = 0; dst < p->world; dst++) {
            if (p->bounds[dst + 1] == p->bounds[dst]) continue;
            uint32_t g0, gn, h0, hn;
            grown(*p, dst, SR_SPATIAL_HALO, g0, gn);
            grown(*p, dst, (uint32_t)std::min<uint64_t>((uint64_t)SR_SPATIAL_HALO + motion_halo, 0xFFFFFFFFull), h0, hn);
            const uint32_t band[2][2] = {{h0, g0}, {g0 + gn, h0 + hn}};       // before and after the traced region
            for (int b = 0; b < 2; b++)
                for (uint32_t src = 0; src < p->world; src++) {
                    if (src == dst) continue;
                    const uint32_t x0 = std::max(band[b][0], p->bounds[src]), x1 = std::min(band[b][1], p->bounds[src + 1]);
                    if (x1 > x0) {
                        if (n < cap) out[n] = SrStripTransfer{src, dst, x0, x1 - x0};
                        n++;
                    }
                }
        }
    }
    *count = n;
    return SR_OK;
}

// Launch rectangles of rank's share of one frame: the RIS pass covers the strip grown by the spatial halo (one launch;
// world == 1: the strip itself) and counts rays for the strip only; the final pass covers the strip.
int sr_strip_rects(const SrPartition* p, uint32_t rank, SrStripRects* out) {
    if (!p || !out) return mfail(SR_ERR_INVALID_ARG, "sr_strip_rects: null argument");
    if (rank >= p->world) return mfail(SR_ERR_INVALID_ARG, "sr_strip_rects: no such rank");
    const uint32_t a0 = p->bounds[rank], n = p->bounds[rank + 1] - a0;
    uint32_t g0 = a0, gn = n;
    if (p->world > 1) grown(*p, rank, SR_SPATIAL_HALO, g0, gn);
    SrStripRects r;
    r.empty = n == 0 ? 1u : 0u;
    r.count_window = (p->world > 1 && n > 0) ? 1u : 0u;
    if (p->axis == SR_AXIS_COLS) {
        r.ris_y0 = 0; r.ris_h = p->height; r.ris_x0 = g0; r.ris_w = gn;
        r.final_y0 = 0; r.final_h = p->height; r.final_x0 = a0; r.final_w = n;
        r.count_y0 = 0; r.count_rows = 0; r.count_x0 = r.count_window ? a0 : 0; r.count_cols = r.count_window ? n : 0;
    } else {
        r.ris_y0 = g0; r.ris_h = gn; r.ris_x0 = 0; r.ris_w = p->width;
        r.final_y0 = a0; r.final_h = n; r.final_x0 = 0; r.final_w = p->width;
        r.count_x0 = 0; r.count_cols = 0; r.count_y0 = r.count_window ? a0 : 0; r.count_rows = r.count_window ? n : 0;
    }
    *out = r;
    return SR_OK;
}

// sr_trace_ris / sr_trace_final of rank's share: `params` describes the whole frame (its tile_* and config.count_* fields
// are replaced); an empty strip is a no-op.
int sr_strip_trace_ris(const SrRtParams* params, const SrPartition* p, uint32_t rank, void* stream) {
    if (!params) return mfail(SR_ERR_INVALID_ARG, "sr_strip_trace_ris: params is null");
    SrStripRects r;
    int rc = sr_strip_rects(p, rank, &r);
    if (rc != SR_OK) return rc;
    if (p->width != params->width || p->height != params->height) return mfail(SR_ERR_INVALID_ARG, "sr_strip_trace_ris: the partition was made for another extent");
    if (r.empty || !params->config.enable_restir) return SR_OK;
    SrRtParams q = *params;
    q.tile_y0 = r.ris_y0; q.tile_h = r.ris_h; q.tile_x0 = r.ris_x0; q.tile_w = r.ris_w;
    if (r.count_window) { q.config.count_y0 = r.count_y0; q.config.count_rows = r.count_rows; q.config.count_x0 = r.count_x0; q.config.count_cols = r.count_cols; }
    return sr_trace_ris(&q, stream);
}

int sr_strip_trace_final(const SrRtParams* params, const SrPartition* p, uint32_t rank, void* stream) {
    if (!params) return mfail(SR_ERR_INVALID_ARG, "sr_strip_trace_final: params is null");
    SrStripRects r;
    int rc = sr_strip_rects(p, rank, &r);
    if (rc != SR_OK) return rc;
    if (p->width != params->width || p->height != params->height) return mfail(SR_ERR_INVALID_ARG, "sr_strip_trace_final: the partition was made for another extent");
    if (r.empty) return SR_OK;
    SrRtParams q = *params;
    q.tile_y0 = r.final_y0; q.tile_h = r.final_h; q.tile_x0 = r.final_x0; q.tile_w = r.final_w;
    return sr_trace_final(&q, stream);
}

}  // extern "C"
