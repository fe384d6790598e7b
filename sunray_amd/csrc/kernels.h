// Kernel argument blocks and host-side launchers shared by kernels.hip and api.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include "traverse.h"

namespace srd {

// Everything one per-pixel pass needs, passed by value in the kernarg segment — the counterpart of
// the 144-byte push constant both reference passes receive (rt_types.slang:151-190).
struct PassArgs {
    DevScene sc;
    SrMatrices mats;
    float* raw_color;
    uint16_t* depth_img;
    uint32_t* normal_img;
    uint32_t* diffuse_img;
    uint32_t* motion_vec_img;
    const uint8_t* blue_noise_tex;
    uint32_t blue_noise_w, blue_noise_h;
    SrReservoir* reservoirs[2];
    SrReservoirGI* reservoirs_gi[2];
    // primary-hit hand-off (SrRtParams.primary_payload): the RIS pass stores the shaded payload of its virtual bounce 0,
    // the final pass starts from it instead of tracing the same camera ray again; null = the final pass traces it
    SrRayPayload* primary_payload;
    uint32_t frame_count;
    uint32_t width, height;
    uint32_t y0, y1;              // rows [y0, y1) and
    uint32_t x0, x1;              // columns [x0, x1) of the image are traced by this launch
    uint32_t tiles_x, tiles_y;    // pixel tiles covering (x1 - x0) x (y1 - y0); filled in by srk_launch_pass
    uint32_t order_cap;           // blocks per XCD = entries per XCD in tile_order (srk_pass_order_cap)
    // cost-ordered tile schedule (kernels.hip): this launch's per-tile cost is written to tile_cost, the order derived
    // from the PREVIOUS launch's costs is read from tile_order (null on the first launch of a geometry)
    uint32_t* tile_cost;
    const uint32_t* tile_order;
    SrTraceConfig cfg;
};

}  // namespace srd

int srk_launch_trace(const srd::DevScene& sc, const SrRay* rays, uint32_t n, SrHit* hits, uint32_t* occluded,
                     int any, int stats, int two_level, int stack_entries, hipStream_t stream);
int srk_launch_shade(const srd::DevScene& sc, const SrHit* hits, uint32_t n, SrRayPayload* out, hipStream_t stream);
int srk_launch_any_hit(const srd::DevScene& sc, const SrHit* hits, uint32_t n, uint32_t* ignored, hipStream_t stream);
int srk_launch_pass(const srd::PassArgs& args, int which, int stats, int textured, int two_level, int stack_entries, hipStream_t stream);
// Tile schedule of the next launch from this launch's costs: per XCD band, tiles in descending cost (64 buckets).
int srk_lds_rows(int stack_entries, int two_level);
uint32_t srk_pass_tile_count(uint32_t width, uint32_t rows);
uint32_t srk_pass_order_cap(uint32_t width, uint32_t rows);
int srk_launch_tile_order(const uint32_t* tile_cost, uint32_t* tile_order, uint32_t width, uint32_t rows, hipStream_t stream);

int srk_launch_post_temporal(const SrPostParams& p, hipStream_t stream);
int srk_launch_post_denoise(const SrPostParams& p, hipStream_t stream);
int srk_launch_post_tonemap(const SrPostParams& p, hipStream_t stream);
