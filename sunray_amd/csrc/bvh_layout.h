// Layout of the quantised wide BVH node shared by the host builder, the device builder / refit and the traversal.
//   dword 0..2   origin (fp32)            dword 3   exponents ex | ey << 8 | ez << 16 (biased, plane = fma(q, 2^e, origin))
//   then six planes LX LY LZ HX HY HZ, one byte per child (kPlaneDwords dwords each),
//   then kBvhWidth child references at kChildOffset (>= 0 inner node index, < 0 leaf ~((first << 3) | count), ~0 = unused).
// Width 4: 64-byte nodes (dwords 10, 11 unused). Width 8: 96 bytes used, padded to 128 so a node never straddles a cache line.
#pragma once
#ifndef SR_BVH_WIDTH
#define SR_BVH_WIDTH 4
#endif
namespace srl {
constexpr int kBvhWidth = SR_BVH_WIDTH;
static_assert(kBvhWidth == 4 || kBvhWidth == 8, "SR_BVH_WIDTH must be 4 or 8");
constexpr int kPlaneDwords = kBvhWidth / 4;
constexpr int kPlaneOffset = 4;
constexpr int kChildOffset = kBvhWidth == 4 ? 12 : 16;
constexpr int kNodeDwords = kBvhWidth == 4 ? 16 : 32;
constexpr int kNodeBytes = kNodeDwords * 4;
// Triangles per leaf (the leaf reference holds the count in 3 bits: <= 7). Measured on the bench frame with the host SAH
// tree: 1 -> 3.11, 2 -> 3.65, 3 -> 3.64, 4 -> 3.50, 6 -> 3.36 Gray/s (fewer triangle tests per ray beat fewer nodes).
#ifndef SR_LEAF_MAX
#define SR_LEAF_MAX 2
#endif
constexpr unsigned kLeafMax = SR_LEAF_MAX;
}  // namespace srl
