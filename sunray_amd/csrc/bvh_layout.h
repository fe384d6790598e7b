// Layout of the quantised 4-wide BVH node shared by the host builder, the device builder / refit and the traversal.
// 64 bytes = one cache-line half, four 16-byte gathers:
//   dword 0..2   origin (fp32)            dword 3   exponents ex | ey << 8 | ez << 16 (biased, plane = fma(q, 2^e, origin))
//   dword 4..9   six planes LX LY LZ HX HY HZ, one byte per child          dword 10, 11 unused
//   dword 12..15 child references: >= 0 inner node index, < 0 leaf ~((first << 3) | count), ~0 = unused
// (An 8-wide variant of this layout — 128-byte nodes — was built and measured in round 1: 34 % fewer node steps, each
// 2.1x as expensive, frame 3.92 vs 2.87 ms; DESIGN.md section 5. The kernels read 4-wide nodes only.)
#pragma once
namespace srl {
constexpr int kBvhWidth = 4;
constexpr int kPlaneDwords = kBvhWidth / 4;
constexpr int kPlaneOffset = 4;
constexpr int kChildOffset = 12;
constexpr int kNodeDwords = 16;
constexpr int kNodeBytes = kNodeDwords * 4;
// Triangles per leaf (the leaf reference holds the count in 3 bits: <= 7). Measured on the bench frame with the host SAH
// tree: 1 -> 3.11, 2 -> 3.65, 3 -> 3.64, 4 -> 3.50, 6 -> 3.36 Gray/s (fewer triangle tests per ray beat fewer nodes).
#ifndef SR_LEAF_MAX
#define SR_LEAF_MAX 2
#endif
constexpr unsigned kLeafMax = SR_LEAF_MAX;
}  // namespace srl
