// Device side of the acceleration-structure maintenance: what the reference gets from
// vkCmdBuildAccelerationStructuresKHR in UPDATE mode (tlas.rs:124-140, accel.rs:263-267) — new instance
// transforms, same topology — done here as (1) re-flattening the instances' triangles to world space, in the
// same operation order as the host (bvh_build.cpp flatten_instances) so the records are bit-identical to a
// rebuild's, and (2) a bottom-up refit of the quantised 4-wide nodes. Box culling is conservative on every
// side, so query results do not depend on which tree answers them.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "bvh_gpu.h"

namespace srd {

__device__ __forceinline__ void world_triangle(const SrMeshInfo* meshes, const FlatInstance& inst, uint32_t prim, float v0[3], float e1[3], float e2[3]) {
    const SrMeshInfo mi = meshes[inst.mesh_slot];
    const uint32_t* idx = (const uint32_t*)(uintptr_t)mi.indices;
    const SrVertex* vtx = (const SrVertex*)(uintptr_t)mi.vertices;
    const float* m = inst.o2w;
    float w[3][3];
    for (int j = 0; j < 3; j++) {
        const float* q = vtx[idx[3 * prim + j]].position;
        // transform_point (rt_utils.slang:278-281): rows dotted with (p, 1), left to right
        w[j][0] = ((m[0] * q[0] + m[1] * q[1]) + m[2] * q[2]) + m[3] * 1.0f;
        w[j][1] = ((m[4] * q[0] + m[5] * q[1]) + m[6] * q[2]) + m[7] * 1.0f;
        w[j][2] = ((m[8] * q[0] + m[9] * q[1]) + m[10] * q[2]) + m[11] * 1.0f;
    }
    for (int a = 0; a < 3; a++) { v0[a] = w[0][a]; e1[a] = w[1][a] - w[0][a]; e2[a] = w[2][a] - w[0][a]; }
}

// One thread per leaf-order slot: the slot keeps its triangle (global id), only the world-space record changes.
__global__ void flatten_slots_kernel(float4* tris, const float4* shade, const SrMeshInfo* meshes, const FlatInstance* instances, uint32_t n_tris) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_tris) return;
    const uint32_t gid = __float_as_uint(tris[(size_t)slot * 3 + 2].y);   // record: (v0, e1, e2, gid, 0, 0), bvh_build.cpp
    const uint32_t ii = __float_as_uint(shade[(size_t)slot * 3 + 2].y);
    const FlatInstance inst = instances[ii];
    if (gid < inst.tri_offset) return;      // cannot happen for a tree built from this layout; never index out of bounds
    float v0[3], e1[3], e2[3];
    world_triangle(meshes, inst, gid - inst.tri_offset, v0, e1, e2);
    tris[(size_t)slot * 3 + 0] = make_float4(v0[0], v0[1], v0[2], e1[0]);
    tris[(size_t)slot * 3 + 1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
    tris[(size_t)slot * 3 + 2] = make_float4(e2[2], __uint_as_float(gid), 0.0f, 0.0f);
}

// Padded box of one triangle record, as the host builder bounds it (bvh_build.cpp: fattened by the triangle
// test's barycentric slack, then one ulp each way).
__device__ __forceinline__ void tri_box(const float4* tris, uint32_t slot, float lo[3], float hi[3]) {
    const float4 a = tris[(size_t)slot * 3 + 0], b = tris[(size_t)slot * 3 + 1], c = tris[(size_t)slot * 3 + 2];
    const float v0[3] = {a.x, a.y, a.z}, e1[3] = {a.w, b.x, b.y}, e2[3] = {b.z, b.w, c.x};
    for (int k = 0; k < 3; k++) {
        const float p1 = v0[k] + e1[k], p2 = v0[k] + e2[k];
        const float pad = 4e-6f * (fabsf(e1[k]) + fabsf(e2[k]));
        const float l = fminf(v0[k], fminf(p1, p2)) - pad, h = fmaxf(v0[k], fmaxf(p1, p2)) + pad;
        lo[k] = fminf(lo[k], nextafterf(l, -INFINITY));
        hi[k] = fmaxf(hi[k], nextafterf(h, INFINITY));
    }
}

// One thread per node of one tree level (deepest level first): child boxes from the triangles (leaf children) or
// from the already refitted child nodes (node_box), then the same quantisation the host collapser applies
// (bvh_build.cpp Collapser::emit): origin = node min, per-axis power-of-two grid, planes rounded outward and
// verified with the decode expression fmaf(q, 2^e, origin).
__global__ void refit_level_kernel(uint32_t* nodes, const float4* tris, float* node_box, const uint32_t* level_nodes, uint32_t first, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t node = level_nodes[first + i];
    uint32_t* q = nodes + (size_t)node * 16;
    float lo[4][3], hi[4][3];
    bool real[4];
    float lo_n[3] = {INFINITY, INFINITY, INFINITY}, hi_n[3] = {-INFINITY, -INFINITY, -INFINITY};
    int n_real = 0;
    for (int c = 0; c < 4; c++) {
        const int ref = (int)q[12 + c];
        for (int a = 0; a < 3; a++) { lo[c][a] = INFINITY; hi[c][a] = -INFINITY; }
        real[c] = false;
        if (ref >= 0) {
            const float* b = node_box + (size_t)ref * 6;
            for (int a = 0; a < 3; a++) { lo[c][a] = b[a]; hi[c][a] = b[3 + a]; }
            real[c] = lo[c][0] <= hi[c][0];
        } else {
            const uint32_t v = ~(uint32_t)ref, t0 = v >> 3, cnt = v & 7u;
            for (uint32_t t = 0; t < cnt; t++) tri_box(tris, t0 + t, lo[c], hi[c]);
            real[c] = cnt != 0u;
        }
        if (!real[c]) continue;
        n_real++;
        for (int a = 0; a < 3; a++) { lo_n[a] = fminf(lo_n[a], lo[c][a]); hi_n[a] = fmaxf(hi_n[a], hi[c][a]); }
    }
    uint32_t plane[6] = {0, 0, 0, 0, 0, 0}, exps = 0;
    float origin[3] = {0.0f, 0.0f, 0.0f};
    if (n_real > 0) {
        for (int a = 0; a < 3; a++) {
            origin[a] = lo_n[a];
            const float ext = hi_n[a] - lo_n[a];
            int e = -126;
            if (ext > 0.0f && ext < INFINITY) { int fe; (void)frexpf(ext / 255.0f, &fe); e = max(fe, -126); }
            if (!(ext < INFINITY)) e = 127;
            for (; e < 127; e++) {   // grow the grid until every child's upper plane fits in a byte
                const float scale = ldexpf(1.0f, e);
                bool ok = true;
                for (int c = 0; c < 4 && ok; c++) {
                    if (!real[c]) continue;
                    int qh = (int)ceil(((double)hi[c][a] - (double)origin[a]) / (double)scale);
                    qh = max(qh, 0);
                    while (qh <= 255 && fmaf((float)qh, scale, origin[a]) < hi[c][a]) qh++;
                    if (qh > 255) ok = false;
                }
                if (ok) break;
            }
            const float scale = ldexpf(1.0f, e);
            exps |= (uint32_t)(e + 127) << (8 * a);
            for (int c = 0; c < 4; c++) {
                uint32_t ql = 255u, qh = 0u;   // inverted box for unused children
                if (real[c]) {
                    int l = (int)floor(((double)lo[c][a] - (double)origin[a]) / (double)scale);
                    l = min(max(l, 0), 255);
                    while (l > 0 && fmaf((float)l, scale, origin[a]) > lo[c][a]) l--;
                    int h = (int)ceil(((double)hi[c][a] - (double)origin[a]) / (double)scale);
                    h = min(max(h, 0), 255);
                    while (h < 255 && fmaf((float)h, scale, origin[a]) < hi[c][a]) h++;
                    ql = (uint32_t)l; qh = (uint32_t)h;
                }
                plane[a] |= ql << (8 * c);
                plane[3 + a] |= qh << (8 * c);
            }
        }
    } else {
        for (int a = 0; a < 3; a++) { plane[a] = 0xFFFFFFFFu; plane[3 + a] = 0u; exps |= 127u << (8 * a); }
    }
    q[0] = __float_as_uint(origin[0]); q[1] = __float_as_uint(origin[1]); q[2] = __float_as_uint(origin[2]); q[3] = exps;
    q[4] = plane[0]; q[5] = plane[1]; q[6] = plane[2]; q[7] = plane[3];
    q[8] = plane[4]; q[9] = plane[5];
    float* b = node_box + (size_t)node * 6;
    for (int a = 0; a < 3; a++) { b[a] = lo_n[a]; b[3 + a] = hi_n[a]; }
}

}  // namespace srd

using namespace srd;

int srk_launch_flatten_slots(float4* tris, const float4* shade, const SrMeshInfo* meshes, const FlatInstance* instances, uint32_t n_tris, hipStream_t stream) {
    if (n_tris == 0) return 0;
    flatten_slots_kernel<<<dim3((n_tris + 255) / 256), dim3(256), 0, stream>>>(tris, shade, meshes, instances, n_tris);
    return (int)hipGetLastError();
}

int srk_launch_refit(uint32_t* nodes, const float4* tris, float* node_box, const uint32_t* level_nodes, const uint32_t* level_offsets_host,
                     uint32_t n_levels, hipStream_t stream) {
    for (uint32_t l = 0; l < n_levels; l++) {     // level_offsets_host[l] .. [l+1]: deepest level first
        const uint32_t first = level_offsets_host[l], count = level_offsets_host[l + 1] - first;
        if (count == 0) continue;
        refit_level_kernel<<<dim3((count + 63) / 64), dim3(64), 0, stream>>>(nodes, tris, node_box, level_nodes, first, count);
    }
    return (int)hipGetLastError();
}
