// weasal_amd/csrc/ws_grid.h -- the uniform cell grid of the radius search, shared by K1 (neighbors.hip) and the
// table-free KPConv backward (kpconv.hip).  Every floating-point step that decides a cell or a distance sits
// under `#pragma clang fp contract(off)`, so the two translation units produce the same bits whatever their
// -ffp-contract setting (the reference's recipe has no FMA: nanoflann.hpp:432-440; HIP's __fmul_rn / __fadd_rn
// are plain operators and do NOT stop the contraction).
#pragma once
#include "ws_common.h"

struct CloudGrid {        // one per batch element, device resident
    float lo[3];
    float inv_cell;
    int nx, ny, nz;
    int cell_base;        // first cell of this element in the global cell arrays
    int cell_cap;         // cells reserved for this element
    int s_base, s_len;
    int q_base, q_len;
};

#ifdef __HIPCC__
// cell coordinate; clamped so that far-away queries cannot overflow the int conversion
__device__ __forceinline__ int cell_coord(float v, float lo, float inv)
{
#pragma clang fp contract(off)
    const float d = v - lo;
    const float t = floorf(d * inv);
    return (int)fminf(fmaxf(t, -2.0f), 1.0e6f);
}

// exact reference recipe (nanoflann.hpp:432-440): result = 0; result += diff*diff, diff = query - support
__device__ __forceinline__ float ref_d2(float qx, float qy, float qz, const float4& c)
{
#pragma clang fp contract(off)
    const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float r = xx;
    r = r + yy;
    r = r + zz;
    return r;
}

// layout of an exported grid (ws_radius_neighbors_grid_export): [CloudGrid nb | cell_start cells+2 | sorted ns]
__host__ __device__ inline int64_t ws_grid_blob_align(int64_t b) { return (b + 255) / 256 * 256; }
__host__ __device__ inline int64_t ws_grid_blob_cells_off(int nb) { return ws_grid_blob_align((int64_t)nb * (int64_t)sizeof(CloudGrid)); }
__host__ __device__ inline int64_t ws_grid_blob_sorted_off(int nb, int64_t cells)
{
    return ws_grid_blob_cells_off(nb) + ws_grid_blob_align((cells + 2) * (int64_t)sizeof(int32_t));
}
__host__ __device__ inline int64_t ws_grid_blob_bytes(int nb, int64_t cells, int64_t ns)
{
    return ws_grid_blob_sorted_off(nb, cells) + ns * (int64_t)sizeof(float4);
}
#endif
