// weasal_amd/csrc/neighbors.hip -- batched fixed-radius neighbour search on gfx950.
// (compiled with -ffp-contract=off: the distance recipe must round like the reference's, no FMA)
//
// Replaces batch_nanoflann_neighbors (cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332):
// the kd-tree + per-query std::sort become
//   1. per batch element: bounding box -> uniform cell grid (cell slightly > radius), supports
//      counting-sorted by cell (x-fastest), stored as float4 (x,y,z,index) so that a run of cells
//      is one contiguous, coalesced segment;
//   2. one wave per query: the 3x3x3 cell block is 9 contiguous runs; 64 lanes test 64 candidates
//      per step with the reference's exact f32 recipe  ((dx*dx + dy*dy) + dz*dz) < r*r,
//      wave ballot + popcount-prefix compacts the hits into the wave's LDS slab;
//   3. the slab is bitonic-sorted on the 64-bit key (d2 bits << 32 | index)  (d2 >= 0, so the
//      IEEE bit pattern is monotonic; ties fall back to the index) and written as one row.
// Pass A (plan) only counts (-> max_count, the data-dependent width, neighbors.cpp:296-304),
// pass B (fill) sorts and writes int32 or int64 rows padded with ns (neighbors.cpp:324).
#include "ws_scan.h"
#include "ws_grid.h"
#include <vector>
#include <algorithm>

namespace {

__global__ __launch_bounds__(1024) void nb_bbox_kernel(const float* __restrict__ pts, CloudGrid* __restrict__ grids,
                                                        float* __restrict__ bbox /*[nb][6]*/)
{
    __shared__ float red[6][16];
    const CloudGrid g = grids[blockIdx.x];
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = threadIdx.x; i < g.s_len; i += blockDim.x) {
        const float* p = pts + 3 * (int64_t)(g.s_base + i);
#pragma unroll
        for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], p[d]); mx[d] = fmaxf(mx[d], p[d]); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], o, 64));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o, 64));
        }
        if (lane == 0) { red[d][wave] = mn[d]; red[3 + d][wave] = mx[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        float v = red[d][0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = d < 3 ? fminf(v, red[d][w]) : fmaxf(v, red[d][w]);
        bbox[blockIdx.x * 6 + d] = v;
    }
}

// The per-element table travels as a kernel argument (copied at launch): no host staging buffer
// whose lifetime would force a stream synchronisation.
constexpr int GRID_TABLE_MAX = 48;
struct GridTable { CloudGrid g[GRID_TABLE_MAX]; };
__global__ void nb_upload_kernel(GridTable t, int nb, CloudGrid* __restrict__ dst)
{
    const int b = threadIdx.x;
    if (b < nb) dst[b] = t.g[b];
}

// grid reuse: only the query ranges of the per-element table change
struct QueryTable { int q_base[GRID_TABLE_MAX]; int q_len[GRID_TABLE_MAX]; };
__global__ void nb_update_queries_kernel(QueryTable t, int nb, CloudGrid* __restrict__ grids)
{
    const int b = threadIdx.x;
    if (b < nb) { grids[b].q_base = t.q_base[b]; grids[b].q_len = t.q_len[b]; }
}

__global__ void nb_grid_setup_kernel(CloudGrid* __restrict__ grids, const float* __restrict__ bbox, int nb, float radius)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    CloudGrid g = grids[b];
    if (g.s_len <= 0) {
        g.lo[0] = g.lo[1] = g.lo[2] = 0.f; g.inv_cell = 1.f; g.nx = g.ny = g.nz = 1;
        grids[b] = g;
        return;
    }
    const float* bb = bbox + 6 * b;
    // cell > radius by a margin that dominates the rounding of (v - lo) * inv_cell (dims <= 512)
    float cell = (radius > 0.f ? radius : 1.f) * 1.001f;
    int nx, ny, nz;
    for (;;) {
        const float inv = 1.0f / cell;
        nx = (int)floorf((bb[3] - bb[0]) * inv) + 1;
        ny = (int)floorf((bb[4] - bb[1]) * inv) + 1;
        nz = (int)floorf((bb[5] - bb[2]) * inv) + 1;
        if (nx <= 512 && ny <= 512 && nz <= 512 && (int64_t)nx * ny * nz <= (int64_t)g.cell_cap) break;
        cell *= 1.26f;
    }
    g.lo[0] = bb[0]; g.lo[1] = bb[1]; g.lo[2] = bb[2];
    g.inv_cell = 1.0f / cell;
    g.nx = nx; g.ny = ny; g.nz = nz;
    grids[b] = g;
}

// one thread per support: cell id + histogram
__global__ __launch_bounds__(256) void nb_bin_count_kernel(const float* __restrict__ pts, const CloudGrid* __restrict__ grids,
                                                            int nb, int64_t ns, int32_t* __restrict__ cell_of,
                                                            int32_t* __restrict__ cell_count)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ns; i += (int64_t)gridDim.x * 256) {
        int b = 0;
        while (b + 1 < nb && i >= grids[b].s_base + grids[b].s_len) ++b;
        const CloudGrid& g = grids[b];
        const float* p = pts + 3 * i;
        int cx = min(max(cell_coord(p[0], g.lo[0], g.inv_cell), 0), g.nx - 1);
        int cy = min(max(cell_coord(p[1], g.lo[1], g.inv_cell), 0), g.ny - 1);
        int cz = min(max(cell_coord(p[2], g.lo[2], g.inv_cell), 0), g.nz - 1);
        const int c = g.cell_base + (cz * g.ny + cy) * g.nx + cx;
        cell_of[i] = c;
        atomicAdd(&cell_count[c], 1);
    }
}

__global__ __launch_bounds__(256) void nb_bin_fill_kernel(const float* __restrict__ pts, int64_t ns,
                                                           const int32_t* __restrict__ cell_of,
                                                           int32_t* __restrict__ cursor, float4* __restrict__ sorted)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ns; i += (int64_t)gridDim.x * 256) {
        const int pos = atomicAdd(&cursor[cell_of[i]], 1);
        sorted[pos] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __int_as_float((int)i));
    }
}

// Enumerate the candidates of query (qx,qy,qz): calls f(float4 cand) for all lanes in lock-step;
// every lane of the wave takes part, `cand.w` carries the global support index; inactive lanes
// get active=false.
template <typename F>
__device__ __forceinline__ void for_each_candidate(const CloudGrid& g, float qx, float qy, float qz,
                                                   const int32_t* __restrict__ cell_start,
                                                   const float4* __restrict__ sorted, int lane, F&& f)
{
    const int cx = cell_coord(qx, g.lo[0], g.inv_cell);
    const int cy = cell_coord(qy, g.lo[1], g.inv_cell);
    const int cz = cell_coord(qz, g.lo[2], g.inv_cell);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    if (x0 > x1) return;
    for (int z = max(cz - 1, 0); z <= min(cz + 1, g.nz - 1); ++z) {
        for (int y = max(cy - 1, 0); y <= min(cy + 1, g.ny - 1); ++y) {
            const int row = g.cell_base + (z * g.ny + y) * g.nx;
            const int beg = cell_start[row + x0], end = cell_start[row + x1 + 1];
            for (int p0 = beg; p0 < end; p0 += 64) {
                const int p = p0 + lane;
                const bool active = p < end;
                float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                if (active) c = sorted[p];
                f(c, active);
            }
        }
    }
}

// The running maximum of the row lengths lives in ONE word: 16 384 waves each ending in an atomicMax on it serialise at
// ~7.5 ns per atomic (measured: 0.9 ms per DALES pyramid, most of it exposed in the mid-size launches).  The maximum saturates
// after a few hundred rows, so a wave first reads the word (a stale value is only ever too small) and skips the atomic
// unless it would raise it.
__device__ __forceinline__ void nb_publish_max(int32_t* __restrict__ max_count, int local_max, int lane)
{
    if (lane == 0 && local_max > 0 &&
        local_max > __hip_atomic_load(max_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(max_count, local_max);
}

__device__ __forceinline__ int find_cloud_q(const CloudGrid* __restrict__ grids, int nb, int64_t q)
{
    int b = 0;
    while (b + 1 < nb && q >= grids[b].q_base + grids[b].q_len) ++b;
    return b;
}

__global__ __launch_bounds__(256) void nb_count_kernel(const float* __restrict__ queries, int64_t nq,
                                                        const CloudGrid* __restrict__ grids, int nb,
                                                        const int32_t* __restrict__ cell_start,
                                                        const float4* __restrict__ sorted, float r2,
                                                        const int32_t* __restrict__ qorder,
                                                        int32_t* __restrict__ counts, int32_t* __restrict__ max_count)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int local_max = 0;
    int64_t ibeg, iend;
    ws_block_range(nq, ibeg, iend);
    for (int64_t it = ibeg + wave; it < iend; it += 4) {
        const int64_t q = qorder ? (int64_t)qorder[it] : it;
        const int b = find_cloud_q(grids, nb, q);
        const CloudGrid g = grids[b];
        const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
        int cnt = 0;
        if (g.s_len > 0) {
            for_each_candidate(g, qx, qy, qz, cell_start, sorted, lane, [&](const float4& c, bool active) {
                const bool hit = active && ref_d2(qx, qy, qz, c) < r2;
                cnt += __builtin_popcountll(__ballot(hit));
            });
        }
        if (lane == 0) counts[q] = cnt;
        local_max = max(local_max, cnt);
    }
    nb_publish_max(max_count, local_max, lane);
}

template <int CAP, typename OutT>
__global__ __launch_bounds__(256) void nb_fill_kernel(const float* __restrict__ queries, int64_t nq,
                                                       const CloudGrid* __restrict__ grids, int nb,
                                                       const int32_t* __restrict__ cell_start,
                                                       const float4* __restrict__ sorted, float r2, int64_t ns,
                                                       int width, const int32_t* __restrict__ qorder, OutT* __restrict__ out,
                                                       int32_t* __restrict__ counts, int32_t* __restrict__ max_count,
                                                       unsigned long long* __restrict__ key_last)
{
    __shared__ unsigned long long slab_all[4][CAP];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned long long* slab = slab_all[wave];
    int local_max = 0;
    int64_t ibeg, iend;
    ws_block_range(nq, ibeg, iend);
    for (int64_t it = ibeg + wave; it < iend; it += 4) {
        const int64_t q = qorder ? (int64_t)qorder[it] : it;
        const int b = find_cloud_q(grids, nb, q);
        const CloudGrid g = grids[b];
        const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
        int cnt = 0;
        if (g.s_len > 0) {
            for_each_candidate(g, qx, qy, qz, cell_start, sorted, lane, [&](const float4& c, bool active) {
                const float d2 = ref_d2(qx, qy, qz, c);
                const bool hit = active && d2 < r2;
                const unsigned long long m = __ballot(hit);
                if (hit) {
                    const int pos = cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (pos < CAP)
                        slab[pos] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(c.w);
                }
                cnt += __builtin_popcountll(m);
            });
        }
        if (counts) {      // fused search: the true count (may exceed CAP -> the host reruns wider)
            if (lane == 0) counts[q] = cnt;
            local_max = max(local_max, cnt);
        }
        cnt = min(cnt, CAP);
        int m = 64;
        while (m < cnt) m <<= 1;
        for (int i = cnt + lane; i < m; i += 64) slab[i] = ~0ull;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (cnt > 1) {
            for (int k = 2; k <= m; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = lane; t < (m >> 1); t += 64) {
                        // t-th compare-exchange of this stage: insert a 0 bit at position log2(j)
                        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                        const int l = i | j;
                        const unsigned long long a = slab[i], bb = slab[l];
                        const bool up = (i & k) == 0;
                        if ((a > bb) == up) { slab[i] = bb; slab[l] = a; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        for (int j = lane; j < width; j += 64)
            out[q * width + j] = j < cnt ? (OutT)(unsigned)(slab[j] & 0xffffffffull) : (OutT)ns;
        // key of the last neighbour kept when the row is truncated, "infinity" when every neighbour is kept
        if (key_last && lane == 0) key_last[q] = cnt > width ? slab[width - 1] : ~0ull;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (max_count) nb_publish_max(max_count, local_max, lane);
}

// Fast path for rows of at most 128 neighbours (every layer of the KP-FCNN pyramids):
//  * the 9 cell runs of the 3x3x3 block are resolved first (scalar loads of the 18 run bounds) and
//    their first 64 candidates fetched together, so nine independent gathers are in flight instead
//    of nine dependent round trips;
//  * hits are compacted through the wave's LDS slab, then each lane takes two keys (i and i+64) and
//    finds their sorted positions by counting the smaller keys (LDS broadcast reads).
template <typename OutT, bool BUCKET = false>
__global__ __launch_bounds__(256) void nb_fill128_kernel(const float* __restrict__ queries, int64_t nq,
                                                          const CloudGrid* __restrict__ grids, int nb,
                                                          const int32_t* __restrict__ cell_start,
                                                          const float4* __restrict__ sorted, float r2, int64_t ns,
                                                          int width, const int32_t* __restrict__ qorder, OutT* __restrict__ out,
                                                          int32_t* __restrict__ counts, int32_t* __restrict__ max_count,
                                                          unsigned long long* __restrict__ key_last)
{
    constexpr int CAP = 128;
    __shared__ unsigned long long slab_all[4][CAP + 64];      // + one dummy slot per lane (unconditional writes)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned long long* slab = slab_all[wave];
    int local_max = 0;
    int64_t ibeg, iend;
    ws_block_range(nq, ibeg, iend);
    // queries come cloud by cloud (stacked order, and the cell order of self-queries follows the clouds'
    // cell ranges), so the element index only ever advances: keep it and its grid in registers
    int b = 0;
    CloudGrid g = grids[0];
    for (int64_t it = ibeg + wave; it < iend; it += 4) {
        const int64_t q = qorder ? (int64_t)qorder[it] : it;
        if (q < g.q_base || q >= g.q_base + g.q_len) {
            b = find_cloud_q(grids, nb, q);
            g = grids[b];
        }
        const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
        int cnt = 0;
        auto take = [&](const float4& c, bool active) {
            const float d2 = ref_d2(qx, qy, qz, c);
            const bool hit = active && d2 < r2;
            const unsigned long long m = __ballot(hit);
            // branch free: misses and overflow land in the lane's dummy slot (exec-mask branches cost the
            // scalar unit more than the store costs the LDS)
            const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            slab[(hit && pos < CAP) ? pos : CAP + lane] =
                ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(c.w);
            cnt += __builtin_popcountll(m);
        };
        if (g.s_len > 0) {
            const int cx = cell_coord(qx, g.lo[0], g.inv_cell);
            const int cy = cell_coord(qy, g.lo[1], g.inv_cell);
            const int cz = cell_coord(qz, g.lo[2], g.inv_cell);
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
            int rb[9], re[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
                const bool ok = x0 <= x1 && z >= 0 && z < g.nz && y >= 0 && y < g.ny;
                const int row = g.cell_base + ((ok ? z : 0) * g.ny + (ok ? y : 0)) * g.nx;
                rb[r] = ok ? cell_start[row + x0] : 0;
                re[r] = ok ? cell_start[row + x1 + 1] : 0;
            }
            float4 c[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int p = rb[r] + lane;
                c[r] = sorted[p < re[r] ? p : 0];          // unconditional (entry 0 exists: s_len > 0), masked by `active`
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                take(c[r], rb[r] + lane < re[r]);
                for (int p0 = rb[r] + 64; p0 < re[r]; p0 += 64) {      // runs longer than one wave (dense cells)
                    const int p = p0 + lane;
                    float4 cc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p < re[r]) cc = sorted[p];
                    take(cc, p < re[r]);
                }
            }
        }
        if (counts) {
            if (lane == 0) counts[q] = cnt;
            local_max = max(local_max, cnt);
        }
        cnt = min(cnt, CAP);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // rank by counting: the keys are distinct (the index is part of the key), so the number of smaller
        // keys IS the sorted position.  Every lane compares its (up to) two keys with key i, read as an LDS
        // broadcast -- independent iterations, no cross-lane shuffle and no dependent latency chain (the
        // 128-key bitonic network this replaces was 28 serially dependent shuffle stages).
        const unsigned long long a = lane < cnt ? slab[lane] : ~0ull;
        const unsigned long long bkey = lane + 64 < cnt ? slab[lane + 64] : ~0ull;
        int ra = 0, rb = 0;
        if constexpr (BUCKET) {
            // round 3: one level of buckets in front of the counting (as in nb_fill_wide_kernel): 64 buckets over [0, r^2),
            // bucket = (int)(d2 * 64 / r^2) is monotone in d2, so rank = keys in smaller buckets + smaller keys of the own
            // bucket -- ~1 key per bucket: a 64-wide scan and a two- or three-step count instead of cnt (60 .. 90) steps
            __shared__ int hist_all[4][64];
            __shared__ unsigned char member_all[4][128];
            int* hist = hist_all[wave];
            unsigned char* member = member_all[wave];
            hist[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float bscale = 64.0f / r2;
            const int ba = min(63, (int)(__uint_as_float((unsigned)(a >> 32)) * bscale));
            const int bb = min(63, (int)(__uint_as_float((unsigned)(bkey >> 32)) * bscale));
            int pa = 0, pb = 0;
            if (lane < cnt) pa = atomicAdd(&hist[ba], 1);
            if (lane + 64 < cnt) pb = atomicAdd(&hist[bb], 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int own = hist[lane];
            int incl = own;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                incl += lane >= o ? t : 0;
            }
            int maxb = own;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) maxb = max(maxb, __shfl_xor(maxb, o, 64));
            maxb = __builtin_amdgcn_readfirstlane(maxb);
            __shared__ int start_all[4][64];
            int* start = start_all[wave];
            start[lane] = incl - own;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int sa = start[ba], sb = start[bb];
            const int na = lane < cnt ? hist[ba] : 0, nbk = lane + 64 < cnt ? hist[bb] : 0;
            if (lane < cnt) member[sa + pa] = (unsigned char)lane;
            if (lane + 64 < cnt) member[sb + pb] = (unsigned char)(lane + 64);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            ra = sa; rb = sb;
            for (int t = 0; t < maxb; ++t) {
                const bool oka = t < na, okb = t < nbk;
                const unsigned long long ka = slab[member[oka ? sa + t : 0]];
                const unsigned long long kb = slab[member[okb ? sb + t : 0]];
                ra += (oka && ka < a) ? 1 : 0;
                rb += (okb && kb < bkey) ? 1 : 0;
            }
        } else if (cnt <= 64) {
#pragma unroll 8
            for (int i = 0; i < cnt; ++i) ra += slab[i] < a ? 1 : 0;
        } else {
#pragma unroll 8
            for (int i = 0; i < cnt; ++i) {
                const unsigned long long kk = slab[i];
                ra += kk < a ? 1 : 0;
                rb += kk < bkey ? 1 : 0;
            }
        }
        OutT* orow = out + q * width;
        if (lane < cnt && ra < width) orow[ra] = (OutT)(unsigned)(a & 0xffffffffull);
        if (lane + 64 < cnt && rb < width) orow[rb] = (OutT)(unsigned)(bkey & 0xffffffffull);
        for (int j = cnt + lane; j < width; j += 64) orow[j] = (OutT)ns;
        if (key_last) {     // the key at sorted position width-1 of a truncated row, "infinity" otherwise
            if (cnt > width) {
                if (lane < cnt && ra == width - 1) key_last[q] = a;
                if (lane + 64 < cnt && rb == width - 1) key_last[q] = bkey;
            } else if (lane == 0) {
                key_last[q] = ~0ull;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (max_count) nb_publish_max(max_count, local_max, lane);
}

// Nearest neighbour only: column 0 of the row the full search would write (the smallest (d2, index) key inside the radius),
// or ns.  What KP-FCNN reads of an UPSAMPLING matrix (closest_pool / nearest upsampling: models/blocks.py:80-92 take
// inds[:, 0]) -- without the compaction and the sort of a 60 .. 550-entry row.  Opt-in (ws_radius_neighbors_nearest_async).
template <typename OutT>
__global__ __launch_bounds__(256) void nb_nearest_kernel(const float* __restrict__ queries, int64_t nq,
                                                          const CloudGrid* __restrict__ grids, int nb,
                                                          const int32_t* __restrict__ cell_start,
                                                          const float4* __restrict__ sorted, float r2, int64_t ns,
                                                          const int32_t* __restrict__ qorder, OutT* __restrict__ out,
                                                          int32_t* __restrict__ any_hit)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int64_t ibeg, iend;
    ws_block_range(nq, ibeg, iend);
    int b = 0;
    CloudGrid g = grids[0];
    int found_any = 0;
    for (int64_t it = ibeg + wave; it < iend; it += 4) {
        const int64_t q = qorder ? (int64_t)qorder[it] : it;
        if (q < g.q_base || q >= g.q_base + g.q_len) {
            b = find_cloud_q(grids, nb, q);
            g = grids[b];
        }
        const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
        unsigned long long best = ~0ull;
        if (g.s_len > 0) {
            const int cx = cell_coord(qx, g.lo[0], g.inv_cell);
            const int cy = cell_coord(qy, g.lo[1], g.inv_cell);
            const int cz = cell_coord(qz, g.lo[2], g.inv_cell);
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
            int rb[9], re[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
                const bool ok = x0 <= x1 && z >= 0 && z < g.nz && y >= 0 && y < g.ny;
                const int row = g.cell_base + ((ok ? z : 0) * g.ny + (ok ? y : 0)) * g.nx;
                rb[r] = ok ? cell_start[row + x0] : 0;
                re[r] = ok ? cell_start[row + x1 + 1] : 0;
            }
            float4 c[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int p = rb[r] + lane;
                c[r] = sorted[p < re[r] ? p : 0];
            }
            auto take = [&](const float4& cc, bool active) {
                const float d2 = ref_d2(qx, qy, qz, cc);
                const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(cc.w);
                if (active && d2 < r2 && key < best) best = key;
            };
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                take(c[r], rb[r] + lane < re[r]);
                for (int p0 = rb[r] + 64; p0 < re[r]; p0 += 64) {
                    const int p = p0 + lane;
                    float4 cc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p < re[r]) cc = sorted[p];
                    take(cc, p < re[r]);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)best, o, 64), hi = __shfl_xor((unsigned)(best >> 32), o, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            best = other < best ? other : best;
        }
        if (lane == 0) out[q] = best != ~0ull ? (OutT)(unsigned)(best & 0xffffffffull) : (OutT)ns;
        found_any |= best != ~0ull ? 1 : 0;
    }
    if (any_hit && lane == 0 && found_any) nb_publish_max(any_hit, 1, 0);
}

// Rows of 129 .. 1024 neighbours: the deformable radius of BASELINE config 5 (datasets/common.py:500-502: every level is
// searched at 2 r, about 8 x the neighbours; calibrated limits 422 / 519 / 472).  Same candidate walk as nb_fill128; what
// changes is the sort.  Rank by counting is quadratic (500 keys: 250 000 comparisons per query) and the bitonic network
// of nb_fill_kernel<2048> runs 45 dependent LDS stages on a 64 KB slab (2 waves per SIMD): 4.2 ms per launch in round 2,
// 55 ms per config-5 step.  Here the keys go through ONE level of buckets first:
//   bucket(key) = min(NB - 1, (int)(d2 * NB / r^2))      -- monotone in d2 (a float multiply by a positive constant and
//                                                           the conversion both are), equal d2 share a bucket
//   rank(key)   = #keys in smaller buckets + #keys of the same bucket that are smaller
// so the sorted position is still a pure function of the (distinct) keys -- the LDS atomics below only hand out arbitrary
// slots INSIDE a bucket, which the final within-bucket count makes irrelevant: bit-identical rows from run to run.
// With NB = 256 a 500-key row has ~2 keys per bucket (~8 in the fullest): the within-bucket count is a handful of LDS
// reads per key, taken for all of a lane's keys together (independent chains).
template <typename OutT, int CAP>
__global__ __launch_bounds__(256) void nb_fill_wide_kernel(const float* __restrict__ queries, int64_t nq,
                                                           const CloudGrid* __restrict__ grids, int nb,
                                                           const int32_t* __restrict__ cell_start,
                                                           const float4* __restrict__ sorted, float r2, int64_t ns,
                                                           int width, const int32_t* __restrict__ qorder, OutT* __restrict__ out,
                                                           int32_t* __restrict__ counts, int32_t* __restrict__ max_count,
                                                           unsigned long long* __restrict__ key_last)
{
    constexpr int NB = 256, KPL = CAP / 64;       // CAP = 576 / 704 / 1024 keys per query: 8.3 / 9.5 / 12.7 KB of LDS per wave = 4 / 4 / 3 workgroups per CU
    __shared__ unsigned long long slab_all[4][CAP + 64];      // keys in arrival order (+ one dummy slot per lane)
    __shared__ unsigned short member_all[4][CAP];             // slab positions grouped by bucket
    __shared__ int hist_all[4][NB];
    __shared__ int start_all[4][NB];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned long long* slab = slab_all[wave];
    unsigned short* member = member_all[wave];
    int* hist = hist_all[wave];
    int* start = start_all[wave];
    const float bscale = (float)NB / r2;
    auto wsync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    int local_max = 0;
    int64_t ibeg, iend;
    ws_block_range(nq, ibeg, iend);
    int b = 0;
    CloudGrid g = grids[0];
    for (int64_t it = ibeg + wave; it < iend; it += 4) {
        const int64_t q = qorder ? (int64_t)qorder[it] : it;
        if (q < g.q_base || q >= g.q_base + g.q_len) {
            b = find_cloud_q(grids, nb, q);
            g = grids[b];
        }
        const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
        int cnt = 0;
        auto take = [&](const float4& c, bool active) {
            const float d2 = ref_d2(qx, qy, qz, c);
            const bool hit = active && d2 < r2;
            const unsigned long long m = __ballot(hit);
            const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            slab[(hit && pos < CAP) ? pos : CAP + lane] =
                ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(c.w);
            cnt += __builtin_popcountll(m);
        };
        for (int i = lane; i < NB; i += 64) hist[i] = 0;
        if (g.s_len > 0) {
            const int cx = cell_coord(qx, g.lo[0], g.inv_cell);
            const int cy = cell_coord(qy, g.lo[1], g.inv_cell);
            const int cz = cell_coord(qz, g.lo[2], g.inv_cell);
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
            int rb[9], re[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
                const bool ok = x0 <= x1 && z >= 0 && z < g.nz && y >= 0 && y < g.ny;
                const int row = g.cell_base + ((ok ? z : 0) * g.ny + (ok ? y : 0)) * g.nx;
                rb[r] = ok ? cell_start[row + x0] : 0;
                re[r] = ok ? cell_start[row + x1 + 1] : 0;
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                // runs of a few hundred candidates: two batches of 64 in flight per iteration
                for (int p0 = rb[r]; p0 < re[r]; p0 += 128) {
                    const int pa = p0 + lane, pb = p0 + 64 + lane;
                    const float4 ca = sorted[pa < re[r] ? pa : rb[r]];
                    const float4 cb = sorted[pb < re[r] ? pb : rb[r]];
                    take(ca, pa < re[r]);
                    if (p0 + 64 < re[r]) take(cb, pb < re[r]);
                }
            }
        }
        if (counts) {
            if (lane == 0) counts[q] = cnt;
            local_max = max(local_max, cnt);
        }
        const int cntc = min(cnt, CAP);
        wsync();
        // ---- A: bucket of every key, slot inside the bucket from an LDS atomic
        unsigned long long key[KPL];
        int bk[KPL], pk[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            key[j] = ~0ull; bk[j] = 0; pk[j] = 0;
            if (64 * j < cntc) {
                const int i = lane + 64 * j;
                if (i < cntc) {
                    key[j] = slab[i];
                    const float d2 = __uint_as_float((unsigned)(key[j] >> 32));
                    bk[j] = min(NB - 1, (int)(d2 * bscale));
                    pk[j] = atomicAdd(&hist[bk[j]], 1);
                }
            }
        }
        wsync();
        // ---- B: exclusive scan of the histogram (lane owns 4 consecutive buckets), largest bucket
        const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
        const int own = (h0 + h1) + (h2 + h3);
        int incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            incl += lane >= o ? t : 0;
        }
        const int excl = incl - own;
        start[4 * lane] = excl; start[4 * lane + 1] = excl + h0; start[4 * lane + 2] = excl + h0 + h1; start[4 * lane + 3] = excl + h0 + h1 + h2;
        int maxb = max(max(h0, h1), max(h2, h3));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxb = max(maxb, __shfl_xor(maxb, o, 64));
        maxb = __builtin_amdgcn_readfirstlane(maxb);
        wsync();
        // ---- C: slab positions grouped by bucket
        int sb[KPL], nbk[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            sb[j] = 0; nbk[j] = 0;
            if (64 * j < cntc) {
                const int i = lane + 64 * j;
                if (i < cntc) {
                    sb[j] = start[bk[j]];
                    nbk[j] = hist[bk[j]];
                    member[sb[j] + pk[j]] = (unsigned short)i;
                }
            }
        }
        wsync();
        // ---- D: sorted position = keys in smaller buckets + smaller keys of the own bucket
        int rank[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) rank[j] = sb[j];
        for (int t = 0; t < maxb; ++t) {
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                if (64 * j < cntc) {
                    const bool ok = t < nbk[j];
                    const unsigned long long other = slab[member[ok ? sb[j] + t : 0]];
                    rank[j] += (ok && other < key[j]) ? 1 : 0;
                }
            }
        }
        OutT* orow = out + q * width;
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            if (64 * j < cntc) {
                const int i = lane + 64 * j;
                if (i < cntc && rank[j] < width) orow[rank[j]] = (OutT)(unsigned)(key[j] & 0xffffffffull);
                if (key_last && cnt > width && i < cntc && rank[j] == width - 1) key_last[q] = key[j];
            }
        }
        for (int j = cntc + lane; j < width; j += 64) orow[j] = (OutT)ns;
        if (key_last && cnt <= width && lane == 0) key_last[q] = ~0ull;
        wsync();
    }
    if (max_count) nb_publish_max(max_count, local_max, lane);
}

// The bin fill hands out slots inside a cell through an atomic cursor (arrival order).  One thread per entry: its place
// among the entries of its cell by support index (rank by counting over the cell's few entries), written to the final copy.
// The cell-sorted copy, the cell order and the summation order of the grid-walk backward then are functions of the input alone.
__global__ __launch_bounds__(256) void nb_cell_rank_kernel(const float4* __restrict__ arrived, int64_t ns,
                                                            const int32_t* __restrict__ cell_of,
                                                            const int32_t* __restrict__ cell_start, float4* __restrict__ sorted)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < ns; p += (int64_t)gridDim.x * 256) {
        const float4 v = arrived[p];
        const int idx = __float_as_int(v.w);
        const int c = cell_of[idx];
        const int beg = cell_start[c], end = cell_start[c + 1];
        int rank = 0;
        for (int j = beg; j < end; ++j) rank += __float_as_int(arrived[j].w) < idx ? 1 : 0;
        sorted[beg + rank] = v;
    }
}

__global__ __launch_bounds__(256) void nb_order_kernel(const float4* __restrict__ sorted, int64_t ns, int32_t* __restrict__ order)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ns; i += (int64_t)gridDim.x * 256)
        order[i] = __float_as_int(sorted[i].w);
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return WS_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 64;
        WS_HIP(hipMalloc((void**)&p, want * sizeof(T)));
        cap = want;
        return WS_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

// A/B switch (WEASAL_NB_BUCKET=0/1): bucketed rank in the <= 128-neighbour search
extern "C" int ws_nb_bucket128 = 1;

struct ws_neighbors_ws {
    DevBuf<CloudGrid> grids;
    DevBuf<float> bbox;
    DevBuf<int32_t> cell_of, cell_start, cursor, counts, scan_scratch, max_count;
    DevBuf<float4> sorted, sorted_alt;      // (sorted_alt: the arrival-ordered copy the bin fill writes; see nb_cell_rank_kernel)
    DevBuf<int32_t> order;        // supports in cell order (work order of self-queries)
    bool self_query = false;      // queries == supports: walk the queries in cell order
    // state of the last plan
    const float* queries = nullptr;
    int64_t nq = 0, ns = 0;
    int nb = 0;
    float r2 = 0.f;
    int max_count_host = 0;
    int64_t cells = 0;                         // cells of the last plan (all elements)
    int32_t* max_count_word = nullptr;         // device word holding the maximum row length of the current plan
    unsigned long long* key_last = nullptr;    // one-shot output of the next fill (ws_radius_neighbors_set_key_last)
    bool reuse_next = false;                   // one-shot: the next plan keeps the support grid (ws_radius_neighbors_reuse_grid)
    const float* supports = nullptr;           // supports of the current grid
    float radius = 0.f;
    std::vector<int32_t> s_lens;               // per-element support counts of the current grid
};

extern "C" {

int ws_neighbors_ws_create(ws_neighbors_ws** ws)
{
    WS_REQUIRE(ws, "NULL argument");
    *ws = new ws_neighbors_ws();
    return WS_OK;
}

void ws_neighbors_ws_destroy(ws_neighbors_ws* ws)
{
    if (!ws) return;
    ws->grids.release(); ws->bbox.release(); ws->cell_of.release(); ws->cell_start.release();
    ws->cursor.release(); ws->counts.release(); ws->scan_scratch.release(); ws->max_count.release();
    ws->sorted.release(); ws->sorted_alt.release(); ws->order.release();
    delete ws;
}

// bounding boxes, cell grids and the counting sort of the supports (no query work, no sync
// besides the staging copy of the per-element table)
static int nb_prepare(ws_neighbors_ws* ws, const float* queries, int64_t nq, const float* supports, int64_t ns,
                      const int32_t* h_q_lens, const int32_t* h_s_lens, int32_t nb, float radius, hipStream_t st)
{
    WS_REQUIRE(ws && h_q_lens && h_s_lens, "NULL argument");
    WS_REQUIRE(nb >= 1 && nq >= 0 && ns >= 0, "bad sizes nb=%d nq=%lld ns=%lld", nb, (long long)nq, (long long)ns);
    WS_REQUIRE(ns < (1ll << 30) && nq < (1ll << 31), "point count exceeds int32 range");
    ws->max_count_host = 0;
    const bool reuse = ws->reuse_next;
    ws->reuse_next = false;
    if (reuse && ws->supports == supports && ws->ns == ns && ws->nb == nb && ws->radius == radius && nb <= GRID_TABLE_MAX &&
        ws->max_count_word && nq > 0 && (int)ws->s_lens.size() == nb &&
        std::equal(ws->s_lens.begin(), ws->s_lens.end(), h_s_lens)) {
        // same supports, same radius as the previous plan (the caller vouches that the support DATA is unchanged):
        // bounding boxes, bins and the cell-sorted copy stay; only the query ranges and the counters are new
        QueryTable qt;
        int64_t qsum = 0;
        for (int b = 0; b < nb; ++b) {
            WS_REQUIRE(h_q_lens[b] >= 0, "negative batch length");
            qt.q_base[b] = (int)qsum; qt.q_len[b] = h_q_lens[b];
            qsum += h_q_lens[b];
        }
        WS_REQUIRE(qsum == nq, "batch lengths do not sum to the query count (%lld/%lld)", (long long)qsum, (long long)nq);
        WS_REQUIRE(queries, "NULL argument");
        int rc2;
        if ((rc2 = ws->counts.ensure((size_t)nq))) return rc2;
        nb_update_queries_kernel<<<1, 64, 0, st>>>(qt, nb, ws->grids.p);
        WS_LAUNCH_CHECK();
        WS_HIP(hipMemsetAsync(ws->max_count_word, 0, sizeof(int32_t), st));
        ws->self_query = (queries == supports && nq == ns);
        ws->queries = queries; ws->nq = nq;
        return WS_OK;
    }
    ws->nq = 0;
    std::vector<CloudGrid> hg((size_t)nb);
    int64_t qsum = 0, ssum = 0, cells = 0;
    for (int b = 0; b < nb; ++b) {
        WS_REQUIRE(h_q_lens[b] >= 0 && h_s_lens[b] >= 0, "negative batch length");
        CloudGrid& g = hg[(size_t)b];
        g = CloudGrid{};
        g.q_base = (int)qsum; g.q_len = h_q_lens[b];
        g.s_base = (int)ssum; g.s_len = h_s_lens[b];
        g.cell_base = (int)cells;
        g.cell_cap = 4 * h_s_lens[b] + 64;
        cells += g.cell_cap;
        qsum += h_q_lens[b]; ssum += h_s_lens[b];
    }
    WS_REQUIRE(qsum == nq && ssum == ns, "batch lengths do not sum to the point counts (%lld/%lld, %lld/%lld)",
               (long long)qsum, (long long)nq, (long long)ssum, (long long)ns);
    WS_REQUIRE(cells < (1ll << 31) - 2, "too many grid cells");
    if (nq == 0 || ns == 0) return ws_fail(WS_ERR_EMPTY, "Error");   // wrapper.cpp:201-205
    WS_REQUIRE(queries && supports, "NULL argument");

    int rc;
    if ((rc = ws->grids.ensure((size_t)nb))) return rc;
    if ((rc = ws->bbox.ensure((size_t)nb * 6))) return rc;
    if ((rc = ws->cell_of.ensure((size_t)ns))) return rc;
    if ((rc = ws->cell_start.ensure((size_t)cells + 2))) return rc;
    if ((rc = ws->cursor.ensure((size_t)cells + 2))) return rc;
    if ((rc = ws->counts.ensure((size_t)nq))) return rc;
    if ((rc = ws->scan_scratch.ensure((size_t)ws_scan_scratch_items(cells + 1)))) return rc;
    if ((rc = ws->max_count.ensure(1))) return rc;
    if ((rc = ws->sorted.ensure((size_t)ns))) return rc;
    if ((rc = ws->sorted_alt.ensure((size_t)ns))) return rc;
    if ((rc = ws->order.ensure((size_t)ns))) return rc;

    if (nb <= GRID_TABLE_MAX) {
        GridTable tbl;
        for (int b = 0; b < nb; ++b) tbl.g[b] = hg[(size_t)b];
        nb_upload_kernel<<<1, 64, 0, st>>>(tbl, nb, ws->grids.p);
        WS_LAUNCH_CHECK();
    } else {
        WS_HIP(hipMemcpyAsync(ws->grids.p, hg.data(), sizeof(CloudGrid) * (size_t)nb, hipMemcpyHostToDevice, st));
        WS_HIP(hipStreamSynchronize(st));   // hg is a stack-lifetime staging buffer
    }
    nb_bbox_kernel<<<nb, 1024, 0, st>>>(supports, ws->grids.p, ws->bbox.p);
    WS_LAUNCH_CHECK();
    nb_grid_setup_kernel<<<(nb + 63) / 64, 64, 0, st>>>(ws->grids.p, ws->bbox.p, nb, radius);
    WS_LAUNCH_CHECK();
    // one memset clears the cell histogram AND the max-count word (kept in the spare slot behind the scan output)
    WS_HIP(hipMemsetAsync(ws->cell_start.p, 0, sizeof(int32_t) * (size_t)(cells + 2), st));
    ws->max_count_word = ws->cell_start.p + cells + 1;
    nb_bin_count_kernel<<<ws_grid(ns, 256), 256, 0, st>>>(supports, ws->grids.p, nb, ns, ws->cell_of.p, ws->cell_start.p);
    WS_LAUNCH_CHECK();
    if ((rc = ws_exclusive_scan_i32(ws->cell_start.p, ws->cell_start.p, cells, ws->scan_scratch.p, st))) return rc;
    WS_HIP(hipMemcpyAsync(ws->cursor.p, ws->cell_start.p, sizeof(int32_t) * (size_t)(cells + 1), hipMemcpyDeviceToDevice, st));
    nb_bin_fill_kernel<<<ws_grid(ns, 256), 256, 0, st>>>(supports, ns, ws->cell_of.p, ws->cursor.p, ws->sorted_alt.p);
    WS_LAUNCH_CHECK();
    // entries of a cell in index order (the bin fill wrote them in arrival order): bit-identical grids from run to run
    nb_cell_rank_kernel<<<ws_grid(ns, 256), 256, 0, st>>>(ws->sorted_alt.p, ns, ws->cell_of.p, ws->cell_start.p, ws->sorted.p);
    WS_LAUNCH_CHECK();
    nb_order_kernel<<<ws_grid(ns, 256), 256, 0, st>>>(ws->sorted.p, ns, ws->order.p);
    WS_LAUNCH_CHECK();
    ws->self_query = (queries == supports && nq == ns);
    ws->queries = queries; ws->nq = nq; ws->ns = ns; ws->nb = nb; ws->cells = cells;
    ws->supports = supports; ws->radius = radius;
    ws->s_lens.assign(h_s_lens, h_s_lens + nb);
    ws->r2 = radius * radius;   // neighbors.cpp:226
    return WS_OK;
}

int ws_radius_neighbors_plan(ws_neighbors_ws* ws, const float* queries, int64_t nq, const float* supports,
                             int64_t ns, const int32_t* h_q_lens, const int32_t* h_s_lens, int32_t nb,
                             float radius, int32_t* h_max_count, void* stream)
{
    WS_REQUIRE(h_max_count, "NULL argument");
    *h_max_count = 0;
    hipStream_t st = (hipStream_t)stream;
    int rc = nb_prepare(ws, queries, nq, supports, ns, h_q_lens, h_s_lens, nb, radius, st);
    if (rc) { if (ws) ws->nq = 0; return rc; }
    nb_count_kernel<<<ws_grid(nq, 4), 256, 0, st>>>(queries, nq, ws->grids.p, nb, ws->cell_start.p, ws->sorted.p, ws->r2,
                                                    ws->self_query ? ws->order.p : nullptr, ws->counts.p, ws->max_count_word);
    WS_LAUNCH_CHECK();
    int32_t mc = 0;
    WS_HIP(hipMemcpyAsync(&mc, ws->max_count_word, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    WS_HIP(hipStreamSynchronize(st));
    ws->max_count_host = mc;
    *h_max_count = mc;
    if (mc == 0) return ws_fail(WS_ERR_EMPTY, "Error");
    return WS_OK;
}

// the key_last request is one-shot: cleared when the public entry that consumed it returns
struct KeyLastGuard {
    ws_neighbors_ws* ws;
    ~KeyLastGuard() { if (ws) ws->key_last = nullptr; }
};

// lab switches (tools/k1_lab.py): grid cap and queries per workgroup of the fill launch
extern "C" int ws_nb_wide_caps = 1;       // 1: slab of the wide asynchronous search sized to the width (576 / 704 / 1024), 0: always 1024 (A/B: WEASAL_NB_WIDE_CAPS)
// grid cap of the fill launch on the ASYNCHRONOUS entries (the pyramid builders; the synchronous drop-in entries keep 4 096).  1 024 workgroups (4 per CU: the fill's queries are walked in cell order either way) instead of
// 4 096: alone the 13 searches of a pyramid take 1.29 instead of 1.08 ms (tools/k1_lab2.py), but they run under the training
// stream, which gets wave slots back -- DALES step 13.53 -> 13.42 ms, inference 5.20 -> 5.09 ms, config 5 38.7 -> 38.4 ms (A/B: WEASAL_NB_MAX_BLOCKS, 0 = 4 096)
extern "C" int ws_nb_max_blocks = 1024;
extern "C" int ws_nb_queries_per_block = 0;

// beside: the call comes through the asynchronous entries, i.e. from a pyramid builder working beside a training stream: the
// grid cap above applies; the synchronous entries (the drop-in `batch_query` and plan / fill) have the GPU to themselves: 4 096
static int nb_launch_fill(ws_neighbors_ws* ws, int cap, int32_t width, int32_t* out_i32, int64_t* out_i64,
                          bool with_counts, hipStream_t st, bool beside = false)
{
    const int grid = ws_grid(ws->nq, ws_nb_queries_per_block > 0 ? ws_nb_queries_per_block : 4,
                             (beside && ws_nb_max_blocks > 0) ? ws_nb_max_blocks : 256 * 16);
    const int32_t* qo = ws->self_query ? ws->order.p : nullptr;
    int32_t* cn = with_counts ? ws->counts.p : nullptr;
    int32_t* mx = with_counts ? ws->max_count_word : nullptr;
    unsigned long long* kl = ws->key_last;
#define WS_NB_FILL(CAP)                                                                                            \
    do {                                                                                                           \
        if (out_i32)                                                                                               \
            nb_fill_kernel<CAP, int32_t><<<grid, 256, 0, st>>>(ws->queries, ws->nq, ws->grids.p, ws->nb, ws->cell_start.p, \
                                                               ws->sorted.p, ws->r2, ws->ns, width, qo, out_i32, cn, mx, kl);  \
        else                                                                                                       \
            nb_fill_kernel<CAP, int64_t><<<grid, 256, 0, st>>>(ws->queries, ws->nq, ws->grids.p, ws->nb, ws->cell_start.p, \
                                                               ws->sorted.p, ws->r2, ws->ns, width, qo, out_i64, cn, mx, kl);  \
    } while (0)
    if (cap <= 128) {
#define WS_NB128(OT, BK, OUT)                                                                                                \
    nb_fill128_kernel<OT, BK><<<grid, 256, 0, st>>>(ws->queries, ws->nq, ws->grids.p, ws->nb, ws->cell_start.p, ws->sorted.p, \
                                                    ws->r2, ws->ns, width, qo, OUT, cn, mx, kl)
        if (ws_nb_bucket128) { if (out_i32) WS_NB128(int32_t, true, out_i32); else WS_NB128(int64_t, true, out_i64); }
        else { if (out_i32) WS_NB128(int32_t, false, out_i32); else WS_NB128(int64_t, false, out_i64); }
#undef WS_NB128
    }
    else if (cap <= 1024) {
#define WS_NBW(OT, CAPV, OUT)                                                                                          \
    nb_fill_wide_kernel<OT, CAPV><<<grid, 256, 0, st>>>(ws->queries, ws->nq, ws->grids.p, ws->nb, ws->cell_start.p,    \
                                                        ws->sorted.p, ws->r2, ws->ns, width, qo, OUT, cn, mx, kl)
        if (cap <= 576) { if (out_i32) WS_NBW(int32_t, 576, out_i32); else WS_NBW(int64_t, 576, out_i64); }
        else if (cap <= 704) { if (out_i32) WS_NBW(int32_t, 704, out_i32); else WS_NBW(int64_t, 704, out_i64); }
        else { if (out_i32) WS_NBW(int32_t, 1024, out_i32); else WS_NBW(int64_t, 1024, out_i64); }
#undef WS_NBW
    }
    else if (cap <= 2048) WS_NB_FILL(2048);
    else return ws_fail(WS_ERR_UNSUPPORTED, "max neighbour count %d exceeds the 2048-entry sort slab", cap);
#undef WS_NB_FILL
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_radius_neighbors_search(ws_neighbors_ws* ws, const float* queries, int64_t nq, const float* supports,
                               int64_t ns, const int32_t* h_q_lens, const int32_t* h_s_lens, int32_t nb,
                               float radius, int32_t width, int32_t* out_i32, int64_t* out_i64,
                               int32_t* h_max_count, void* stream)
{
    KeyLastGuard guard{ws};
    WS_REQUIRE(h_max_count, "NULL argument");
    WS_REQUIRE((out_i32 != nullptr) != (out_i64 != nullptr), "exactly one of out_i32 / out_i64 must be given");
    WS_REQUIRE(width >= 1, "width must be >= 1");
    *h_max_count = 0;
    hipStream_t st = (hipStream_t)stream;
    int rc = nb_prepare(ws, queries, nq, supports, ns, h_q_lens, h_s_lens, nb, radius, st);
    if (rc) { if (ws) ws->nq = 0; return rc; }
    int cap = width > 128 ? 1024 : 128;      // wide rows are asked for: the bucketed sort from the start
    for (;;) {
        WS_HIP(hipMemsetAsync(ws->max_count_word, 0, sizeof(int32_t), st));
        if ((rc = nb_launch_fill(ws, cap, width, out_i32, out_i64, true, st))) return rc;
        int32_t mc = 0;
        WS_HIP(hipMemcpyAsync(&mc, ws->max_count_word, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        WS_HIP(hipStreamSynchronize(st));
        ws->max_count_host = mc;
        *h_max_count = mc;
        if (mc == 0) return ws_fail(WS_ERR_EMPTY, "Error");
        if (mc <= cap) return WS_OK;
        cap = mc;      // a row overflowed the sort slab: run once more with a slab that fits
    }
}

// Row slab of the asynchronous search for rows of `width` entries.  The slab must hold EVERY neighbour inside the radius (the row
// keeps the nearest `width` of them); the calibrated limits are 90th percentiles of the counts, the largest count of a batch lies
// 5-20 % above them (config 5: limits 422 / 519 / 472, maxima 491 / 552 / 565), so 5/4 of the width rounded up to the next
// kernel variant covers it with fewer LDS bytes per wave (4 instead of 3 workgroups per CU); a batch that does overflow is
// reported through d_max_count and redone by the caller, as before.
int32_t ws_radius_neighbors_async_cap(int32_t width)
{
    if (width <= 128) return 128;
    if (!ws_nb_wide_caps) return 1024;
    const int need = width + width / 4;
    return need <= 576 ? 576 : (need <= 704 ? 704 : 1024);
}

int ws_radius_neighbors_search_async(ws_neighbors_ws* ws, const float* queries, int64_t nq, const float* supports,
                                     int64_t ns, const int32_t* h_q_lens, const int32_t* h_s_lens, int32_t nb,
                                     float radius, int32_t width, int32_t* out_i32, int64_t* out_i64,
                                     int32_t* d_max_count, void* stream)
{
    KeyLastGuard guard{ws};
    WS_REQUIRE(d_max_count, "NULL argument");
    WS_REQUIRE((out_i32 != nullptr) != (out_i64 != nullptr), "exactly one of out_i32 / out_i64 must be given");
    WS_REQUIRE(width >= 1, "width must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    int rc = nb_prepare(ws, queries, nq, supports, ns, h_q_lens, h_s_lens, nb, radius, st);
    if (rc) { if (ws) ws->nq = 0; return rc; }
    // rows wider than the 128-entry fast path are asked for (deformable radius): the 1024-key bucketed sort from the start,
    // instead of a 128-entry pass the caller would have to repeat
    const int cap = ws_radius_neighbors_async_cap(width);
    if ((rc = nb_launch_fill(ws, cap, width, out_i32, out_i64, true, st, true))) return rc;   // max-count word cleared by nb_prepare
    WS_HIP(hipMemcpyAsync(d_max_count, ws->max_count_word, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    ws->max_count_host = cap;   // unknown on the host; rows beyond the slab are reported through d_max_count
    return WS_OK;
}

int ws_radius_neighbors_nearest_async(ws_neighbors_ws* ws, const float* queries, int64_t nq, const float* supports,
                                      int64_t ns, const int32_t* h_q_lens, const int32_t* h_s_lens, int32_t nb,
                                      float radius, int32_t* out_i32, int64_t* out_i64, int32_t* d_any, void* stream)
{
    KeyLastGuard guard{ws};
    WS_REQUIRE(d_any, "NULL argument");
    WS_REQUIRE((out_i32 != nullptr) != (out_i64 != nullptr), "exactly one of out_i32 / out_i64 must be given");
    hipStream_t st = (hipStream_t)stream;
    int rc = nb_prepare(ws, queries, nq, supports, ns, h_q_lens, h_s_lens, nb, radius, st);
    if (rc) { if (ws) ws->nq = 0; return rc; }
    const int grid = ws_grid(ws->nq, 4);
    const int32_t* qo = ws->self_query ? ws->order.p : nullptr;
    if (out_i32)
        nb_nearest_kernel<int32_t><<<grid, 256, 0, st>>>(ws->queries, ws->nq, ws->grids.p, ws->nb, ws->cell_start.p, ws->sorted.p, ws->r2,
                                                         ws->ns, qo, out_i32, ws->max_count_word);
    else
        nb_nearest_kernel<int64_t><<<grid, 256, 0, st>>>(ws->queries, ws->nq, ws->grids.p, ws->nb, ws->cell_start.p, ws->sorted.p, ws->r2,
                                                         ws->ns, qo, out_i64, ws->max_count_word);
    WS_LAUNCH_CHECK();
    WS_HIP(hipMemcpyAsync(d_any, ws->max_count_word, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    ws->max_count_host = 1;
    return WS_OK;
}

int ws_radius_neighbors_fill(ws_neighbors_ws* ws, int32_t width, int32_t* out_i32, int64_t* out_i64, void* stream)
{
    KeyLastGuard guard{ws};
    WS_REQUIRE(ws && ws->nq > 0 && ws->max_count_host > 0, "no successful plan to fill from");
    WS_REQUIRE((out_i32 != nullptr) != (out_i64 != nullptr), "exactly one of out_i32 / out_i64 must be given");
    WS_REQUIRE(width >= 1 && width <= ws->max_count_host, "width %d outside [1, max_count=%d]", width, ws->max_count_host);
    return nb_launch_fill(ws, ws->max_count_host, width, out_i32, out_i64, false, (hipStream_t)stream);
}

int ws_radius_neighbors_order(const ws_neighbors_ws* ws, int32_t* out_order, void* stream)
{
    WS_REQUIRE(ws && ws->ns > 0 && out_order, "no plan / NULL argument");
    WS_HIP(hipMemcpyAsync(out_order, ws->order.p, sizeof(int32_t) * (size_t)ws->ns, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return WS_OK;
}

const int32_t* ws_radius_neighbors_counts(const ws_neighbors_ws* ws) { return ws ? ws->counts.p : nullptr; }

int ws_radius_neighbors_reuse_grid(ws_neighbors_ws* ws, int32_t on)
{
    WS_REQUIRE(ws, "NULL argument");
    ws->reuse_next = on != 0;
    return WS_OK;
}

int ws_radius_neighbors_set_key_last(ws_neighbors_ws* ws, uint64_t* d_key_last)
{
    WS_REQUIRE(ws, "NULL argument");
    ws->key_last = reinterpret_cast<unsigned long long*>(d_key_last);
    return WS_OK;
}

int ws_radius_neighbors_grid_info(const ws_neighbors_ws* ws, int32_t* nb, int64_t* cells, int64_t* ns, int64_t* blob_bytes)
{
    WS_REQUIRE(ws && ws->nq > 0 && nb && cells && ns && blob_bytes, "no plan / NULL argument");
    *nb = ws->nb; *cells = ws->cells; *ns = ws->ns;
    *blob_bytes = ws_grid_blob_bytes(ws->nb, ws->cells, ws->ns);
    return WS_OK;
}

int ws_radius_neighbors_grid_export(const ws_neighbors_ws* ws, void* blob, void* stream)
{
    WS_REQUIRE(ws && ws->nq > 0 && blob, "no plan / NULL argument");
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)blob;
    WS_HIP(hipMemcpyAsync(base, ws->grids.p, sizeof(CloudGrid) * (size_t)ws->nb, hipMemcpyDeviceToDevice, st));
    WS_HIP(hipMemcpyAsync(base + ws_grid_blob_cells_off(ws->nb), ws->cell_start.p, sizeof(int32_t) * (size_t)(ws->cells + 2),
                          hipMemcpyDeviceToDevice, st));
    WS_HIP(hipMemcpyAsync(base + ws_grid_blob_sorted_off(ws->nb, ws->cells), ws->sorted.p, sizeof(float4) * (size_t)ws->ns,
                          hipMemcpyDeviceToDevice, st));
    return WS_OK;
}

}  // extern "C"
