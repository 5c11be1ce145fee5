// weasal_amd/csrc/csr.hip -- transposed neighbour table (support -> sorted list of pair ids).
//
// The autograd of the reference's gather (models/blocks.py:36-67, used at :281,:360 and in
// max_pool/closest_pool :80-111) is a scatter_add over the [nq,h] index matrix.  Here the matrix
// is inverted once per (layer, batch) by a counting sort, and every backward kernel *gathers*
// through it: deterministic, no float atomics.
//   1. histogram of inds (int atomics)      2. exclusive scan -> t_offsets [ns+2]
//   3. fill with an atomic cursor           4. per-list sort by pair id (restores a fixed order)
#include "ws_scan.h"

namespace {

// pass 1: histogram; the returning atomic also gives every pair its rank inside its list
__global__ __launch_bounds__(256) void tr_count(const int64_t* __restrict__ inds, int64_t np, int64_t ns,
                                                 int32_t* __restrict__ counts, int32_t* __restrict__ rank)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (int64_t)gridDim.x * 256) {
        const int64_t s = inds[p];
        int r = -1;                               // shadow pairs carry no gradient: not tabulated
        if (s >= 0 && s < ns) r = atomicAdd(&counts[s], 1);
        rank[p] = r;
    }
}

// pass 2 (after the scan): plain scatter, no atomics
__global__ __launch_bounds__(256) void tr_fill(const int64_t* __restrict__ inds, int64_t np,
                                                const int32_t* __restrict__ offsets, const int32_t* __restrict__ rank,
                                                int32_t* __restrict__ pairs)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (int64_t)gridDim.x * 256) {
        const int r = rank[p];
        if (r >= 0) pairs[offsets[inds[p]] + r] = (int32_t)p;
    }
}

// bitonic sort of one list per wave, in the wave's LDS slab (lists up to SORT_CAP), ascending.
constexpr int SORT_CAP = 2048;
__global__ __launch_bounds__(256) void tr_sort_lists(const int32_t* __restrict__ offsets, int64_t nlists,
                                                      int32_t* __restrict__ pairs)
{
    __shared__ int32_t slab_all[4][SORT_CAP];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int32_t* slab = slab_all[wave];
    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < nlists; s += (int64_t)gridDim.x * 4) {
        const int beg = offsets[s], end = offsets[s + 1];
        const int n = end - beg;
        if (n <= 1) continue;
        if (n <= 64) {
            // rank by counting: pair ids are distinct, so the number of smaller ids is the sorted position.
            // Key i is broadcast with v_readlane (uniform loop index): independent compare + add per
            // iteration instead of 21 dependent shuffle stages of a bitonic network.
            const int v = lane < n ? pairs[beg + lane] : 0x7fffffff;
            int r = 0;
            for (int i = 0; i < n; ++i) r += __builtin_amdgcn_readlane(v, i) < v ? 1 : 0;
            if (lane < n) pairs[beg + r] = v;
        } else if (n <= SORT_CAP) {
            int m = 64;
            while (m < n) m <<= 1;
            for (int i = lane; i < m; i += 64) slab[i] = i < n ? pairs[beg + i] : 0x7fffffff;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int k = 2; k <= m; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = lane; i < m; i += 64) {
                        const int l = i ^ j;
                        if (l > i) {
                            const int a = slab[i], b = slab[l];
                            const bool up = (i & k) == 0;
                            if ((a > b) == up) { slab[i] = b; slab[l] = a; }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            for (int i = lane; i < n; i += 64) pairs[beg + i] = slab[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            // very long list: odd-even transposition in global memory by one wave (rare; correct, slow)
            for (int pass = 0; pass < n; ++pass) {
                for (int i = (pass & 1) + 2 * lane; i + 1 < n; i += 128) {
                    const int a = pairs[beg + i], b = pairs[beg + i + 1];
                    if (a > b) { pairs[beg + i] = b; pairs[beg + i + 1] = a; }
                }
                __threadfence();   // global memory hand-off between lanes: agent scope
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

}  // namespace

extern "C" {

int64_t ws_transpose_scratch_bytes(int64_t nq, int32_t h, int64_t ns)
{
        // rank [nq*h] + scan scratch
    return (int64_t)sizeof(int32_t) * (nq * (int64_t)h + ws_scan_scratch_items(ns + 1)) + 64;
}

int ws_transpose_build(const int64_t* inds, int64_t nq, int32_t h, int64_t ns, int32_t* t_offsets,
                       int32_t* t_pairs, void* scratch, void* stream)
{
    WS_REQUIRE(nq >= 0 && ns >= 0 && h >= 1, "bad sizes nq=%lld ns=%lld h=%d", (long long)nq, (long long)ns, h);
    WS_REQUIRE(t_offsets && scratch && (nq == 0 || (inds && t_pairs)), "NULL argument");
    const int64_t np = nq * (int64_t)h;
    WS_REQUIRE(np < (1ll << 31) && ns < (1ll << 31) - 2, "nq*h or ns exceeds int32");
    hipStream_t st = (hipStream_t)stream;
    int32_t* rank = (int32_t*)scratch;
    int32_t* scan_scratch = rank + np;
    const int64_t nlists = ns + 1;   // list ns = shadow pairs
    WS_HIP(hipMemsetAsync(t_offsets, 0, sizeof(int32_t) * (ns + 2), st));
    if (np > 0) {
        tr_count<<<ws_grid(np, 256), 256, 0, st>>>(inds, np, ns, t_offsets, rank);
        WS_LAUNCH_CHECK();
    }
    int rc = ws_exclusive_scan_i32(t_offsets, t_offsets, nlists, scan_scratch, st);
    if (rc) return rc;
    if (np > 0) {
        tr_fill<<<ws_grid(np, 256), 256, 0, st>>>(inds, np, t_offsets, rank, t_pairs);
        WS_LAUNCH_CHECK();
        // the shadow list (slot ns) is never read by a backward kernel: not sorted
        tr_sort_lists<<<ws_grid(ns, 4), 256, 0, st>>>(t_offsets, ns, t_pairs);
        WS_LAUNCH_CHECK();
    }
    return WS_OK;
}

}  // extern "C"
