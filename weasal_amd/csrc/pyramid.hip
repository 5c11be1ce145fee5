// weasal_amd/csrc/pyramid.hip -- the whole input pyramid of one batch behind ONE call (host code: no kernels of its own).
//
// Replaces the per-layer loop of PointCloudDataset.segmentation_inputs (datasets/common.py:461-577): per level one
// batch_neighbors for the convolutions, batch_grid_subsampling (with the random grid orientation of :77-135) and two more
// batch_neighbors for pooling / upsampling, matrices cropped to the calibrated neighbourhood limits (:336-346).  The
// schedule, radii, cell sizes and crops are those of weasal_amd/pyramid.py (which stays the general form: no limits, host
// tensors); what changes is WHO issues the ~300 launches: this function, from C, in one go.  The caller's interpreter is
// released for the whole build, so the thread that trains is not interrupted ~500 times per batch by the thread that
// builds the next pyramid (two Python threads hand the interpreter lock back and forth at every library call: measured on
// BASELINE config 2, 3.1 ms training + 1.9 ms pyramid alone became 6-7 ms together).
//
// Memory: every output lives in ONE caller-owned arena (bump-allocated here, offsets returned), temporaries in a caller-
// owned scratch.  The sizes depend on the subsampled point counts, so the subsampling of all levels runs first; if the arena
// turns out too small the call returns WS_ERR_CAPACITY with `needed_bytes` set and the caller repeats it with a larger one.
#include "ws_common.h"
#include "ws_grid.h"
#include <string.h>
#include <algorithm>

namespace {

inline int64_t al256(int64_t b) { return (b + 255) / 256 * 256; }

struct Bump {
    char* base; int64_t cap, used;
    int64_t take(int64_t bytes) { const int64_t o = used; used += al256(bytes > 0 ? bytes : 1); return o; }
};

// rows cropped to the true maximum width (what the reference's crop yields when no row reaches the limit): packed copy
__global__ __launch_bounds__(256) void pyr_trim_kernel(const int64_t* __restrict__ src, int64_t rows, int w, int mc,
                                                        int64_t* __restrict__ dst)
{
    const int64_t total = rows * (int64_t)mc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        dst[i] = src[(i / mc) * w + (i % mc)];
}

}  // namespace

extern "C" int64_t ws_pyramid_desc_bytes(void) { return (int64_t)sizeof(ws_pyramid_desc); }

extern "C" int ws_pyramid_build(ws_neighbors_ws* nws, ws_subsample_ws* sws, ws_pyramid_desc* d, void* stream)
{
    WS_REQUIRE(nws && sws && d, "NULL argument");
    WS_REQUIRE(d->n_levels >= 1 && d->n_levels <= WS_PYRAMID_MAX_LEVELS, "1 <= n_levels <= %d", WS_PYRAMID_MAX_LEVELS);
    WS_REQUIRE(d->nb >= 1 && d->nb <= WS_PYRAMID_MAX_BATCH, "1 <= nb <= %d batch elements", WS_PYRAMID_MAX_BATCH);
    WS_REQUIRE(d->points && d->arena && d->scratch && d->n0 >= 1, "NULL argument / empty batch");
    hipStream_t st = (hipStream_t)stream;
    const int L = d->n_levels, nb = d->nb;
    int64_t sum0 = 0;
    for (int b = 0; b < nb; ++b) { WS_REQUIRE(d->lens[0][b] >= 0, "negative batch length"); sum0 += d->lens[0][b]; }
    WS_REQUIRE(sum0 == d->n0, "batch lengths do not sum to the point count");
    WS_REQUIRE(d->scratch_bytes >= 2 * al256(d->n0 * 12), "scratch too small: 2 x n0 x 12 bytes (256-byte aligned) needed");
    WS_REQUIRE(((uintptr_t)d->scratch & 255) == 0 && ((uintptr_t)d->arena & 255) == 0, "arena / scratch must be 256-byte aligned");
    for (int l = 0; l < L; ++l) {
        WS_REQUIRE(d->limit[l] >= 1, "neighbourhood limits must be given (>= 1) for every level");
        if (d->pool_on[l]) WS_REQUIRE(l + 1 < L && d->limit[l + 1] >= 1 && d->dl[l] > 0.0f, "a pooling level needs a next level");
    }
    Bump A{(char*)d->arena, d->arena_bytes, 0};
    d->needed_bytes = 0;
    d->n[0] = d->n0;
    d->off_points[0] = -1;                 // level 0 is the caller's tensor
    const float* pts[WS_PYRAMID_MAX_LEVELS];
    pts[0] = d->points;
    float* tmp_a = (float*)d->scratch;
    float* tmp_b = (float*)((char*)d->scratch + al256(d->n0 * 12));
    int rc;

    // ---- phase A: subsample every level (each plan synchronises: the next level's size is data)
    int last_level = 0;
    for (int l = 0; l < L; ++l) {
        last_level = l;
        if (!d->pool_on[l]) break;
        const float* src = pts[l];
        const float* rot = d->h_rot ? d->h_rot + (size_t)l * nb * 9 : nullptr;
        if (rot) {
            if ((rc = ws_rotate_clouds_host(src, d->n[l], d->lens[l], nb, rot, 0, tmp_a, st))) return rc;
            src = tmp_a;
        }
        int64_t m = 0;
        if ((rc = ws_grid_subsample_plan(sws, src, d->n[l], d->lens[l], nb, d->dl[l], 0, WS_ORDER_REFERENCE, d->lens[l + 1], &m, st)))
            return rc;
        d->n[l + 1] = m;
        d->off_points[l + 1] = A.take(m * 12);
        const bool fits = A.used <= A.cap;
        // an arena that is already too small: keep going through the scratch halves only, to learn every level's size
        // (`needed_bytes` for the caller's second attempt); the scratch half that holds this level's input is never the target
        float* spare = (pts[l] == tmp_b) ? tmp_a : tmp_b;
        float* dst = fits ? (float*)(A.base + d->off_points[l + 1]) : nullptr;
        if (rot) {
            // src = tmp_a (rotated copy); the subsampled points go to tmp_b, then back-rotated to their place.  Without a
            // place: tmp_b -> tmp_a (the rotated input is not needed any more), and tmp_a is the next level's input
            if ((rc = ws_grid_subsample_fill(sws, nullptr, 0, nullptr, 0, tmp_b, nullptr, nullptr, nullptr, nullptr, st))) return rc;
            float* back = dst ? dst : tmp_a;
            if ((rc = ws_rotate_clouds_host(tmp_b, m, d->lens[l + 1], nb, rot, 1, back, st))) return rc;
            if (!dst) {     // the next round rotates pts[l + 1] INTO tmp_a: move it out of the way first
                WS_HIP(hipMemcpyAsync(tmp_b, tmp_a, (size_t)m * 12, hipMemcpyDeviceToDevice, st));
                pts[l + 1] = tmp_b;
            } else pts[l + 1] = dst;
        } else {
            float* out = dst ? dst : spare;
            if ((rc = ws_grid_subsample_fill(sws, nullptr, 0, nullptr, 0, out, nullptr, nullptr, nullptr, nullptr, st))) return rc;
            pts[l + 1] = out;
        }
        if (m == 0) return ws_fail(WS_ERR_EMPTY, "Error");
    }
    const int levels = last_level + 1;
    WS_REQUIRE(levels == L, "n_levels = %d but the pooling flags end the pyramid at level %d", L, levels);

    // ---- every output's place
    for (int l = 0; l < L; ++l) {
        const int64_t n = d->n[l];
        d->off_neighbors[l] = d->conv_on[l] ? A.take(n * d->limit[l] * 8) : -1;
        d->off_order[l] = d->conv_on[l] ? A.take(n * 4) : -1;
        d->off_key_last[l] = -1; d->off_blob[l] = -1; d->blob_bytes[l] = 0; d->grid_cells[l] = 0;
        if (d->conv_on[l] && d->want_grids) {
            d->off_key_last[l] = A.take(n * 8);
            d->grid_cells[l] = 4 * n + 64 * (int64_t)nb;                   // nb_prepare: cell_cap = 4 s_len + 64 per element
            d->blob_bytes[l] = ws_grid_blob_bytes(nb, d->grid_cells[l], n);
            d->off_blob[l] = A.take(d->blob_bytes[l]);
        }
        d->off_pools[l] = d->pool_on[l] ? A.take(d->n[l + 1] * d->limit[l] * 8) : -1;
        d->off_upsamples[l] = d->pool_on[l] ? A.take(n * (d->nearest_up ? 1 : d->limit[l + 1]) * 8) : -1;
    }
    for (int l = 0; l < L; ++l)
        for (int k = 0; k < 3; ++k) { d->off_toffsets[3 * l + k] = -1; d->off_tpairs[3 * l + k] = -1; d->final_width[3 * l + k] = 0; }
    int64_t trim_bytes = 0, tr_bytes = 0;          // scratch the post-processing needs
    for (int l = 0; l < L; ++l) {
        const int64_t n = d->n[l];
        if (d->conv_on[l]) trim_bytes = std::max(trim_bytes, n * d->limit[l] * 8);
        if (d->pool_on[l]) trim_bytes = std::max(trim_bytes, std::max(d->n[l + 1] * d->limit[l] * 8, n * d->limit[l + 1] * 8));
        if (!d->want_tables) continue;
        if (d->conv_on[l] && !d->want_grids) {     // (with grids the self-query layers need no table; a rare fallback builds it lazily)
            d->off_toffsets[3 * l] = A.take((n + 2) * 4);
            d->off_tpairs[3 * l] = A.take(n * d->limit[l] * 4);
            tr_bytes = std::max(tr_bytes, ws_transpose_scratch_bytes(n, d->limit[l], n));
        }
        if (d->pool_on[l]) {
            const int64_t m = d->n[l + 1];
            d->off_toffsets[3 * l + 1] = A.take((n + 2) * 4);
            d->off_tpairs[3 * l + 1] = A.take(m * d->limit[l] * 4);
            tr_bytes = std::max(tr_bytes, ws_transpose_scratch_bytes(m, d->limit[l], n));
            d->off_toffsets[3 * l + 2] = A.take((m + 2) * 4);                  // first column of the upsampling matrix
            d->off_tpairs[3 * l + 2] = A.take(n * 4);
            tr_bytes = std::max(tr_bytes, al256(n * 8) + ws_transpose_scratch_bytes(n, 1, m));
        }
    }
    WS_REQUIRE(d->scratch_bytes >= std::max(trim_bytes, tr_bytes), "scratch too small: %lld bytes needed for the cropped copies / tables",
               (long long)std::max(trim_bytes, tr_bytes));
    d->off_lens = A.take((int64_t)L * nb * 4);
    d->off_slots = A.take((int64_t)(3 * L + L) * 4);                       // max-count slots, then one overflow flag per level
    d->needed_bytes = A.used;
    if (A.used > A.cap) return ws_fail(WS_ERR_CAPACITY, "pyramid arena too small: %lld bytes needed, %lld given", (long long)A.used, (long long)A.cap);

    // ---- phase B: the searches, in the reference's order (conv, pool, upsample per level); a search whose supports and
    //      radius are those of the previous one reuses its grid (conv -> pool of a level; upsample -> conv of the next)
    int32_t* slots = (int32_t*)(A.base + d->off_slots);
    WS_HIP(hipMemsetAsync(slots, 0, (size_t)(4 * L) * 4, st));
    const float* last_s = nullptr;
    float last_r = -1.0f;
    for (int l = 0; l < L; ++l) {
        const int64_t n = d->n[l];
        for (int k = 0; k < 3; ++k) { d->max_count[3 * l + k] = -1; d->width[3 * l + k] = 0; }
        if (d->conv_on[l]) {
            if (last_s == pts[l] && last_r == d->r_conv[l]) { if ((rc = ws_radius_neighbors_reuse_grid(nws, 1))) return rc; }
            if (d->want_grids)
                if ((rc = ws_radius_neighbors_set_key_last(nws, (uint64_t*)(A.base + d->off_key_last[l])))) return rc;
            if ((rc = ws_radius_neighbors_search_async(nws, pts[l], n, pts[l], n, d->lens[l], d->lens[l], nb, d->r_conv[l], d->limit[l],
                                                       nullptr, (int64_t*)(A.base + d->off_neighbors[l]), slots + 3 * l, st)))
                return rc;
            last_s = pts[l]; last_r = d->r_conv[l];
            d->width[3 * l] = d->limit[l];
            if ((rc = ws_radius_neighbors_order(nws, (int32_t*)(A.base + d->off_order[l]), st))) return rc;
            if (d->want_grids) {
                int32_t gnb = 0; int64_t cells = 0, gns = 0, bytes = 0;
                if ((rc = ws_radius_neighbors_grid_info(nws, &gnb, &cells, &gns, &bytes))) return rc;
                WS_REQUIRE(bytes == d->blob_bytes[l] && cells == d->grid_cells[l], "grid export size differs from the planned one");
                if ((rc = ws_radius_neighbors_grid_export(nws, A.base + d->off_blob[l], st))) return rc;
            }
        }
        if (d->pool_on[l]) {
            const int64_t m = d->n[l + 1];
            if (last_s == pts[l] && last_r == d->r_pool[l]) { if ((rc = ws_radius_neighbors_reuse_grid(nws, 1))) return rc; }
            if ((rc = ws_radius_neighbors_search_async(nws, pts[l + 1], m, pts[l], n, d->lens[l + 1], d->lens[l], nb, d->r_pool[l],
                                                       d->limit[l], nullptr, (int64_t*)(A.base + d->off_pools[l]), slots + 3 * l + 1, st)))
                return rc;
            last_s = pts[l]; last_r = d->r_pool[l];
            d->width[3 * l + 1] = d->limit[l];
            const float r_up = d->r_up[l];
            if (d->nearest_up) {
                // opt-in: only the nearest support of every point (what closest_pool reads of an upsampling matrix)
                if ((rc = ws_radius_neighbors_nearest_async(nws, pts[l], n, pts[l + 1], m, d->lens[l], d->lens[l + 1], nb, r_up, nullptr,
                                                            (int64_t*)(A.base + d->off_upsamples[l]), slots + 3 * l + 2, st)))
                    return rc;
            } else if ((rc = ws_radius_neighbors_search_async(nws, pts[l], n, pts[l + 1], m, d->lens[l], d->lens[l + 1], nb, r_up,
                                                              d->limit[l + 1], nullptr, (int64_t*)(A.base + d->off_upsamples[l]),
                                                              slots + 3 * l + 2, st)))
                return rc;
            last_s = pts[l + 1]; last_r = r_up;
            d->width[3 * l + 2] = d->nearest_up ? 1 : d->limit[l + 1];
        }
    }
    // lengths of every level in one copy; true maximum row lengths back to the host; one synchronisation for all of it
    for (int l = 0; l < L; ++l)
        WS_HIP(hipMemcpyAsync(A.base + d->off_lens + (int64_t)l * nb * 4, d->lens[l], (size_t)nb * 4, hipMemcpyHostToDevice, st));
    int32_t host_slots[3 * WS_PYRAMID_MAX_LEVELS];
    WS_HIP(hipMemcpyAsync(host_slots, slots, (size_t)(3 * L) * 4, hipMemcpyDeviceToHost, st));
    WS_HIP(hipStreamSynchronize(st));
    for (int l = 0; l < L; ++l)
        for (int k = 0; k < 3; ++k)
            if (d->width[3 * l + k] > 0) d->max_count[3 * l + k] = host_slots[3 * l + k];
    // ---- the crop to the true width (datasets/common.py:336-346 crops to min(limit, widest row)), then the transposed
    //      tables of the pooling / upsampling matrices (every backward of the batch needs them; built on this stream)
    for (int l = 0; l < L; ++l)
        for (int k = 0; k < 3; ++k) {
            const int w = d->width[3 * l + k], mc = d->max_count[3 * l + k];
            if (w <= 0) continue;
            const int cap = ws_radius_neighbors_async_cap(w);
            if (mc <= 0 || mc > cap) continue;                       // empty result / slab overflow: the caller's business
            d->final_width[3 * l + k] = w;
            if (mc < w) {
                const int64_t rows = k == 1 ? d->n[l + 1] : d->n[l];
                int64_t* mat = (int64_t*)(A.base + (k == 0 ? d->off_neighbors[l] : (k == 1 ? d->off_pools[l] : d->off_upsamples[l])));
                pyr_trim_kernel<<<ws_grid(rows * mc, 256), 256, 0, st>>>(mat, rows, w, mc, (int64_t*)d->scratch);
                WS_LAUNCH_CHECK();
                WS_HIP(hipMemcpyAsync(mat, d->scratch, (size_t)(rows * mc) * 8, hipMemcpyDeviceToDevice, st));
                d->final_width[3 * l + k] = mc;
            }
        }
    if (d->want_tables)
        for (int l = 0; l < L; ++l) {
            const int64_t n = d->n[l];
            if (d->off_toffsets[3 * l] >= 0 && d->final_width[3 * l] > 0) {
                if ((rc = ws_transpose_build((const int64_t*)(A.base + d->off_neighbors[l]), n, d->final_width[3 * l], n,
                                             (int32_t*)(A.base + d->off_toffsets[3 * l]), (int32_t*)(A.base + d->off_tpairs[3 * l]),
                                             d->scratch, st)))
                    return rc;
            } else d->off_toffsets[3 * l] = -1;
            if (!d->pool_on[l]) continue;
            const int64_t m = d->n[l + 1];
            if (d->final_width[3 * l + 1] > 0) {
                if ((rc = ws_transpose_build((const int64_t*)(A.base + d->off_pools[l]), m, d->final_width[3 * l + 1], n,
                                             (int32_t*)(A.base + d->off_toffsets[3 * l + 1]), (int32_t*)(A.base + d->off_tpairs[3 * l + 1]),
                                             d->scratch, st)))
                    return rc;
            } else d->off_toffsets[3 * l + 1] = -1;
            if (d->final_width[3 * l + 2] > 0) {
                int64_t* col = (int64_t*)d->scratch;
                pyr_trim_kernel<<<ws_grid(n, 256), 256, 0, st>>>((const int64_t*)(A.base + d->off_upsamples[l]), n,
                                                                 d->final_width[3 * l + 2], 1, col);
                WS_LAUNCH_CHECK();
                if ((rc = ws_transpose_build(col, n, 1, m, (int32_t*)(A.base + d->off_toffsets[3 * l + 2]),
                                             (int32_t*)(A.base + d->off_tpairs[3 * l + 2]), (char*)d->scratch + al256(n * 8), st)))
                    return rc;
            } else d->off_toffsets[3 * l + 2] = -1;
        }
    return WS_OK;
}
