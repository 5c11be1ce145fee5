/* include/weasal_hip.h -- C ABI of libweasal_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary of the KPConv hot path: plain pointers and sizes, no
 * torch / numpy / C++ types.  Every pointer is a DEVICE pointer unless the parameter
 * name starts with `h_` (host).  `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  All functions are asynchronous on `stream` unless documented
 * otherwise, never allocate device memory unless documented, and return a ws_status.
 *
 * Each entry names the reference interface it replaces (paths relative to the reference
 * repository JohannesErnst/WeaSAL).  INTEGRATION.md shows the Python-side bindings.
 */
#ifndef WEASAL_HIP_H
#define WEASAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    WS_OK = 0,
    WS_ERR_INVALID = 1,      /* bad shape / NULL pointer / inconsistent sizes          */
    WS_ERR_UNSUPPORTED = 2,  /* valid request this build has no kernel for             */
    WS_ERR_HIP = 3,          /* a HIP runtime call failed (see ws_last_error)          */
    WS_ERR_EMPTY = 4,        /* empty result: the reference raises RuntimeError("Error")
                                (cpp_wrappers/cpp_neighbors/wrapper.cpp:201-205,
                                 cpp_wrappers/cpp_subsampling/wrapper.cpp:266-270)     */
    WS_ERR_CAPACITY = 5      /* a caller-provided buffer is too small                  */
} ws_status;

/* KP_influence / aggregation_mode of models/blocks.py:147 */
enum { WS_INFLUENCE_LINEAR = 0, WS_INFLUENCE_CONSTANT = 1, WS_INFLUENCE_GAUSSIAN = 2 };
enum { WS_AGGREGATION_SUM = 0, WS_AGGREGATION_CLOSEST = 1 };

const char* ws_last_error(void);
/* kernels launched by this library since it was loaded (all threads, all streams): launches-per-step accounting */
int64_t ws_launch_count(void);   /* thread-local message of the last failing call */
const char* ws_version(void);
int ws_device_count(void);         /* number of HIP devices visible (no context is created) */

/* ------------------------------------------------------------------------------------------
 * KPConv (models/blocks.py:238-374).
 *
 * Shapes:  q_pts [nq,3] f32, s_pts [ns,3] f32, inds [nq,h] int64 with values in [0,ns]
 * (ns = shadow neighbour: point (1e6,1e6,1e6), zero feature, blocks.py:278,357),
 * x [ns,ci] f32, kernel_points [k,3] f32.
 *
 * ws_kpconv_gather_fwd: the fused neighbour-gather -> kernel-point influence -> feature
 *   aggregate (blocks.py:278-363):   wf[q,kk,c] = sum_h w(q,h,kk) * x[inds[q,h], c]
 *   with w = clamp(1 - sqrt(d2)/extent, 0) (linear, :337), 1 (constant, :332) or
 *   exp(-d2 / (2 (0.3 extent)^2 + 1e-9)) (gaussian, :343 and :70-77); aggregation CLOSEST
 *   keeps only the arg-min kernel point of each neighbour (:349-351).
 *   wf is [nq, k, ci] row-major, so that  out = wf.reshape(nq, k*ci) @ weights.reshape(k*ci, co)
 *   is the dense contraction of blocks.py:370-374 (done by the caller on MFMA).
 *   Deformable variant (blocks.py:244-325): deformed_kp [nq,k,3] (= kernel_points + offsets,
 *   :288) replaces kernel_points when non-NULL; neighbours with no kernel point within
 *   `extent` contribute nothing (:301-325); min_d2 [nq,k] (:304) is written when non-NULL;
 *   modulations [nq,k] (:256,:367) scale wf[q,kk,:] when non-NULL.
 * ws_kpconv_gather_bwd_x: dx[s,c] = sum over (q,h) with inds[q,h]==s of
 *   sum_kk w(q,h,kk) * (mod[q,kk]) * dwf[q,kk,c]   -- the autograd of the gather
 *   (a scatter_add in the reference), computed as a deterministic gather over the transposed
 *   neighbour table built by ws_transpose_build.
 * ws_kpconv_gather_bwd_geom (deformable only): gradients w.r.t. deformed_kp [nq,k,3] and
 *   modulations [nq,k], from dwf and from d(min_d2).
 * ------------------------------------------------------------------------------------------ */
int ws_kpconv_gather_fwd(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                         const int64_t* inds, int32_t h,
                         const float* x, int32_t ci,
                         const float* kernel_points, int32_t k,
                         const float* deformed_kp,    /* NULL = rigid */
                         const float* modulations,    /* NULL = none  */
                         float extent, int32_t influence, int32_t aggregation,
                         const int32_t* order,        /* NULL or a permutation of [0,nq): the order in which
                                                         queries are scheduled (results do not depend on it;
                                                         a spatial order keeps gathered rows in L2) */
                         float* wf,                   /* out [nq,k,ci] */
                         float* min_d2,               /* out [nq,k] or NULL */
                         void* stream);

int ws_kpconv_gather_bwd_x(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                           const int64_t* inds, int32_t h,
                           const int32_t* t_offsets,  /* [ns+2] from ws_transpose_build */
                           const int32_t* t_pairs,    /* [nq*h]  from ws_transpose_build */
                           const float* dwf, int32_t ci,      /* [nq,k,ci] */
                           const float* kernel_points, int32_t k,
                           const float* deformed_kp, const float* modulations,
                           float extent, int32_t influence, int32_t aggregation,
                           const int32_t* order,      /* NULL or a permutation of [0,ns): scheduling order */
                           float* dx,                 /* out [ns,ci] (fully overwritten) */
                           void* stream);

int ws_kpconv_gather_bwd_geom(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                              const int64_t* inds, int32_t h,
                              const float* x, int32_t ci,
                              const float* dwf,          /* [nq,k,ci] */
                              const float* kernel_points, int32_t k,
                              const float* deformed_kp,  /* [nq,k,3], required */
                              const float* modulations,  /* [nq,k] or NULL */
                              const float* d_min_d2,     /* [nq,k] or NULL */
                              float extent, int32_t influence, int32_t aggregation,
                              float* d_deformed_kp,      /* out [nq,k,3] */
                              float* d_modulations,      /* out [nq,k] or NULL */
                              void* stream);

/* Transposed neighbour table (support -> list of flat pair ids q*h+col), the deterministic
 * replacement of the scatter_add in the autograd of blocks.gather (blocks.py:36-67).
 * t_offsets [ns+2] int32: entries of support s are t_pairs[t_offsets[s] .. t_offsets[s+1]);
 * shadow pairs (index == ns) are not tabulated (slot ns stays empty).  Lists are sorted by pair id.
 * Requires nq*h < 2^31.
 * scratch: at least ws_transpose_scratch_bytes(nq,h,ns) bytes. */
int64_t ws_transpose_scratch_bytes(int64_t nq, int32_t h, int64_t ns);
int ws_transpose_build(const int64_t* inds, int64_t nq, int32_t h, int64_t ns,
                       int32_t* t_offsets, int32_t* t_pairs, void* scratch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pooling helpers (models/blocks.py:80-111).
 * ws_max_pool_fwd: out[q,c] = max_h xpad[inds[q,h], c] with xpad = [x; 0] (blocks.py:95-111);
 *   arg[q,c] = the column h of the (first) maximum, kept for the backward.
 * ws_max_pool_bwd: dx[s,c] = sum of dy[q,c] over (q,h) with inds[q,h]==s and arg[q,c]==h.
 * ws_closest_pool_fwd: out[q,:] = xpad[inds[q,0], :] (blocks.py:80-92).
 * ws_closest_pool_bwd: dx[s,:] = sum of dy[q,:] over q with inds[q,0]==s.
 * The backward forms take the transposed table of `inds` (ws_transpose_build).
 * ------------------------------------------------------------------------------------------ */
int ws_max_pool_fwd(const float* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                    float* out, int32_t* arg, void* stream);
int ws_max_pool_bwd(const float* dy, const int32_t* arg, int64_t nq, int32_t h, int32_t c,
                    const int32_t* t_offsets, const int32_t* t_pairs, int64_t ns,
                    float* dx, void* stream);
/* the same with a spatially coherent walking order (int32 permutation: the cell order of the queries' / supports' level,
 * ws_radius_neighbors_order; NULL = index order): results do not depend on it, only the memory traffic does -- consecutive
 * work items gather overlapping rows, and the workgroups of an XCD walk its part of the order together (pools.hip). */
int ws_max_pool_fwd_ordered(const float* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h, float* out,
                            int32_t* arg, const int32_t* order_q, void* stream);
int ws_max_pool_bwd_ordered(const float* dy, const int32_t* arg, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                            const int32_t* t_pairs, int64_t ns, float* dx, const int32_t* order_s, void* stream);
int ws_max_pool_fwd_ordered_bf16(const uint16_t* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h, uint16_t* out,
                                 int32_t* arg, const int32_t* order_q, void* stream);
int ws_max_pool_bwd_ordered_bf16(const uint16_t* dy, const int32_t* arg, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                                 const int32_t* t_pairs, int64_t ns, uint16_t* dx, const int32_t* order_s, void* stream);
int ws_closest_pool_fwd(const float* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                        float* out, void* stream);
int ws_closest_pool_bwd(const float* dy, int64_t nq, int32_t h, int32_t c,
                        const int32_t* t_offsets, const int32_t* t_pairs, int64_t ns,
                        float* dx, void* stream);

/* ------------------------------------------------------------------------------------------
 * Tall-skinny fp32 GEMMs on the f32-input MFMA -- the dense parts of the path:
 *   the unary 1x1 MLPs  y = x W^T (models/blocks.py:490-501, nn.Linear without bias) and the kernel
 *   contraction  out = wf[N,15Ci] @ weights[15Ci,Co]  (blocks.py:370-374), plus their autograd.
 * ws_gemm_xb :  y[m,n] = x[m,k] @ b[k,n]        (b row-major, ld = n; x/y row-major with ldx/ldy)
 * ws_gemm_xty:  out[k,n] = x[m,k]^T @ y[m,n]    (reduction over the tall dimension m; partial sums per
 *               row chunk in `scratch` (>= ws_gemm_xty_scratch_bytes), added in a fixed order)
 * Exact fp32 products and sums (v_mfma_f32_32x32x2_f32), i.e. rocBLAS-equivalent up to summation order.
 * ------------------------------------------------------------------------------------------ */
int ws_gemm_xb(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n,
               float* y, int64_t ldy, void* stream);
/* y = act(x @ b + bias[n] + residual[m,n]) -- the BatchNormBlock bias (models/blocks.py:465), the
 * residual sum and LeakyReLU(0.1) of the blocks (blocks.py:497-500, :563-564, :709) applied in the GEMM
 * epilogue.  bias / residual may be NULL; act: 0 = none, 1 = LeakyReLU(slope). */
int ws_gemm_xb_epilogue(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n,
                        const float* bias, const float* residual, int64_t ldr, int32_t act, float slope,
                        float* y, int64_t ldy, void* stream);
/* the same with split-K for short, deep products (m below ~32 k rows, k >= 512: the bottom of the pyramid):
 * up to 16 workgroup layers contract disjoint k ranges into `scratch` (>= ws_gemm_xb_scratch_bytes, may be
 * 0 = never split) and a second kernel adds them in a fixed order and applies the epilogue. */
int64_t ws_gemm_xb_scratch_bytes(int64_t m, int32_t k, int32_t n);
int ws_gemm_xb_epilogue_splitk(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n,
                               const float* bias, const float* residual, int64_t ldr, int32_t act, float slope,
                               float* y, int64_t ldy, void* scratch, int64_t scratch_bytes, void* stream);
/* backward of that epilogue in one pass over [m,n]: dz = dy * (y > 0 ? 1 : slope)  (LeakyReLU backward from
 * the OUTPUT y, models/blocks.py:500,564,709; y == NULL: no activation, dz untouched) and
 * colsum[n] = sum over rows of dz (the gradient of the BatchNormBlock bias, blocks.py:465; NULL: skipped).
 * Partial sums per row chunk in `scratch` (>= ws_act_bwd_colsum_scratch_bytes), added in a fixed order. */
int64_t ws_act_bwd_colsum_scratch_bytes(int64_t m, int32_t n);
int ws_act_bwd_colsum(const float* dy, int64_t m, int32_t n, int64_t lddy, const float* y, int64_t ldy, float slope,
                      float* dz, int64_t lddz, float* colsum, void* scratch, void* stream);
int64_t ws_gemm_xty_scratch_bytes(int64_t m, int32_t k, int32_t n);
int ws_gemm_xty(const float* x, int64_t m, int32_t k, int64_t ldx, const float* y, int32_t n, int64_t ldy,
                float* out, void* scratch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Radius neighbours -- replaces cpp_wrappers/cpp_neighbors (radius_neighbors.batch_query,
 * wrapper.cpp:58-238 -> batch_nanoflann_neighbors, neighbors/neighbors.cpp:211-332).
 * For every query of batch element b: supports j of the same element with
 *   fl(fl(fl(dx*dx)+fl(dy*dy))+fl(dz*dz)) < fl(radius*radius)     (no FMA contraction)
 * sorted ascending by that d2 (ties: by index), as global support indices; rows padded with ns.
 *
 * Two-call protocol (the width is data dependent, neighbors.cpp:296-304):
 *   ws_radius_neighbors_plan   builds the per-element cell grid in `ws` and counts; synchronises
 *                              `stream` and returns *h_max_count (host).  WS_ERR_EMPTY if it is 0
 *                              or nq == 0.
 *   ws_radius_neighbors_fill   writes out[nq, width] with width <= max_count columns (cropping
 *                              columns is what datasets/common.py:336-346 does afterwards);
 *                              exactly one of out_i32 / out_i64 is non-NULL (int64 is what
 *                              common.py:551-553 converts to).
 * q_lens / s_lens are DEVICE int32 [nb]; h_q_lens / h_s_lens the same values on the host.
 * ------------------------------------------------------------------------------------------ */
typedef struct ws_neighbors_ws ws_neighbors_ws;     /* opaque workspace, owns device scratch */
int ws_neighbors_ws_create(ws_neighbors_ws** ws);
void ws_neighbors_ws_destroy(ws_neighbors_ws* ws);
int ws_radius_neighbors_plan(ws_neighbors_ws* ws,
                             const float* queries, int64_t nq, const float* supports, int64_t ns,
                             const int32_t* h_q_lens, const int32_t* h_s_lens, int32_t nb,
                             float radius, int32_t* h_max_count, void* stream);
int ws_radius_neighbors_fill(ws_neighbors_ws* ws, int32_t width,
                             int32_t* out_i32, int64_t* out_i64, void* stream);
/* One-call form for callers that crop anyway (datasets/common.py:336-346): builds the grid and makes
 * ONE query pass that writes out[nq,width] (rows padded with ns) and counts at the same time;
 * synchronises and returns the true *h_max_count -- if it is smaller than `width` the caller keeps the
 * first max_count columns (the rest is padding).  WS_ERR_EMPTY as above. */
int ws_radius_neighbors_search(ws_neighbors_ws* ws,
                               const float* queries, int64_t nq, const float* supports, int64_t ns,
                               const int32_t* h_q_lens, const int32_t* h_s_lens, int32_t nb,
                               float radius, int32_t width, int32_t* out_i32, int64_t* out_i64,
                               int32_t* h_max_count, void* stream);
/* Same single pass without any host synchronisation: the true max count is written to the DEVICE
 * int32 *d_max_count.  The rows are exact when that value is <= 128 (the sort slab of the fast
 * kernel); the caller checks the value later (e.g. once per batch for all its searches) and repeats
 * a search with ws_radius_neighbors_search if it was larger, or trims columns if it was smaller than
 * `width`.  An empty result shows as *d_max_count == 0. */
/* capacity (neighbours inside the radius per query) of the row slab ws_radius_neighbors_search_async uses for rows of `width`
 * entries: 128 up to width 128, else 576 / 704 / 1024; a maximum count beyond it (d_max_count) means truncated candidates:
 * the caller repeats that search with the two-call protocol. */
int32_t ws_radius_neighbors_async_cap(int32_t width);
int ws_radius_neighbors_search_async(ws_neighbors_ws* ws,
                                     const float* queries, int64_t nq, const float* supports, int64_t ns,
                                     const int32_t* h_q_lens, const int32_t* h_s_lens, int32_t nb,
                                     float radius, int32_t width, int32_t* out_i32, int64_t* out_i64,
                                     int32_t* d_max_count, void* stream);
/* Nearest neighbour only (opt-in): out [nq] = column 0 of the row ws_radius_neighbors_search_async would write -- the support
 * with the smallest (distance, index) inside `radius`, or ns -- without building and sorting the row.  For the UPSAMPLING
 * searches of the pyramid, of which KP-FCNN reads the first column only (models/blocks.py:80-92); *d_any (device) becomes
 * 1 if any query found a support, else 0.  Same grid reuse (ws_radius_neighbors_reuse_grid) as the full search. */
int ws_radius_neighbors_nearest_async(ws_neighbors_ws* ws, const float* queries, int64_t nq, const float* supports,
                                      int64_t ns, const int32_t* h_q_lens, const int32_t* h_s_lens, int32_t nb,
                                      float radius, int32_t* out_i32, int64_t* out_i64, int32_t* d_any, void* stream);
/* supports of the last plan/search in cell order (device int32 [ns], a permutation): a spatially
 * coherent scheduling order for the kernels that take `order`. */
int ws_radius_neighbors_order(const ws_neighbors_ws* ws, int32_t* out_order, void* stream);
/* per-query neighbour counts of the last plan (device int32 [nq]); valid until the next plan.
 * Feeds the neighbourhood-limit calibration (datasets/DALES_PseudoLabel.py:1238-1240). */
const int32_t* ws_radius_neighbors_counts(const ws_neighbors_ws* ws);

/* Table-free backward of self-query KPConv layers (queries == supports): the cell grid of the last search and the
 * key of the last neighbour kept per query replace the transposed table (ws_transpose_build).
 *   ws_radius_neighbors_set_key_last  one-shot request: the NEXT search/fill also writes key_last[nq]
 *                                     = (d2 bits << 32 | index) of the last kept neighbour, ~0 for untruncated rows
 *   ws_radius_neighbors_grid_info / _grid_export   copy the grid of the last plan (per-element table, cell offsets,
 *                                     cell-sorted supports) into a caller-owned device blob of `blob_bytes`
 *   ws_kpconv_gather_bwd_x_grid       = ws_kpconv_gather_bwd_x for q_pts == s_pts, pairs re-derived per support from
 *                                     that blob: q -> s iff d2 < radius^2 and (d2, s) <= key_last[q]; same pair order,
 *                                     same sums.  `overflow` (device int32, zero-initialised by the caller) receives the
 *                                     in-degree of a support with more than 192 incoming pairs (result then invalid);
 *                                     impossible when the search reported max_count <= 128 for that matrix. */
int ws_radius_neighbors_set_key_last(ws_neighbors_ws* ws, uint64_t* d_key_last);
/* one-shot hint: the NEXT plan / search uses the same supports (pointer, lengths, radius AND unchanged contents -- the
 * caller vouches) as the previous one and keeps its cell grid; ignored when anything checkable differs.  In the pyramid
 * of datasets/common.py:487-545 the conv, pool and (previous level's) upsample searches share one support set and one
 * radius per level: 5 grids instead of 13. */
int ws_radius_neighbors_reuse_grid(ws_neighbors_ws* ws, int32_t on);
int ws_radius_neighbors_grid_info(const ws_neighbors_ws* ws, int32_t* nb, int64_t* cells, int64_t* ns, int64_t* blob_bytes);
int ws_radius_neighbors_grid_export(const ws_neighbors_ws* ws, void* blob, void* stream);
int ws_kpconv_gather_bwd_x_grid(const float* s_pts, int64_t ns, const void* grid_blob, int32_t nb, int64_t cells,
                                const uint64_t* key_last, float radius, const float* dwf, int32_t ci,
                                const float* kernel_points, int32_t k, const float* deformed_kp, const float* modulations,
                                float extent, int32_t influence, int32_t aggregation, const int32_t* order, float* dx,
                                int32_t* overflow, void* stream);

/* K4 / K4G with the activation backward of the preceding unary block folded into the store: dx * LeakyReLU'(gate_y),
 * gate_y [ns, ci] = that block's activated output (this layer's input x); gate_y NULL = the plain entries.
 * K4G also takes `rows` [ns, rows_h] (NULL = none): the index matrix of the same self-query search.  A support whose own
 * row was not truncated (key_last[s] = all ones) finds the queries that kept it among its row's entries (the distance is
 * symmetric bit for bit) instead of walking the 27 cells around it; truncated rows still walk the grid.
 * Replaces the autograd of nn.LeakyReLU between unary1 and KPConv (models/blocks.py:676-683). */
int ws_kpconv_gather_bwd_x_gated(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                                 const int64_t* inds, int32_t h, const int32_t* t_offsets, const int32_t* t_pairs,
                                 const float* dwf, int32_t ci, const float* kernel_points, int32_t k,
                                 const float* deformed_kp, const float* modulations, float extent,
                                 int32_t influence, int32_t aggregation, const int32_t* order, const float* gate_y,
                                 float gate_slope, float* dx, void* stream);
int ws_kpconv_gather_bwd_x_grid_gated(const float* s_pts, int64_t ns, const void* grid_blob, int32_t nb, int64_t cells,
                                      const uint64_t* key_last, float radius, const float* dwf, int32_t ci,
                                      const float* kernel_points, int32_t k, const float* deformed_kp, const float* modulations,
                                      float extent, int32_t influence, int32_t aggregation, const int32_t* order,
                                      const float* gate_y, float gate_slope, const int64_t* rows, int32_t rows_h, float* dx,
                                      int32_t* overflow, void* stream);

/* ------------------------------------------------------------------------------------------
 * Grid subsampling -- replaces cpp_wrappers/cpp_subsampling (grid_subsampling.subsample_batch /
 * subsample, wrapper.cpp:62-333,338-566 -> grid_subsampling.cpp:5-106,109-211).
 * Per batch element: origin = floor(min*(1/dl))*dl, cell key = iX + nX*iY + nX*nY*iZ (all f32,
 * IEEE divide), per cell the SEQUENTIAL f32 sum of its points in input order times
 * (float)(1.0/count); features: sequential sum / (float)count; labels: arg-max of the per-cell
 * histogram (first-seen label wins ties -- the reference's tie order is implementation-defined).
 * order_mode WS_ORDER_REFERENCE reproduces the reference's row order (the iteration order of a
 * libstdc++ unordered_map filled in first-occurrence order); WS_ORDER_FIRST_SEEN orders cells by
 * first occurrence (cheaper; same set of rows).
 *
 * Two-call protocol:
 *   ws_grid_subsample_plan  bins the points, synchronises and returns the per-element row counts
 *                           h_out_lens[nb] (after max_p truncation) and *h_m = their sum.
 *   ws_grid_subsample_fill  writes out_points [m,3] (+ out_features [m,fd], out_labels [m,ld]).
 * ------------------------------------------------------------------------------------------ */
enum { WS_ORDER_REFERENCE = 0, WS_ORDER_FIRST_SEEN = 1 };
typedef struct ws_subsample_ws ws_subsample_ws;
int ws_subsample_ws_create(ws_subsample_ws** ws);
void ws_subsample_ws_destroy(ws_subsample_ws* ws);
int ws_grid_subsample_plan(ws_subsample_ws* ws, const float* points, int64_t n,
                           const int32_t* h_lens, int32_t nb, float dl, int32_t max_p,
                           int32_t order_mode, int32_t* h_out_lens, int64_t* h_m, void* stream);
int ws_grid_subsample_fill(ws_subsample_ws* ws,
                           const float* features, int32_t fd,   /* [n,fd] or NULL */
                           const int32_t* labels, int32_t ld,   /* [n,ld] or NULL */
                           float* out_points, float* out_features, int32_t* out_labels,
                           uint64_t* out_keys,                  /* [m] cell keys or NULL (test aid) */
                           int32_t* out_counts,                 /* [m] points per cell or NULL */
                           void* stream);

/* Rotation of stacked clouds by one 3x3 matrix per batch element, f32, no FMA:
 * out[i,j] = (p0*R[b,0,j] + p1*R[b,1,j]) + p2*R[b,2,j]   (transpose=1 uses R[b,j,:]),
 * the arithmetic of datasets/common.py:116-119 and :131-135 (random grid orientation). */
int ws_rotate_clouds(const float* points, int64_t n, const int32_t* lens /*device [nb]*/, int32_t nb,
                     const float* rot /*device [nb,3,3]*/, int32_t transpose, float* out, void* stream);

/* same with HOST lengths / matrices (h_lens [nb], h_rot [nb,3,3]), passed as a kernel argument: no
 * host-to-device copy and no synchronisation; nb <= 64 (WS_ERR_UNSUPPORTED beyond). */
int ws_rotate_clouds_host(const float* points, int64_t n, const int32_t* h_lens, int32_t nb,
                          const float* h_rot, int32_t transpose, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * The whole input pyramid of one batch in ONE call: the per-layer loop of PointCloudDataset.segmentation_inputs
 * (datasets/common.py:461-577) -- conv / pool / upsample searches and the grid subsampling with its random grid
 * orientation (:77-135) for every level, matrices cropped to the neighbourhood limits (:336-346).  Same kernels and
 * schedule as the per-call entries above (which it calls); the caller's interpreter stays released for the whole build.
 * All outputs live in the caller's `arena` at the returned byte offsets (-1: not produced): points of level l >= 1
 * (float [n[l],3]), neighbors (int64 [n[l], limit[l]]), pools (int64 [n[l+1], limit[l]]), upsamples (int64 [n[l],
 * limit[l+1]]), the cell order of every searched level (int32 [n[l]]), with want_grids the exported search grid and
 * key_last of the self-query searches (ws_radius_neighbors_grid_export / _set_key_last), the lengths (int32
 * [n_levels][nb]) and 4 n_levels int32 slots (maximum row lengths; then n_levels zeroed flags for the grid backward).
 * max_count [3 l + {0 conv, 1 pool, 2 upsample}] = true maximum row length of that search (host; -1: no such search):
 * 0 = the reference's empty-result error, < width = the caller may trim, > 128 (1024 for limits > 128) = the row slab
 * of the asynchronous search overflowed and the caller repeats that search with the two-call protocol.
 * Matrices whose widest row is shorter than the limit are cropped in place to that width (final_width; 0 = left to the
 * caller: empty / overflowed).  want_tables: the transposed tables (ws_transpose_build) of the pooling matrices, of the
 * first column of the upsampling matrices and -- without want_grids -- of the convolution matrices, at off_toffsets /
 * off_tpairs [3 l + kind] (-1: none).  scratch must also hold the largest matrix and the largest table's scratch.
 * WS_ERR_CAPACITY: the arena is too small; needed_bytes is set, nothing else is valid.  Synchronises the stream.
 * ------------------------------------------------------------------------------------------ */
#define WS_PYRAMID_MAX_LEVELS 8
#define WS_PYRAMID_MAX_BATCH 64
typedef struct ws_pyramid_desc {
    /* in */
    int32_t n_levels, nb, want_grids, want_tables;
    const float* points;                 /* level 0: [n0,3] device */
    int64_t n0;
    const float* h_rot;                  /* host [n_levels-1][nb][3][3] grid orientations (common.py:99-121) or NULL */
    void* arena; int64_t arena_bytes;    /* device, 256-byte aligned */
    void* scratch; int64_t scratch_bytes;/* device, >= 2 * align256(12 n0) */
    int32_t conv_on[WS_PYRAMID_MAX_LEVELS], pool_on[WS_PYRAMID_MAX_LEVELS];
    float r_conv[WS_PYRAMID_MAX_LEVELS], r_pool[WS_PYRAMID_MAX_LEVELS], r_up[WS_PYRAMID_MAX_LEVELS], dl[WS_PYRAMID_MAX_LEVELS];
    int32_t limit[WS_PYRAMID_MAX_LEVELS + 1];
    int32_t nearest_up;                  /* 1: upsampling matrices hold the nearest support only, [n, 1] (ws_radius_neighbors_nearest_async) */
    /* in (level 0) / out (levels >= 1) */
    int32_t lens[WS_PYRAMID_MAX_LEVELS][WS_PYRAMID_MAX_BATCH];
    /* out */
    int64_t needed_bytes;
    int64_t n[WS_PYRAMID_MAX_LEVELS];
    int64_t off_points[WS_PYRAMID_MAX_LEVELS], off_neighbors[WS_PYRAMID_MAX_LEVELS], off_pools[WS_PYRAMID_MAX_LEVELS],
            off_upsamples[WS_PYRAMID_MAX_LEVELS], off_order[WS_PYRAMID_MAX_LEVELS], off_key_last[WS_PYRAMID_MAX_LEVELS],
            off_blob[WS_PYRAMID_MAX_LEVELS], blob_bytes[WS_PYRAMID_MAX_LEVELS], grid_cells[WS_PYRAMID_MAX_LEVELS];
    int64_t off_lens, off_slots;
    int32_t max_count[3 * WS_PYRAMID_MAX_LEVELS], width[3 * WS_PYRAMID_MAX_LEVELS], final_width[3 * WS_PYRAMID_MAX_LEVELS];
    int32_t reserved2[3 * WS_PYRAMID_MAX_LEVELS];
    int64_t off_toffsets[3 * WS_PYRAMID_MAX_LEVELS], off_tpairs[3 * WS_PYRAMID_MAX_LEVELS];
} ws_pyramid_desc;
int ws_pyramid_build(ws_neighbors_ws* nws, ws_subsample_ws* sws, ws_pyramid_desc* desc, void* stream);
int64_t ws_pyramid_desc_bytes(void);   /* sizeof(ws_pyramid_desc): lets a binding check its mirror of the struct */



/* ------------------------------------------------------------------------------------------
 * Supervised contrastive loss of the pseudo-label trainer, the [N, slc_con] part of
 * KPFCNN.contrast_loss (models/architectures.py:455-497), fused: per point i
 *   loss[i] = -T * mean over the positive slice columns of log-softmax(<on_i, xs_j>/T over the usable columns)
 * with usable = (slc_idx[j] != i) and certain[slc_idx[j]] == certain[i], positive = usable and
 * lbl[slc_idx[j]] == lbl[i].  on [n,c] = L2-normalised logits (:475), xs [s,c] = on[slc_idx] (:476),
 * certain [n] uint8 (:433-435), lbl [n] int64 pseudo labels (:438-439).  rowmax / den / npos [n] are saved
 * for the backward, which returns d loss / d on (d_on [n,c], the slice rows' own contribution excluded) and
 * d loss / d xs (d_xs [s,c]; per-chunk partials in `scratch`, added in a fixed order).  c <= 16, s <= 1024.
 * ------------------------------------------------------------------------------------------ */
int ws_contrast_rows_fwd(const float* on, int64_t n, int32_t c, const float* xs, int32_t s, const int64_t* slc_idx,
                         const uint8_t* certain, const int64_t* lbl, float temperature, float eps, float* loss,
                         float* rowmax, float* den, float* npos, void* stream);
int64_t ws_contrast_rows_bwd_scratch_bytes(int64_t n, int32_t c, int32_t s);
int ws_contrast_rows_bwd(const float* on, int64_t n, int32_t c, const float* xs, int32_t s, const int64_t* slc_idx,
                         const uint8_t* certain, const int64_t* lbl, float temperature, const float* rowmax,
                         const float* den, const float* npos, const float* g, float* d_on, float* d_xs, void* scratch,
                         void* stream);

/* The rest of KPFCNN.contrast_loss around the [N, slc_con] part (models/architectures.py:425-454 before it, :475-476 the
 * normalisation, :498-504 after it), so that the whole loss is a few launches and never synchronises with the host.
 * head_fwd: x [n, c] logits (row stride ldx), labels [n] int64 (values >= 10 = unlabelled, :430-433), threshold
 *   (config.contrast_thd / 100).  Writes on [n,c] = F.normalize(x) (:475), inv_norm [n] = 1 / max(|x|, 1e-12), certain [n]
 *   uint8 = (max softmax > threshold) | labelled, lbl [n] int64 = the label where given else the arg-max class, then draws the
 *   slice: slot j of s takes the r_j-th valid point (0-based, in index order), r_j = r_given[j], or from the uniforms u [s] as
 *   floor(u_j * num_valid) -- with fewer than s valid points slot j < num_valid takes point j (:450-454) -- clamped to
 *   num_valid - 1; slc_idx [s] int64 and xs [s,c] = on[slc_idx] (:476).  state [2] int32: number of valid points, and the
 *   tail's arrival counter (zeroed here).  No valid point: slc_idx = n - 1 everywhere and the tail returns 0 (:441-443).
 * tail_fwd: pts_loss [n] from ws_contrast_rows_fwd -> per_class [n_cls] (mean of the points with loss > 0 per label),
 *   loss [1] = mean of the per_class entries > 0 (:498-504), w_cls [n_cls] = d loss / d pts_loss of a kept point of the class.
 * tail_bwd: g [1] -> g_row [n] (the `g` operand of ws_contrast_rows_bwd).
 * head_bwd: d_on [n,c] (modified in place: the slice rows' gradients d_xs [s,c] are added onto their points, duplicates in
 *   slot order) -> d_x [n, c] (row stride ldd) through the backward of the normalisation.   c <= 16, s <= 2048, n_cls <= 16. */
int64_t ws_contrast_head_scratch_bytes(int64_t n);
int ws_contrast_head_fwd(const float* x, int64_t n, int32_t c, int64_t ldx, const int64_t* labels, float threshold,
                         const float* u, const int64_t* r_given, int32_t s, float* on, float* inv_norm, uint8_t* certain,
                         int64_t* lbl, int64_t* slc_idx, float* xs, int32_t* state, void* scratch, void* stream);
int64_t ws_contrast_tail_scratch_bytes(int64_t n);
int ws_contrast_tail_fwd(const float* pts_loss, const int64_t* lbl, int64_t n, int32_t n_cls, int32_t* state, float* per_class,
                         float* w_cls, float* loss, void* scratch, void* stream);
int ws_contrast_tail_bwd(const float* pts_loss, const int64_t* lbl, int64_t n, int32_t n_cls, const float* w_cls, const float* g,
                         float* g_row, void* stream);
int ws_contrast_head_bwd(float* d_on, const float* d_xs, const int64_t* slc_idx, int32_t s, const float* on,
                         const float* inv_norm, int64_t n, int32_t c, float* d_x, int64_t ldd, void* stream);


/* ------------------------------------------------------------------------------------------
 * bf16-feature path (BASELINE.json configs[4]: "deformable-KPConv ... bf16"; SURVEY.md section 8d C5:
 * feature rows / weights bf16 in HBM, fp32 accumulate, geometry fp32).  The reference has no reduced-precision
 * path; these entries are the bf16-row forms of the operators above and replace the same reference lines
 * (models/blocks.py:36-134 pools, :238-374 KPConv, :370-374 / :490-501 the dense products).  Feature rows are
 * bf16 (uint16_t bit patterns, round-to-nearest-even on store), everything geometric (points, kernel points,
 * deformed_kp, modulations, min_d2 and their gradients) stays f32; sums run in f32 and are rounded once.
 * Rows need c % 4 == 0 and 8-byte alignment (WS_ERR_INVALID otherwise -- the 3-channel input layer stays f32).
 * ------------------------------------------------------------------------------------------ */
int ws_kpconv_gather_fwd_bf16(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                              const uint16_t* x, int32_t ci, const float* kernel_points, int32_t k, const float* deformed_kp,
                              const float* modulations, float extent, int32_t influence, int32_t aggregation,
                              const int32_t* order, uint16_t* wf, float* min_d2, void* stream);
int ws_kpconv_gather_bwd_x_bf16(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                                const int32_t* t_offsets, const int32_t* t_pairs, const uint16_t* dwf, int32_t ci,
                                const float* kernel_points, int32_t k, const float* deformed_kp, const float* modulations,
                                float extent, int32_t influence, int32_t aggregation, const int32_t* order, uint16_t* dx,
                                void* stream);
int ws_kpconv_gather_bwd_x_grid_bf16(const float* s_pts, int64_t ns, const void* grid_blob, int32_t nb, int64_t cells,
                                     const uint64_t* key_last, float radius, const uint16_t* dwf, int32_t ci,
                                     const float* kernel_points, int32_t k, const float* deformed_kp, const float* modulations,
                                     float extent, int32_t influence, int32_t aggregation, const int32_t* order, uint16_t* dx,
                                     int32_t* overflow, void* stream);
int ws_kpconv_gather_bwd_geom_bf16(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                                   const uint16_t* x, int32_t ci, const uint16_t* dwf, const float* kernel_points, int32_t k,
                                   const float* deformed_kp, const float* modulations, const float* d_min_d2, float extent,
                                   int32_t influence, int32_t aggregation, float* d_deformed_kp, float* d_modulations,
                                   void* stream);
int ws_max_pool_fwd_bf16(const uint16_t* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                         uint16_t* out, int32_t* arg, void* stream);
int ws_max_pool_bwd_bf16(const uint16_t* dy, const int32_t* arg, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                         const int32_t* t_pairs, int64_t ns, uint16_t* dx, void* stream);
int ws_closest_pool_fwd_bf16(const uint16_t* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                             uint16_t* out, void* stream);
int ws_closest_pool_bwd_bf16(const uint16_t* dy, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                             const int32_t* t_pairs, int64_t ns, uint16_t* dx, void* stream);

/* Y[m,n] = act( X[m,k] * Bt[n,k]^T + bias + residual ) on v_mfma_f32_32x32x16_bf16: X, Bt, residual bf16 rows
 * (K-contiguous: Bt is nn.Linear's own [out,in] layout, blocks.py:490), bias f32 [n], fp32 accumulate, one rounding
 * to bf16 on store -- or f32 output (out_f32 = 1: the logits, the offsets of deformable KPConv blocks.py:244-267).
 * k % 32 == 0, ldx % 8 == 0, ldbt % 8 == 0, 16-byte aligned x / bt (WS_ERR_INVALID otherwise; callers keep such
 * products on the f32 kernels).  act: 0 = identity, 1 = LeakyReLU(slope). */
int ws_gemm_xbt_bf16(const uint16_t* x, int64_t m, int32_t k, int64_t ldx, const uint16_t* bt, int32_t n, int64_t ldbt,
                     const float* bias, const uint16_t* residual, int64_t ldr, int32_t act, float slope,
                     void* y, int64_t ldy, int32_t out_f32, void* stream);
/* out[k,n] f32 = X[m,k]^T * Y[m,n] with bf16 rows (the fp32 master gradient dW; exact products, fp32 sums on the
 * f32 MFMA); scratch as ws_gemm_xty_scratch_bytes(m, k, n). */
int ws_gemm_xty_bf16(const uint16_t* x, int64_t m, int32_t k, int64_t ldx, const uint16_t* y, int32_t n, int64_t ldy,
                     float* out, void* scratch, void* stream);
/* dz = dy * LeakyReLU'(y) as bf16 rows (y = the layer's bf16 output, NULL = identity: then dz may be NULL) and the f32
 * column sums of dz (colsum NULL = not wanted); dy is bf16, or f32 when dy_f32 = 1.  n % 4 == 0. */
int64_t ws_act_bwd_colsum_bf16_scratch_bytes(int64_t m, int32_t n);
int ws_act_bwd_colsum_bf16(const void* dy, int32_t dy_f32, int64_t m, int32_t n, int64_t lddy, const uint16_t* y, int64_t ldy,
                           float slope, uint16_t* dz, int64_t lddz, float* colsum, void* scratch, void* stream);


/* ------------------------------------------------------------------------------------------
 * Whole network blocks behind ONE call each (rigid KPConv, linear influence, sum aggregation, f32 rows).
 *
 * The reference runs a block as ~10 separate torch ops plus their autograd nodes (models/blocks.py:510-564
 * SimpleBlock, :624-709 ResnetBottleneckBlock; the decoder step models/architectures.py:339-343: nearest_upsample ->
 * concat(skip) -> unary); driven from Python each launch costs ~15 us of host time, more than most of these kernels
 * take.  ws_kpblock_fwd / _bwd launch the whole sequence from C on one stream:
 *
 *   x1   = lrelu(feat @ w1^T + b1)                          unary1   (absent when w1 == NULL: x1 = feat)
 *   wf   = gather(x1)                                       K3, ws_kpconv_gather_fwd
 *   x2   = lrelu(wf @ wk + bk)                              the kernel contraction + BatchNormBlock bias + LeakyReLU
 *   out  = lrelu(x2 @ w2^T + b2 + shortcut)                 unary2 + residual sum (absent when w2 == NULL: out = x2)
 *   shortcut = max_pool(feat, inds) if strided else feat,   then @ ws^T + bs when ws != NULL
 *
 * and in the backward every gradient accumulation is the residual operand of a GEMM epilogue (no separate adds),
 * LeakyReLU' and the bias gradients share one pass (ws_act_bwd_colsum), dX of the KPConv goes through the search grid
 * of the level (ws_kpconv_gather_bwd_x_grid) when grid_blob != NULL, else through the transposed table.
 * Linear weights are nn.Linear's [out,in] row-major; wk is KPConv.weights [k, conv_in, conv_out] (blocks.py:171).
 * Biases may be NULL (BatchNormBlock with use_bn is an identity, blocks.py:453-463).
 * Requirements (WS_ERR_UNSUPPORTED otherwise; the caller then runs the operators one by one): in_dim, conv_out,
 * out_dim multiples of 4 and conv_in a multiple of 4 (or the 3..4-channel input layer without unary1), k = 15.
 * All buffers are caller-owned; `scratch` must hold ws_kpblock_*_scratch_bytes(desc) bytes.
 * ------------------------------------------------------------------------------------------ */
typedef struct ws_kpblock {
    /* geometry of the convolution */
    const float* q_pts; int64_t nq;
    const float* s_pts; int64_t ns;
    const int64_t* inds; int32_t h;
    const float* kernel_points; int32_t k; float extent;
    const int32_t* order_q;              /* scheduling hints (NULL allowed) */
    const int32_t* order_s;
    /* dX route: search grid of a self-query level ... */
    const void* grid_blob; int32_t grid_nb; int64_t grid_cells; const uint64_t* key_last; float grid_radius;
    int32_t* grid_overflow;
    /* ... or the transposed table of inds (also used by the max-pool shortcut of strided blocks) */
    const int32_t* t_offsets; const int32_t* t_pairs;
    /* widths */
    int32_t in_dim, conv_in, conv_out, out_dim, strided;
    float slope;
    /* parameters */
    const float *w1, *b1, *wk, *bk, *w2, *b2, *ws, *bs;
    /* activations: feat [ns,in_dim] in; x1 [ns,conv_in] (w1 != NULL), wf [nq,k*conv_in], x2 [nq,conv_out] (w2 != NULL),
     * pooled [nq,in_dim] + arg (strided with w2), out [nq,out_dim]: written by fwd, read by bwd */
    const float* feat; float* x1; float* wf; float* x2; float* pooled; int32_t* arg; float* out;
    /* backward: dout [nq,out_dim] in; gradients out (dfeat NULL = not wanted; parameter gradients NULL where the
     * parameter is NULL; the shortcut bias gradient equals db2) */
    const float* dout; float* dfeat; float *dw1, *db1, *dwk, *dbk, *dw2, *db2, *dws;
    int32_t timed;                       /* 1: bracket the K3 launch with HIP events (ws_timer_*) */
    int32_t rows_sorted;                 /* 1: the rows of inds are sorted by distance from their query (radius search output):
                                            the gather stops at the reach of the kernel points (ws_kpconv_gather_fwd_ex) */
    int32_t infer;                       /* 1: forward only -- no backward will follow: `wf` may be NULL and layers the fused
                                            forward kernel covers (ws_kpconv_layer_fwd_fused) run as one launch */
    int32_t dout_pregated;               /* backward: 1 = `dout` already carries this block's output LeakyReLU' (the consumer of
                                            `out` multiplied it into the gradient it wrote: its gate_dfeat): the block's first
                                            activation-backward pass is skipped (bias gradients: column sums of dout) */
    int32_t gate_dfeat;                  /* backward: 1 = write dfeat * LeakyReLU'(feat) -- feat is the activated output of the
                                            block that produced it, which then runs with dout_pregated; the multiplication
                                            rides on the kernel that writes dfeat */
    const float* dfeat_add;              /* backward, optional [ns,in_dim]: a second gradient of `feat` (the decoder's skip
                                            connection reads the same tensor, architectures.py:339-341), summed into dfeat by
                                            the kernel that writes the shortcut's share instead of by a pass of its own */
} ws_kpblock;

int64_t ws_kpblock_fwd_scratch_bytes(const ws_kpblock* d);
int64_t ws_kpblock_bwd_scratch_bytes(const ws_kpblock* d);
int ws_kpblock_fwd(const ws_kpblock* d, void* scratch, int64_t scratch_bytes, void* stream);
int ws_kpblock_bwd(const ws_kpblock* d, void* scratch, int64_t scratch_bytes, void* stream);

/* Decoder step (models/architectures.py:339-343 + blocks.py:473-507): with the unary's weight w = [wx | wsk]
 * ([out, c_up + c_skip], leading dimension ldw), evaluated as
 *   yc  = xc @ wx^T                        at the coarse resolution [nc,out]
 *   out = lrelu(skip @ wsk^T + b + yc[ups[:,0]])          (closest_pool = nearest upsampling, blocks.py:80-92)
 * identical in exact arithmetic to upsample -> concat -> unary (a row gather commutes with a per-row linear map).
 * Backward: dw [out, c_up + c_skip] (same layout as w), db, dxc [nc,c_up], dskip [nf,c_skip]; t_offsets / t_pairs =
 * transposed table of column 0 of `ups`. */
typedef struct ws_upunary {
    const float* xc; int64_t nc; int32_t c_up;
    const float* skip; int64_t nf; int32_t c_skip;
    const int64_t* ups; int32_t h_up;
    const int32_t* t_offsets; const int32_t* t_pairs;
    const float* w; int64_t ldw; const float* b; int32_t out_dim; int32_t relu; float slope;
    float* yc; float* out;               /* fwd outputs (yc is scratch-like but caller-owned: not needed by bwd) */
    const float* dout; float* dxc; float* dskip; float* dw; float* db;
    /* nn.Dropout on `out` (models/architectures.py:345-346: the droplayer in front of the head), fused: drop_p > 0 makes the
     * forward's last epilogue write dropout(out) (keep decision = function of (drop_seed, element index), the bits of
     * ws_dropout_apply) and the backward treat `dout` as the gradient of that dropped tensor.  Needs relu != 0. */
    float drop_p; uint64_t drop_seed;
    int32_t dout_pregated;               /* as in ws_kpblock: dout already multiplied by LeakyReLU'(out) by the consumer (and run
                                            through the dropout backward when drop_p > 0: ws_gemm_xb_gate_dropout) */
    int32_t gate_dxc;                    /* backward: write dxc * LeakyReLU'(xc) (xc = the activated output of its producer) */
} ws_upunary;

int64_t ws_upunary_fwd_scratch_bytes(const ws_upunary* d);
int64_t ws_upunary_bwd_scratch_bytes(const ws_upunary* d);
int ws_upunary_fwd(const ws_upunary* d, void* scratch, int64_t scratch_bytes, void* stream);
int ws_upunary_bwd(const ws_upunary* d, void* scratch, int64_t scratch_bytes, void* stream);

/* y = act(x @ B + bias + residual) with a strided small matrix: B[kk][col] = b[kk*b_row_stride + col*b_col_stride]
 * (b_row_stride = 1, b_col_stride = K reads nn.Linear's [N,K] weight as its transpose in place).  Strided forms need
 * k % 32 == 0, n % 4 == 0 and 16-byte aligned rows. */
int ws_gemm_xb_epilogue_strided(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int64_t b_row_stride,
                                int64_t b_col_stride, int32_t n, const float* bias, const float* residual, int64_t ldr,
                                int32_t act, float slope, float* y, int64_t ldy, void* scratch, int64_t scratch_bytes,
                                void* stream);

/* The same product with nn.Dropout applied to the activated output in the epilogue: y = keep ? act(..) / (1 - p) : 0, the keep
 * decision of element (row, col) being that of ws_dropout_apply for index row * n + col and the same seed (bit-identical to
 * the product followed by ws_dropout_apply on a contiguous [m, n] tensor; no mask is stored). */
int ws_gemm_xb_dropout_strided(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int64_t b_row_stride,
                               int64_t b_col_stride, int32_t n, const float* bias, const float* residual, int64_t ldr,
                               int32_t act, float slope, float drop_p, uint64_t drop_seed, float* y, int64_t ldy, void* scratch,
                               int64_t scratch_bytes, void* stream);

/* dX = dZ @ B (B row-major [K, N]) as the gradient of a layer INPUT that came out of a LeakyReLU and, optionally, an nn.Dropout
 * after it: y = dropout_bwd(x @ b) * LeakyReLU'(gate_y), in that order (the order of the separate backward passes: same bits);
 * gate_y [m, n] = the activated (and dropped: same signs where kept) tensor, NULL = no gate; drop_p = 0 = no dropout.  The
 * producer of gate_y then needs no activation-backward pass of its own (ws_kpblock / ws_upunary: dout_pregated). */
int ws_gemm_xb_gate_dropout(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n, const float* gate_y,
                            int64_t ldg, float gate_slope, float drop_p, uint64_t drop_seed, float* y, int64_t ldy, void* scratch,
                            int64_t scratch_bytes, void* stream);

/* ws_act_bwd_colsum for a tensor that went through that dropout: dy is the gradient of dropout(y') where y' is the activated
 * output and y = dropout(y') is what was kept (same sign wherever the mask keeps): dz = dropout_bwd(dy) * act'(y), column sums. */
int ws_act_bwd_colsum_dropout(const float* dy, int64_t m, int32_t n, int64_t lddy, const float* y, int64_t ldy, float slope,
                              float drop_p, uint64_t drop_seed, float* dz, int64_t lddz, float* colsum, void* scratch, void* stream);

/* The same product with multiplicative gates on the epilogue (after bias / residual / activation):
 *   gate_y [m, n] (pitch ldg, NULL = none): y *= LeakyReLU'(gate_y) = (gate_y > 0 ? 1 : gate_slope) -- when this product is the
 *     gradient dX = dY W^T of a layer whose input came out of a LeakyReLU, gate_y is that activated tensor and the
 *     activation backward (models/blocks.py:473-507, autograd of nn.LeakyReLU) costs no pass of its own;
 *   mask [m, n] bytes (pitch ldm, NULL = none): y = mask ? y * mask_scale : 0 -- nn.Dropout (models/architectures.py:354-356),
 *     forward on the activated output and backward on the gradient. */
int ws_gemm_xb_gated_strided(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int64_t b_row_stride,
                             int64_t b_col_stride, int32_t n, const float* bias, const float* residual, int64_t ldr,
                             int32_t act, float slope, const float* gate_y, int64_t ldg, float gate_slope, const uint8_t* mask,
                             int64_t ldm, float mask_scale, float* y, int64_t ldy, void* scratch, int64_t scratch_bytes,
                             void* stream);

/* HIP-event timing of selected kernel launches inside the block calls (bench.py's roofline figure): events are
 * recorded on the launch stream around the K3 launch of blocks with timed = 1.  ws_timer_read synchronises on the
 * events of record i and returns its key (nq, h, ci) and the elapsed milliseconds. */
int ws_timer_reset(void);
int ws_timer_count(void);
int ws_timer_read(int32_t i, int64_t* nq, int32_t* h, int32_t* ci, float* ms);
/* the same record, from the start of the K3 launch to the end of the contraction that follows it: the whole KPConv layer
 * (models/blocks.py:278-374), the unit SURVEY section 8d's B_fwd describes */
int ws_timer_read_layer(int32_t i, float* ms);


/* ------------------------------------------------------------------------------------------
 * Forward-only callers of the hot path: the voting test loop and the sampler potentials.
 * ws_vote_update: utils/tester_PseudoLabel.py:168-194 -- for every point i of a sphere (n rows of `logits` [n,c]):
 *   probs[inds[i], :] = smooth * probs[inds[i], :] + (1 - smooth) * softmax(logits[i, :]); with radius_mask > 0 only the
 *   points with |points[i]|^2 < radius_mask^2 take part (test_radius_ratio * in_radius).  inds are unique inside a sphere.
 * ws_project_confusion: tester_PseudoLabel.py:283-307 + utils/metrics.py:35-110 -- preds[i] = argmax_k probs[proj[i], k]
 *   (first maximum; proj NULL = identity), confusion[t, p] += 1 for t = labels[i] in [0, nc) (int64 [nc, nc], must be
 *   zeroed by the caller; accumulated over calls).
 * ws_potentials_update: datasets/DALES_PseudoLabel.py:335-350 -- potentials[i] += (1 - d2/r^2)^2 for the coarse points
 *   within `radius` of h_center (host double[3]), float64 arithmetic of the KDTree-based reference; then the new minimum
 *   (value, first index) into out_min / out_argmin (device).  scratch: ws_potentials_scratch_bytes(n).
 * The nearest-neighbour projection indices themselves (DALES_PseudoLabel.py:888-892) are column 0 of the K1 radius search.
 * ------------------------------------------------------------------------------------------ */
int ws_vote_update(const float* logits, int64_t n, int32_t c, const float* points, float radius_mask, const int64_t* inds,
                   float* probs, int64_t n_cloud, float smooth, void* stream);
int ws_project_confusion(const float* probs, int32_t c, const int32_t* proj, int64_t m, const int32_t* labels, int32_t* preds,
                         int32_t nc, int64_t* confusion, void* stream);
int64_t ws_potentials_scratch_bytes(int64_t n);
int ws_potentials_update(const float* pot_points, int64_t n, const double* h_center, double radius, double* potentials,
                         double* out_min, int64_t* out_argmin, void* scratch, void* stream);

/* nn.Dropout on the decoder output in front of the head, training mode (models/architectures.py:345-346): out[i] = keep(i) ? in[i] / (1 - p) : 0,
 * keep(i) a pure function of (seed, i) (counter-based; Bernoulli(1 - p)): the backward is the same call on the incoming
 * gradient with the same seed, no mask is stored.  in / out float32 [n], 16-byte aligned; in == out allowed. */
int ws_dropout_apply(const float* in, int64_t n, float p, uint64_t seed, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * The two ends of the training step around the network (forward + backward + SGD).
 * ws_softmax_ce_fwd / _bwd: models/architectures.py:362-373 (KPFCNN.loss) -- the label mapping (:362-365; lut [lut_n]
 *   maps raw label values to class positions, its last entry is the spare -1 every out-of-table label takes; lut NULL =
 *   labels are positions, < 0 ignored) and torch.nn.CrossEntropyLoss(weight = class_w or NULL, ignore_index = -1) over the
 *   n rows of logits [n, c] (row pitch ldl), c <= 64: loss[0] = weighted mean over the valid rows, wsum[0] = sum of their
 *   weights (both device floats; no valid row -> nan like the stock loss).  The backward takes d loss as a device
 *   float (grad_loss[0]) so that nothing synchronises; rows with ignored labels get zeros.
 * ws_sgd_step: utils/trainer_PseudoLabel.py:216-218 with the optimizer of :72-82 -- clip_grad_value_(clip_value; <= 0 =
 *   off), weight decay, momentum buffer (first != 0: buf = g, the first step of torch.optim.SGD) and p -= lr * buf for
 *   `count` tensors in one pass; h_params / h_grads / h_bufs / h_sizes are HOST arrays of device pointers / element counts
 *   (one parameter group per call: lr, momentum and weight_decay are per group).  No dampening, no Nesterov.
 * ------------------------------------------------------------------------------------------ */
int64_t ws_softmax_ce_scratch_bytes(int64_t n);
int ws_softmax_ce_fwd(const float* logits, int64_t n, int32_t c, int64_t ldl, const int64_t* labels, const int64_t* lut,
                      int32_t lut_n, const float* class_w, float* loss, float* wsum, void* scratch, void* stream);
int ws_softmax_ce_bwd(const float* logits, int64_t n, int32_t c, int64_t ldl, const int64_t* labels, const int64_t* lut,
                      int32_t lut_n, const float* class_w, const float* grad_loss, const float* wsum, float* dlogits, int64_t ldd,
                      void* stream);
int ws_sgd_step(float* const* h_params, const float* const* h_grads, float* const* h_bufs, const int64_t* h_sizes, int32_t count,
                float lr, float momentum, float weight_decay, float clip_value, int32_t first, void* stream);

/* ------------------------------------------------------------------------------------------
 * Deformable KPConv, the fast path of BASELINE config 5 (models/blocks.py:244-325, 366-367 with KP_influence = 'linear',
 * aggregation_mode = 'sum'; neighbour rows of several hundred columns: datasets/common.py:500-502).
 *
 * ws_kpconv_deform_prepare: blocks.py:250-267, 287-288 in one pass.  offset_features [n, od] f32 (od = 3k, or 4k when
 *   modulated) -> deformed_kp [n,k,3] = offset * extent + kernel_points (two roundings, like the reference's mul and
 *   add), modulations [n,k] = 2 sigmoid(last k columns) (NULL / ignored when not modulated), and kp4 [n,k] float4 =
 *   (x, y, z, modulation or 1): the packed operand of the entries below (16-byte aligned).  deformed_kp may be NULL.
 *   kp_rmax (device float, may be NULL) receives max |deformed kernel point| of the call: ws_kpconv_gather_bwd_x_grid_wide
 *   drops the pairs farther apart than kp_rmax + extent (no kernel point reaches them).
 * rows_sorted != 0 (ws_kpconv_gather_fwd_def, _bwd_geom_def, ws_kpconv_gather_fwd_ex): the caller vouches that every index
 *   row is sorted by distance from its query (what the radius search delivers).  A neighbour beyond the reach of every
 *   kernel point (max |kp| + extent) has 15 zero influences and one beyond max_k (sqrt(min_d2[k]) + |kp_k|) cannot lower
 *   a minimum: the walk over the row stops there.  Exact (the skipped terms are zeros); with the deformable search radius
 *   (2 r against a reach of ~1.1 r) it skips most of every row.
 * ws_kpconv_deform_prepare_bwd: d offset_features [n, od] from d_kp4 [n,k,4] (NULL = zero) and / or a second gradient
 *   d_deformed_kp [n,k,3] of the positions (NULL = none; the regulariser's).
 * ws_kpconv_gather_fwd_def / _bwd_x_def / _bwd_geom_def: ws_kpconv_gather_fwd / _bwd_x / _bwd_geom for that mode with
 *   kp4 in place of (kernel_points, deformed_kp, modulations).  The in-range filter of blocks.py:301-325 is implied: with
 *   the linear influence a neighbour without a kernel point inside the extent has 15 zero influences.  _bwd_geom_def
 *   (the dense product dwf[q] . x[neighbours]^T on the matrix core; ci % 16 == 0) writes d_kp4 [nq,k,4] =
 *   (d x, d y, d z, d modulation), including the min_d2 path when d_min_d2 != NULL.  rows_bf16 != 0: the feature rows
 *   x / wf / dwf / dx are bf16.
 * ws_kpconv_gather_bwd_x_grid_wide: ws_kpconv_gather_bwd_x_grid for ANY in-degree (rows wider than 128 neighbours): the
 *   candidate slab is a queue that is consumed 64 pairs at a time.  Rigid (kp4 NULL, kernel_points given) or deformable
 *   (kp4); linear influence, sum aggregation.  rows / rows_h as in ws_kpconv_gather_bwd_x_grid_gated (NULL = walk the grid).
 * ws_p2p_regularizer_fwd / _bwd: models/architectures.py:24-57 for ONE layer.  out2[0] = mean |min_d2 / extent^2|,
 *   out2[1] = sum_i mean_n |sum_{j != i} min(|kp_i - kp_j| / extent - repulse_extent, 0)^2| / k (the other points
 *   detached); positions from deformed_kp [n,k,3] or, when that is NULL, from kp4.  The backward takes g2 = (d loss /
 *   d out2[0], d loss / d out2[1]) as DEVICE floats and writes d_min_d2 [n,k], d_deformed_kp [n,k,3].
 *   scratch: ws_p2p_regularizer_scratch_bytes(n).
 * ------------------------------------------------------------------------------------------ */
int ws_kpconv_deform_prepare(const float* offset_features, int64_t n, int32_t od, const float* kernel_points, int32_t k,
                             float extent, int32_t modulated, float* deformed_kp, float* modulations, float* kp4, float* kp_rmax,
                             void* stream);
int ws_kpconv_deform_prepare_bwd(const float* d_kp4, const float* d_deformed_kp, const float* kp4, int64_t n, int32_t od, int32_t k,
                                 float extent, int32_t modulated, float* d_offset_features, void* stream);
int ws_kpconv_gather_fwd_def(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                             const void* x, int32_t ci, const float* kp4, int32_t k, float extent, const int32_t* order,
                             void* wf, float* min_d2, int32_t rows_bf16, int32_t rows_sorted, void* stream);
int ws_kpconv_gather_fwd_ex(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                            const void* x, int32_t ci, const float* kernel_points, int32_t k, const float* deformed_kp,
                            const float* modulations, float extent, int32_t influence, int32_t aggregation, const int32_t* order,
                            void* wf, float* min_d2, int32_t rows_bf16, int32_t rows_sorted, void* stream);
int ws_kpconv_gather_bwd_x_def(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, int32_t h,
                               const int32_t* t_offsets, const int32_t* t_pairs, const void* dwf, int32_t ci, const float* kp4,
                               int32_t k, float extent, const int32_t* order, void* dx, int32_t rows_bf16, void* stream);
int ws_kpconv_gather_bwd_x_grid_wide(const float* s_pts, int64_t ns, const void* grid_blob, int32_t nb, int64_t cells,
                                     const uint64_t* key_last, float radius, const void* dwf, int32_t ci,
                                     const float* kernel_points, int32_t k, const float* kp4, const float* kp_rmax, float extent,
                                     const int32_t* order, const int64_t* rows, int32_t rows_h, void* dx, int32_t rows_bf16,
                                     void* stream);
int ws_kpconv_gather_bwd_geom_def(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                                  const void* x, int32_t ci, const void* dwf, const float* kp4, int32_t k, const float* d_min_d2,
                                  float extent, const int32_t* order, float* d_kp4, int32_t rows_bf16, int32_t rows_sorted,
                                  void* stream);
/* A whole forward KPConv layer in one launch, out [nq, co] = act(KPConv(x) + bias) with the contraction inside the gather
 * kernel (models/blocks.py:278-374 + the BatchNormBlock bias / LeakyReLU of :556-563); rigid, linear influence, sum, f32,
 * ci = co = 32 (else WS_ERR_UNSUPPORTED).  Forward only: nothing is kept for a backward pass (the weighted features
 * never reach memory) -- the testers' forward passes, utils/tester_PseudoLabel.py:164. */
int ws_kpconv_layer_fwd_fused(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                              const float* x, int32_t ci, const float* kernel_points, int32_t k, float extent,
                              const int32_t* order, const float* weights, int32_t co, const float* bias, int32_t act, float slope,
                              float* out, void* stream);
/* name of the forward gather kernel the library launches for a layer of ci channels (mode 0 rigid, 1 deformable through the
 * generic entries, 2 deformable fast path): for reports, no device work */
int ws_kpconv_gather_fwd_variant(int32_t ci, int32_t mode, int32_t influence, int32_t aggregation, int32_t rows_bf16,
                                 int32_t rows_sorted, char* out, int32_t cap);
int64_t ws_p2p_regularizer_scratch_bytes(int64_t n);
int ws_p2p_regularizer_fwd(const float* deformed_kp, const float* kp4, const float* min_d2, int64_t n, int32_t k, float extent,
                           float repulse_extent, float* out2, void* scratch, void* stream);
int ws_p2p_regularizer_bwd(const float* deformed_kp, const float* kp4, const float* min_d2, int64_t n, int32_t k, float extent,
                           float repulse_extent, const float* g2, float* d_min_d2, float* d_deformed_kp, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WEASAL_HIP_H */
