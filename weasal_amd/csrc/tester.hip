// weasal_amd/csrc/tester.hip -- forward-only callers of the hot path: the voting test loop and the potentials of the
// sphere sampler (second half).
//
//   ws_vote_update        utils/tester_PseudoLabel.py:168-194 : softmax of the logits, optional radius mask, exponential
//                         smoothing of the per-cloud class probabilities at the sphere's input indices
//   ws_project_confusion  utils/tester_PseudoLabel.py:283-307 + utils/metrics.py:35-110 : probabilities of the sub-cloud
//                         re-projected onto the full cloud (nearest sub-cloud point, indices from the K1 search), arg-max,
//                         confusion matrix against the labels
//   ws_potentials_update  datasets/DALES_PseudoLabel.py:335-350 : Tukey update of the sampling potentials around a sphere
//                         centre and the new minimum (float64, the arithmetic of the KDTree-based reference:
//                         dist = sqrt(sum of squares), d2 = dist^2, tukey = (1 - d2/r^2)^2 for dist <= r)
// The nearest-neighbour projection itself (DALES_PseudoLabel.py:888-892, KDTree.query(k = 1)) is the K1 radius search with
// one column (rows are sorted by distance): weasal_amd/tester.py.
#include "ws_common.h"

namespace {

__global__ __launch_bounds__(256) void vote_update_kernel(const float* __restrict__ logits, int64_t n, int c,
                                                          const float* __restrict__ points, float r2,
                                                          const int64_t* __restrict__ inds, float* __restrict__ probs,
                                                          int64_t n_cloud, float smooth)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (r2 > 0.0f) {        // test_radius_ratio mask: sum(points^2) < (ratio * in_radius)^2, f32 like numpy on f32 input
            const float x = points[3 * i], y = points[3 * i + 1], z = points[3 * i + 2];
            if (!(((x * x + y * y) + z * z) < r2)) continue;
        }
        const int64_t dst = inds[i];
        if (dst < 0 || dst >= n_cloud) continue;
        const float* l = logits + i * c;
        float m = l[0];
        for (int k = 1; k < c; ++k) m = fmaxf(m, l[k]);
        float s = 0.0f;
        for (int k = 0; k < c; ++k) s += expf(l[k] - m);
        float* p = probs + dst * c;
        for (int k = 0; k < c; ++k) p[k] = smooth * p[k] + (1.0f - smooth) * (expf(l[k] - m) / s);
    }
}

// confusion[t * nc + p] += 1 with p = arg-max class of probs[proj[i]] (first maximum, like np.argmax), t = labels[i];
// classes are positions in the sorted label_values; labels outside [0, nc) are skipped.  Per-workgroup LDS histogram,
// integer atomics (exact).
__global__ __launch_bounds__(256) void project_confusion_kernel(const float* __restrict__ probs, int c,
                                                                const int32_t* __restrict__ proj, int64_t m,
                                                                const int32_t* __restrict__ labels, int32_t* __restrict__ preds,
                                                                int nc, long long* __restrict__ confusion)
{
    extern __shared__ int hist[];
    for (int e = threadIdx.x; e < nc * nc; e += 256) hist[e] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
        const float* p = probs + (int64_t)(proj ? proj[i] : i) * c;
        int best = 0;
        float bv = p[0];
        for (int k = 1; k < c; ++k)
            if (p[k] > bv) { bv = p[k]; best = k; }
        if (preds) preds[i] = best;
        if (labels) {
            const int t = labels[i];
            if (t >= 0 && t < nc && best < nc) atomicAdd(&hist[t * nc + best], 1);
        }
    }
    __syncthreads();
    if (confusion)
        for (int e = threadIdx.x; e < nc * nc; e += 256)
            if (hist[e]) atomicAdd((unsigned long long*)&confusion[e], (unsigned long long)hist[e]);
}

__global__ __launch_bounds__(256) void potentials_update_kernel(const float* __restrict__ pts, int64_t n, double cx, double cy,
                                                                double cz, double radius, double* __restrict__ pot)
{
    const double r2 = radius * radius;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double dx = (double)pts[3 * i] - cx, dy = (double)pts[3 * i + 1] - cy, dz = (double)pts[3 * i + 2] - cz;
        const double rd = dx * dx + dy * dy + dz * dz;       // the tree's reduced distance
        if (rd <= r2) {                                      // query_radius is inclusive
            const double dist = sqrt(rd);
            const double d2 = dist * dist;                   // np.square(dists)
            const double t = 1.0 - d2 / r2;
            pot[i] += (d2 > r2) ? 0.0 : t * t;
        }
    }
}

// (min value, first index of it) of a float64 array: per-workgroup partials, then one workgroup
__global__ __launch_bounds__(256) void argmin_partial_kernel(const double* __restrict__ v, int64_t n, double* __restrict__ pv,
                                                             long long* __restrict__ pi)
{
    __shared__ double sv[256];
    __shared__ long long si[256];
    double best = 1.0e308;
    long long bi = -1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = v[i];
        if (x < best || (x == best && (bi < 0 || i < bi))) { best = x; bi = i; }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double ov = sv[threadIdx.x + o];
            const long long oi = si[threadIdx.x + o];
            if (oi >= 0 && (ov < sv[threadIdx.x] || (ov == sv[threadIdx.x] && (si[threadIdx.x] < 0 || oi < si[threadIdx.x])))) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { pv[blockIdx.x] = sv[0]; pi[blockIdx.x] = si[0]; }
}

// final step over the partial (value, index) pairs: smallest value, smallest original index among equal values
__global__ __launch_bounds__(256) void argmin_final_kernel(const double* __restrict__ pv, const long long* __restrict__ pi, int nparts,
                                                           double* __restrict__ out_min, long long* __restrict__ out_argmin)
{
    __shared__ double sv[256];
    __shared__ long long si[256];
    double best = 1.0e308;
    long long bi = -1;
    for (int e = threadIdx.x; e < nparts; e += 256) {
        const double x = pv[e];
        const long long xi = pi[e];
        if (xi >= 0 && (x < best || (x == best && (bi < 0 || xi < bi)))) { best = x; bi = xi; }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double ov = sv[threadIdx.x + o];
            const long long oi = si[threadIdx.x + o];
            if (oi >= 0 && (ov < sv[threadIdx.x] || (ov == sv[threadIdx.x] && (si[threadIdx.x] < 0 || oi < si[threadIdx.x])))) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { *out_min = sv[0]; *out_argmin = si[0]; }
}

}  // namespace

extern "C" {

int ws_vote_update(const float* logits, int64_t n, int32_t c, const float* points, float radius_mask, const int64_t* inds,
                   float* probs, int64_t n_cloud, float smooth, void* stream)
{
    WS_REQUIRE(n >= 0 && c >= 1 && c <= 64 && n_cloud >= 0, "bad sizes n=%lld c=%d", (long long)n, c);
    if (n == 0) return WS_OK;
    WS_REQUIRE(logits && inds && probs && (radius_mask <= 0.0f || points), "NULL argument");
    vote_update_kernel<<<ws_grid(n, 256), 256, 0, (hipStream_t)stream>>>(logits, n, c, points,
                                                                        radius_mask > 0.0f ? radius_mask * radius_mask : 0.0f, inds,
                                                                        probs, n_cloud, smooth);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_project_confusion(const float* probs, int32_t c, const int32_t* proj, int64_t m, const int32_t* labels, int32_t* preds,
                         int32_t nc, int64_t* confusion, void* stream)
{
    WS_REQUIRE(m >= 0 && c >= 1 && nc >= 1 && nc <= 64, "bad sizes m=%lld c=%d nc=%d", (long long)m, c, nc);
    if (m == 0) return WS_OK;
    WS_REQUIRE(probs && (preds || (labels && confusion)), "NULL argument");
    WS_REQUIRE(!labels || confusion, "labels need a confusion matrix");
    project_confusion_kernel<<<ws_grid(m, 256, 1024), 256, sizeof(int) * nc * nc, (hipStream_t)stream>>>(
        probs, c, proj, m, labels, preds, nc, (long long*)confusion);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int64_t ws_potentials_scratch_bytes(int64_t n)
{
    (void)n;
    return 1024 * (int64_t)(sizeof(double) + sizeof(long long));
}

int ws_potentials_update(const float* pot_points, int64_t n, const double* h_center, double radius, double* potentials,
                         double* out_min, int64_t* out_argmin, void* scratch, void* stream)
{
    WS_REQUIRE(n >= 1 && radius > 0.0, "bad sizes n=%lld radius=%g", (long long)n, radius);
    WS_REQUIRE(pot_points && h_center && potentials && out_min && out_argmin && scratch, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    potentials_update_kernel<<<ws_grid(n, 256), 256, 0, st>>>(pot_points, n, h_center[0], h_center[1], h_center[2], radius, potentials);
    WS_LAUNCH_CHECK();
    const int blocks = ws_grid(n, 256, 1024);
    double* pv = (double*)scratch;
    long long* pi = (long long*)((char*)scratch + 1024 * sizeof(double));
    argmin_partial_kernel<<<blocks, 256, 0, st>>>(potentials, n, pv, pi);
    WS_LAUNCH_CHECK();
    argmin_final_kernel<<<1, 256, 0, st>>>(pv, pi, blocks, out_min, (long long*)out_argmin);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

}  // extern "C"
