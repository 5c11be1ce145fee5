// weasal_amd/csrc/contrast.hip -- the per-point part of the supervised contrastive loss of the pseudo-label
// trainer, KPFCNN.contrast_loss (models/architectures.py:455-497), fused.
//
// The reference materialises six [N, 1000] float matrices (three masks, the logits, exp_logits, log_prob) and
// their autograd copies; N = 400 000 makes that ~1.6 GB each.  Here a thread owns one point i, the slice table
// (1000 normalised logit rows + their index / certainty / pseudo label) lives in LDS, and the row statistics
//   m_i = max_j mul_ij,  E_i = sum_j use_ij exp(mul_ij - m_i),  P_i = sum_j pos_ij,  S_i = sum_j pos_ij (mul_ij - m_i)
//   mul_ij = <o_i, s_j> / T,   use_ij = (slc_idx_j != i) & (certain_slc_j == certain_i),   pos_ij = use_ij & (lbl_slc_j == lbl_i)
//   loss_i = -T * (S_i - P_i log(E_i + eps)) / (P_i + 1e-12)                          (:478-497)
// are two passes over LDS broadcasts; nothing of size [N, 1000] exists.  Backward: d mul_ij = g_i * (-T / (P_i + 1e-12))
// * (pos_ij - P_i use_ij exp(mul_ij - m_i) / (E_i + eps)), with the max detached as in the reference (:486); the
// gradient of the point rows is row-parallel again, the gradient of the slice rows is column-parallel over
// staged row tiles with per-chunk partials added in a fixed order (deterministic, no float atomics).
#include "ws_common.h"

namespace {

constexpr int CT_ROWS = 256;        // points per workgroup (one per thread)
constexpr int CT_SMAX = 1024;       // slice columns supported (the reference uses 1000)

template <int CP>
struct SliceTable {
    float xs[CT_SMAX][CP];
    int idx[CT_SMAX];
    int tag[CT_SMAX];               // (label << 1) | certain
};

template <int CP>
__device__ __forceinline__ void stage_slice(SliceTable<CP>& tb, const float* __restrict__ xs, int s, int c,
                                            const int64_t* __restrict__ slc_idx, const uint8_t* __restrict__ certain,
                                            const int64_t* __restrict__ lbl)
{
    for (int e = threadIdx.x; e < s * CP; e += blockDim.x) {
        const int j = e / CP, cc = e % CP;
        tb.xs[j][cc] = cc < c ? xs[(int64_t)j * c + cc] : 0.0f;
    }
    for (int j = threadIdx.x; j < s; j += blockDim.x) {
        const int64_t p = slc_idx[j];
        tb.idx[j] = (int)p;
        tb.tag[j] = ((int)lbl[p] << 1) | (certain[p] ? 1 : 0);
    }
}

constexpr int RB = 2;               // points per thread in the row-parallel kernels

template <int CP>
__device__ __forceinline__ float dotc(const float (&o)[CP], const float* __restrict__ s)
{
    float d = 0.0f;
#pragma unroll
    for (int cc = 0; cc < CP; ++cc) d += o[cc] * s[cc];
    return d;
}

template <int CP>
__global__ __launch_bounds__(CT_ROWS) void contrast_fwd_kernel(const float* __restrict__ on, int64_t n, int c,
                                                                const float* __restrict__ xs, int s,
                                                                const int64_t* __restrict__ slc_idx,
                                                                const uint8_t* __restrict__ certain,
                                                                const int64_t* __restrict__ lbl, float temperature, float eps,
                                                                float* __restrict__ loss, float* __restrict__ rowmax,
                                                                float* __restrict__ den, float* __restrict__ npos)
{
    __shared__ SliceTable<CP> tb;
    stage_slice<CP>(tb, xs, s, c, slc_idx, certain, lbl);
    __syncthreads();
    // RB points per thread: every LDS broadcast of a slice row serves RB dot products (the broadcasts, not
    // the arithmetic, bound this kernel)
    int64_t i[RB];
    float o[RB][CP];
    int mytag[RB];
    float m[RB], E[RB], P[RB], S[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        i[b] = ((int64_t)blockIdx.x * RB + b) * CT_ROWS + threadIdx.x;
        const int64_t ic = i[b] < n ? i[b] : n - 1;
#pragma unroll
        for (int cc = 0; cc < CP; ++cc) o[b][cc] = cc < c ? on[ic * c + cc] : 0.0f;
        mytag[b] = ((int)lbl[ic] << 1) | (certain[ic] ? 1 : 0);
        m[b] = -3.0e38f; E[b] = 0.0f; P[b] = 0.0f; S[b] = 0.0f;
    }
    // ONE pass over the slice (round 3; the first form took the row maximum in a pass of its own).  The rows are unit
    // vectors, so every logit lies in [-1/T, 1/T]: the sums are taken against that bound,
    //   E' = sum_j use_ij exp(mul_ij - 1/T),   S' = sum_j pos_ij mul_ij,   m = max_j mul_ij   (over ALL columns, :484-486),
    // and rebased on the true maximum afterwards: E = E' exp(1/T - m), S = S' - P m.  exp(mul - 1/T) >= exp(-2/T) = 2e-9
    // at T = 0.1: no underflow, the same relative rounding as the two-pass form.  (The maximum cannot be dropped
    // altogether: eps enters log(E + eps) after the subtraction, :490.)
    const float inv_t = 1.0f / temperature;
    for (int j = 0; j < s; ++j) {
        float sj[CP];
#pragma unroll
        for (int cc = 0; cc < CP; ++cc) sj[cc] = tb.xs[j][cc];
        const int tg = tb.tag[j], sid = tb.idx[j];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const float mul = dotc<CP>(o[b], sj) * inv_t;
            m[b] = fmaxf(m[b], mul);
            const bool use = sid != (int)i[b] && ((tg ^ mytag[b]) & 1) == 0;
            const bool pos = use && tg == mytag[b];
            E[b] += use ? __expf(mul - inv_t) : 0.0f;
            P[b] += pos ? 1.0f : 0.0f;
            S[b] += pos ? mul : 0.0f;
        }
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        E[b] *= __expf(inv_t - m[b]);
        S[b] -= P[b] * m[b];
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        if (i[b] >= n) continue;
        const float d = E[b] + eps;
        loss[i[b]] = -temperature * ((S[b] - P[b] * logf(d)) / (P[b] + 1e-12f));
        rowmax[i[b]] = m[b];
        den[i[b]] = d;
        npos[i[b]] = P[b];
    }
}

// gradient of the point rows: d_on[i, :] = sum_j dmul_ij * s_j / T
template <int CP>
__global__ __launch_bounds__(CT_ROWS) void contrast_bwd_rows_kernel(const float* __restrict__ on, int64_t n, int c,
                                                                     const float* __restrict__ xs, int s,
                                                                     const int64_t* __restrict__ slc_idx,
                                                                     const uint8_t* __restrict__ certain,
                                                                     const int64_t* __restrict__ lbl, float temperature,
                                                                     const float* __restrict__ rowmax,
                                                                     const float* __restrict__ den, const float* __restrict__ npos,
                                                                     const float* __restrict__ g, float* __restrict__ d_on)
{
    __shared__ SliceTable<CP> tb;
    stage_slice<CP>(tb, xs, s, c, slc_idx, certain, lbl);
    __syncthreads();
    int64_t i[RB];
    float o[RB][CP], acc[RB][CP];
    int mytag[RB];
    float m[RB], P[RB], rden[RB], gs[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        i[b] = ((int64_t)blockIdx.x * RB + b) * CT_ROWS + threadIdx.x;
        const int64_t ic = i[b] < n ? i[b] : n - 1;
#pragma unroll
        for (int cc = 0; cc < CP; ++cc) {
            o[b][cc] = cc < c ? on[ic * c + cc] : 0.0f;
            acc[b][cc] = 0.0f;
        }
        mytag[b] = ((int)lbl[ic] << 1) | (certain[ic] ? 1 : 0);
        m[b] = rowmax[ic]; P[b] = npos[ic]; rden[b] = 1.0f / den[ic];
        gs[b] = i[b] < n ? g[ic] * (-temperature / (P[b] + 1e-12f)) / temperature : 0.0f;
    }
    const float inv_t = 1.0f / temperature;
    for (int j = 0; j < s; ++j) {
        float sj[CP];
#pragma unroll
        for (int cc = 0; cc < CP; ++cc) sj[cc] = tb.xs[j][cc];
        const int tg = tb.tag[j], sid = tb.idx[j];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const float lg = dotc<CP>(o[b], sj) * inv_t - m[b];
            const bool use = sid != (int)i[b] && ((tg ^ mytag[b]) & 1) == 0;
            const bool pos = use && tg == mytag[b];
            const float w = gs[b] * ((pos ? 1.0f : 0.0f) - (use ? P[b] * __expf(lg) * rden[b] : 0.0f));
#pragma unroll
            for (int cc = 0; cc < CP; ++cc) acc[b][cc] += w * sj[cc];
        }
    }
#pragma unroll
    for (int b = 0; b < RB; ++b)
        if (i[b] < n)
            for (int cc = 0; cc < c; ++cc) d_on[i[b] * c + cc] = acc[b][cc];
}

// gradient of the slice rows: partial[chunk][j, :] = sum over the chunk's points of dmul_ij * o_i / T.
// Thread = slice column(s) j = t, t + 256, ...; the chunk's points are staged 256 at a time and broadcast.
template <int CP>
__global__ __launch_bounds__(CT_ROWS) __attribute__((amdgpu_waves_per_eu(2, 2))) void contrast_bwd_slice_kernel(const float* __restrict__ on, int64_t n, int c,
                                                                      const float* __restrict__ xs, int s,
                                                                      const int64_t* __restrict__ slc_idx,
                                                                      const uint8_t* __restrict__ certain,
                                                                      const int64_t* __restrict__ lbl, float temperature,
                                                                      const float* __restrict__ rowmax,
                                                                      const float* __restrict__ den, const float* __restrict__ npos,
                                                                      const float* __restrict__ g, int64_t chunk,
                                                                      float* __restrict__ partial)
{
    constexpr int JT = CT_SMAX / CT_ROWS;                 // columns per thread
    __shared__ float ro[CT_ROWS][CP];
    __shared__ float rm[CT_ROWS], rrden[CT_ROWS], rP[CT_ROWS], rgs[CT_ROWS];
    __shared__ int rtag[CT_ROWS];
    const int t = threadIdx.x;
    float sx[JT][CP], acc[JT][CP];
    int sidx[JT], stag[JT];
#pragma unroll
    for (int q = 0; q < JT; ++q) {
        const int j = t + CT_ROWS * q;
        const bool live = j < s;
        const int64_t p = live ? slc_idx[j] : 0;
        sidx[q] = live ? (int)p : -1;
        stag[q] = live ? (((int)lbl[p] << 1) | (certain[p] ? 1 : 0)) : -2;
#pragma unroll
        for (int cc = 0; cc < CP; ++cc) {
            sx[q][cc] = (live && cc < c) ? xs[(int64_t)j * c + cc] : 0.0f;
            acc[q][cc] = 0.0f;
        }
    }
    const float inv_t = 1.0f / temperature;
    const int64_t beg = (int64_t)blockIdx.x * chunk;
    const int64_t end = beg + chunk < n ? beg + chunk : n;
    for (int64_t r0 = beg; r0 < end; r0 += CT_ROWS) {
        const int64_t i = r0 + t;
        const bool live = i < end;
        const float gi = live ? g[i] : 0.0f;
        const float P = live ? npos[i] : 0.0f;
#pragma unroll
        for (int cc = 0; cc < CP; ++cc) ro[t][cc] = (live && cc < c) ? on[i * c + cc] : 0.0f;
        rm[t] = live ? rowmax[i] : 0.0f;
        rrden[t] = live ? 1.0f / den[i] : 0.0f;
        rP[t] = P;
        rgs[t] = gi * (-temperature / (P + 1e-12f)) / temperature;
        rtag[t] = live ? (((int)lbl[i] << 1) | (certain[i] ? 1 : 0)) : -4;
        __syncthreads();
        const int rows = (int)(end - r0 < CT_ROWS ? end - r0 : CT_ROWS);
        for (int r = 0; r < rows; ++r) {
            const float gs = rgs[r];
            if (gs == 0.0f) continue;                     // block-uniform: the row is a broadcast
            float o[CP];
#pragma unroll
            for (int cc = 0; cc < CP; ++cc) o[cc] = ro[r][cc];
            const float m = rm[r], P2 = rP[r], rd = rrden[r];
            const int tg = rtag[r];
            const int row = (int)(r0 + r);
#pragma unroll
            for (int q = 0; q < JT; ++q) {
                const float lg = dotc<CP>(o, sx[q]) * inv_t - m;
                const bool use = sidx[q] >= 0 && sidx[q] != row && ((stag[q] ^ tg) & 1) == 0;
                const bool pos = use && stag[q] == tg;
                const float w = gs * ((pos ? 1.0f : 0.0f) - (use ? P2 * __expf(lg) * rd : 0.0f));
#pragma unroll
                for (int cc = 0; cc < CP; ++cc) acc[q][cc] += w * o[cc];
            }
        }
        __syncthreads();
    }
    float* out = partial + (int64_t)blockIdx.x * s * c;
#pragma unroll
    for (int q = 0; q < JT; ++q) {
        const int j = t + CT_ROWS * q;
        if (j < s)
            for (int cc = 0; cc < c; ++cc) out[(int64_t)j * c + cc] = acc[q][cc];
    }
}

__global__ __launch_bounds__(256) void contrast_reduce_kernel(const float* __restrict__ partial, int64_t elems, int chunks,
                                                               float* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    float s = 0.0f;
    int cidx = 0;
    for (; cidx + 3 < chunks; cidx += 4) {                 // four loads in flight, added in a fixed order
        const float v0 = partial[(int64_t)cidx * elems + e], v1 = partial[(int64_t)(cidx + 1) * elems + e];
        const float v2 = partial[(int64_t)(cidx + 2) * elems + e], v3 = partial[(int64_t)(cidx + 3) * elems + e];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; cidx < chunks; ++cidx) s += partial[(int64_t)cidx * elems + e];
    out[e] = s;
}

int64_t slice_chunk(int64_t n)
{
    int64_t c = ws_ceil_div(n, 2048);                      // <= 2048 chunks: enough workgroups to fill every SIMD twice
    c = ws_ceil_div(c, CT_ROWS) * CT_ROWS;
    return c < CT_ROWS ? CT_ROWS : c;
}

}  // namespace

// the matrix-core form of the same three passes (contrast_mfma.hip), the default
extern "C" int ws_contrast_variant;
extern "C" int ws_contrast_mfma_fwd(const float* on, int64_t n, int32_t c, const float* xs, int32_t s, const int64_t* slc_idx,
                                    const uint8_t* certain, const int64_t* lbl, float temperature, float eps, float* loss,
                                    float* rowmax, float* den, float* npos, void* stream);
extern "C" int64_t ws_contrast_mfma_bwd_scratch_bytes(int64_t n, int32_t c, int32_t s);
extern "C" int ws_contrast_mfma_bwd(const float* on, int64_t n, int32_t c, const float* xs, int32_t s, const int64_t* slc_idx,
                                    const uint8_t* certain, const int64_t* lbl, float temperature, const float* rowmax,
                                    const float* den, const float* npos, const float* g, float* d_on, float* d_xs, void* scratch,
                                    void* stream);

extern "C" {

int ws_contrast_rows_fwd(const float* on, int64_t n, int32_t c, const float* xs, int32_t s, const int64_t* slc_idx,
                         const uint8_t* certain, const int64_t* lbl, float temperature, float eps, float* loss,
                         float* rowmax, float* den, float* npos, void* stream)
{
    WS_REQUIRE(n >= 0 && c >= 1 && s >= 1, "bad sizes n=%lld c=%d s=%d", (long long)n, c, s);
    if (c > 16 || s > CT_SMAX) return ws_fail(WS_ERR_UNSUPPORTED, "contrast rows: c=%d (<= 16) s=%d (<= %d)", c, s, CT_SMAX);
    WS_REQUIRE(n < (1ll << 31), "n exceeds int32");
    if (n == 0) return WS_OK;
    WS_REQUIRE(on && xs && slc_idx && certain && lbl && loss && rowmax && den && npos, "NULL argument");
    if (ws_contrast_variant == 2)
        return ws_contrast_mfma_fwd(on, n, c, xs, s, slc_idx, certain, lbl, temperature, eps, loss, rowmax, den, npos, stream);
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ws_ceil_div(n, CT_ROWS * RB);
#define WS_CF(CPV) contrast_fwd_kernel<CPV><<<grid, CT_ROWS, 0, st>>>(on, n, c, xs, s, slc_idx, certain, lbl, temperature, eps, loss, rowmax, den, npos)
    if (c <= 4) WS_CF(4);
    else if (c <= 8) WS_CF(8);
    else if (c <= 12) WS_CF(12);
    else WS_CF(16);
#undef WS_CF
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int64_t ws_contrast_rows_bwd_scratch_bytes(int64_t n, int32_t c, int32_t s)
{
    const int64_t a = ws_ceil_div(n > 0 ? n : 1, slice_chunk(n)) * (int64_t)s * c * (int64_t)sizeof(float);
    const int64_t b = ws_contrast_mfma_bwd_scratch_bytes(n, c, s);
    return a > b ? a : b;          // (either variant may run)
}

int ws_contrast_rows_bwd(const float* on, int64_t n, int32_t c, const float* xs, int32_t s, const int64_t* slc_idx,
                         const uint8_t* certain, const int64_t* lbl, float temperature, const float* rowmax,
                         const float* den, const float* npos, const float* g, float* d_on, float* d_xs, void* scratch,
                         void* stream)
{
    WS_REQUIRE(n >= 0 && c >= 1 && s >= 1, "bad sizes n=%lld c=%d s=%d", (long long)n, c, s);
    if (c > 16 || s > CT_SMAX) return ws_fail(WS_ERR_UNSUPPORTED, "contrast rows: c=%d (<= 16) s=%d (<= %d)", c, s, CT_SMAX);
    WS_REQUIRE(d_xs, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        WS_HIP(hipMemsetAsync(d_xs, 0, sizeof(float) * (size_t)s * c, st));
        return WS_OK;
    }
    WS_REQUIRE(on && xs && slc_idx && certain && lbl && rowmax && den && npos && g && d_on && scratch, "NULL argument");
    if (ws_contrast_variant == 2)
        return ws_contrast_mfma_bwd(on, n, c, xs, s, slc_idx, certain, lbl, temperature, rowmax, den, npos, g, d_on, d_xs, scratch, stream);
    const unsigned grid = (unsigned)ws_ceil_div(n, CT_ROWS * RB);
    const int64_t chunk = slice_chunk(n);
    const int chunks = (int)ws_ceil_div(n, chunk);
    float* partial = chunks == 1 ? d_xs : (float*)scratch;
#define WS_CB(CPV)                                                                                                       \
    do {                                                                                                                 \
        contrast_bwd_rows_kernel<CPV><<<grid, CT_ROWS, 0, st>>>(on, n, c, xs, s, slc_idx, certain, lbl, temperature, rowmax, \
                                                                den, npos, g, d_on);                                     \
        contrast_bwd_slice_kernel<CPV><<<chunks, CT_ROWS, 0, st>>>(on, n, c, xs, s, slc_idx, certain, lbl, temperature,  \
                                                                   rowmax, den, npos, g, chunk, partial);               \
    } while (0)
    if (c <= 4) WS_CB(4);
    else if (c <= 8) WS_CB(8);
    else if (c <= 12) WS_CB(12);
    else WS_CB(16);
#undef WS_CB
    WS_LAUNCH_CHECK();
    if (chunks > 1) {
        const int64_t elems = (int64_t)s * c;
        contrast_reduce_kernel<<<(unsigned)ws_ceil_div(elems, 256), 256, 0, st>>>(partial, elems, chunks, d_xs);
        WS_LAUNCH_CHECK();
    }
    return WS_OK;
}

}  // extern "C"
