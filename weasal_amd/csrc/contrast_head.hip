// weasal_amd/csrc/contrast_head.hip -- everything of KPFCNN.contrast_loss (models/architectures.py:405-504) AROUND the
// [N, slc_con] part (contrast.hip / contrast_mfma.hip): a handful of kernels instead of ~75 framework launches per step.
//
//   head, forward  (:425-454, :475-476)   softmax -> pseudo_logits = max prob, pseudo label = argmax (given labels < 10 win),
//                                         certain = (pseudo_logits > threshold) | labelled, L2-normalised rows `on`;
//                                         then the slice of `s` valid points WITHOUT host synchronisation: the r-th valid point
//                                         is located through per-block counts of valid points (no [N] prefix sum in memory)
//   tail, forward  (:498-504)             points with loss <= 0 dropped, mean per pseudo label, classes with mean <= 0 (or empty)
//                                         dropped, mean of the rest; fixed summation order (no float atomics)
//   tail / head, backward                 per-point gradient coefficient; slice-row gradients added back onto their points
//                                         (duplicates of a point in the slice summed in slice order), backward of the normalisation
//
// `state` (2 x int32, device): [0] = number of valid points, [1] = arrival counter of the tail's reduction -- written by the
// head's selection kernel, so the tail needs no memset of its own.
#include "ws_common.h"

namespace {

constexpr int CH_CMAX = 16;         // classes (columns of the logits) supported, = the [N, slc_con] kernels' limit
constexpr int CH_BINS = 16;         // label bins of the tail (max(C, 10) in the reference's setting)
constexpr int CH_MAXBLK = 8192;     // per-block counts the selection kernel keeps in LDS

__global__ __launch_bounds__(256) void contrast_prepare_kernel(const float* __restrict__ x, int64_t n, int c, int64_t ldx,
                                                                const int64_t* __restrict__ labels, float threshold, int rpb,
                                                                float* __restrict__ on, float* __restrict__ inv_norm,
                                                                uint8_t* __restrict__ certain, int64_t* __restrict__ lbl,
                                                                int32_t* __restrict__ blk_cnt)
{
    __shared__ int wsum[4];
    const int64_t row0 = (int64_t)blockIdx.x * rpb;
    int mine = 0;
    for (int off = threadIdx.x; off < rpb; off += 256) {
        const int64_t p = row0 + off;
        if (p >= n) break;
        float v[CH_CMAX], e[CH_CMAX];
        float m = -INFINITY, ss = 0.0f;
#pragma unroll
        for (int k = 0; k < CH_CMAX; ++k) {
            v[k] = k < c ? x[p * ldx + k] : 0.0f;
            if (k < c) { m = fmaxf(m, v[k]); ss += v[k] * v[k]; }
        }
#pragma unroll
        for (int k = 0; k < CH_CMAX; ++k) e[k] = k < c ? expf(v[k] - m) : 0.0f;
        // the sum in the order of the framework's row softmax (a 16-lane butterfly: i with i^8, ^4, ^2, ^1); zeros are exact
        float t8[8], t4[4], t2[2];
#pragma unroll
        for (int k = 0; k < 8; ++k) t8[k] = e[k] + e[k + 8];
#pragma unroll
        for (int k = 0; k < 4; ++k) t4[k] = t8[k] + t8[k + 4];
#pragma unroll
        for (int k = 0; k < 2; ++k) t2[k] = t4[k] + t4[k + 2];
        const float sum = t2[0] + t2[1];
        float best = -1.0f;
        int arg = 0;
#pragma unroll
        for (int k = 0; k < CH_CMAX; ++k) {
            const float q = e[k] / sum;
            if (k < c && q > best) { best = q; arg = k; }          // first maximum, as torch.argmax
        }
        const int64_t lab = labels[p];
        const bool labelled = lab < 10;                            // > 10 = unlabelled (:430-433)
        const bool cert = (best > threshold) || labelled;
        certain[p] = cert ? 1 : 0;
        lbl[p] = labelled ? lab : (int64_t)arg;
        const float nrm = sqrtf(ss);
        const float inv = 1.0f / fmaxf(nrm, 1e-12f);               // F.normalize: x / max(|x|, eps)
        inv_norm[p] = inv;
#pragma unroll
        for (int k = 0; k < CH_CMAX; ++k)
            if (k < c) on[p * c + k] = v[k] / fmaxf(nrm, 1e-12f);
        mine += cert ? 1 : 0;
    }
    // block count of valid points
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// grid = ceil(s / 64) workgroups of 16 waves; every workgroup rebuilds the (small) prefix of the block counts, then each
// wave locates 4 slots: binary search over the prefix, then the block's `certain` bytes 256 at a time (4 per lane)
__global__ __launch_bounds__(1024) void contrast_select_kernel(const int32_t* __restrict__ blk_cnt, int nblk, int rpb, int64_t n,
                                                                const uint8_t* __restrict__ certain,
                                                                const float* __restrict__ u, const int64_t* __restrict__ r_given,
                                                                int s, const float* __restrict__ on, int c,
                                                                int64_t* __restrict__ slc_idx, float* __restrict__ xs,
                                                                int32_t* __restrict__ state)
{
    __shared__ int pre[CH_MAXBLK];          // exclusive prefix of the block counts
    __shared__ int part[1024];
    const int t = threadIdx.x;
    constexpr int PER = CH_MAXBLK / 1024;
    int loc[PER];
    int acc = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int b = t * PER + k;
        loc[k] = acc;
        acc += b < nblk ? blk_cnt[b] : 0;
    }
    part[t] = acc;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {        // inclusive scan of the per-thread sums
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const int base = t > 0 ? part[t - 1] : 0;
    const int nv = part[1023];
#pragma unroll
    for (int k = 0; k < PER; ++k) pre[t * PER + k] = base + loc[k];
    __syncthreads();
    if (blockIdx.x == 0 && t == 0) { state[0] = nv; state[1] = 0; }
    const int wave = t >> 6, lane = t & 63;
    for (int jj = 0; jj < 4; ++jj) {
        const int j = blockIdx.x * 64 + wave * 4 + jj;
        if (j >= s) break;                                          // wave-uniform
        int64_t r;
        if (r_given) r = r_given[j];
        else {
            const int64_t rr = (int64_t)floorf(u[j] * (float)nv);
            r = (nv < s && j < nv) ? (int64_t)j : rr;               // fewer valid points than slots: each once, then repeats (:450-454)
        }
        const int64_t cap = nv > 0 ? (int64_t)nv - 1 : 0;
        if (r > cap) r = cap;
        if (r < 0) r = 0;
        int64_t found = n - 1;                                      // no valid point at all: searchsorted past the end, clamped
        if (nv > 0) {
            int lo = 0, hi = nblk - 1;                              // last block whose exclusive prefix is <= r
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if ((int64_t)pre[mid] <= r) lo = mid; else hi = mid - 1;
            }
            int rem = (int)(r - pre[lo]);
            const int64_t a = (int64_t)lo * rpb;
            const int64_t bnd = a + rpb < n ? a + rpb : n;
            for (int64_t p0 = a; p0 < bnd; p0 += 256) {
                const int64_t p = p0 + 4 * lane;
                unsigned bits = 0;                                  // bit k: row p + k is valid
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (p + k < bnd && certain[p + k]) bits |= 1u << k;
                const int nz = __builtin_popcount(bits);
                int incl = nz;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o, 64);
                    incl += lane >= o ? v : 0;
                }
                const int total = __shfl(incl, 63, 64);
                if (rem < total) {
                    const int excl = incl - nz;
                    const bool mine = rem >= excl && rem < incl;
                    int64_t cand = 0;
                    if (mine) {
                        int need = rem - excl;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (bits & (1u << k)) { if (need == 0) cand = p + k; --need; }
                    }
                    const unsigned long long m = __ballot(mine);
                    const int src = __builtin_ctzll(m);
                    found = ((int64_t)__shfl((int)(cand >> 32), src, 64) << 32) | (unsigned)__shfl((int)(cand & 0xffffffffll), src, 64);
                    break;
                }
                rem -= total;
            }
        }
        if (lane == 0) slc_idx[j] = found;
        if (lane < c) xs[(int64_t)j * c + lane] = on[found * c + lane];
    }
}

// tail: per-label sums / counts of the kept points, block partials in a fixed order; the last block to arrive finishes
__global__ __launch_bounds__(256) void contrast_tail_kernel(const float* __restrict__ pts_loss, const int64_t* __restrict__ lbl,
                                                             int64_t n, int n_cls, int rpb, int32_t* __restrict__ state,
                                                             float* __restrict__ partial /*[nblk][2][CH_BINS]*/,
                                                             float* __restrict__ per_class, float* __restrict__ w_cls,
                                                             float* __restrict__ loss)
{
    __shared__ float bs[CH_BINS][257];
    __shared__ float bc[CH_BINS][257];
    __shared__ int last;
    const int t = threadIdx.x;
    for (int k = 0; k < n_cls; ++k) { bs[k][t] = 0.0f; bc[k][t] = 0.0f; }
    const int64_t row0 = (int64_t)blockIdx.x * rpb;
    for (int off = t; off < rpb; off += 256) {
        const int64_t p = row0 + off;
        if (p >= n) break;
        const float v = pts_loss[p];
        int k = (int)lbl[p];
        k = k < 0 ? 0 : (k >= n_cls ? n_cls - 1 : k);
        if (v > 0.0f) { bs[k][t] += v; bc[k][t] += 1.0f; }
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o)
            for (int k = 0; k < n_cls; ++k) { bs[k][t] += bs[k][t + o]; bc[k][t] += bc[k][t + o]; }
        __syncthreads();
    }
    if (t < n_cls) {
        partial[((int64_t)blockIdx.x * 2 + 0) * CH_BINS + t] = bs[t][0];
        partial[((int64_t)blockIdx.x * 2 + 1) * CH_BINS + t] = bc[t][0];
    }
    __threadfence();
    __syncthreads();
    if (t == 0) last = (atomicAdd(&state[1], 1) == (int)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!last) return;
    __threadfence();
    __shared__ float pc[CH_BINS], sl[CH_BINS], cn[CH_BINS];
    // the partials of all workgroups: fetched by all threads (independent loads), summed per class in block order
    float* stage = &bs[0][0];                                       // CH_BINS x 257 floats, free again
    float ssum = 0.0f, csum = 0.0f;
    const int per_pass = (CH_BINS * 257) / (2 * CH_BINS);           // workgroups per pass through the staging area
    for (unsigned b0 = 0; b0 < gridDim.x; b0 += per_pass) {
        const unsigned nb_here = min((unsigned)per_pass, gridDim.x - b0);
        __syncthreads();
        for (unsigned e = t; e < nb_here * 2 * CH_BINS; e += 256)
            stage[e] = __hip_atomic_load(&partial[(int64_t)b0 * 2 * CH_BINS + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (t < n_cls)
            for (unsigned b = 0; b < nb_here; ++b) {
                ssum += stage[(b * 2 + 0) * CH_BINS + t];
                csum += stage[(b * 2 + 1) * CH_BINS + t];
            }
    }
    if (t < n_cls) {
        const float mean = ssum / fmaxf(csum, 1.0f);
        pc[t] = mean;
        sl[t] = mean > 0.0f ? 1.0f : 0.0f;
        cn[t] = fmaxf(csum, 1.0f);
        per_class[t] = mean;
    }
    __syncthreads();
    if (t == 0) {
        float num = 0.0f, den = 0.0f;
        for (int k = 0; k < n_cls; ++k) { num += pc[k] * sl[k]; den += sl[k]; }
        const bool none = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0;                            // no valid point: the reference returns 0 first (:441-443)
        loss[0] = none ? 0.0f : num / den;
        for (int k = 0; k < n_cls; ++k) w_cls[k] = none ? 0.0f : sl[k] / (den * cn[k]);
        state[1] = 0;
    }
}

__global__ __launch_bounds__(256) void contrast_tail_bwd_kernel(const float* __restrict__ pts_loss, const int64_t* __restrict__ lbl,
                                                                 int64_t n, int n_cls, const float* __restrict__ w_cls,
                                                                 const float* __restrict__ g, float* __restrict__ g_row)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int k = (int)lbl[p];
    k = k < 0 ? 0 : (k >= n_cls ? n_cls - 1 : k);
    g_row[p] = pts_loss[p] > 0.0f ? g[0] * w_cls[k] : 0.0f;
}

// slice-row gradients back onto their points: the FIRST slot of a point adds all its slots in slot order (deterministic).
// One wave per slot: its 64 lanes compare the slot's point with 64 table entries at a time (two ballots per round: an earlier
// slot of the same point?  a later one?); later duplicates are rare (about one per 1000 draws out of 400 000) and then added
// in slot order by the lanes that own a channel.
__global__ __launch_bounds__(256) void contrast_slice_add_kernel(const float* __restrict__ d_xs, const int64_t* __restrict__ slc_idx,
                                                                  int s, int c, float* __restrict__ d_on)
{
    __shared__ int idx[2048];
    for (int j = threadIdx.x; j < s; j += 256) idx[j] = (int)slc_idx[j];
    __syncthreads();
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= s) return;                                              // (wave-uniform)
    const int me = idx[j];
    bool earlier = false, later = false;
    for (int q0 = 0; q0 < s; q0 += 64) {
        const int q = q0 + lane;
        const bool same = q < s && idx[q] == me;
        earlier = earlier || __ballot(same && q < j) != 0ull;
        later = later || __ballot(same && q > j) != 0ull;
    }
    if (earlier) return;                                             // (wave-uniform: all 64 lanes stay for the ballots below)
    const int ch = lane < c ? lane : 0;
    float a = d_on[(int64_t)me * c + ch] + d_xs[(int64_t)j * c + ch];
    if (later)          // rare; 64 table entries per round again (a slot-by-slot loop here kept ONE wave busy for ~50 us)
        for (int q0 = (j + 1) & ~63; q0 < s; q0 += 64) {
            const int q = q0 + lane;
            unsigned long long m = __ballot(q < s && q > j && idx[q] == me);
            while (m) {                                              // ascending slots
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                a += d_xs[(int64_t)(q0 + b) * c + ch];
            }
        }
    if (lane < c) d_on[(int64_t)me * c + lane] = a;
}

__global__ __launch_bounds__(256) void contrast_normalize_bwd_kernel(const float* __restrict__ d_on, const float* __restrict__ on,
                                                                      const float* __restrict__ inv_norm, int64_t n, int c,
                                                                      float* __restrict__ d_x, int64_t ldd)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    float g[CH_CMAX], o[CH_CMAX];
    float dot = 0.0f;
#pragma unroll
    for (int k = 0; k < CH_CMAX; ++k) {
        g[k] = k < c ? d_on[p * c + k] : 0.0f;
        o[k] = k < c ? on[p * c + k] : 0.0f;
        dot += g[k] * o[k];
    }
    const float inv = inv_norm[p];
    const bool clamped = inv >= 1e12f;                              // |x| <= eps: the clamp passes no gradient to the norm
#pragma unroll
    for (int k = 0; k < CH_CMAX; ++k)
        if (k < c) d_x[p * ldd + k] = clamped ? g[k] * inv : (g[k] - o[k] * dot) * inv;
}

inline int rows_per_block(int64_t n)
{
    int rpb = 256;
    while (ws_ceil_div(n > 0 ? n : 1, rpb) > CH_MAXBLK) rpb *= 2;
    return rpb;
}

}  // namespace

extern "C" {

int64_t ws_contrast_head_scratch_bytes(int64_t n)
{
    return (ws_ceil_div(n > 0 ? n : 1, rows_per_block(n)) + 4) * (int64_t)sizeof(int32_t);
}

int ws_contrast_head_fwd(const float* x, int64_t n, int32_t c, int64_t ldx, const int64_t* labels, float threshold,
                         const float* u, const int64_t* r_given, int32_t s, float* on, float* inv_norm, uint8_t* certain,
                         int64_t* lbl, int64_t* slc_idx, float* xs, int32_t* state, void* scratch, void* stream)
{
    WS_REQUIRE(x && labels && on && inv_norm && certain && lbl && slc_idx && xs && state && scratch, "NULL argument");
    WS_REQUIRE((u != nullptr) != (r_given != nullptr), "exactly one of u / r_given must be given");
    WS_REQUIRE(n >= 1 && n < (int64_t)2147483647, "n out of range");
    WS_REQUIRE(c >= 1 && c <= CH_CMAX, "1 <= c <= 16 classes supported");
    WS_REQUIRE(s >= 1 && s <= 2048, "1 <= s <= 2048 slice rows supported");
    WS_REQUIRE(ldx >= c, "ldx < c");
    hipStream_t st = (hipStream_t)stream;
    const int rpb = rows_per_block(n);
    const int nblk = (int)ws_ceil_div(n, rpb);
    int32_t* blk_cnt = (int32_t*)scratch;
    contrast_prepare_kernel<<<nblk, 256, 0, st>>>(x, n, c, ldx, labels, threshold, rpb, on, inv_norm, certain, lbl, blk_cnt);
    WS_LAUNCH_CHECK();
    contrast_select_kernel<<<(unsigned)ws_ceil_div(s, 64), 1024, 0, st>>>(blk_cnt, nblk, rpb, n, certain, u, r_given, s, on, c, slc_idx, xs, state);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int64_t ws_contrast_tail_scratch_bytes(int64_t n)
{
    return ws_ceil_div(n > 0 ? n : 1, 4096) * 2 * CH_BINS * (int64_t)sizeof(float);
}

int ws_contrast_tail_fwd(const float* pts_loss, const int64_t* lbl, int64_t n, int32_t n_cls, int32_t* state, float* per_class,
                         float* w_cls, float* loss, void* scratch, void* stream)
{
    WS_REQUIRE(pts_loss && lbl && state && per_class && w_cls && loss && scratch, "NULL argument");
    WS_REQUIRE(n >= 1, "n out of range");
    WS_REQUIRE(n_cls >= 1 && n_cls <= CH_BINS, "1 <= n_cls <= 16 label bins supported");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)ws_ceil_div(n, 4096);
    contrast_tail_kernel<<<nblk, 256, 0, st>>>(pts_loss, lbl, n, n_cls, 4096, state, (float*)scratch, per_class, w_cls, loss);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_contrast_tail_bwd(const float* pts_loss, const int64_t* lbl, int64_t n, int32_t n_cls, const float* w_cls, const float* g,
                         float* g_row, void* stream)
{
    WS_REQUIRE(pts_loss && lbl && w_cls && g && g_row, "NULL argument");
    WS_REQUIRE(n >= 1 && n_cls >= 1 && n_cls <= CH_BINS, "size out of range");
    contrast_tail_bwd_kernel<<<(unsigned)ws_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(pts_loss, lbl, n, n_cls, w_cls, g, g_row);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_contrast_head_bwd(float* d_on, const float* d_xs, const int64_t* slc_idx, int32_t s, const float* on,
                         const float* inv_norm, int64_t n, int32_t c, float* d_x, int64_t ldd, void* stream)
{
    WS_REQUIRE(d_on && d_xs && slc_idx && on && inv_norm && d_x, "NULL argument");
    WS_REQUIRE(n >= 1 && c >= 1 && c <= CH_CMAX && s >= 1 && s <= 2048 && ldd >= c, "size out of range");
    hipStream_t st = (hipStream_t)stream;
    contrast_slice_add_kernel<<<(unsigned)ws_ceil_div(s, 4), 256, 0, st>>>(d_xs, slc_idx, s, c, d_on);
    WS_LAUNCH_CHECK();
    contrast_normalize_bwd_kernel<<<(unsigned)ws_ceil_div(n, 256), 256, 0, st>>>(d_on, on, inv_norm, n, c, d_x, ldd);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

}  // extern "C"
