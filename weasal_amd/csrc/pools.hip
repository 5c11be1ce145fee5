// weasal_amd/csrc/pools.hip -- max_pool / closest_pool (models/blocks.py:80-111) forward and
// backward.  Rows are contiguous (4*c bytes): lanes run along the channel axis so that every
// access is a coalesced row segment; the backward forms gather through the transposed table
// (csr.hip) instead of the reference's scatter_add.
#include "ws_common.h"

namespace {

// one wave per query row; lanes stride over channels
__global__ __launch_bounds__(256) void max_pool_fwd_kernel(const float* __restrict__ x, int64_t ns, int c,
                                                            const int64_t* __restrict__ inds, int64_t nq, int h,
                                                            float* __restrict__ out, int32_t* __restrict__ arg)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < nq; q += (int64_t)gridDim.x * 4) {
        for (int c0 = 0; c0 < c; c0 += 64) {
            const int ch = c0 + lane;
            float best = -3.4e38f;
            int bi = 0;
            for (int j = 0; j < h; ++j) {
                const int64_t s = inds[q * h + j];           // wave-uniform
                float v = 0.0f;                               // shadow row = zeros (blocks.py:104)
                if (s >= 0 && s < ns && ch < c) v = x[s * c + ch];
                if (v > best) { best = v; bi = j; }           // first maximum wins
            }
            if (ch < c) {
                out[q * c + ch] = best;
                if (arg) arg[q * c + ch] = bi;
            }
        }
    }
}

// one wave per support row
__global__ __launch_bounds__(256) void max_pool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ arg,
                                                            int h, int c, const int32_t* __restrict__ t_offsets,
                                                            const int32_t* __restrict__ t_pairs, int64_t ns,
                                                            float* __restrict__ dx)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < ns; s += (int64_t)gridDim.x * 4) {
        const int beg = t_offsets[s], end = t_offsets[s + 1];
        for (int c0 = 0; c0 < c; c0 += 64) {
            const int ch = c0 + lane;
            float acc = 0.0f;
            if (ch < c) {
                for (int p = beg; p < end; ++p) {
                    const int pair = t_pairs[p];
                    const int q = pair / h, col = pair - q * h;
                    if (arg[(int64_t)q * c + ch] == col) acc += dy[(int64_t)q * c + ch];
                }
                dx[s * c + ch] = acc;
            }
        }
    }
}

__global__ __launch_bounds__(256) void closest_pool_fwd_kernel(const float* __restrict__ x, int64_t ns, int c,
                                                                const int64_t* __restrict__ inds, int64_t nq, int h,
                                                                float* __restrict__ out)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < nq; q += (int64_t)gridDim.x * 4) {
        const int64_t s = inds[q * h];
        const bool real = s >= 0 && s < ns;
        for (int ch = lane; ch < c; ch += 64) out[q * c + ch] = real ? x[s * c + ch] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void closest_pool_bwd_kernel(const float* __restrict__ dy, int h, int c,
                                                                const int32_t* __restrict__ t_offsets,
                                                                const int32_t* __restrict__ t_pairs, int64_t ns,
                                                                float* __restrict__ dx)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < ns; s += (int64_t)gridDim.x * 4) {
        const int beg = t_offsets[s], end = t_offsets[s + 1];
        for (int c0 = 0; c0 < c; c0 += 64) {
            const int ch = c0 + lane;
            float acc = 0.0f;
            if (ch < c) {
                for (int p = beg; p < end; ++p) {
                    const int pair = t_pairs[p];
                    const int q = pair / h;
                    if (pair - q * h == 0) acc += dy[(int64_t)q * c + ch];
                }
                dx[s * c + ch] = acc;
            }
        }
    }
}

}  // namespace

extern "C" {

int ws_max_pool_fwd(const float* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                    float* out, int32_t* arg, void* stream)
{
    WS_REQUIRE(ns >= 0 && nq >= 0 && c >= 1 && h >= 1, "bad sizes");
    if (nq == 0) return WS_OK;
    WS_REQUIRE(inds && out && (ns == 0 || x), "NULL argument");
    max_pool_fwd_kernel<<<ws_grid(nq, 4), 256, 0, (hipStream_t)stream>>>(x, ns, c, inds, nq, h, out, arg);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_max_pool_bwd(const float* dy, const int32_t* arg, int64_t nq, int32_t h, int32_t c,
                    const int32_t* t_offsets, const int32_t* t_pairs, int64_t ns, float* dx, void* stream)
{
    WS_REQUIRE(ns >= 0 && nq >= 0 && c >= 1 && h >= 1, "bad sizes");
    if (ns == 0) return WS_OK;
    WS_REQUIRE(t_offsets && dx && (nq == 0 || (dy && arg && t_pairs)), "NULL argument");
    max_pool_bwd_kernel<<<ws_grid(ns, 4), 256, 0, (hipStream_t)stream>>>(dy, arg, h, c, t_offsets, t_pairs, ns, dx);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_closest_pool_fwd(const float* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                        float* out, void* stream)
{
    WS_REQUIRE(ns >= 0 && nq >= 0 && c >= 1 && h >= 1, "bad sizes");
    if (nq == 0) return WS_OK;
    WS_REQUIRE(inds && out && (ns == 0 || x), "NULL argument");
    closest_pool_fwd_kernel<<<ws_grid(nq, 4), 256, 0, (hipStream_t)stream>>>(x, ns, c, inds, nq, h, out);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_closest_pool_bwd(const float* dy, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                        const int32_t* t_pairs, int64_t ns, float* dx, void* stream)
{
    WS_REQUIRE(ns >= 0 && nq >= 0 && c >= 1 && h >= 1, "bad sizes");
    if (ns == 0) return WS_OK;
    WS_REQUIRE(t_offsets && dx && (nq == 0 || (dy && t_pairs)), "NULL argument");
    closest_pool_bwd_kernel<<<ws_grid(ns, 4), 256, 0, (hipStream_t)stream>>>(dy, h, c, t_offsets, t_pairs, ns, dx);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

}  // extern "C"
