// weasal_amd/csrc/pools.hip -- max_pool / closest_pool (models/blocks.py:80-111) forward and
// backward.  Rows are contiguous (4*c bytes): lanes run along the channel axis so that every
// access is a coalesced row segment; the backward forms gather through the transposed table
// (csr.hip) instead of the reference's scatter_add.
#include "ws_common.h"
#include <algorithm>
#include "ws_bf16.h"

namespace {

__device__ __forceinline__ int64_t ws_ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }
bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__device__ __forceinline__ int4 load_arg4(const int32_t* p) { return *reinterpret_cast<const int4*>(p); }
__device__ __forceinline__ int4 load_arg4(const uint8_t* p)
{
    const unsigned w = *reinterpret_cast<const unsigned*>(p);
    return make_int4((int)(w & 0xffu), (int)((w >> 8) & 0xffu), (int)((w >> 16) & 0xffu), (int)(w >> 24));
}

// VEC path (c % 4 == 0, 16-byte aligned rows): G = c/4 lanes (capped at 64) cover one row with
// float4 pieces, 64/G query rows per wave, the neighbour loop unrolled by 4 so that four row
// gathers are in flight per lane; shadow columns read nothing (zero row, blocks.py:104).
// AT = the element type of the arg-max record: int32 (the C ABI), or uint8 inside the block calls when h <= 255 -- the
// backward reads one arg piece per (incoming pair, row piece), 2.7 GB at level 0 with 4-byte elements
// Work assignment of the vectorised pools.  ilv = 0: every workgroup takes one contiguous chunk of the (spatially ordered)
// group list (ws_block_range).  ilv > 0 (workgroups per XCD; grid = 8 x ilv): the workgroups of an XCD walk ITS eighth of
// the list together, group g = base + (local workgroup) * 4 + wave + t * (ilv * 4): at any moment the XCD works on ~ilv * 4
// * S neighbouring queries, whose gathered rows fit its 4 MB L2 (with contiguous chunks the resident workgroups sit at
// chunk starts spread over the whole cloud: every row came from the fabric again, FETCH_SIZE ~ the logical bytes).
__device__ __forceinline__ void pool_groups(int64_t ngroups, int ilv, int wave, int64_t& g0, int64_t& gstep, int64_t& gend)
{
    if (ilv > 0 && (gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7, lb = blockIdx.x >> 3, nbx = gridDim.x >> 3;
        const int64_t per8 = (ngroups + 7) / 8;
        const int64_t b8 = (int64_t)x * per8;
        gend = b8 + per8 < ngroups ? b8 + per8 : ngroups;
        g0 = b8 + (int64_t)lb * 4 + wave;
        gstep = (int64_t)nbx * 4;
    } else {
        int64_t ibeg, iend;
        ws_block_range(ngroups, ibeg, iend);
        g0 = ibeg + wave; gstep = 4; gend = iend;
    }
}

template <int G, typename T, typename AT = int32_t, int U = 4>
__global__ __launch_bounds__(256) void max_pool_fwd_vec_kernel(const T* __restrict__ x, int64_t ns, int c,
                                                                const int64_t* __restrict__ inds, int64_t nq, int h,
                                                                T* __restrict__ out, AT* __restrict__ arg,
                                                                const int32_t* __restrict__ order = nullptr, int ilv = 0, int split = 1)
{
    // split > 1 (wide rows of few queries: the deep levels): a group = (query, one 4 G-channel chunk) instead of a query with
    // its chunks one after the other -- 275 queries x 1 024 channels are 1 100 independent waves instead of 275 four times as long
    // order (optional): a spatially coherent permutation of the queries (the cell order of their level's search).  The
    // pooled points come out of the grid subsampling in hash-table order: walked by index, consecutive queries gather
    // rows from all over the cloud and every 512-byte row is fetched ~3.6 times from HBM (PMC, round 2); in cell order
    // the waves of a workgroup share their neighbourhoods in L2.
    constexpr int S = 64 / G;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane % G, slot = lane / G;
    int64_t g0, gstep, gend;
    pool_groups(ws_ceil_div_dev(nq, S) * split, ilv, wave, g0, gstep, gend);
    for (int64_t grp = g0; grp < gend; grp += gstep) {
        const int64_t qi = (grp / split) * S + slot;
        const bool qok = qi < nq;
        const int64_t q = (qok && order) ? (int64_t)order[qi] : qi;
        const int cbeg = split > 1 ? (int)(grp % split) * (4 * G) : 0;
        const int cend = split > 1 ? min(cbeg + 4 * G, c) : c;
        for (int c0 = cbeg; c0 < cend; c0 += 4 * G) {
            const int ch = c0 + 4 * j;
            const bool ok = qok && ch < c;
            float4 best = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
            int bi[4] = {0, 0, 0, 0};
            for (int h0 = 0; h0 < h; h0 += U) {            // U neighbour rows in flight per lane
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int col = h0 + u;
                    int64_t s = (ok && col < h) ? inds[q * h + col] : -1;
                    if (s >= 0 && s < ns) v[u] = ld4(x + s * c + ch);
                    else if (col >= h) v[u] = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {               // first maximum wins
                    if (v[u].x > best.x) { best.x = v[u].x; bi[0] = h0 + u; }
                    if (v[u].y > best.y) { best.y = v[u].y; bi[1] = h0 + u; }
                    if (v[u].z > best.z) { best.z = v[u].z; bi[2] = h0 + u; }
                    if (v[u].w > best.w) { best.w = v[u].w; bi[3] = h0 + u; }
                }
            }
            if (ok) {
                st4(out + q * c + ch, best);
                if (arg) {
                    if constexpr (sizeof(AT) == 4)
                        *reinterpret_cast<int4*>(arg + q * c + ch) = make_int4(bi[0], bi[1], bi[2], bi[3]);
                    else
                        *reinterpret_cast<unsigned*>(arg + q * c + ch) =
                            (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
                }
            }
        }
    }
}

// generic path: one wave per query row; lanes stride over channels
template <typename T>
__global__ __launch_bounds__(256) void max_pool_fwd_kernel(const T* __restrict__ x, int64_t ns, int c,
                                                            const int64_t* __restrict__ inds, int64_t nq, int h,
                                                            T* __restrict__ out, int32_t* __restrict__ arg)
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
                if (s >= 0 && s < ns && ch < c) v = ld1(x + s * c + ch);
                if (v > best) { best = v; bi = j; }           // first maximum wins
            }
            if (ch < c) {
                st1(out + q * c + ch, best);
                if (arg) arg[q * c + ch] = bi;
            }
        }
    }
}

// backward, VEC path: G lanes x float4 per support row, 64/G supports per wave
template <int G, typename T, typename AT = int32_t>
__global__ __launch_bounds__(256) void max_pool_bwd_vec_kernel(const T* __restrict__ dy, const AT* __restrict__ arg,
                                                                int h, int c, const int32_t* __restrict__ t_offsets,
                                                                const int32_t* __restrict__ t_pairs, int64_t ns,
                                                                T* __restrict__ dx, const int32_t* __restrict__ order = nullptr,
                                                                int ilv = 0, const T* __restrict__ add = nullptr, int split = 1)
{
    // split: as in the forward -- a group = (support, one channel chunk)
    // add (same shape as dx, or NULL): a second gradient of the same rows, summed into the store -- dx = pool gradient + add
    constexpr int S = 64 / G;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane % G, slot = lane / G;
    int64_t g0, gstep, gend;
    pool_groups(ws_ceil_div_dev(ns, S) * split, ilv, wave, g0, gstep, gend);
    for (int64_t grp = g0; grp < gend; grp += gstep) {
        const int64_t si = (grp / split) * S + slot;
        const bool sok = si < ns;
        const int64_t s = (sok && order) ? (int64_t)order[si] : si;       // supports in their level's cell order (see the forward)
        const int beg = sok ? t_offsets[s] : 0, end = sok ? t_offsets[s + 1] : 0;
        const int cbeg = split > 1 ? (int)(grp % split) * (4 * G) : 0;
        const int cend = split > 1 ? min(cbeg + 4 * G, c) : c;
        for (int c0 = cbeg; c0 < cend; c0 += 4 * G) {
            const int ch = c0 + 4 * j;
            if (!(sok && ch < c)) continue;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            // a channel of a pooled row takes its maximum from ONE of the h neighbours: of the incoming pairs of s only
            // ~ 1 / h carry a gradient for a given channel.  So the arg-max pieces of four pairs are fetched together and
            // a piece of dy only where one of its four channels matches (same sums, in the same order; half the bytes)
            int p = beg;
            for (; p + 3 < end; p += 4) {
                int q4[4], col4[4];
                int4 a4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pair = t_pairs[p + u];
                    q4[u] = pair / h;
                    col4[u] = pair - q4[u] * h;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) a4[u] = load_arg4(arg + (int64_t)q4[u] * c + ch);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int4 a = a4[u];
                    const int col = col4[u];
                    if (a.x == col || a.y == col || a.z == col || a.w == col) {
                        const float4 g = ld4(dy + (int64_t)q4[u] * c + ch);
                        if (a.x == col) acc.x += g.x;
                        if (a.y == col) acc.y += g.y;
                        if (a.z == col) acc.z += g.z;
                        if (a.w == col) acc.w += g.w;
                    }
                }
            }
            for (; p < end; ++p) {
                const int pair = t_pairs[p];
                const int q = pair / h, col = pair - q * h;
                const int4 a = load_arg4(arg + (int64_t)q * c + ch);
                if (a.x == col || a.y == col || a.z == col || a.w == col) {
                    const float4 g = ld4(dy + (int64_t)q * c + ch);
                    if (a.x == col) acc.x += g.x;
                    if (a.y == col) acc.y += g.y;
                    if (a.z == col) acc.z += g.z;
                    if (a.w == col) acc.w += g.w;
                }
            }
            if (add) { const float4 o = ld4(add + s * c + ch); acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
            st4(dx + s * c + ch, acc);
        }
    }
}

// one wave per support row
template <typename T>
__global__ __launch_bounds__(256) void max_pool_bwd_kernel(const T* __restrict__ dy, const int32_t* __restrict__ arg,
                                                            int h, int c, const int32_t* __restrict__ t_offsets,
                                                            const int32_t* __restrict__ t_pairs, int64_t ns,
                                                            T* __restrict__ dx)
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
                    if (arg[(int64_t)q * c + ch] == col) acc += ld1(dy + (int64_t)q * c + ch);
                }
                st1(dx + s * c + ch, acc);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void closest_pool_fwd_kernel(const T* __restrict__ x, int64_t ns, int c,
                                                                const int64_t* __restrict__ inds, int64_t nq, int h,
                                                                T* __restrict__ out)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < nq; q += (int64_t)gridDim.x * 4) {
        const int64_t s = inds[q * h];
        const bool real = s >= 0 && s < ns;
        for (int ch = lane; ch < c; ch += 64) st1(out + q * c + ch, real ? ld1(x + s * c + ch) : 0.0f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void closest_pool_bwd_kernel(const T* __restrict__ dy, int h, int c,
                                                                const int32_t* __restrict__ t_offsets,
                                                                const int32_t* __restrict__ t_pairs, int64_t ns,
                                                                T* __restrict__ dx)
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
                    if (pair - q * h == 0) acc += ld1(dy + (int64_t)q * c + ch);
                }
                st1(dx + s * c + ch, acc);
            }
        }
    }
}


// float rows of c = 4 G channels, G | 64: G lanes x float4 per incoming row, S = 64 / G rows of a support in flight side by side
// (and two trips unrolled), the S partial sums added in a fixed order at the end.  The scalar form above walks a support's
// fine points one dependent 4-byte-per-lane load at a time: 95 us for the 400 000 -> 71 000 level of the DALES step.
template <int G>
__global__ __launch_bounds__(256) void closest_pool_bwd_vec_kernel(const float* __restrict__ dy, int h, int c,
                                                                    const int32_t* __restrict__ t_offsets,
                                                                    const int32_t* __restrict__ t_pairs, int64_t ns,
                                                                    float* __restrict__ dx)
{
    constexpr int S = 64 / G;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane % G, slot = lane / G;
    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < ns; s += (int64_t)gridDim.x * 4) {
        const int beg = t_offsets[s], end = t_offsets[s + 1];
        for (int c0 = 0; c0 < c; c0 += 4 * G) {              // (G < 64: c = 4 G, one trip; G = 64: 256 channels per trip)
            const int ch = c0 + 4 * j;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int p = beg + slot;
            for (; p + S < end; p += 2 * S) {
                const int pa = t_pairs[p], pb = t_pairs[p + S];
                const int qa = pa / h, qb = pb / h;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a;
                if (pa - qa * h == 0) a = *reinterpret_cast<const float4*>(dy + (int64_t)qa * c + ch);
                if (pb - qb * h == 0) b4 = *reinterpret_cast<const float4*>(dy + (int64_t)qb * c + ch);
                acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
                acc.x += b4.x; acc.y += b4.y; acc.z += b4.z; acc.w += b4.w;
            }
            if (p < end) {
                const int pa = t_pairs[p];
                const int qa = pa / h;
                if (pa - qa * h == 0) {
                    const float4 a = *reinterpret_cast<const float4*>(dy + (int64_t)qa * c + ch);
                    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
                }
            }
#pragma unroll
            for (int o = G; o < 64; o <<= 1) {
                acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
                acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
            }
            if (slot == 0) *reinterpret_cast<float4*>(dx + s * c + ch) = acc;
        }
    }
}

// workgroups per XCD of the interleaved assignment (0 = contiguous chunks; A/B switch WEASAL_POOL_INTERLEAVE).  Level-0 max-pool
// of the DALES step (71 000 x 59 rows of 512 bytes): forward 245 -> 195 us in the step, 136 -> 97 us alone (tools/pool_lab.py)
extern "C" int ws_pool_interleave = 256;
extern "C" int ws_pool_unroll = 8;          // neighbour rows in flight per lane of the 128-channel max-pool forward: 8 (level 0 of the DALES step: 184 -> 172 us alone) or 4
extern "C" int ws_pool_split_rows = 8192;   // max-pools over fewer rows than this give every 256-channel chunk of a row a wave of its own (0 = never)
static int pool_split(int64_t rows, int c)
{
    return (ws_pool_split_rows > 0 && rows < ws_pool_split_rows && c > 256) ? (int)ws_ceil_div(c, 256) : 1;
}
extern "C" int ws_closest_bwd_vec = 1;      // nearest-upsampling backward: 1 = float4 lanes, several incoming rows side by side (A/B: WEASAL_CLOSEST_BWD_VEC)
static inline int pool_grid(int64_t groups, int ilv)
{
    if (ilv <= 0) return ws_grid(groups, 4);
    const int64_t need = (groups + 31) / 32;              // never more workgroups per XCD than groups / 4 / 8
    return 8 * (int)std::max<int64_t>(1, std::min<int64_t>(ilv, need));
}

template <typename T>
int max_pool_fwd_impl(const T* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                      T* out, int32_t* arg, void* stream, const int32_t* order = nullptr)
{
    WS_REQUIRE(ns >= 0 && nq >= 0 && c >= 1 && h >= 1, "bad sizes");
    if (nq == 0) return WS_OK;
    WS_REQUIRE(inds && out && (ns == 0 || x), "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (c % 4 == 0) && ws_row_aligned<T>(x) && ws_row_aligned<T>(out) && (!arg || al16p(arg));
    const int ilv = order ? ws_pool_interleave : 0;
    if (vec && c <= 16) max_pool_fwd_vec_kernel<4, T><<<pool_grid(ws_ceil_div(nq, 16), ilv), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg, order, ilv);
    else if (vec && c <= 32) max_pool_fwd_vec_kernel<8, T><<<pool_grid(ws_ceil_div(nq, 8), ilv), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg, order, ilv);
    else if (vec && c <= 64) max_pool_fwd_vec_kernel<16, T><<<pool_grid(ws_ceil_div(nq, 4), ilv), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg, order, ilv);
    else if (vec && c <= 128) max_pool_fwd_vec_kernel<32, T><<<pool_grid(ws_ceil_div(nq, 2), ilv), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg, order, ilv);
    else if (vec) {
        const int sp = pool_split(nq, c);
        max_pool_fwd_vec_kernel<64, T><<<pool_grid(nq * sp, ilv), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg, order, ilv, sp);
    }
    else max_pool_fwd_kernel<T><<<ws_grid(nq, 4), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

template <typename T>
int max_pool_bwd_impl(const T* dy, const int32_t* arg, int64_t nq, int32_t h, int32_t c,
                      const int32_t* t_offsets, const int32_t* t_pairs, int64_t ns, T* dx, void* stream,
                      const int32_t* order = nullptr)
{
    WS_REQUIRE(ns >= 0 && nq >= 0 && c >= 1 && h >= 1, "bad sizes");
    if (ns == 0) return WS_OK;
    WS_REQUIRE(t_offsets && dx && (nq == 0 || (dy && arg && t_pairs)), "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (c % 4 == 0) && ws_row_aligned<T>(dy) && ws_row_aligned<T>(dx) && al16p(arg);
    const int ilv = order ? ws_pool_interleave : 0;
    if (vec && c <= 32) max_pool_bwd_vec_kernel<8, T><<<pool_grid(ws_ceil_div(ns, 8), ilv), 256, 0, st>>>(dy, arg, h, c, t_offsets, t_pairs, ns, dx, order, ilv);
    else if (vec && c <= 64) max_pool_bwd_vec_kernel<16, T><<<pool_grid(ws_ceil_div(ns, 4), ilv), 256, 0, st>>>(dy, arg, h, c, t_offsets, t_pairs, ns, dx, order, ilv);
    else if (vec && c <= 128) max_pool_bwd_vec_kernel<32, T><<<pool_grid(ws_ceil_div(ns, 2), ilv), 256, 0, st>>>(dy, arg, h, c, t_offsets, t_pairs, ns, dx, order, ilv);
    else if (vec) {
        const int sp = pool_split(ns, c);
        max_pool_bwd_vec_kernel<64, T><<<pool_grid(ns * sp, ilv), 256, 0, st>>>(dy, arg, h, c, t_offsets, t_pairs, ns, dx, order, ilv, nullptr, sp);
    }
    else max_pool_bwd_kernel<T><<<ws_grid(ns, 4), 256, 0, st>>>(dy, arg, h, c, t_offsets, t_pairs, ns, dx);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

// the block calls' private form: arg-max record in bytes (h <= 255, c % 4 == 0, aligned rows: the caller checks)
int max_pool_fwd_u8_impl(const float* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h, float* out, uint8_t* arg,
                         const int32_t* order, hipStream_t st)
{
    if (nq == 0) return WS_OK;
    const int ilv = order ? ws_pool_interleave : 0;       // (only with a spatial order is a neighbouring group a neighbouring place)
    if (c <= 16) max_pool_fwd_vec_kernel<4, float, uint8_t><<<pool_grid(ws_ceil_div(nq, 16), ilv), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg, order, ilv);
    else if (c <= 32) max_pool_fwd_vec_kernel<8, float, uint8_t><<<pool_grid(ws_ceil_div(nq, 8), ilv), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg, order, ilv);
    else if (c <= 64) max_pool_fwd_vec_kernel<16, float, uint8_t><<<pool_grid(ws_ceil_div(nq, 4), ilv), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg, order, ilv);
    else if (c <= 128 && ws_pool_unroll == 8) max_pool_fwd_vec_kernel<32, float, uint8_t, 8><<<pool_grid(ws_ceil_div(nq, 2), ilv), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg, order, ilv);
    else if (c <= 128) max_pool_fwd_vec_kernel<32, float, uint8_t><<<pool_grid(ws_ceil_div(nq, 2), ilv), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg, order, ilv);
    else {
        const int sp = pool_split(nq, c);
        max_pool_fwd_vec_kernel<64, float, uint8_t><<<pool_grid(nq * sp, ilv), 256, 0, st>>>(x, ns, c, inds, nq, h, out, arg, order, ilv, sp);
    }
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int max_pool_bwd_u8_impl(const float* dy, const uint8_t* arg, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                         const int32_t* t_pairs, int64_t ns, float* dx, const int32_t* order, hipStream_t st, const float* add = nullptr)
{
    (void)nq;
    if (ns == 0) return WS_OK;
    const int ilv = order ? ws_pool_interleave : 0;       // (only with a spatial order is a neighbouring group a neighbouring place)
    if (c <= 32) max_pool_bwd_vec_kernel<8, float, uint8_t><<<pool_grid(ws_ceil_div(ns, 8), ilv), 256, 0, st>>>(dy, arg, h, c, t_offsets, t_pairs, ns, dx, order, ilv, add);
    else if (c <= 64) max_pool_bwd_vec_kernel<16, float, uint8_t><<<pool_grid(ws_ceil_div(ns, 4), ilv), 256, 0, st>>>(dy, arg, h, c, t_offsets, t_pairs, ns, dx, order, ilv, add);
    else if (c <= 128) max_pool_bwd_vec_kernel<32, float, uint8_t><<<pool_grid(ws_ceil_div(ns, 2), ilv), 256, 0, st>>>(dy, arg, h, c, t_offsets, t_pairs, ns, dx, order, ilv, add);
    else {
        const int sp = pool_split(ns, c);
        max_pool_bwd_vec_kernel<64, float, uint8_t><<<pool_grid(ns * sp, ilv), 256, 0, st>>>(dy, arg, h, c, t_offsets, t_pairs, ns, dx, order, ilv, add, sp);
    }
    WS_LAUNCH_CHECK();
    return WS_OK;
}

template <typename T>
int closest_pool_fwd_impl(const T* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                          T* out, void* stream)
{
    WS_REQUIRE(ns >= 0 && nq >= 0 && c >= 1 && h >= 1, "bad sizes");
    if (nq == 0) return WS_OK;
    WS_REQUIRE(inds && out && (ns == 0 || x), "NULL argument");
    closest_pool_fwd_kernel<T><<<ws_grid(nq, 4), 256, 0, (hipStream_t)stream>>>(x, ns, c, inds, nq, h, out);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

template <typename T>
int closest_pool_bwd_impl(const T* dy, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                          const int32_t* t_pairs, int64_t ns, T* dx, void* stream)
{
    WS_REQUIRE(ns >= 0 && nq >= 0 && c >= 1 && h >= 1, "bad sizes");
    if (ns == 0) return WS_OK;
    WS_REQUIRE(t_offsets && dx && (nq == 0 || (dy && t_pairs)), "NULL argument");
    if constexpr (sizeof(T) == 4) {
        const bool al = ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15u) == 0;
        const float* dyf = reinterpret_cast<const float*>(dy);
        float* dxf = reinterpret_cast<float*>(dx);
        hipStream_t st = (hipStream_t)stream;
#define WS_CPB(GV) closest_pool_bwd_vec_kernel<GV><<<ws_grid(ns, 4), 256, 0, st>>>(dyf, h, c, t_offsets, t_pairs, ns, dxf)
        if (ws_closest_bwd_vec && al && (c == 32 || c == 64 || c == 128 || c % 256 == 0)) {
            if (c == 32) WS_CPB(8);
            else if (c == 64) WS_CPB(16);
            else if (c == 128) WS_CPB(32);
            else WS_CPB(64);
            WS_LAUNCH_CHECK();
            return WS_OK;
        }
#undef WS_CPB
    }
    closest_pool_bwd_kernel<T><<<ws_grid(ns, 4), 256, 0, (hipStream_t)stream>>>(dy, h, c, t_offsets, t_pairs, ns, dx);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

}  // namespace

extern "C" {

int ws_max_pool_fwd(const float* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                    float* out, int32_t* arg, void* stream)
{
    return max_pool_fwd_impl<float>(x, ns, c, inds, nq, h, out, arg, stream);
}

int ws_max_pool_bwd(const float* dy, const int32_t* arg, int64_t nq, int32_t h, int32_t c,
                    const int32_t* t_offsets, const int32_t* t_pairs, int64_t ns, float* dx, void* stream)
{
    return max_pool_bwd_impl<float>(dy, arg, nq, h, c, t_offsets, t_pairs, ns, dx, stream);
}

int ws_max_pool_fwd_ordered(const float* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h, float* out, int32_t* arg,
                            const int32_t* order_q, void* stream)
{
    return max_pool_fwd_impl<float>(x, ns, c, inds, nq, h, out, arg, stream, order_q);
}

int ws_max_pool_bwd_ordered(const float* dy, const int32_t* arg, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                            const int32_t* t_pairs, int64_t ns, float* dx, const int32_t* order_s, void* stream)
{
    return max_pool_bwd_impl<float>(dy, arg, nq, h, c, t_offsets, t_pairs, ns, dx, stream, order_s);
}

int ws_max_pool_fwd_ordered_bf16(const uint16_t* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h, uint16_t* out,
                                 int32_t* arg, const int32_t* order_q, void* stream)
{
    return max_pool_fwd_impl<bf16_t>(reinterpret_cast<const bf16_t*>(x), ns, c, inds, nq, h, reinterpret_cast<bf16_t*>(out), arg, stream,
                                     order_q);
}

int ws_max_pool_bwd_ordered_bf16(const uint16_t* dy, const int32_t* arg, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                                 const int32_t* t_pairs, int64_t ns, uint16_t* dx, const int32_t* order_s, void* stream)
{
    return max_pool_bwd_impl<bf16_t>(reinterpret_cast<const bf16_t*>(dy), arg, nq, h, c, t_offsets, t_pairs, ns,
                                     reinterpret_cast<bf16_t*>(dx), stream, order_s);
}

int ws_closest_pool_fwd(const float* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                        float* out, void* stream)
{
    return closest_pool_fwd_impl<float>(x, ns, c, inds, nq, h, out, stream);
}

int ws_closest_pool_bwd(const float* dy, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                        const int32_t* t_pairs, int64_t ns, float* dx, void* stream)
{
    return closest_pool_bwd_impl<float>(dy, nq, h, c, t_offsets, t_pairs, ns, dx, stream);
}

// bf16 feature rows (BASELINE config 5): same kernels, 8-byte row pieces
int ws_max_pool_fwd_bf16(const uint16_t* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                         uint16_t* out, int32_t* arg, void* stream)
{
    return max_pool_fwd_impl<bf16_t>(reinterpret_cast<const bf16_t*>(x), ns, c, inds, nq, h, reinterpret_cast<bf16_t*>(out), arg, stream);
}

int ws_max_pool_bwd_bf16(const uint16_t* dy, const int32_t* arg, int64_t nq, int32_t h, int32_t c,
                         const int32_t* t_offsets, const int32_t* t_pairs, int64_t ns, uint16_t* dx, void* stream)
{
    return max_pool_bwd_impl<bf16_t>(reinterpret_cast<const bf16_t*>(dy), arg, nq, h, c, t_offsets, t_pairs, ns,
                                     reinterpret_cast<bf16_t*>(dx), stream);
}

int ws_closest_pool_fwd_bf16(const uint16_t* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h,
                             uint16_t* out, void* stream)
{
    return closest_pool_fwd_impl<bf16_t>(reinterpret_cast<const bf16_t*>(x), ns, c, inds, nq, h, reinterpret_cast<bf16_t*>(out), stream);
}

int ws_closest_pool_bwd_bf16(const uint16_t* dy, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                             const int32_t* t_pairs, int64_t ns, uint16_t* dx, void* stream)
{
    return closest_pool_bwd_impl<bf16_t>(reinterpret_cast<const bf16_t*>(dy), nq, h, c, t_offsets, t_pairs, ns,
                                         reinterpret_cast<bf16_t*>(dx), stream);
}

// private to the library (declared in ws_common.h, not part of include/weasal_hip.h): the block calls keep their arg-max
// record in bytes.  Preconditions (checked by the caller): h <= 255, c % 4 == 0, 16-byte aligned rows, 4-byte aligned arg.
int ws_priv_max_pool_fwd_u8(const float* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h, float* out,
                            uint8_t* arg, const int32_t* order_q, void* stream)
{
    return max_pool_fwd_u8_impl(x, ns, c, inds, nq, h, out, arg, order_q, (hipStream_t)stream);
}

int ws_priv_max_pool_bwd_u8(const float* dy, const uint8_t* arg, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                            const int32_t* t_pairs, int64_t ns, float* dx, const int32_t* order_s, const float* add, void* stream)
{
    return max_pool_bwd_u8_impl(dy, arg, nq, h, c, t_offsets, t_pairs, ns, dx, order_s, (hipStream_t)stream, add);
}

}  // extern "C"
