// weasal_amd/csrc/contrast_mfma.hip -- the [N, slc_con] part of KPFCNN.contrast_loss (models/architectures.py:455-497) on the
// matrix core (round 3; contrast.hip holds the VALU form and the description of the arithmetic, which is unchanged).
//
// The similarities  mul = O (N x C) . S^T (C x 1000) / T  are a dense product; so are both gradients,
//     d O = W (N x 1000) . S (1000 x C) / T        d S = W^T (1000 x N) . O (N x C) / T,
// with W = d loss / d mul evaluated element-wise from the similarities and the row statistics.  The VALU form spends
// 12 + 12 fused multiply-adds and ~12 other instructions per (point, slice column) in each of three kernels; here a
// 16 x 16 tile of similarities costs 3-4 v_mfma_f32_16x16x4_f32 (C <= 16 padded to a multiple of 4), the element-wise part
// runs on the 4 values a lane holds of it, and the two gradient products take the tile of W back in as an MFMA operand
// (through a 1 KB LDS scratch per wave: the D layout of one product is not the A / B layout of the next).
//   forward    point tiles outer, slice tiles inner; the slice table (rows + index / tag) in LDS; one pass (see contrast.hip)
//   backward   ONE kernel for both gradients: slice tiles outer, the wave's 8 point tiles (128 points, staged in LDS)
//              inner; d O accumulates in registers across the outer loop, d S per slice tile is summed over the four
//              waves through LDS and written as the workgroup's partial (fixed-order reduction afterwards: deterministic).
#include "ws_common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int CM_SMAX = 1024;          // slice columns supported (the reference uses 1000)
constexpr int CM_CP = 17;              // channel pitch of the staged rows (C <= 16; odd: row- and column-wise operand reads both spread over the banks)

__device__ __forceinline__ void lds_order()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---- forward -------------------------------------------------------------------------------------------------------
constexpr int CF_TILES = 2;            // point tiles (of 16) per wave (8: 371 us, 4: 358 us, 2: 318 us at N = 400 000: more workgroups hide more latency)
constexpr int CF_WAVES = 8;            // waves per workgroup: they share one 56 KB slice table (two workgroups per CU = 4 waves per SIMD)

template <int NS>                      // NS = ceil(C / 4) MFMA steps per tile
__global__ __launch_bounds__(512) void contrast_fwd_mfma_kernel(const float* __restrict__ on, int64_t n, int c,
                                                                 const float* __restrict__ xs, int s,
                                                                 const int64_t* __restrict__ slc_idx, const uint8_t* __restrict__ certain,
                                                                 const int64_t* __restrict__ lbl, float temperature, float eps,
                                                                 float* __restrict__ loss, float* __restrict__ rowmax,
                                                                 float* __restrict__ den, float* __restrict__ npos)
{
    __shared__ float sx[CM_SMAX][4 * NS];
    __shared__ int sidx[CM_SMAX], stag[CM_SMAX];
    const int ncol = (s + 15) & ~15;
    for (int e = threadIdx.x; e < ncol * 4 * NS; e += 512) {
        const int j = e / (4 * NS), cc = e % (4 * NS);
        sx[j][cc] = (j < s && cc < c) ? xs[(int64_t)j * c + cc] : 0.0f;
    }
    for (int j = threadIdx.x; j < ncol; j += 512) {
        const int64_t p = j < s ? slc_idx[j] : 0;
        sidx[j] = j < s ? (int)p : -1;
        stag[j] = j < s ? (((int)lbl[p] << 1) | (certain[p] ? 1 : 0)) : -2;      // -2: a padding column (never usable)
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 15, kk = lane >> 4;              // A: point i of the tile; B / D: column j = i, D: points 4 kk + r
    const float inv_t = 1.0f / temperature;
    for (int t = 0; t < CF_TILES; ++t) {
        const int64_t p0 = (((int64_t)blockIdx.x * CF_WAVES + wave) * CF_TILES + t) * 16;
        if (p0 >= n) break;
        float a[NS];
        {
            const int64_t pi = p0 + i < n ? p0 + i : n - 1;
#pragma unroll
            for (int st = 0; st < NS; ++st) a[st] = (4 * st + kk) < c ? on[pi * c + 4 * st + kk] : 0.0f;
        }
        int ptag[4], pidx[4];
        float m[4], E[4], P[4], S[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t p = p0 + 4 * kk + r;
            const int64_t pc = p < n ? p : n - 1;
            pidx[r] = (int)p;
            ptag[r] = ((int)lbl[pc] << 1) | (certain[pc] ? 1 : 0);
            m[r] = -3.0e38f; E[r] = 0.0f; P[r] = 0.0f; S[r] = 0.0f;
        }
        for (int c0 = 0; c0 < ncol; c0 += 16) {
            f32x4v d = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < NS; ++st) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st], sx[c0 + i][4 * st + kk], d, 0, 0, 0);
            const int tg = stag[c0 + i], sid = sidx[c0 + i];
            const bool colok = tg != -2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float mul = d[r] * inv_t;
                m[r] = colok ? fmaxf(m[r], mul) : m[r];
                const bool use = colok && sid != pidx[r] && ((tg ^ ptag[r]) & 1) == 0;
                const bool pos = use && tg == ptag[r];
                E[r] += use ? __expf(mul - inv_t) : 0.0f;
                P[r] += pos ? 1.0f : 0.0f;
                S[r] += pos ? mul : 0.0f;
            }
        }
        // the 16 lanes of a group hold the partial statistics of the same 4 points over different columns
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                m[r] = fmaxf(m[r], __shfl_xor(m[r], o, 64));
                E[r] += __shfl_xor(E[r], o, 64);
                P[r] += __shfl_xor(P[r], o, 64);
                S[r] += __shfl_xor(S[r], o, 64);
            }
            const int64_t p = p0 + 4 * kk + r;
            if (i == 0 && p < n) {
                const float Er = E[r] * __expf(inv_t - m[r]);         // rebased on the true maximum (contrast.hip)
                const float dd = Er + eps;
                loss[p] = -temperature * (((S[r] - P[r] * m[r]) - P[r] * logf(dd)) / (P[r] + 1e-12f));
                rowmax[p] = m[r];
                den[p] = dd;
                npos[p] = P[r];
            }
        }
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------
constexpr int CB_TILES = 4;            // point tiles per wave: 64 points per wave, 256 per workgroup (8: 771 us, 4: 561 us + 8 us more reduction)
constexpr int CB_WLD = 17;             // pitch of the W scratch tile

template <int NS>
__global__ __launch_bounds__(256) void contrast_bwd_mfma_kernel(const float* __restrict__ on, int64_t n, int c,
                                                                 const float* __restrict__ xs, int s,
                                                                 const int64_t* __restrict__ slc_idx, const uint8_t* __restrict__ certain,
                                                                 const int64_t* __restrict__ lbl, float temperature,
                                                                 const float* __restrict__ rowmax, const float* __restrict__ den,
                                                                 const float* __restrict__ npos, const float* __restrict__ g,
                                                                 float* __restrict__ d_on, float* __restrict__ partial)
{
    __shared__ float po_all[4][16 * CB_TILES][CM_CP];      // the wave's points (rows padded to 16 channels)
    __shared__ float4 pst_all[4][16 * CB_TILES];            // (rowmax, npos, 1 / den, g * (-T / (P + 1e-12)) / T) per point
    __shared__ int ptg_all[4][16 * CB_TILES];               // tag per point (-4: beyond n)
    __shared__ float wt_all[4][16 * CB_WLD];                // W tile [point][column]
    __shared__ float ds_all[4][16][16];                     // d S tile of every wave [channel][column]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 15, kk = lane >> 4;
    float (*po)[CM_CP] = po_all[wave];
    float4* pst = pst_all[wave];
    int* ptg = ptg_all[wave];
    float* wt = wt_all[wave];
    const int64_t pw0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * CB_TILES);
    const float inv_t = 1.0f / temperature;
    for (int e = lane; e < 16 * CB_TILES * CM_CP; e += 64) {
        const int r = e / CM_CP, cc = e % CM_CP;
        const int64_t p = pw0 + r;
        po[r][cc] = (p < n && cc < c) ? on[p * c + cc] : 0.0f;
    }
    for (int r = lane; r < 16 * CB_TILES; r += 64) {
        const int64_t p = pw0 + r;
        const bool live = p < n;
        const float P = live ? npos[p] : 0.0f;
        pst[r] = make_float4(live ? rowmax[p] : 0.0f, P, live ? 1.0f / den[p] : 0.0f,
                             live ? g[p] * (-temperature / (P + 1e-12f)) / temperature : 0.0f);
        ptg[r] = live ? (((int)lbl[p] << 1) | (certain[p] ? 1 : 0)) : -4;
    }
    lds_order();
    f32x4v don[CB_TILES];
#pragma unroll
    for (int t = 0; t < CB_TILES; ++t) don[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int ncol = (s + 15) & ~15;
    for (int c0 = 0; c0 < ncol; c0 += 16) {
        // this slice tile's operands: B of the similarity product (lane (j, kk): xs[c0 + j][4 st + kk]), B of d O = W . S
        // (lane (j = channel, kk): xs[c0 + 4 st + kk][j]), the column's index / tag
        float b1[NS], b2[4];
        const int col = c0 + i;
        const int colc = col < s ? col : s - 1;
#pragma unroll
        for (int st = 0; st < NS; ++st) b1[st] = (col < s && 4 * st + kk < c) ? xs[(int64_t)colc * c + 4 * st + kk] : 0.0f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int cj = c0 + 4 * st + kk;
            b2[st] = (cj < s && i < c) ? xs[(int64_t)cj * c + i] : 0.0f;
        }
        const int64_t sp = slc_idx[colc];
        const int sid = col < s ? (int)sp : -1;
        const int tg = col < s ? (((int)lbl[sp] << 1) | (certain[sp] ? 1 : 0)) : -2;
        f32x4v dsa = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < CB_TILES; ++t) {
            f32x4v d = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < NS; ++st) d = __builtin_amdgcn_mfma_f32_16x16x4f32(po[16 * t + i][4 * st + kk], b1[st], d, 0, 0, 0);
            lds_order();                                         // the previous tile's readers of wt are done
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pr = 16 * t + 4 * kk + r;
                const float4 st4 = pst[pr];
                const int ptag = ptg[pr];
                const float lg = d[r] * inv_t - st4.x;
                const bool use = tg != -2 && ptag != -4 && sid != (int)(pw0 + pr) && ((tg ^ ptag) & 1) == 0;
                const bool pos = use && tg == ptag;
                const float w = st4.w * ((pos ? 1.0f : 0.0f) - (use ? st4.y * __expf(lg) * st4.z : 0.0f));
                wt[(4 * kk + r) * CB_WLD + i] = w;               // [point][column]
            }
            lds_order();
            // d O tile += W (16 points x 16 columns) . S (16 columns x C):  A lane (point i, kk) = W[i][4 st + kk]
            // d S^T tile += O^T (C x 16 points) . W:  A lane (channel i, kk) = O[4 st + kk][i],  B lane (column j, kk) = W[4 st + kk][j]
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                don[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[i * CB_WLD + 4 * st + kk], b2[st], don[t], 0, 0, 0);
                dsa = __builtin_amdgcn_mfma_f32_16x16x4f32(po[16 * t + 4 * st + kk][i], wt[(4 * st + kk) * CB_WLD + i], dsa, 0, 0, 0);
            }
        }
        // the four waves' d S tiles of this slice tile: summed in a fixed order, one partial per workgroup
#pragma unroll
        for (int r = 0; r < 4; ++r) ds_all[wave][4 * kk + r][i] = dsa[r];       // D: lane (column i, kk): channels 4 kk + r
        __syncthreads();
        {
            const int ch = threadIdx.x >> 4, cj = threadIdx.x & 15;                 // 256 threads = 16 channels x 16 columns
            if (ch < c && c0 + cj < s)
                partial[((int64_t)blockIdx.x * s + c0 + cj) * c + ch] = (ds_all[0][ch][cj] + ds_all[1][ch][cj]) + (ds_all[2][ch][cj] + ds_all[3][ch][cj]);
        }
        __syncthreads();
    }
    // d O: lane (channel i, kk) holds points 4 kk + r of every tile
#pragma unroll
    for (int t = 0; t < CB_TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t p = pw0 + 16 * t + 4 * kk + r;
            if (p < n && i < c) d_on[p * c + i] = don[t][r];
        }
}

__global__ __launch_bounds__(1024) void contrast_reduce2_kernel(const float* __restrict__ partial, int64_t elems, int chunks,
                                                                 float* __restrict__ out)
{
    // 32 elements x 32 chunk lanes per workgroup: lane cl sums chunks cl, cl + 32, ... in order, then a fixed tree over cl
    __shared__ float red[32][33];
    const int el = threadIdx.x & 31, cl = threadIdx.x >> 5;
    const int64_t e = (int64_t)blockIdx.x * 32 + el;
    float sacc = 0.0f;
    if (e < elems)
        for (int cidx = cl; cidx < chunks; cidx += 32) sacc += partial[(int64_t)cidx * elems + e];
    red[cl][el] = sacc;
    __syncthreads();
    for (int o = 16; o > 0; o >>= 1) {
        if (cl < o) red[cl][el] += red[cl + o][el];
        __syncthreads();
    }
    if (cl == 0 && e < elems) out[e] = red[0][el];
}

}  // namespace

// 2 = matrix-core kernels (this file), 1 = the VALU form of contrast.hip (A/B switch: WEASAL_CONTRAST_VARIANT)
extern "C" int ws_contrast_variant = 2;

extern "C" int ws_contrast_mfma_fwd(const float* on, int64_t n, int32_t c, const float* xs, int32_t s, const int64_t* slc_idx,
                                    const uint8_t* certain, const int64_t* lbl, float temperature, float eps, float* loss,
                                    float* rowmax, float* den, float* npos, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ws_ceil_div(n, CF_WAVES * CF_TILES * 16);
    if (c <= 4) contrast_fwd_mfma_kernel<1><<<grid, 64 * CF_WAVES, 0, st>>>(on, n, c, xs, s, slc_idx, certain, lbl, temperature, eps, loss, rowmax, den, npos);
    else if (c <= 8) contrast_fwd_mfma_kernel<2><<<grid, 64 * CF_WAVES, 0, st>>>(on, n, c, xs, s, slc_idx, certain, lbl, temperature, eps, loss, rowmax, den, npos);
    else if (c <= 12) contrast_fwd_mfma_kernel<3><<<grid, 64 * CF_WAVES, 0, st>>>(on, n, c, xs, s, slc_idx, certain, lbl, temperature, eps, loss, rowmax, den, npos);
    else contrast_fwd_mfma_kernel<4><<<grid, 64 * CF_WAVES, 0, st>>>(on, n, c, xs, s, slc_idx, certain, lbl, temperature, eps, loss, rowmax, den, npos);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

extern "C" int64_t ws_contrast_mfma_bwd_scratch_bytes(int64_t n, int32_t c, int32_t s)
{
    return ws_ceil_div(n > 0 ? n : 1, 4 * CB_TILES * 16) * (int64_t)s * c * (int64_t)sizeof(float);
}

extern "C" int ws_contrast_mfma_bwd(const float* on, int64_t n, int32_t c, const float* xs, int32_t s, const int64_t* slc_idx,
                                    const uint8_t* certain, const int64_t* lbl, float temperature, const float* rowmax,
                                    const float* den, const float* npos, const float* g, float* d_on, float* d_xs, void* scratch,
                                    void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int chunks = (int)ws_ceil_div(n, 4 * CB_TILES * 16);
    float* partial = chunks == 1 ? d_xs : (float*)scratch;
#define WS_CBM(NSV) contrast_bwd_mfma_kernel<NSV><<<chunks, 256, 0, st>>>(on, n, c, xs, s, slc_idx, certain, lbl, temperature, rowmax, den, npos, g, d_on, partial)
    if (c <= 4) WS_CBM(1);
    else if (c <= 8) WS_CBM(2);
    else if (c <= 12) WS_CBM(3);
    else WS_CBM(4);
#undef WS_CBM
    WS_LAUNCH_CHECK();
    if (chunks > 1) {
        const int64_t elems = (int64_t)s * c;
        contrast_reduce2_kernel<<<(unsigned)ws_ceil_div(elems, 32), 1024, 0, st>>>(partial, elems, chunks, d_xs);
        WS_LAUNCH_CHECK();
    }
    return WS_OK;
}
