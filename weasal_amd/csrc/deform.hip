// weasal_amd/csrc/deform.hip -- the element-wise ends of a deformable KPConv (gfx950).
//
// Reference: models/blocks.py:244-267, 287-291 (offset features -> offsets, modulations, deformed kernel points) and
// models/architectures.py:24-57 (p2p_fitting_regularizer).  The reference runs them as ~10 + ~25 torch ops per layer and
// as many autograd nodes; its regulariser materialises 15 x [N, 14, 3] difference tensors per layer.  Here:
//   ws_kpconv_deform_prepare / _bwd   offset features [N, 3K (+K)] -> deformed_kp [N,K,3], modulations [N,K] and the packed
//                                     operand kp4 [N,K] float4 = (x, y, z, modulation) the MODE-2 gather kernels read
//   ws_p2p_regularizer_fwd / _bwd     fitting + repulsive loss of one layer and their gradients, thread = point, the 15
//                                     kernel points of the point in registers, fixed-order reduction (bit-reproducible)
#include "ws_common.h"

namespace {

constexpr int KP = 15;

__global__ __launch_bounds__(256) void deform_prepare_kernel(const float* __restrict__ off, int64_t n, int32_t od,
                                                             const float* __restrict__ kernel_points, float extent, int modulated,
                                                             float* __restrict__ deformed_kp, float* __restrict__ modulations,
                                                             float4* __restrict__ kp4, int* __restrict__ rmax_bits)
{
#pragma clang fp contract(off)
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;            // (point, kernel point)
    const bool tail = e >= n * KP;                                  // (tail lanes redo the last element: the wave reduction
    if (tail) e = n * KP - 1;                                       //  below needs every lane; identical values are rewritten)
    const int64_t p = e / KP;
    const int k = (int)(e - p * KP);
    const float* row = off + p * od;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // offsets = unscaled * KP_extent (blocks.py:267); deformed_KP = offsets + kernel_points (:288): two roundings, no FMA
        // (the pragma above: __fmul_rn / __fadd_rn alone do not stop the contraction)
        const float o = row[3 * k + c] * extent;
        v[c] = o + kernel_points[3 * k + c];
    }
    float m = 1.0f;
    if (modulated) m = 2.0f * (1.0f / (1.0f + __expf(-row[3 * KP + k])));      // 2 * sigmoid (blocks.py:256)
    if (deformed_kp) { deformed_kp[3 * e] = v[0]; deformed_kp[3 * e + 1] = v[1]; deformed_kp[3 * e + 2] = v[2]; }
    if (modulations && modulated) modulations[e] = m;
    kp4[e] = make_float4(v[0], v[1], v[2], m);
    if (rmax_bits) {
        // largest |kp| of the launch (non-negative floats order like their bit patterns): one atomic per wave
        float r = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) r = fmaxf(r, __shfl_xor(r, o, 64));
        // (a stale read only costs an atomic that changes nothing: the word grows monotonically; without the test
        //  94 000 waves of a level-0 launch queue on one address -- 1 ms)
        if ((threadIdx.x & 63) == 0 && __float_as_int(r) > *(volatile int*)rmax_bits) atomicMax(rmax_bits, __float_as_int(r));
    }
}

// d offset_features from d kp4 (the gather kernels' geometry gradient: xyz and modulation) and, optionally, a second
// gradient of the kernel-point positions (the regulariser's)
__global__ __launch_bounds__(256) void deform_prepare_bwd_kernel(const float4* __restrict__ d_kp4, const float* __restrict__ d_dkp,
                                                                 const float4* __restrict__ kp4, int64_t n, int32_t od, float extent,
                                                                 int modulated, float* __restrict__ d_off)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * KP) return;
    const int64_t p = e / KP;
    const int k = (int)(e - p * KP);
    float4 g = d_kp4 ? d_kp4[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (d_dkp) { g.x += d_dkp[3 * e]; g.y += d_dkp[3 * e + 1]; g.z += d_dkp[3 * e + 2]; }
    float* row = d_off + p * od;
    row[3 * k + 0] = g.x * extent;
    row[3 * k + 1] = g.y * extent;
    row[3 * k + 2] = g.z * extent;
    if (modulated) {
        const float m = kp4[e].w;                       // m = 2 s  ->  d m / d logit = 2 s (1 - s) = m (1 - m / 2)
        row[3 * KP + k] = g.w * m * (1.0f - 0.5f * m);
    }
}

// ---- regulariser ----------------------------------------------------------------------------------------------------
// fitting   = mean_{n,k} |min_d2[n,k] / extent^2|                                   (architectures.py:36-42)
// repulsive = sum_i mean_n | sum_{j != i} min(|loc_i - loc_j| - repulse_extent, 0)^2 | / K,  loc = deformed_kp / extent,
//             the other points detached (:45-51)
// one thread per point; per-workgroup partial sums, added in a fixed order by ws_p2p_regularizer_fwd's final kernel.
__device__ __forceinline__ void load_locs(const float* __restrict__ dkp, const float4* __restrict__ kp4, int64_t p, float inv_extent,
                                          float (&lx)[KP], float (&ly)[KP], float (&lz)[KP])
{
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (kp4) {
            const float4 v = kp4[p * KP + k];
            lx[k] = v.x * inv_extent; ly[k] = v.y * inv_extent; lz[k] = v.z * inv_extent;
        } else {
            const float* v = dkp + (p * KP + k) * 3;
            lx[k] = v[0] * inv_extent; ly[k] = v[1] * inv_extent; lz[k] = v[2] * inv_extent;
        }
    }
}

__global__ __launch_bounds__(256) void p2p_reg_fwd_kernel(const float* __restrict__ dkp, const float4* __restrict__ kp4,
                                                          const float* __restrict__ min_d2, int64_t n, float extent,
                                                          float repulse_extent, float* __restrict__ partial /*[blocks][2]*/)
{
    __shared__ float red[2][4];
    const float inv_extent = 1.0f / extent, inv_e2 = 1.0f / (extent * extent);
    float fit = 0.0f, rep = 0.0f;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
        float lx[KP], ly[KP], lz[KP];
        load_locs(dkp, kp4, p, inv_extent, lx, ly, lz);
#pragma unroll
        for (int k = 0; k < KP; ++k) fit += fabsf(min_d2[p * KP + k] * inv_e2);
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            float ri = 0.0f;
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                if (j == i) continue;
                const float dx = lx[j] - lx[i], dy = ly[j] - ly[i], dz = lz[j] - lz[i];
                const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
                const float c = fminf(d - repulse_extent, 0.0f);
                ri += c * c;
            }
            rep += fabsf(ri);
        }
    }
    fit = ws_wave_sum(fit);
    rep = ws_wave_sum(rep);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = fit; red[1][wave] = rep; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void p2p_reg_final_kernel(const float* __restrict__ partial, int blocks, int64_t n,
                                                            float* __restrict__ out /*[2]: fitting, repulsive*/)
{
    __shared__ double red[2][4];
    double fit = 0.0, rep = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 256) { fit += partial[2 * b]; rep += partial[2 * b + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { fit += __shfl_xor(fit, o, 64); rep += __shfl_xor(rep, o, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = fit; red[1][wave] = rep; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nk = (double)n * KP;
        out[0] = (float)(((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / nk);
        out[1] = (float)(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / nk);
    }
}

// gradients: d fitting / d min_d2 = sign(min_d2) / (extent^2 N K) * g_fit;
//            d repulsive / d deformed_kp[n,i] = g_rep / (N K extent) * sign(rep_i) * sum_{j != i} 2 c_ij (loc_i - loc_j) / d_ij
// g = (g_fit, g_rep) are DEVICE scalars (the upstream gradient times the loss weights): no host synchronisation.
__global__ __launch_bounds__(256) void p2p_reg_bwd_kernel(const float* __restrict__ dkp, const float4* __restrict__ kp4,
                                                          const float* __restrict__ min_d2, int64_t n, float extent,
                                                          float repulse_extent, const float* __restrict__ g,
                                                          float* __restrict__ d_min_d2, float* __restrict__ d_dkp)
{
    const float inv_extent = 1.0f / extent, inv_e2 = 1.0f / (extent * extent);
    const float nk = (float)((double)n * KP);
    const float g_fit = g[0] * inv_e2 / nk, g_rep = g[1] * inv_extent / nk;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
        float lx[KP], ly[KP], lz[KP];
        load_locs(dkp, kp4, p, inv_extent, lx, ly, lz);
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const float v = min_d2[p * KP + k];
            d_min_d2[p * KP + k] = v > 0.0f ? g_fit : (v < 0.0f ? -g_fit : 0.0f);
        }
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            float gx = 0.0f, gy = 0.0f, gz = 0.0f;
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                if (j == i) continue;
                const float dx = lx[i] - lx[j], dy = ly[i] - ly[j], dz = lz[i] - lz[j];
                const float d2 = (dx * dx + dy * dy) + dz * dz;
                const float d = sqrtf(d2);
                const float c = fminf(d - repulse_extent, 0.0f);
                const float f = d > 0.0f ? 2.0f * c / d : 0.0f;
                gx = fmaf(f, dx, gx); gy = fmaf(f, dy, gy); gz = fmaf(f, dz, gz);
            }
            float* o = d_dkp + (p * KP + i) * 3;      // rep_i >= 0: its |.| has derivative 1 wherever a term is live, 0 otherwise
            o[0] = g_rep * gx; o[1] = g_rep * gy; o[2] = g_rep * gz;
        }
    }
}

}  // namespace

extern "C" {

int ws_kpconv_deform_prepare(const float* offset_features, int64_t n, int32_t od, const float* kernel_points, int32_t k,
                             float extent, int32_t modulated, float* deformed_kp, float* modulations, float* kp4, float* kp_rmax,
                             void* stream)
{
    if (k != KP) return ws_fail(WS_ERR_UNSUPPORTED, "num_kernel_points=%d: this build instantiates K=15 only", k);
    WS_REQUIRE(n >= 0 && od == (modulated ? 4 : 3) * KP, "offset features must have %d columns (got %d)", (modulated ? 4 : 3) * KP, od);
    if (n == 0) return WS_OK;
    WS_REQUIRE(offset_features && kernel_points && kp4 && ((uintptr_t)kp4 & 15u) == 0, "NULL / unaligned argument");
    if (kp_rmax) WS_HIP(hipMemsetAsync(kp_rmax, 0, sizeof(float), (hipStream_t)stream));
    deform_prepare_kernel<<<(unsigned)ws_ceil_div(n * KP, 256), 256, 0, (hipStream_t)stream>>>(
        offset_features, n, od, kernel_points, extent, modulated, deformed_kp, modulations, reinterpret_cast<float4*>(kp4),
        reinterpret_cast<int*>(kp_rmax));
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_kpconv_deform_prepare_bwd(const float* d_kp4, const float* d_deformed_kp, const float* kp4, int64_t n, int32_t od, int32_t k,
                                 float extent, int32_t modulated, float* d_offset_features, void* stream)
{
    if (k != KP) return ws_fail(WS_ERR_UNSUPPORTED, "num_kernel_points=%d: this build instantiates K=15 only", k);
    WS_REQUIRE(n >= 0 && od == (modulated ? 4 : 3) * KP, "offset features must have %d columns (got %d)", (modulated ? 4 : 3) * KP, od);
    if (n == 0) return WS_OK;
    WS_REQUIRE(kp4 && d_offset_features && (d_kp4 || d_deformed_kp), "NULL argument");
    deform_prepare_bwd_kernel<<<(unsigned)ws_ceil_div(n * KP, 256), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(d_kp4), d_deformed_kp, reinterpret_cast<const float4*>(kp4), n, od, extent, modulated,
        d_offset_features);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

static int reg_blocks(int64_t n) { return ws_grid(n, 256, 1024); }

int64_t ws_p2p_regularizer_scratch_bytes(int64_t n) { return (int64_t)reg_blocks(n) * 2 * (int64_t)sizeof(float); }

int ws_p2p_regularizer_fwd(const float* deformed_kp, const float* kp4, const float* min_d2, int64_t n, int32_t k, float extent,
                           float repulse_extent, float* out2, void* scratch, void* stream)
{
    if (k != KP) return ws_fail(WS_ERR_UNSUPPORTED, "num_kernel_points=%d: this build instantiates K=15 only", k);
    WS_REQUIRE(n >= 1 && (deformed_kp || kp4) && min_d2 && out2 && scratch && extent > 0.0f, "NULL argument / empty layer");
    const int blocks = reg_blocks(n);
    hipStream_t st = (hipStream_t)stream;
    p2p_reg_fwd_kernel<<<blocks, 256, 0, st>>>(deformed_kp, reinterpret_cast<const float4*>(kp4), min_d2, n, extent, repulse_extent,
                                               (float*)scratch);
    WS_LAUNCH_CHECK();
    p2p_reg_final_kernel<<<1, 256, 0, st>>>((const float*)scratch, blocks, n, out2);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_p2p_regularizer_bwd(const float* deformed_kp, const float* kp4, const float* min_d2, int64_t n, int32_t k, float extent,
                           float repulse_extent, const float* g2, float* d_min_d2, float* d_deformed_kp, void* stream)
{
    if (k != KP) return ws_fail(WS_ERR_UNSUPPORTED, "num_kernel_points=%d: this build instantiates K=15 only", k);
    WS_REQUIRE(n >= 1 && (deformed_kp || kp4) && min_d2 && g2 && d_min_d2 && d_deformed_kp && extent > 0.0f, "NULL argument / empty layer");
    p2p_reg_bwd_kernel<<<reg_blocks(n), 256, 0, (hipStream_t)stream>>>(deformed_kp, reinterpret_cast<const float4*>(kp4), min_d2, n, extent,
                                                                       repulse_extent, g2, d_min_d2, d_deformed_kp);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

}  // extern "C"
