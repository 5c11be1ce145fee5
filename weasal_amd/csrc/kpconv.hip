// weasal_amd/csrc/kpconv.hip -- fused KPConv gather kernels for gfx950 (wave64, LDS-staged rows).
//
// Reference semantics: models/blocks.py:238-374 (KPConv.forward) and its autograd.
//
// K3  kpconv_gather_fwd :  wf[q,k,c] = sum_h w(q,h,k) * x[inds[q,h], c]
//     One wave owns one query at a time.  Phase 1: lane = neighbour column; each lane loads its
//     index and neighbour xyz and evaluates the K kernel-point influences in registers.  Phase 2:
//     the feature rows of the neighbours that have any influence are staged in the wave's LDS
//     slab with 16-byte accesses (a row is contiguous in HBM, 4*ci bytes).  Phase 3: lane =
//     (entry slot, channel); for every kernel point the non-zero influences are enumerated from a
//     wave ballot (ctz over the mask, weights broadcast by v_readlane) and accumulated from LDS.
//     The [N,H,K,3] / [N,H,K] / [N,H,Ci] intermediates of the reference never exist.
//     With `linear` influence only ~1 of the 15 kernel points is non-zero per neighbour, so the
//     accumulate runs over the sparse entries instead of the dense 15 x H matrix.
// K4  kpconv_gather_bwd_x : dx[s,c] = sum_{(q,h)->s} sum_k w * dwf[q,k,c]   (transposed table, no atomics)
// K6  kpconv_gather_bwd_geom : d deformed_kp, d modulations (deformable only)
#include "ws_common.h"

namespace {

constexpr float WS_SHADOW = 1e6f;

struct GeomParams {
    float extent;
    int influence;
    int aggregation;
    int deformable;   // 1: apply the in-range filter of blocks.py:301-325
};

// Influence of the K kernel points on one neighbour offset n = s - q.  kp is wave-uniform.
// Returns w[K] (0 where no influence) and d2[K].
template <int K>
__device__ __forceinline__ void kp_influence(float nx, float ny, float nz, const float* __restrict__ kp,
                                             const GeomParams& g, bool live, float (&w)[K], float (&d2)[K])
{
    float best = 3.4e38f;
    int arg = 0;
    bool inrange = false;
    const float e2 = g.extent * g.extent;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float dx = nx - kp[3 * k + 0];
        const float dy = ny - kp[3 * k + 1];
        const float dz = nz - kp[3 * k + 2];
        const float d = (dx * dx + dy * dy) + dz * dz;
        d2[k] = d;
        if (d < best) { best = d; arg = k; }
        inrange |= d < e2;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float v;
        if (g.influence == WS_INFLUENCE_LINEAR) {
            v = fmaxf(1.0f - sqrtf(d2[k]) / g.extent, 0.0f);
        } else if (g.influence == WS_INFLUENCE_CONSTANT) {
            v = 1.0f;
        } else {
            const float sig = g.extent * 0.3f;
            v = expf(-d2[k] / (2.0f * sig * sig + 1e-9f));
        }
        if (g.aggregation == WS_AGGREGATION_CLOSEST && k != arg) v = 0.0f;
        if (!live || (g.deformable && !inrange)) v = 0.0f;
        w[k] = v;
    }
}

__device__ __forceinline__ void wave_lds_sync()
{
    // the slab is private to one wave: order the wave's own LDS writes before its reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------
// K3 forward, ci >= 5.  CC = channel chunk (16 or 32), ES = 64/CC entry slots.
// ---------------------------------------------------------------------------------------------
template <int K, int CC>
__global__ __launch_bounds__(256) void kpconv_gather_fwd_kernel(
    const float* __restrict__ q_pts, int64_t nq, const float* __restrict__ s_pts, int64_t ns,
    const int64_t* __restrict__ inds, int h, const float* __restrict__ x, int ci,
    const float* __restrict__ kernel_points, const float* __restrict__ deformed_kp,
    const float* __restrict__ modulations, GeomParams g, float* __restrict__ wf,
    float* __restrict__ min_d2, int vec4)
{
    constexpr int ES = 64 / CC;
    constexpr int PIECES = CC / 4;            // 16-byte pieces per staged row
    constexpr int ROWS_PER_I = 64 / PIECES;   // rows staged per wave-instruction
    __shared__ __attribute__((aligned(16))) float slab_all[4][64 * CC];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    float* slab = slab_all[wave];
    const int c = lane % CC;       // channel inside the chunk
    const int slot = lane / CC;    // entry slot

    for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < nq; q += (int64_t)gridDim.x * 4) {
        const float qx = q_pts[3 * q + 0], qy = q_pts[3 * q + 1], qz = q_pts[3 * q + 2];
        const float* kp = deformed_kp ? deformed_kp + q * (3 * K) : kernel_points;
        float mind[K];
        if (min_d2) {
#pragma unroll
            for (int k = 0; k < K; ++k) mind[k] = 3.4e38f;
        }
        for (int cc0 = 0; cc0 < ci; cc0 += CC) {
            float acc[K];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = 0.0f;
            for (int h0 = 0; h0 < h; h0 += 64) {
                // ---- phase 1: lane = neighbour column
                const int col = h0 + lane;
                const bool incol = col < h;
                int64_t idx = incol ? inds[q * h + col] : ns;
                const bool real = incol && idx < ns && idx >= 0;
                float px = WS_SHADOW, py = WS_SHADOW, pz = WS_SHADOW;
                if (real) { px = s_pts[3 * idx]; py = s_pts[3 * idx + 1]; pz = s_pts[3 * idx + 2]; }
                float w[K], d2[K];
                kp_influence<K>(px - qx, py - qy, pz - qz, kp, g, real, w, d2);
                if (min_d2 && cc0 == 0) {
#pragma unroll
                    for (int k = 0; k < K; ++k) if (incol) mind[k] = fminf(mind[k], d2[k]);
                }
                bool any = false;
#pragma unroll
                for (int k = 0; k < K; ++k) any |= w[k] != 0.0f;
                const int idx32 = real ? (int)idx : 0;
                // ---- phase 2: stage rows [64][CC] (only rows with some influence)
#pragma unroll
                for (int r0 = 0; r0 < 64; r0 += ROWS_PER_I) {
                    const int r = r0 + lane / PIECES;
                    const int j = lane % PIECES;
                    const int ridx = __shfl(idx32, r, 64);
                    const int need = __shfl((int)any, r, 64);
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int ch = cc0 + 4 * j;
                    if (need) {
                        const float* src = x + (int64_t)ridx * ci + ch;
                        if (vec4 && ch + 3 < ci) {
                            v = *reinterpret_cast<const float4*>(src);
                        } else {
                            if (ch + 0 < ci) v.x = src[0];
                            if (ch + 1 < ci) v.y = src[1];
                            if (ch + 2 < ci) v.z = src[2];
                            if (ch + 3 < ci) v.w = src[3];
                        }
                    }
                    *reinterpret_cast<float4*>(&slab[r * CC + 4 * j]) = v;
                }
                wave_lds_sync();
                // ---- phase 3: lane = (slot, channel); sparse accumulate per kernel point
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    unsigned long long m = __ballot(w[k] != 0.0f);
                    while (m) {
                        int hh = 0;
                        float ww = 0.0f;
#pragma unroll
                        for (int e = 0; e < ES; ++e) {
                            if (m) {
                                const int hb = __builtin_ctzll(m);
                                m &= m - 1;
                                const float wb = ws_readlane_f(w[k], hb);
                                if (slot == e) { hh = hb; ww = wb; }
                            }
                        }
                        acc[k] = fmaf(ww, slab[hh * CC + c], acc[k]);
                    }
                }
                wave_lds_sync();
            }
            // ---- combine entry slots and write wf[q, k, cc0 + c]
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float a = acc[k];
#pragma unroll
                for (int o = CC; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
                if (modulations) a *= modulations[q * K + k];
                if (slot == 0 && cc0 + c < ci) wf[(q * K + k) * ci + cc0 + c] = a;
            }
        }
        if (min_d2) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float m = mind[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
                if (lane == 0) min_d2[q * K + k] = m;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K3 forward, ci <= 4 (input layer: in_features_dim 1..4).  lane = neighbour column holds its own
// feature row in registers; wf[q,k,c] is a wave reduction.  No LDS.
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void kpconv_gather_fwd_small_kernel(
    const float* __restrict__ q_pts, int64_t nq, const float* __restrict__ s_pts, int64_t ns,
    const int64_t* __restrict__ inds, int h, const float* __restrict__ x, int ci,
    const float* __restrict__ kernel_points, const float* __restrict__ deformed_kp,
    const float* __restrict__ modulations, GeomParams g, float* __restrict__ wf,
    float* __restrict__ min_d2)
{
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < nq; q += (int64_t)gridDim.x * 4) {
        const float qx = q_pts[3 * q + 0], qy = q_pts[3 * q + 1], qz = q_pts[3 * q + 2];
        const float* kp = deformed_kp ? deformed_kp + q * (3 * K) : kernel_points;
        float acc[K][4];
        float mind[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            mind[k] = 3.4e38f;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) acc[k][cc] = 0.0f;
        }
        for (int h0 = 0; h0 < h; h0 += 64) {
            const int col = h0 + lane;
            const bool incol = col < h;
            int64_t idx = incol ? inds[q * h + col] : ns;
            const bool real = incol && idx < ns && idx >= 0;
            float px = WS_SHADOW, py = WS_SHADOW, pz = WS_SHADOW;
            float xv[4] = {0.f, 0.f, 0.f, 0.f};
            if (real) {
                px = s_pts[3 * idx]; py = s_pts[3 * idx + 1]; pz = s_pts[3 * idx + 2];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) if (cc < ci) xv[cc] = x[idx * ci + cc];
            }
            float w[K], d2[K];
            kp_influence<K>(px - qx, py - qy, pz - qz, kp, g, real, w, d2);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (incol) mind[k] = fminf(mind[k], d2[k]);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) acc[k][cc] = fmaf(w[k], xv[cc], acc[k][cc]);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float mod = modulations ? modulations[q * K + k] : 1.0f;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                if (cc < ci) {
                    const float a = ws_wave_sum(acc[k][cc]);
                    if (lane == 0) wf[(q * K + k) * ci + cc] = a * mod;
                }
            }
            if (min_d2) {
                float m = mind[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
                if (lane == 0) min_d2[q * K + k] = m;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K4 backward w.r.t. x through the transposed table.  One wave per support point s:
//   phase 1: lane = incoming pair (q,col): influence of the K kernel points (recomputed);
//            non-zero (row = q*K+k, weight) entries are compacted into the wave's LDS list;
//   phase 2: lane = (slot, channel): dx[s, c] = sum_e weight_e * dwf[row_e, c], rows read straight
//            from HBM/L2 (each row is 4*ci contiguous bytes), UNROLL entries in flight per slot.
// ---------------------------------------------------------------------------------------------
template <int K, int CC>
__global__ __launch_bounds__(256) void kpconv_gather_bwd_x_kernel(
    const float* __restrict__ q_pts, int64_t nq, const float* __restrict__ s_pts, int64_t ns,
    int h, const int32_t* __restrict__ t_offsets, const int32_t* __restrict__ t_pairs,
    const float* __restrict__ dwf, int ci, const float* __restrict__ kernel_points,
    const float* __restrict__ deformed_kp, const float* __restrict__ modulations, GeomParams g,
    float* __restrict__ dx)
{
    constexpr int ES = 64 / CC;
    constexpr int LIST = 64 * K;   // worst case: every pair touches every kernel point
    __shared__ int l_row_all[4][LIST];
    __shared__ float l_w_all[4][LIST];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    int* l_row = l_row_all[wave];
    float* l_w = l_w_all[wave];
    const int c = lane % CC;
    const int slot = lane / CC;

    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < ns; s += (int64_t)gridDim.x * 4) {
        const float sx = s_pts[3 * s + 0], sy = s_pts[3 * s + 1], sz = s_pts[3 * s + 2];
        const int beg = t_offsets[s], end = t_offsets[s + 1];
        for (int cc0 = 0; cc0 < ci; cc0 += CC) {
            float acc = 0.0f;
            for (int p0 = beg; p0 < end; p0 += 64) {
                // ---- phase 1
                const int p = p0 + lane;
                const bool live = p < end;
                const int pair = live ? t_pairs[p] : 0;
                const int q = pair / h;
                float w[K], d2[K];
                {
                    const float nx = sx - q_pts[3 * (int64_t)q + 0];
                    const float ny = sy - q_pts[3 * (int64_t)q + 1];
                    const float nz = sz - q_pts[3 * (int64_t)q + 2];
                    if (deformed_kp) {
                        // per-lane kernel points (not wave-uniform here)
                        float kpl[3 * K];
#pragma unroll
                        for (int t = 0; t < 3 * K; ++t) kpl[t] = deformed_kp[(int64_t)q * (3 * K) + t];
                        kp_influence<K>(nx, ny, nz, kpl, g, live, w, d2);
                    } else {
                        kp_influence<K>(nx, ny, nz, kernel_points, g, live, w, d2);
                    }
                }
                int total = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float wk = w[k];
                    if (modulations && wk != 0.0f) wk *= modulations[(int64_t)q * K + k];
                    const bool nz_ = wk != 0.0f;
                    const unsigned long long m = __ballot(nz_);
                    if (nz_) {
                        const int rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
                        l_row[total + rank] = q * K + k;
                        l_w[total + rank] = wk;
                    }
                    total += __builtin_popcountll(m);
                }
                wave_lds_sync();
                // ---- phase 2
                for (int e0 = 0; e0 < total; e0 += ES * 4) {
                    float xv[4], wv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = e0 + u * ES + slot;
                        const bool ok = e < total && (cc0 + c) < ci;
                        const int row = ok ? l_row[e] : 0;
                        wv[u] = ok ? l_w[e] : 0.0f;
                        xv[u] = ok ? dwf[(int64_t)row * ci + cc0 + c] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = fmaf(wv[u], xv[u], acc);
                }
                wave_lds_sync();
            }
#pragma unroll
            for (int o = CC; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
            if (slot == 0 && cc0 + c < ci) dx[s * ci + cc0 + c] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K6 geometry backward (deformable): gradients of deformed_kp and modulations.
//   dL/dw(q,h,k)   = mod[q,k] * sum_c dwf[q,k,c] * x[idx,c]
//   dL/dmod[q,k]   = sum_h w(q,h,k) * sum_c dwf[q,k,c] * x[idx,c]
//   linear:  dw/dkp = (n - kp) / (extent * sqrt(d2))   where 0 < w
//   gaussian: dw/dkp = w * (n - kp) / (sigma^2 + 0.5e-9)
//   min_d2:  d min_d2[q,k]/dkp = 2 (kp - n_h*) at the arg-min column h*
// One wave per query; lane = neighbour column for the geometry, channels are looped with a wave
// reduction of the per-(h,k) dot products through LDS-staged dwf rows.
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void kpconv_gather_bwd_geom_kernel(
    const float* __restrict__ q_pts, int64_t nq, const float* __restrict__ s_pts, int64_t ns,
    const int64_t* __restrict__ inds, int h, const float* __restrict__ x, int ci,
    const float* __restrict__ dwf, const float* __restrict__ deformed_kp,
    const float* __restrict__ modulations, const float* __restrict__ d_min_d2, GeomParams g,
    float* __restrict__ d_kp, float* __restrict__ d_mod)
{
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < nq; q += (int64_t)gridDim.x * 4) {
        const float qx = q_pts[3 * q + 0], qy = q_pts[3 * q + 1], qz = q_pts[3 * q + 2];
        const float* kp = deformed_kp + q * (3 * K);
        float gk[K][3];
        float gm[K];
        float best[K];      // running (min d2, column) for the min_d2 path
        int bestcol[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            gk[k][0] = gk[k][1] = gk[k][2] = 0.0f;
            gm[k] = 0.0f;
            best[k] = 3.4e38f;
            bestcol[k] = 0x7fffffff;
        }
        for (int h0 = 0; h0 < h; h0 += 64) {
            const int col = h0 + lane;
            const bool incol = col < h;
            int64_t idx = incol ? inds[q * h + col] : ns;
            const bool real = incol && idx < ns && idx >= 0;
            float px = WS_SHADOW, py = WS_SHADOW, pz = WS_SHADOW;
            if (real) { px = s_pts[3 * idx]; py = s_pts[3 * idx + 1]; pz = s_pts[3 * idx + 2]; }
            const float nx = px - qx, ny = py - qy, nz = pz - qz;
            float w[K], d2[K];
            kp_influence<K>(nx, ny, nz, kp, g, real, w, d2);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                // dot[k] = sum_c dwf[q,k,c] * x[idx,c]  (only where the weight is live)
                float dot = 0.0f;
                if (w[k] != 0.0f) {
                    const float* a = dwf + (q * K + k) * ci;
                    const float* b = x + idx * ci;
                    for (int cc = 0; cc < ci; ++cc) dot = fmaf(a[cc], b[cc], dot);
                }
                const float mod = modulations ? modulations[q * K + k] : 1.0f;
                gm[k] += w[k] * dot;
                float coef = 0.0f;   // dL/dw * dw/d(d2) * 2, applied to (kp - n)
                if (w[k] != 0.0f) {
                    const float gw = dot * mod;
                    if (g.influence == WS_INFLUENCE_LINEAR) {
                        const float sd = sqrtf(d2[k]);
                        // w = 1 - sd/ext ; dw/dkp = -(kp - n) / (ext * sd)
                        coef = (sd > 0.0f) ? -gw / (g.extent * sd) : 0.0f;
                    } else if (g.influence == WS_INFLUENCE_GAUSSIAN) {
                        const float sig = g.extent * 0.3f;
                        // w = exp(-d2 / den) ; dw/dkp = -w * 2 (kp - n) / den
                        coef = -gw * w[k] * 2.0f / (2.0f * sig * sig + 1e-9f);
                    }
                }
                gk[k][0] += coef * (kp[3 * k + 0] - nx);
                gk[k][1] += coef * (kp[3 * k + 1] - ny);
                gk[k][2] += coef * (kp[3 * k + 2] - nz);
                if (incol && (d2[k] < best[k])) { best[k] = d2[k]; bestcol[k] = col; }
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            // min_d2 path: wave arg-min (first column wins ties, like torch.min on CPU)
            float gx = ws_wave_sum(gk[k][0]);
            float gy = ws_wave_sum(gk[k][1]);
            float gz = ws_wave_sum(gk[k][2]);
            const float gmod = ws_wave_sum(gm[k]);
            if (d_min_d2) {
                float b = best[k];
                int bc = bestcol[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ob = __shfl_xor(b, o, 64);
                    const int oc = __shfl_xor(bc, o, 64);
                    if (ob < b || (ob == b && oc < bc)) { b = ob; bc = oc; }
                }
                // the winning lane recomputes its offset
                const int64_t idx = inds[q * h + bc];
                float px = WS_SHADOW, py = WS_SHADOW, pz = WS_SHADOW;
                if (idx < ns && idx >= 0) { px = s_pts[3 * idx]; py = s_pts[3 * idx + 1]; pz = s_pts[3 * idx + 2]; }
                const float gmin = d_min_d2[q * K + k];
                gx += gmin * 2.0f * (kp[3 * k + 0] - (px - qx));
                gy += gmin * 2.0f * (kp[3 * k + 1] - (py - qy));
                gz += gmin * 2.0f * (kp[3 * k + 2] - (pz - qz));
            }
            if (lane == 0) {
                d_kp[(q * K + k) * 3 + 0] = gx;
                d_kp[(q * K + k) * 3 + 1] = gy;
                d_kp[(q * K + k) * 3 + 2] = gz;
                if (d_mod) d_mod[q * K + k] = gmod;
            }
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_common(const void* q_pts, int64_t nq, const void* s_pts, int64_t ns, int32_t h, int32_t ci,
                 int32_t k, float extent, int32_t influence, int32_t aggregation)
{
    WS_REQUIRE(nq >= 0 && ns >= 0, "negative point count (nq=%lld ns=%lld)", (long long)nq, (long long)ns);
    WS_REQUIRE(h >= 1 && ci >= 1, "h=%d and ci=%d must be >= 1", h, ci);
    WS_REQUIRE(nq == 0 || q_pts, "q_pts is NULL");
    WS_REQUIRE(ns == 0 || s_pts, "s_pts is NULL");
    WS_REQUIRE(extent > 0.0f, "KP_extent must be > 0");
    WS_REQUIRE(influence >= 0 && influence <= 2, "unknown influence %d", influence);
    WS_REQUIRE(aggregation >= 0 && aggregation <= 1, "unknown aggregation %d", aggregation);
    WS_REQUIRE(ns < (1ll << 31) && nq * (int64_t)k < (1ll << 31), "index range exceeds int32");
    if (k != 15) return ws_fail(WS_ERR_UNSUPPORTED, "num_kernel_points=%d: this build instantiates K=15 only", k);
    return WS_OK;
}

}  // namespace

extern "C" {

int ws_kpconv_gather_fwd(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                         const int64_t* inds, int32_t h, const float* x, int32_t ci,
                         const float* kernel_points, int32_t k, const float* deformed_kp,
                         const float* modulations, float extent, int32_t influence, int32_t aggregation,
                         float* wf, float* min_d2, void* stream)
{
    int rc = check_common(q_pts, nq, s_pts, ns, h, ci, k, extent, influence, aggregation);
    if (rc) return rc;
    if (nq == 0) return WS_OK;
    WS_REQUIRE(inds && x && wf && (kernel_points || deformed_kp), "NULL argument");
    GeomParams g{extent, influence, aggregation, deformed_kp ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    const int grid = ws_grid(nq, 4);
    if (ci <= 4) {
        kpconv_gather_fwd_small_kernel<15><<<grid, 256, 0, st>>>(q_pts, nq, s_pts, ns, inds, h, x, ci, kernel_points,
                                                                 deformed_kp, modulations, g, wf, min_d2);
    } else {
        const int vec4 = (ci % 4 == 0) && aligned16(x);
        if (ci <= 16)
            kpconv_gather_fwd_kernel<15, 16><<<grid, 256, 0, st>>>(q_pts, nq, s_pts, ns, inds, h, x, ci, kernel_points,
                                                                   deformed_kp, modulations, g, wf, min_d2, vec4);
        else
            kpconv_gather_fwd_kernel<15, 32><<<grid, 256, 0, st>>>(q_pts, nq, s_pts, ns, inds, h, x, ci, kernel_points,
                                                                   deformed_kp, modulations, g, wf, min_d2, vec4);
    }
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_kpconv_gather_bwd_x(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                           const int64_t* inds, int32_t h, const int32_t* t_offsets, const int32_t* t_pairs,
                           const float* dwf, int32_t ci, const float* kernel_points, int32_t k,
                           const float* deformed_kp, const float* modulations, float extent,
                           int32_t influence, int32_t aggregation, float* dx, void* stream)
{
    (void)inds;
    int rc = check_common(q_pts, nq, s_pts, ns, h, ci, k, extent, influence, aggregation);
    if (rc) return rc;
    if (ns == 0) return WS_OK;
    WS_REQUIRE(t_offsets && t_pairs && dwf && dx && (kernel_points || deformed_kp), "NULL argument");
    WS_REQUIRE(nq * (int64_t)h < (1ll << 31), "nq*h exceeds int32");
    GeomParams g{extent, influence, aggregation, deformed_kp ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    const int grid = ws_grid(ns, 4);
    if (ci <= 16)
        kpconv_gather_bwd_x_kernel<15, 16><<<grid, 256, 0, st>>>(q_pts, nq, s_pts, ns, h, t_offsets, t_pairs, dwf, ci,
                                                                 kernel_points, deformed_kp, modulations, g, dx);
    else
        kpconv_gather_bwd_x_kernel<15, 32><<<grid, 256, 0, st>>>(q_pts, nq, s_pts, ns, h, t_offsets, t_pairs, dwf, ci,
                                                                 kernel_points, deformed_kp, modulations, g, dx);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_kpconv_gather_bwd_geom(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                              const int64_t* inds, int32_t h, const float* x, int32_t ci, const float* dwf,
                              const float* kernel_points, int32_t k, const float* deformed_kp,
                              const float* modulations, const float* d_min_d2, float extent,
                              int32_t influence, int32_t aggregation, float* d_deformed_kp,
                              float* d_modulations, void* stream)
{
    (void)kernel_points;
    int rc = check_common(q_pts, nq, s_pts, ns, h, ci, k, extent, influence, aggregation);
    if (rc) return rc;
    if (nq == 0) return WS_OK;
    WS_REQUIRE(inds && x && dwf && deformed_kp && d_deformed_kp, "NULL argument");
    GeomParams g{extent, influence, aggregation, 1};
    hipStream_t st = (hipStream_t)stream;
    kpconv_gather_bwd_geom_kernel<15><<<ws_grid(nq, 4), 256, 0, st>>>(q_pts, nq, s_pts, ns, inds, h, x, ci, dwf,
                                                                       deformed_kp, modulations, d_min_d2, g,
                                                                       d_deformed_kp, d_modulations);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

}  // extern "C"
