// weasal_amd/csrc/kpconv.hip -- fused KPConv gather kernels for gfx950 (wave64).
//
// Reference semantics: models/blocks.py:238-374 (KPConv.forward) and its autograd.
//
// K3  kpconv_gather_fwd :  wf[q,k,c] = sum_h w(q,h,k) * x[inds[q,h], c]
//     neighbour gather -> kernel-point influence -> feature aggregate in ONE kernel; the
//     [N,H,3] / [N,H,K,3] / [N,H,K] / [N,H,Ci] intermediates of the reference never exist.
// K4  kpconv_gather_bwd_x : dx[s,c] = sum_{(q,h)->s} sum_k w * dwf[q,k,c]   (transposed table, no atomics)
// K6  kpconv_gather_bwd_geom : d deformed_kp, d modulations (deformable only)
// The structure of K3/K4 (entry pool in LDS, branch-free flush, scheduling order) is described
// above the kernels.
#include "ws_common.h"
#include <algorithm>
#include "ws_grid.h"
#include "ws_bf16.h"

extern "C" int ws_kpconv_table_interleave;
extern "C" int ws_kpconv_gridw_interleave;
extern "C" int ws_kpconv_k6_interleave;
extern "C" int ws_kpconv_grid_interleave;
namespace {

constexpr float WS_SHADOW = 1e6f;

struct GeomParams {
    float extent;
    int influence;
    int aggregation;
    int deformable;   // 1: apply the in-range filter of blocks.py:301-325
    int ablate;       // diagnostics only (tools/kpconv_lab.py): 1 = no wf store, 2 = every row gather reads row 0,
                      // 4 = coordinates of point `lane` instead of the neighbour's, 8 = no index load
    const void* gate; // K4 / K4G only: rows [ns, ci] of the activated output y of the layer that produced x; the stored
    float gate_slope; // gradient is dx * LeakyReLU'(y) (1 where y > 0, gate_slope elsewhere) -- the activation backward
                      // of the preceding unary block folded into the store.  NULL: plain dx.
    const int64_t* rows;  // K4G only: the index matrix [ns, rows_h] of the SAME self-query search the grid belongs to (NULL:
    int rows_h;           // none).  A support whose own row was not truncated (key_last[s] = "infinity") holds every
                          // point within the radius in that row -- a superset of the queries that kept s -- so its
                          // candidates are the row's entries instead of the 27-cell walk.
    const float4* kp4;    // MODE 2 only (deformable / modulated, linear influence, sum): the per-query kernel points packed as
                          // [nq, 15] float4 = (x, y, z, modulation) -- ONE aligned 16-byte load per (query, kernel point)
                          // where the reference-shaped operands deformed_kp [nq,15,3] + modulations [nq,15] need four.
    const float* rmax;    // K4G, MODE 2: device float = max over the queries of (max_k |kp_k|), written by
                          // ws_kpconv_deform_prepare: a pair farther apart than that + extent has no influence (NULL: no bound)
    int cut;              // pool-form K3 (MODE 0): 1 = rows are sorted by distance, stop at the influence reach (see CUT)
    int ilv;              // K4G: workgroups per XCD of the interleaved item assignment (ws_wave_items), 0 = contiguous chunks
    int csplit;           // matrix-core K3 (rigid): > 1 = an item is (query, one of csplit runs of channel blocks) instead of a query
                          // with all its blocks one after the other -- few queries with wide rows (the deep levels: 275 x 512)
                          // otherwise leave most SIMDs without a wave; the influences are recomputed per block either way
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

// LDS-only ordering inside one wave: the wavefront-scope release/acquire fences above make the compiler drain EVERY
// outstanding vector-memory load (s_waitcnt vmcnt(0)), i.e. they expose the full latency of loads that were issued on
// purpose ahead of their use.  A wave's own LDS writes and reads are ordered by the LDS queue; what is needed is that the
// compiler keeps them in program order and that the wave's earlier LDS operations have completed.
__device__ __forceinline__ void wave_lds_order()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// Pool + accumulate structure shared by K3 (forward) and K4 (backward w.r.t. x).
//
// One wave owns one query (K3) / one support (K4) at a time and alternates two phases:
//   list phase   lane = neighbour column (K3) or incoming pair (K4).  The influence of each kernel
//                point is evaluated in registers and the non-zero ones are compacted (v_cmp mask +
//                mbcnt rank) into the wave's LDS entry pool as (row, weight), kernel point after
//                kernel point, so the pool is grouped by kernel point (segs[k] = start of k).
//                With `linear` influence ~1 of the 15 kernel points is non-zero per neighbour: the
//                pool holds ~H entries instead of 15*H.
//   flush phase  lane = (entry slot, 16-byte piece of the row): G lanes read one feature-row
//                segment as float4 straight from L2/HBM (a row is 4*ci contiguous bytes -> 64..256
//                byte coalesced segments, no staging copy), S = 64/G entries per step, register
//                accumulation, no atomics (LDS float atomics were measured at ~1 lane/clk on gfx950).
//
// What bounds these kernels is the scalar unit (one SALU per CU) and divergent-branch bookkeeping,
// not HBM: profiles/r01 showed ~520 SALU instructions per query.  Hence
//   * MODE 0 (rigid kernel, linear influence, sum aggregation -- every shipped configuration) is a
//     separate instantiation: no per-kernel-point mode dispatch, the 15 kernel points live in SGPRs
//     for the whole launch, the list phase is fully unrolled;
//   * pool writes are unconditional (lanes without an entry write to a private dummy slot) and the
//     flush loops are branch free (lanes past the end of their segment carry weight 0 and read
//     row 0), so no exec-mask save/restore sits in any inner loop;
//   * pool overflow (only possible with dense influence modes or collapsed deformed kernels) is
//     handled after the fact: the chunk is redone in four 16-lane sub-chunks that always fit.
// MODE 1 keeps every influence / aggregation / deformable combination behind runtime switches.
// ---------------------------------------------------------------------------------------------
constexpr int POOL = 256;                 // real entries per wave (8 B each), >= 16 * 15
constexpr int POOL_ALLOC = POOL + 8 + 128; // + read-past slack of the flush + two dummy slots per lane

__device__ __forceinline__ int lane_rank(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// influence of ONE kernel point (same formulas as kp_influence)
__device__ __forceinline__ float kp_weight(float d2, const GeomParams& g, float inv_extent)
{
    if (g.influence == WS_INFLUENCE_LINEAR) return fmaxf(1.0f - __builtin_amdgcn_sqrtf(d2) * inv_extent, 0.0f);
    if (g.influence == WS_INFLUENCE_CONSTANT) return 1.0f;
    const float sig = g.extent * 0.3f;
    return __expf(-d2 / (2.0f * sig * sig + 1e-9f));
}

// 16-byte piece of a feature row, branch free.  VEC: the caller guarantees ch+3 < ci (or passes
// ch = 0 together with a zero weight).  !VEC: element-wise with clamped columns; columns >= ci
// are zeroed.
template <bool VEC, typename T>
__device__ __forceinline__ float4 load_row_piece(const T* __restrict__ base, unsigned row, int ci, int ch)
{
    // 32-bit element offset (the launcher checks rows * ci < 2^31): one v_mul_lo_u32 + one 64-bit add
    const T* src = base + (size_t)(row * (unsigned)ci);
    if (VEC) return ld4(src + ch);          // 16 bytes of f32 / 8 bytes of bf16 (ws_bf16.h)
    float4 v;
    const int last = ci - 1;
    v.x = ld1(src + min(ch + 0, last)); v.y = ld1(src + min(ch + 1, last)); v.z = ld1(src + min(ch + 2, last)); v.w = ld1(src + min(ch + 3, last));
    if (ch + 0 >= ci) v.x = 0.f;
    if (ch + 1 >= ci) v.y = 0.f;
    if (ch + 2 >= ci) v.z = 0.f;
    if (ch + 3 >= ci) v.w = 0.f;
    return v;
}

// keeps a pool read unconditional: without it the compiler sinks `ok ? pool[i] : 0` back under an
// exec-mask branch (s_and_saveexec + s_cbranch per entry -- the scalar unit is the bottleneck here)
__device__ __forceinline__ void keep_unconditional(uint2& e) { asm volatile("" : "+v"(e.x), "+v"(e.y)); }

// List phase for one 64-lane chunk.  KPF: kp(k, c) -> coordinate c of kernel point k (SGPR array,
// wave-uniform pointer or per-lane pointer).  Entry row = row_base + k * row_step.
// MINF: optional per-kernel-point hook (k, d2) used by the deformable forward for min_d2.
template <int K, int MODE, typename KPF, typename MINF>
__device__ __forceinline__ void kp_list(float nx, float ny, float nz, bool live, KPF kp, const GeomParams& g,
                                        float inv_extent, unsigned row_base, unsigned row_step, const float* __restrict__ lane_mod,
                                        uint2* pool, int* segs, int lane, int& total, int& maxlen, MINF minf)
{
    const unsigned dummy = POOL + 8 + lane;
    auto emit = [&](int k, float w) {
        const unsigned long long m = __ballot(w != 0.0f);
        const unsigned pos = (unsigned)(total + lane_rank(m));
        const bool ok = (w != 0.0f) && pos < (unsigned)POOL;
        pool[ok ? pos : dummy] = make_uint2(row_base + (unsigned)k * row_step, __float_as_uint(w));
        segs[k] = total;                                   // every lane, same value
        const int c = __builtin_popcountll(m);
        total += c;
        maxlen = max(maxlen, c);
    };
    if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float dx = nx - kp(k, 0), dy = ny - kp(k, 1), dz = nz - kp(k, 2);
            const float d = (dx * dx + dy * dy) + dz * dz;
            float w = fmaxf(1.0f - __builtin_amdgcn_sqrtf(d) * inv_extent, 0.0f);
            w = live ? w : 0.0f;
            emit(k, w);
        }
    } else {
        constexpr int KU = 3;
        static_assert(K % KU == 0, "K must be a multiple of 3");
        const float e2 = g.extent * g.extent;
        int arg = 0;
        if (g.deformable || g.aggregation == WS_AGGREGATION_CLOSEST) {
            float best = 3.4e38f;
            bool inrange = false;
#pragma unroll 1
            for (int kb = 0; kb < K; kb += KU) {
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    const int k = kb + u;
                    const float dx = nx - kp(k, 0), dy = ny - kp(k, 1), dz = nz - kp(k, 2);
                    const float d = (dx * dx + dy * dy) + dz * dz;
                    if (d < best) { best = d; arg = k; }
                    inrange |= d < e2;
                }
            }
            if (g.deformable) live = live && inrange;
        }
#pragma unroll 1
        for (int kb = 0; kb < K; kb += KU) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int k = kb + u;
                const float dx = nx - kp(k, 0), dy = ny - kp(k, 1), dz = nz - kp(k, 2);
                const float d = (dx * dx + dy * dy) + dz * dz;
                minf(k, d);
                float w = kp_weight(d, g, inv_extent);
                if (g.aggregation == WS_AGGREGATION_CLOSEST && k != arg) w = 0.0f;
                w = live ? w : 0.0f;
                if (lane_mod && w != 0.0f) w *= lane_mod[k];
                emit(k, w);
            }
        }
    }
    segs[K] = total;
}

// List phase for TWO 64-lane column chunks at once (rigid / linear / sum only): lane = columns c and
// c + 64.  Rows of 65..128 neighbours (the calibrated limits of the deeper pyramid levels sit at
// 70-75) then need ONE pool and ONE flush instead of a second, nearly empty chunk that re-reads and
// re-writes all K rows of wf[q].
template <int K, typename KPF>
__device__ __forceinline__ void kp_list2(float ax, float ay, float az, bool alive, unsigned arow, float bx, float by, float bz,
                                         bool blive, unsigned brow, KPF kp, float inv_extent, uint2* pool, int* segs, int lane,
                                         int& total, int& maxlen)
{
    const unsigned dummy_a = POOL + 8 + lane, dummy_b = POOL + 8 + 64 + lane;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float kx = kp(k, 0), ky = kp(k, 1), kz = kp(k, 2);
        const float dxa = ax - kx, dya = ay - ky, dza = az - kz;
        const float dxb = bx - kx, dyb = by - ky, dzb = bz - kz;
        float wa = fmaxf(1.0f - __builtin_amdgcn_sqrtf((dxa * dxa + dya * dya) + dza * dza) * inv_extent, 0.0f);
        float wb = fmaxf(1.0f - __builtin_amdgcn_sqrtf((dxb * dxb + dyb * dyb) + dzb * dzb) * inv_extent, 0.0f);
        wa = alive ? wa : 0.0f;
        wb = blive ? wb : 0.0f;
        const unsigned long long ma = __ballot(wa != 0.0f), mb = __ballot(wb != 0.0f);
        const int ca = __builtin_popcountll(ma), cb = __builtin_popcountll(mb);
        const unsigned pa = (unsigned)(total + lane_rank(ma)), pb = (unsigned)(total + ca + lane_rank(mb));
        const bool oka = (wa != 0.0f) && pa < (unsigned)POOL, okb = (wb != 0.0f) && pb < (unsigned)POOL;
        pool[oka ? pa : dummy_a] = make_uint2(arow, __float_as_uint(wa));
        pool[okb ? pb : dummy_b] = make_uint2(brow, __float_as_uint(wb));
        segs[k] = total;
        total += ca + cb;
        maxlen = max(maxlen, ca + cb);
    }
    segs[K] = total;
}

// List phase of the deformable fast path (MODE 2: per-query kernel points + modulations, linear influence, sum
// aggregation; models/blocks.py:244-267, 287-291, 330-346, 366-367).  lane = pair; `kq` = the 15 packed kernel points
// (x, y, z, modulation) of the pair's QUERY: 15 independent 16-byte loads, all issued before the first use.  One pass:
// the in-range filter of blocks.py:301-325 is implied by the linear influence (a neighbour with no kernel point inside
// the extent has 15 zero influences), and the modulation multiplies the weight (d wf / d x = mod * w).
template <int K>
__device__ __forceinline__ void kp_list_def(float nx, float ny, float nz, bool live, const float4* __restrict__ kq,
                                            float inv_extent, unsigned row_base, uint2* pool, int* segs, int lane, int& total,
                                            int& maxlen)
{
    const unsigned dummy = POOL + 8 + lane;
    float4 kp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) kp[k] = kq[k];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float dx = nx - kp[k].x, dy = ny - kp[k].y, dz = nz - kp[k].z;
        const float d = (dx * dx + dy * dy) + dz * dz;
        float w = fmaxf(1.0f - __builtin_amdgcn_sqrtf(d) * inv_extent, 0.0f) * kp[k].w;
        w = live ? w : 0.0f;
        const unsigned long long m = __ballot(w != 0.0f);
        const unsigned pos = (unsigned)(total + lane_rank(m));
        const bool ok = (w != 0.0f) && pos < (unsigned)POOL;
        pool[ok ? pos : dummy] = make_uint2(row_base + (unsigned)k, __float_as_uint(w));
        segs[k] = total;
        const int c = __builtin_popcountll(m);
        total += c;
        maxlen = max(maxlen, c);
    }
    segs[K] = total;
}

// ---------------------------------------------------------------------------------------------
// K3 forward.  In the flush phase entry slot s owns the kernel points s, s+S, s+2S, ...: it walks
// their segments with a float4 register accumulator and stores the finished 16-byte piece of
// wf[q,k,:] straight to HBM (fixed summation order).  The pool is built once per 64-column chunk
// and flushed for every 4*G-channel chunk; further column chunks (H > 64) and overflow sub-chunks
// accumulate onto the rows already written.  The index and xyz loads of the next two items are
// software-prefetched under the current item.
// ---------------------------------------------------------------------------------------------
template <int K, int G, int MODE, bool DEF, bool VEC, int PW = 4, typename T = float>
__global__ __launch_bounds__(256) void kpconv_gather_fwd_kernel(
    const float* __restrict__ q_pts, int64_t nq, const float* __restrict__ s_pts, int64_t ns,
    const int64_t* __restrict__ inds, int h, const T* __restrict__ x, int ci,
    const float* __restrict__ kernel_points, const float* __restrict__ deformed_kp,
    const float* __restrict__ modulations, GeomParams g, T* __restrict__ wf,
    float* __restrict__ min_d2, const int32_t* __restrict__ order)
{
    static_assert(!(DEF && MODE == 0), "deformable layers use MODE 1");
    static_assert(PW == 4 || (PW == 1 && !VEC), "piece width: 16 bytes, or one float for the 3-channel input layer");
    constexpr int CC = PW * G;     // channels per chunk (PW = 1: lane = (kernel point, channel), 4-byte pieces)
    constexpr int S = 64 / G;      // entry slots
    constexpr int KPS = (K + S - 1) / S;   // kernel points per slot
    static_assert(K <= 16 && POOL >= 16 * K, "pool sizing");
    __shared__ uint2 pool_all[4][POOL_ALLOC];
    __shared__ int segs_all[4][K + 1];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint2* pool = pool_all[wave];
    int* segs = segs_all[wave];
    const int j = lane % G;        // 16-byte piece of the row chunk
    const int slot = lane / G;
    const float inv_extent = 1.0f / g.extent;

    int64_t ibeg, iend;
    ws_block_range(nq, ibeg, iend);

    // MODE 0: the rigid kernel points are wave-uniform for the whole launch -> registers (SGPRs)
    float kpr[MODE == 0 ? 3 * K : 1];
    float cut2_pool = 3.4e38f;
    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 3 * K; ++t) kpr[t] = kernel_points[t];
        if (g.cut) {
            float rr = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) rr = fmaxf(rr, (kpr[3 * k] * kpr[3 * k] + kpr[3 * k + 1] * kpr[3 * k + 1]) + kpr[3 * k + 2] * kpr[3 * k + 2]);
            const float R = (__builtin_sqrtf(rr) + g.extent) * 1.0001f;
            cut2_pool = R * R;
        }
    }

    // flush the pool: every slot walks the segments of its kernel points for all channel chunks.
    // maxlen = longest segment (wave-uniform): every round runs that many steps.
    auto flush = [&](int64_t q, bool accumulate, int maxlen) {
        wave_lds_sync();
        if (PW == 1) {
            // narrow rows (the 3-channel input layer): one float per lane, a slot of G lanes per kernel point --
            // 45 of 64 lanes busy with one 4-byte load per entry instead of 15 lanes with four clamped loads
            for (int cc0 = 0; cc0 < ci; cc0 += CC) {
                const int ch = cc0 + j;
                const bool chok = ch < ci;
                const int chl = chok ? ch : 0;
#pragma unroll
                for (int kk = 0; kk < KPS; ++kk) {
                    const int k = slot + kk * S;
                    const bool kok = k < K && chok;
                    const int beg = kok ? segs[k] : 0;
                    const int end = kok ? min(segs[k + 1], POOL) : 0;
                    float a = 0.0f;
                    for (int it = 0; it < maxlen; it += 4) {
                        uint2 e[4];
                        float v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            e[u] = pool[min(beg + it + u, POOL + 7)];
                            keep_unconditional(e[u]);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool ok = beg + it + u < end;
                            e[u].x = ok ? e[u].x : 0u;
                            e[u].y = ok ? e[u].y : 0u;
                            v[u] = ld1(x + (size_t)(e[u].x * (unsigned)ci) + chl);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) a = fmaf(__uint_as_float(e[u].y), v[u], a);
                    }
                    if (kok) {
                        if (modulations) a *= modulations[q * K + k];
                        T* dst = wf + (q * K + k) * ci + ch;
                        st1(dst, (accumulate ? ld1(dst) : 0.f) + a);
                    }
                }
            }
            wave_lds_sync();
            return;
        }
        for (int cc0 = 0; cc0 < ci; cc0 += CC) {
            const int ch = cc0 + 4 * j;
            const bool chok = VEC ? (ch + 3 < ci) : (ch < ci);
            const int chl = chok ? ch : 0;
#pragma unroll
            for (int kk = 0; kk < KPS; ++kk) {
                const int k = slot + kk * S;
                const bool kok = k < K && chok;
                const int beg = kok ? segs[k] : 0;
                const int end = kok ? min(segs[k + 1], POOL) : 0;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int it = 0; it < maxlen; it += 4) {
                    uint2 e[4];
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        e[u] = pool[min(beg + it + u, POOL + 7)];
                        keep_unconditional(e[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool ok = beg + it + u < end;
                        e[u].x = ok ? e[u].x : 0u;
                        e[u].y = ok ? e[u].y : 0u;                    // weight 0.0f
                        v[u] = load_row_piece<VEC>(x, e[u].x, ci, chl);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float w = __uint_as_float(e[u].y);
                        a.x = fmaf(w, v[u].x, a.x); a.y = fmaf(w, v[u].y, a.y);
                        a.z = fmaf(w, v[u].z, a.z); a.w = fmaf(w, v[u].w, a.w);
                    }
                }
                if (kok) {
                    if (modulations) { const float md = modulations[q * K + k]; a.x *= md; a.y *= md; a.z *= md; a.w *= md; }
                    T* dst = wf + (q * K + k) * ci + ch;
                    if (VEC) {
                        if (accumulate) { const float4 o = ld4(dst); a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
                        st4(dst, a);
                    } else {
                        if (ch + 0 < ci) st1(dst + 0, (accumulate ? ld1(dst + 0) : 0.f) + a.x);
                        if (ch + 1 < ci) st1(dst + 1, (accumulate ? ld1(dst + 1) : 0.f) + a.y);
                        if (ch + 2 < ci) st1(dst + 2, (accumulate ? ld1(dst + 2) : 0.f) + a.z);
                        if (ch + 3 < ci) st1(dst + 3, (accumulate ? ld1(dst + 3) : 0.f) + a.w);
                    }
                }
            }
        }
        wave_lds_sync();
    };

    // ---- software pipeline over the items of this wave: idx two items ahead, xyz one item ahead
    auto item_q = [&](int64_t it) -> int64_t { return it < iend ? (order ? (int64_t)order[it] : it) : -1; };
    auto load_idx = [&](int64_t q) -> int {
        if (q < 0 || lane >= h) return -1;
        const int64_t v = inds[q * h + lane];
        return (v >= 0 && v < ns) ? (int)v : -1;
    };
    auto load_pt = [&](int idx, float& px, float& py, float& pz) {
        px = py = pz = WS_SHADOW;
        if (idx >= 0) { px = s_pts[3 * (int64_t)idx]; py = s_pts[3 * (int64_t)idx + 1]; pz = s_pts[3 * (int64_t)idx + 2]; }
    };
    int64_t q0 = item_q(ibeg + wave), q1 = item_q(ibeg + wave + 4);
    int idx0 = load_idx(q0), idx1 = load_idx(q1);
    float p0x, p0y, p0z;
    load_pt(idx0, p0x, p0y, p0z);

    for (int64_t item = ibeg + wave; item < iend; item += 4) {
        const int64_t q = q0;
        const int64_t q2 = item_q(item + 8);
        const int idx2 = load_idx(q2);
        float p1x, p1y, p1z;
        load_pt(idx1, p1x, p1y, p1z);

        const float qx = q_pts[3 * q + 0], qy = q_pts[3 * q + 1], qz = q_pts[3 * q + 2];
        const float* kpp = DEF ? deformed_kp + q * (3 * K) : kernel_points;      // wave-uniform
        auto kp = [&](int k, int c) { return MODE == 0 ? kpr[MODE == 0 ? 3 * k + c : 0] : kpp[3 * k + c]; };
        bool accumulate = false;
        for (int h0 = 0; h0 < h; h0 += 64) {
            const int col = h0 + lane;
            const bool incol = col < h;
            int idx = idx0;
            float px = p0x, py = p0y, pz = p0z;
            if (h0 > 0) {      // further column chunks are loaded on demand
                idx = -1;
                if (incol) { const int64_t v = inds[q * h + col]; idx = (v >= 0 && v < ns) ? (int)v : -1; }
                load_pt(idx, px, py, pz);
            }
            const bool real = incol && idx >= 0;
            const float nx = px - qx, ny = py - qy, nz = pz - qz;
            if (MODE == 0 && g.cut && h0 > 0) {
                // sorted rows: once a chunk holds no neighbour inside the influence reach, neither does any later one
                if (__ballot(real && (nx * nx + ny * ny) + nz * nz <= cut2_pool) == 0ull) break;
            }
            const unsigned row = (unsigned)(real ? idx : 0);
            int total = 0, maxlen = 0;
            if (MODE == 0 && h - h0 > 64) {
                // 65..128 columns left: both chunks into one pool, one flush
                const int colb = col + 64;
                int idxb = -1;
                if (colb < h) { const int64_t v = inds[q * h + colb]; idxb = (v >= 0 && v < ns) ? (int)v : -1; }
                float bx, by, bz;
                load_pt(idxb, bx, by, bz);
                const bool realb = idxb >= 0;
                kp_list2<K>(nx, ny, nz, real, row, bx - qx, by - qy, bz - qz, realb, (unsigned)(realb ? idxb : 0), kp, inv_extent,
                            pool, segs, lane, total, maxlen);
                if (total <= POOL) {
                    flush(q, accumulate, maxlen);
                    accumulate = true;
                    h0 += 64;
                    continue;
                }
                total = 0; maxlen = 0;      // does not fit: fall through, one chunk at a time
            }
            auto minf = [&](int k, float d) {
                if (DEF && min_d2) {
                    float m = incol ? d : 3.4e38f;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
                    if (lane == 0) min_d2[q * K + k] = h0 == 0 ? m : fminf(min_d2[q * K + k], m);
                }
            };
            kp_list<K, MODE>(nx, ny, nz, real, kp, g, inv_extent, row, 0u, nullptr, pool, segs, lane, total, maxlen, minf);
            if (total <= POOL) {
                flush(q, accumulate, maxlen);
                accumulate = true;
            } else {
                // the chunk does not fit the pool: redo it in four 16-lane sub-chunks (16 * K <= POOL)
                auto nomin = [&](int, float) {};
                for (int sub = 0; sub < 4; ++sub) {
                    total = 0; maxlen = 0;
                    kp_list<K, MODE>(nx, ny, nz, real && (lane >> 4) == sub, kp, g, inv_extent, row, 0u, nullptr, pool, segs,
                                     lane, total, maxlen, nomin);
                    flush(q, accumulate, maxlen);
                    accumulate = true;
                }
            }
        }
        // rotate the pipeline
        q0 = q1; q1 = q2;
        idx0 = idx1; idx1 = idx2;
        p0x = p1x; p0y = p1y; p0z = p1z;
    }
}

// ---------------------------------------------------------------------------------------------
// K3 forward on the matrix core.  The weighted-feature product of the reference,
//     weighted_features = torch.matmul(all_weights [N,K,H], neighb_x [N,H,Ci])       (models/blocks.py:363)
// is a small dense matrix product per query: wf[q] (15 x Ci) = W_q^T (15 x H) . X_q (H x Ci).  Here it runs as such on
// v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered fmaf chain, so the sum over the neighbour columns has the same bits as
// the per-kernel-point VALU chains of kpconv_gather_fwd_kernel -- zero weights add exactly nothing):
//   A operand  A[i][k]   lane (i = lane&15, kk = lane>>4) holds the influence of kernel point i on neighbour column
//              4 s + kk of step s: ONE weight, computed by that lane from the neighbour's offset (an LDS broadcast of
//              the staged float4) and its own kernel point (rigid: a register for the whole launch; deformed: 3 loads
//              per query).  No ballot, no compaction, no entry pool: 16 steps x ~10 VALU per 64 columns.
//   B operand  B[k][j]   lane (j = lane&15, kk) loads NT consecutive channels of row inds[q, 4 s + kk] (one 4..64-byte
//              load; the 16 lanes of a row cover 16 NT channels = the whole row for Ci <= 256), tile t = channel
//              NT j + t; loads of a whole group of steps are in flight before the first MFMA needs them.
//   C / D      lane (j, g = lane>>4) ends with kernel points 4 g .. 4 g + 3 of channels NT j .. NT j + NT - 1: rows of wf
//              are written as contiguous 16 NT-channel runs.
// Wider rows (H > 64) simply keep accumulating over further 64-column chunks: no read-modify-write of wf.  Only ~1.3 of
// the 15 influences per neighbour are non-zero, so the matrix core does ~12 x the useful flops -- on a pipe that is
// otherwise idle here, at 32 cycles per 16x16x4 step: 30 MFMAs per query at Ci = 32 against the ~750 VALU/SALU/LDS
// instructions of the pool form.  MODE / DEF as above (MODE 0 = rigid, linear, sum).
// ---------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
#ifndef WS_K3_GS
#define WS_K3_GS 4
#endif
#ifndef WS_FUSE_GS
#define WS_FUSE_GS 2      // steps in flight of the fused-contraction variant: its 60 registers of W leave room for fewer
#endif

template <int NT, typename T> struct RowLoad;
template <int NT> struct RowLoad<NT, float> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[NT])
    {
        if constexpr (NT == 1) v[0] = *p;
        else if constexpr (NT == 2) { const float2 a = *reinterpret_cast<const float2*>(p); v[0] = a.x; v[1] = a.y; }
        else {
#pragma unroll
            for (int t = 0; t < NT; t += 4) {
                const float4 a = *reinterpret_cast<const float4*>(p + t);
                v[t] = a.x; v[t + 1] = a.y; v[t + 2] = a.z; v[t + 3] = a.w;
            }
        }
    }
    static __device__ __forceinline__ void st(float* p, const float (&v)[NT])
    {
        if constexpr (NT == 1) *p = v[0];
        else if constexpr (NT == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
        else {
#pragma unroll
            for (int t = 0; t < NT; t += 4) *reinterpret_cast<float4*>(p + t) = make_float4(v[t], v[t + 1], v[t + 2], v[t + 3]);
        }
    }
};
template <int NT> struct RowLoad<NT, bf16_t> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[NT])
    {
        if constexpr (NT == 1) v[0] = ld1(p);
        else if constexpr (NT == 2) { const unsigned a = *reinterpret_cast<const unsigned*>(p); v[0] = ws_bf_lo(a); v[1] = ws_bf_hi(a); }
        else if constexpr (NT == 4) { const float4 a = ld4(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
        else {
#pragma unroll
            for (int t = 0; t < NT; t += 8) {
                const uint4 a = *reinterpret_cast<const uint4*>(p + t);
                v[t] = ws_bf_lo(a.x); v[t + 1] = ws_bf_hi(a.x); v[t + 2] = ws_bf_lo(a.y); v[t + 3] = ws_bf_hi(a.y);
                v[t + 4] = ws_bf_lo(a.z); v[t + 5] = ws_bf_hi(a.z); v[t + 6] = ws_bf_lo(a.w); v[t + 7] = ws_bf_hi(a.w);
            }
        }
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[NT])
    {
        if constexpr (NT == 1) st1(p, v[0]);
        else if constexpr (NT == 2) *reinterpret_cast<unsigned*>(p) = ws_pack_bf2(v[0], v[1]);
        else if constexpr (NT == 4) st4(p, make_float4(v[0], v[1], v[2], v[3]));
        else {
#pragma unroll
            for (int t = 0; t < NT; t += 8)
                *reinterpret_cast<uint4*>(p + t) = make_uint4(ws_pack_bf2(v[t], v[t + 1]), ws_pack_bf2(v[t + 2], v[t + 3]),
                                                              ws_pack_bf2(v[t + 4], v[t + 5]), ws_pack_bf2(v[t + 6], v[t + 7]));
        }
    }
};

// VECROW: ci is a multiple of NT and rows are aligned for NT-element accesses (else NT == 1 with per-lane masking)
// CUT (MODE 0 / 2 only): the caller vouches that every index row is sorted by distance from its query (what the radius search
// delivers, neighbors.cpp:293).  A neighbour farther from the query than  max_k |kp_k| + extent  has 15 zero influences, and
// one farther than  max_k (sqrt(min_d2[k] so far) + |kp_k|)  cannot lower any minimum (triangle inequality): the walk over
// the row stops there.  Exact -- the skipped terms are zeros -- and decisive where the rows come from the DEFORMABLE radius
// (datasets/common.py:500-502: 2 r, while the kernel points reach 0.69 r + extent = 1.09 r: 84 % of a rigid offset
// convolution's neighbours, ~60 % of a deformed one's, are outside every influence).
// FUSE (forward only; MODE 0, NT = 2, f32, Ci = Co = 32: the level-0 layers, where the time is): the kernel contraction
//   out[q, :] = act(wf[q] (1 x 15 Ci) . W (15 Ci x Co) + bias)                         (models/blocks.py:370-374, 556-563)
// runs inside this kernel and `wf` never reaches memory (N x 15 x Ci x 4 B = 768 MB at enc1, written and read back by the
// two-launch form -- which training keeps, dW needs wf).  The four waves of a workgroup gather one query each per
// iteration into an LDS tile of 16 queries (four iterations); then every wave multiplies its QUARTER of the 480-deep
// contraction -- that quarter of W lives in its registers for the whole launch (60 VGPRs: lane (j, kk) holds
// W[120 w + 4 s + kk][16 n + j]) -- on v_mfma_f32_16x16x4_f32, the four partial [16 x 32] tiles are added through LDS
// with the bias / LeakyReLU epilogue and stored as 16 output rows.  Tile rows are 484 floats apart: the A-operand reads
// (lane (query, kk)) fall on 64 different banks.
struct FuseArgs {
    const float* w;        // [15 * ci, co] row-major (the module's weights [15, ci, co])
    const float* bias;     // [co] or NULL
    float slope;           // LeakyReLU slope (act != 0)
    int act;
    float* out;            // [nq, co]
};

template <int NT, int MODE, bool DEF, bool VECROW, typename T, int GSV = 0, bool CUT = false, bool FUSE = false>
__global__ __launch_bounds__(256) void kpconv_gather_fwd_mfma_kernel(
    const float* __restrict__ q_pts, int64_t nq, const float* __restrict__ s_pts, int64_t ns,
    const int64_t* __restrict__ inds, int h, const T* __restrict__ x, int ci,
    const float* __restrict__ kernel_points, const float* __restrict__ deformed_kp,
    const float* __restrict__ modulations, GeomParams g, T* __restrict__ wf,
    float* __restrict__ min_d2, const int32_t* __restrict__ order, FuseArgs fz = FuseArgs{})
{
    static_assert(!FUSE || (MODE == 0 && NT == 2 && sizeof(T) == 4 && VECROW), "the fused contraction covers rigid 32 -> 32 f32 layers");
    constexpr int TILE_LD = 484;
    constexpr int K = 15;
    constexpr int CB = 16 * NT;                                   // channels per block
    // steps whose loads are in flight together (per buffer; two buffers): more = deeper memory pipelining per wave,
    // fewer = fewer registers = more waves per SIMD.  The kernel is bound by instruction latency x occupancy (with every
    // memory access ablated it still takes 0.53 of its 0.68 ms on the 400k x 59 x 32 layer), so registers win
    constexpr int GS = GSV > 0 ? GSV : (NT <= 2 ? WS_K3_GS : (NT == 4 ? 2 : 1));
    static_assert(!(DEF && MODE == 0), "deformable layers use MODE 1 or 2");
    static_assert(MODE != 2 || DEF, "MODE 2 is the deformable fast path");
    static_assert(!CUT || MODE == 0 || MODE == 2, "the distance cutoff belongs to the linear-influence fast paths");
    static_assert(VECROW || NT == 1, "masked rows use one channel per lane");
    __shared__ float4 nb_all[4][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kk = lane >> 4;                      // A: kernel point i; B / D: channel lane j = i; row group kk
    float4* nb = nb_all[wave];
    const float inv_extent = 1.0f / g.extent;
    const float e2 = g.extent * g.extent;
    const bool haskp = i < K;

    const int csplit = (FUSE || MODE != 0 || g.csplit < 1) ? 1 : g.csplit;      // items per query (see GeomParams::csplit)
    int64_t ibeg, iend;
    ws_block_range(nq * csplit, ibeg, iend);

    float kx = 0.f, ky = 0.f, kz = 0.f;
    if (!DEF && haskp) { kx = kernel_points[3 * i]; ky = kernel_points[3 * i + 1]; kz = kernel_points[3 * i + 2]; }
    if ((MODE == 0 || MODE == 2) && !haskp) kx = ky = kz = 1.0e9f;   // the 16th "kernel point": far from everything, weight 0
    float kmod = 0.0f;                                            // MODE 2: modulation of this lane's kernel point
    // squared cutoff on the neighbour's distance from the query (CUT); rigid: a constant of the launch
    auto group_max = [&](float v) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        return v;
    };
    float cut2_rigid = 3.4e38f;
    if constexpr (CUT && MODE == 0) {
        const float rr = group_max(haskp ? __builtin_sqrtf((kx * kx + ky * ky) + kz * kz) : 0.0f);
        const float R = (rr + g.extent) * 1.0001f;
        cut2_rigid = R * R;
    }

    // ---- software pipeline over the items of this wave.  Everything an item needs before its row loads -- its query index
    // (order[]), its index row, the neighbours' coordinates, its own coordinates -- is a chain of dependent memory
    // accesses of ~1 us each; executed at the item they would leave the wave idle for several microseconds per query (the
    // first forms of this kernel spent 0.47 of 0.68 ms with EVERY memory access and all arithmetic ablated).  So the chain
    // is spread over three items: at the top of item t the wave issues  order[t+3],  inds / q_pts of item t+2,  the
    // neighbour coordinates of item t+1,  and looks at none of the results before the next iteration.  All of them are
    // VECTOR loads, also the lane-uniform ones (order, q_pts: the address carries an opaque zero VGPR): scalar loads share
    // the lgkm counter with LDS and return out of order, so the LDS ordering below would wait for them on the spot.
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    const int64_t it0 = ibeg + wave;
    if (FUSE ? (ibeg >= iend) : (it0 >= iend)) return;            // (FUSE: workgroup-uniform -- every wave meets the barriers)
    // FUSE: every wave runs the same number of iterations (dummy ones redo the workgroup's last query and store nothing)
    const int64_t iend_u = FUSE ? ibeg + ((iend - ibeg + 3) / 4) * 4 : iend;
    float wq[FUSE ? 30 : 1][2];
    if constexpr (FUSE) {
#pragma unroll
        for (int s_ = 0; s_ < 30; ++s_) {
            wq[s_][0] = fz.w[(120 * wave + 4 * s_ + kk) * 32 + i];
            wq[s_][1] = fz.w[(120 * wave + 4 * s_ + kk) * 32 + 16 + i];
        }
    }
    int iter = 0;
    auto item_v = [&](int64_t it) -> int {                        // query of item `it`, in a VGPR (clamped past the end)
        const int64_t itc = it < iend ? it : iend - 1;
        const int64_t qi = csplit > 1 ? itc / csplit : itc;
        return order ? order[qi + vz] : (int)qi + vz;
    };
    const int col0 = lane < h ? lane : 0;
    auto raw_idx = [&](int qv) -> int64_t {
        if (g.ablate & 8) return ((int64_t)qv + 37 * col0) % ns;
        return inds[(int64_t)qv * h + col0];
    };
    auto chk = [&](int64_t raw) -> int { return (lane < h && raw >= 0 && raw < ns) ? (int)raw : -1; };
    auto pt_raw = [&](int idx, float& px, float& py, float& pz) {
        int ii = idx >= 0 ? idx : 0;
        if (g.ablate & 4) ii = lane;
        px = s_pts[3 * (int64_t)ii]; py = s_pts[3 * (int64_t)ii + 1]; pz = s_pts[3 * (int64_t)ii + 2];
    };
    auto q_xyz = [&](int qv, float& x_, float& y_, float& z_) {
        x_ = q_pts[3 * (int64_t)qv + 0]; y_ = q_pts[3 * (int64_t)qv + 1]; z_ = q_pts[3 * (int64_t)qv + 2];
    };
    auto load_idx = [&](int qv, int col) -> int {                  // (further 64-column chunks of wide rows: at use)
        if (col >= h) return -1;
        const int64_t v = inds[(int64_t)qv * h + col];
        return (v >= 0 && v < ns) ? (int)v : -1;
    };
    auto load_pt = [&](int idx, float& px, float& py, float& pz) {
        px = py = pz = WS_SHADOW;
        if (idx >= 0) { px = s_pts[3 * (int64_t)idx]; py = s_pts[3 * (int64_t)idx + 1]; pz = s_pts[3 * (int64_t)idx + 2]; }
    };
    int qv0 = item_v(it0), qv1 = item_v(it0 + 4), qv2 = item_v(it0 + 8);
    int64_t raw1 = raw_idx(qv1);
    int idx0 = chk(raw_idx(qv0));
    float p0x, p0y, p0z, qx, qy, qz, q1x, q1y, q1z;
    pt_raw(idx0, p0x, p0y, p0z);
    p0x = idx0 >= 0 ? p0x : WS_SHADOW; p0y = idx0 >= 0 ? p0y : WS_SHADOW; p0z = idx0 >= 0 ? p0z : WS_SHADOW;
    q_xyz(qv0, qx, qy, qz);
    q_xyz(qv1, q1x, q1y, q1z);
    int idx1 = chk(raw1);

    for (int64_t item = it0; item < iend_u; item += 4) {
        const int64_t q = qv0;                                    // (a VGPR value, the same in every lane)
        const int qv3 = item_v(item + 12);
        const int64_t raw2 = raw_idx(qv2);
        float q2x, q2y, q2z, p1x, p1y, p1z;
        q_xyz(qv2, q2x, q2y, q2z);
        pt_raw(idx1, p1x, p1y, p1z);
        if constexpr (MODE == 2) {
            if (haskp) {
                const float4 kq = g.kp4[q * K + i];
                kx = kq.x; ky = kq.y; kz = kq.z; kmod = kq.w;
            }
        } else if (DEF && haskp) {
            const float* kp = deformed_kp + q * (3 * K) + 3 * i;
            kx = kp[0]; ky = kp[1]; kz = kp[2];
        }
        // channel blocks of this item: all of them, or the item's run of whole blocks
        int cb_lo = 0, cb_hi = ci;
        if (csplit > 1) {
            const int run = ((ci + CB * csplit - 1) / (CB * csplit)) * CB;
            cb_lo = (int)(item % csplit) * run;
            cb_hi = min(ci, cb_lo + run);
        }
        float mind = 3.4e38f;
        float cut2 = cut2_rigid, rk = 0.0f, rq = 0.0f;
        if constexpr (CUT && MODE == 2) {
            rk = haskp ? __builtin_sqrtf((kx * kx + ky * ky) + kz * kz) : 0.0f;
            rq = group_max(rk) + g.extent;                            // beyond it: no influence
            cut2 = 3.4e38f;                                           // the first chunk is walked in full (min_d2 needs a start)
        }
        for (int cb = cb_lo; cb < cb_hi; cb += CB) {
            f32x4v acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
            const int ch = cb + NT * i;                           // first channel of this lane
            const bool chok = VECROW ? (ch < ci) : (ch < ci);
            for (int h0 = 0; h0 < h; h0 += 64) {
                int idx = idx0;
                float px = p0x, py = p0y, pz = p0z;
                if (h0 > 0) {
                    idx = load_idx(qv0, h0 + lane);
                    load_pt(idx, px, py, pz);
                }
                wave_lds_order();                                 // the previous chunk's readers are done
                if constexpr (MODE == 0 || MODE == 2) {
                    // ---- MODE 2 (deformable + modulated, linear, sum; BASELINE config 5) shares this form: the lane's kernel
                    // point is re-loaded per query (one float4 with its modulation), the modulation multiplies the influence
                    // (A operand) instead of the finished row, min_d2 is tracked per lane.  The in-range filter of
                    // blocks.py:301-325 needs no code here: with the linear influence a neighbour without a kernel point
                    // inside the extent has all 15 influences equal to zero already.  Columns past the row are staged
                    // far beyond the shadow point so that they never win the minimum; shadow columns take part in it with
                    // the reference's coordinates (1e6 - q, blocks.py:278-284).
                    // ---- rigid / linear / sum: the instruction-lean form.  This kernel is bound by vector-instruction issue
                    // (SQ counters: the SIMDs issue ~88 % of the time, 435 VALU per query in the first form), so everything
                    // that can be decided once per neighbour is decided by the lane that stages it, not by the 16 lanes that
                    // consume it:  * a column that is not a real neighbour (shadow index, past the row) is staged as the
                    // shadow point (1e6,1e6,1e6) with row offset 0 -- its linear influence is exactly 0, no mask per step;
                    // * the row's element offset idx*ci is computed at staging;  * channel lanes past ci load row 0 and feed
                    // output columns that are never stored, no select;  * two steps share every arithmetic instruction
                    // (v_pk_* on float2): the staging layout puts (x_a,x_b,y_a,y_b) and (z_a,z_b,off_a,off_b) of the step
                    // pair (a, b) of a row group side by side, one ds_read_b128 each.
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    float* nbf = reinterpret_cast<float*>(nb);       // [pair p = 0..7][kk = 0..3][8 floats]
                    float dn2 = 0.0f;
                    {
                        const bool real = idx >= 0 && (h0 + lane < h);
                        float sx = real ? px - qx : WS_SHADOW, sy = real ? py - qy : WS_SHADOW, sz = real ? pz - qz : WS_SHADOW;
                        if constexpr (MODE == 2) {
                            const bool incol = h0 + lane < h;
                            sx = incol ? px - qx : 3.0e18f; sy = incol ? py - qy : 3.0e18f; sz = incol ? pz - qz : 3.0e18f;
                        }
                        const unsigned off = real && !(g.ablate & 2) ? (unsigned)idx * (unsigned)ci : 0u;
                        const int sstep = lane >> 2, skk = lane & 3;  // this lane's neighbour is column 4 sstep + skk
                        float* dst = nbf + (((sstep >> 1) * 4 + skk) * 8) + (sstep & 1);
                        dst[0] = sx; dst[2] = sy; dst[4] = sz; dst[6] = __uint_as_float(off);
                        if constexpr (CUT) dn2 = real ? (sx * sx + sy * sy) + sz * sz : 3.4e38f;
                    }
                    int ncut = 64;
                    if constexpr (CUT) {
                        // sorted row: the columns inside the cutoff are a prefix of the chunk
                        ncut = __builtin_popcountll(__ballot(dn2 <= cut2));
                        if (ncut == 0 && (MODE == 0 || h0 > 0)) break;
                    }
                    wave_lds_order();
                    const int cols = min(min(64, h - h0), ncut);
                    const int steps = (cols + 3) >> 2;
                    const int npairs = (steps + 1) >> 1;
                    constexpr int GP = GS >= 2 ? GS / 2 : 1;         // step pairs whose loads are in flight together
                    const T* xlane = x + (chok ? ch : 0);
                    f2 wb2[2][GP];
                    float xb[2][2 * GP][NT];
                    auto load_pairs = [&](int gi, int slot) {
                        float4 va[GP], vb[GP];
#pragma unroll
                        for (int u = 0; u < GP; ++u) {
                            const float4* src = reinterpret_cast<const float4*>(nbf + ((gi * GP + u) * 4 + kk) * 8);
                            va[u] = src[0];
                            vb[u] = src[1];
                        }
#pragma unroll
                        for (int u = 0; u < GP; ++u) {
                            RowLoad<NT, T>::ld(xlane + __float_as_uint(vb[u].z), xb[slot][2 * u]);
                            RowLoad<NT, T>::ld(xlane + __float_as_uint(vb[u].w), xb[slot][2 * u + 1]);
                        }
#pragma unroll
                        for (int u = 0; u < GP; ++u) {
                            const f2 dx = f2{va[u].x, va[u].y} - kx, dy = f2{va[u].z, va[u].w} - ky, dz = f2{vb[u].x, vb[u].y} - kz;
                            const f2 d2 = (dx * dx + dy * dy) + dz * dz;
                            const f2 sd = f2{__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
                            const f2 w = 1.0f - sd * inv_extent;
                            wb2[slot][u] = f2{fmaxf(w.x, 0.0f), fmaxf(w.y, 0.0f)};
                            if constexpr (MODE == 2) {
                                wb2[slot][u] = wb2[slot][u] * kmod;
                                mind = fminf(mind, fminf(d2.x, d2.y));
                            }
                        }
                    };
                    auto comp_pairs = [&](int slot) {
#pragma unroll
                        for (int u = 0; u < GP; ++u) {
#pragma unroll
                            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb2[slot][u].x, xb[slot][2 * u][t], acc[t], 0, 0, 0);
#pragma unroll
                            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb2[slot][u].y, xb[slot][2 * u + 1][t], acc[t], 0, 0, 0);
                        }
                    };
                    const int ngroups = (npairs + GP - 1) / GP;
                    load_pairs(0, 0);
                    for (int gi = 0; gi < ngroups; gi += 2) {
                        if (gi + 1 < ngroups) load_pairs(gi + 1, 1);
                        comp_pairs(0);
                        if (gi + 1 < ngroups) {
                            if (gi + 2 < ngroups) load_pairs(gi + 2, 0);
                            comp_pairs(1);
                        }
                    }
                    if constexpr (CUT) {
                        if (ncut < 64) break;                         // the row left the cutoff inside this chunk
                        if constexpr (MODE == 2) {
                            if (h0 == 0) {
                                // from here on a neighbour matters only inside max(influence reach, what could still lower a minimum)
                                float m = fminf(mind, __shfl_xor(mind, 16, 64));
                                m = fminf(m, __shfl_xor(m, 32, 64));
                                const float reach = group_max(haskp ? __builtin_sqrtf(m) + rk : 0.0f);
                                const float R = fmaxf(rq, reach) * 1.0001f;
                                cut2 = R * R;
                            }
                        }
                    }
                    continue;
                }
                nb[lane] = make_float4(px - qx, py - qy, pz - qz, __int_as_float((h0 + lane < h) ? (idx >= 0 ? idx : -2) : -1));
                wave_lds_order();
                const int cols = min(64, h - h0);
                const int steps = (cols + 3) >> 2;
                float wb[2][GS];
                float xb[2][GS][NT];
                // three separate passes per group so that nothing waits between the independent accesses: all LDS
                // broadcasts, then all row loads (addresses from the broadcasts), then the weights (VALU under the loads);
                // selects instead of branches throughout (an exec-mask branch per step serialises the LDS round trips)
                auto load_group = [&](int gi, int slot) {
                    float4 nv[GS];
#pragma unroll
                    for (int u = 0; u < GS; ++u)
                        nv[u] = nb[min(4 * (gi * GS + u) + kk, 63)];
#pragma unroll
                    for (int u = 0; u < GS; ++u) {
                        const int s = gi * GS + u;
                        const int nidx = __float_as_int(nv[u].w);     // >= 0 real, -2 shadow column, -1 past the row
                        const bool live = nidx >= 0 && s < steps;
                        const unsigned row = (live && !(g.ablate & 2)) ? (unsigned)nidx : 0u;
                        const T* src = x + (size_t)(row * (unsigned)ci) + (chok ? ch : 0);
                        RowLoad<NT, T>::ld(src, xb[slot][u]);
                    }
#pragma unroll
                    for (int u = 0; u < GS; ++u) {
                        const int s = gi * GS + u;
                        const float4 n = nv[u];
                        const int nidx = __float_as_int(n.w);
                        const bool live = nidx >= 0 && s < steps;
                        // influence of this lane's kernel point on that neighbour
                        const float dx = n.x - kx, dy = n.y - ky, dz = n.z - kz;
                        const float d2 = (dx * dx + dy * dy) + dz * dz;
                        float w;
                        if (MODE == 0) {
                            w = fmaxf(1.0f - __builtin_amdgcn_sqrtf(d2) * inv_extent, 0.0f);
                        } else {
                            w = kp_weight(d2, g, inv_extent);
                            if (g.aggregation == WS_AGGREGATION_CLOSEST) {
                                // arg-min kernel point of this neighbour over the 16 lanes of the row group (first wins)
                                float bd = haskp ? d2 : 3.4e38f;
                                int bi = i;
#pragma unroll
                                for (int o = 1; o < 16; o <<= 1) {
                                    const float od = __shfl_xor(bd, o, 64);
                                    const int oi = __shfl_xor(bi, o, 64);
                                    const bool take = od < bd || (od == bd && oi < bi);
                                    bd = take ? od : bd;
                                    bi = take ? oi : bi;
                                }
                                w = bi != i ? 0.0f : w;
                            }
                            if (DEF) {
                                const unsigned long long m = __ballot(haskp && d2 < e2);
                                w = ((((unsigned)(m >> (16 * kk))) & 0xffffu) == 0u) ? 0.0f : w;     // no kernel point in range (blocks.py:301-325)
                                mind = (nidx != -1 && s < steps && haskp) ? fminf(mind, d2) : mind;
                            }
                        }
                        wb[slot][u] = (live && haskp) ? w : 0.0f;
                    }
                };
                auto comp_group = [&](int slot) {
#pragma unroll
                    for (int u = 0; u < GS; ++u)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[slot][u], chok ? xb[slot][u][t] : 0.0f, acc[t], 0, 0, 0);
                };
                const int ngroups = (steps + GS - 1) / GS;
                load_group(0, 0);
                for (int gi = 0; gi < ngroups; gi += 2) {
                    if (gi + 1 < ngroups) load_group(gi + 1, 1);
                    comp_group(0);
                    if (gi + 1 < ngroups) {
                        if (gi + 2 < ngroups) load_group(gi + 2, 0);
                        comp_group(1);
                    }
                }
            }
            // D: lane (j = i, g = kk) holds kernel points 4 kk + r of its NT channels
            if constexpr (FUSE) {
                __shared__ float tile_all[16 * TILE_LD];
                __shared__ int tq_all[16];
                const int slot = 4 * (iter & 3) + wave;
                const bool valid = item < iend;
                if (lane == 0) tq_all[slot] = valid ? (int)q : -1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * kk + r;
                    if (k < K) *reinterpret_cast<float2*>(&tile_all[slot * TILE_LD + k * 32 + 2 * i]) = make_float2(acc[0][r], acc[1][r]);
                }
                if ((iter & 3) == 3 || item + 4 >= iend_u) {
                    __shared__ float part_all[4 * 16 * 32];
                    __syncthreads();                              // the tile is complete
                    f32x4v o0 = f32x4v{0.f, 0.f, 0.f, 0.f}, o1 = o0;
                    const float* arow = &tile_all[i * TILE_LD + 120 * wave + kk];
#pragma unroll
                    for (int s_ = 0; s_ < 30; ++s_) {
                        const float a = arow[4 * s_];
                        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wq[s_][0], o0, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wq[s_][1], o1, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                 // D: lane (j, g) holds queries 4 g + r of output columns j, 16 + j
                        part_all[(wave * 16 + 4 * kk + r) * 32 + i] = o0[r];
                        part_all[(wave * 16 + 4 * kk + r) * 32 + 16 + i] = o1[r];
                    }
                    __syncthreads();
#pragma unroll
                    for (int e = threadIdx.x; e < 512; e += 256) {
                        const int qs = e >> 5, c = e & 31;
                        const int qq = tq_all[qs];
                        float v = (part_all[(0 * 16 + qs) * 32 + c] + part_all[(1 * 16 + qs) * 32 + c]) +
                                  (part_all[(2 * 16 + qs) * 32 + c] + part_all[(3 * 16 + qs) * 32 + c]);
                        if (fz.bias) v += fz.bias[c];
                        if (fz.act) v = v > 0.0f ? v : v * fz.slope;
                        // slots this tile did not fill (a short last tile) keep the index of an older query: masked by the count
                        const bool filled = qs < 4 * ((iter & 3) + 1);
                        if (qq >= 0 && filled) fz.out[(int64_t)qq * 32 + c] = v;
                    }
                    __syncthreads();                              // tile and partials are free again
                }
                ++iter;
            } else if (chok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * kk + r;
                    if (k < K) {
                        float v[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) v[t] = acc[t][r];
                        if (MODE != 0 && modulations) {
                            const float md = modulations[q * K + k];
#pragma unroll
                            for (int t = 0; t < NT; ++t) v[t] *= md;
                        }
                        if (!(g.ablate & 1) || v[0] == 1.2345e30f) RowLoad<NT, T>::st(wf + (q * K + k) * ci + ch, v);
                    }
                }
            }
        }
        if (DEF && min_d2) {
            mind = fminf(mind, __shfl_xor(mind, 16, 64));
            mind = fminf(mind, __shfl_xor(mind, 32, 64));
            if (kk == 0 && haskp) min_d2[q * K + i] = mind;
        }
        // rotate; the checks / selects on the prefetched values happen here, one item after their loads were issued
        p0x = idx1 >= 0 ? p1x : WS_SHADOW; p0y = idx1 >= 0 ? p1y : WS_SHADOW; p0z = idx1 >= 0 ? p1z : WS_SHADOW;
        idx0 = idx1;
        idx1 = chk(raw2);
        qv0 = qv1; qv1 = qv2; qv2 = qv3;
        qx = q1x; qy = q1y; qz = q1z;
        q1x = q2x; q1y = q2y; q1z = q2z;
    }
}

// ---------------------------------------------------------------------------------------------
// K4 backward w.r.t. x through the transposed table: dx[s, :] = sum_e weight_e * dwf[row_e, :]
// with row_e = q*K + k.  One wave per support; the flush is balanced over the S slots
// (slot s takes entries [s*per, (s+1)*per)) and the S partial sums are combined by shuffles.
// ---------------------------------------------------------------------------------------------
template <int K, int G, int MODE, bool VEC, typename T = float>
__global__ __launch_bounds__(256) void kpconv_gather_bwd_x_kernel(
    const float* __restrict__ q_pts, int64_t nq, const float* __restrict__ s_pts, int64_t ns,
    int h, const int32_t* __restrict__ t_offsets, const int32_t* __restrict__ t_pairs,
    const T* __restrict__ dwf, int ci, const float* __restrict__ kernel_points,
    const float* __restrict__ deformed_kp, const float* __restrict__ modulations, GeomParams g,
    T* __restrict__ dx, const int32_t* __restrict__ order)
{
    constexpr int CC = 4 * G;
    constexpr int S = 64 / G;
    static_assert(K <= 16 && POOL >= 16 * K, "pool sizing");
    __shared__ uint2 pool_all[4][POOL_ALLOC];
    __shared__ int segs_all[4][K + 1];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint2* pool = pool_all[wave];
    int* segs = segs_all[wave];
    const int j = lane % G;
    const int slot = lane / G;
    const float inv_extent = 1.0f / g.extent;

    float kpr[MODE == 0 ? 3 * K : 1];
    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 3 * K; ++t) kpr[t] = kernel_points[t];
    }

    int64_t item0, istep, iend;
    ws_wave_items(ns, order ? g.ilv : 0, wave, item0, istep, iend);
    for (int64_t item = item0; item < iend; item += istep) {
        const int64_t s = order ? (int64_t)order[item] : item;
        const float sx = s_pts[3 * s + 0], sy = s_pts[3 * s + 1], sz = s_pts[3 * s + 2];
        const int beg = t_offsets[s], end = t_offsets[s + 1];
        for (int cc0 = 0; cc0 < ci; cc0 += CC) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int ch = cc0 + 4 * j;
            const bool chok = VEC ? (ch + 3 < ci) : (ch < ci);
            const int chl = chok ? ch : 0;
            // the gate row (GeomParams::gate) is fetched here, ahead of the walk: at the store its latency has long passed
            float4 gq = make_float4(1.f, 1.f, 1.f, 1.f);
            if (g.gate && slot == 0) {
                const T* gy = reinterpret_cast<const T*>(g.gate) + s * ci + ch;
                if (VEC) {
                    if (chok) gq = ld4(gy);
                } else {
                    if (ch + 0 < ci) gq.x = ld1(gy + 0);
                    if (ch + 1 < ci) gq.y = ld1(gy + 1);
                    if (ch + 2 < ci) gq.z = ld1(gy + 2);
                    if (ch + 3 < ci) gq.w = ld1(gy + 3);
                }
            }
            auto flush = [&](int total) {
                wave_lds_sync();
                total = min(total, POOL);
                const int per = (total + S - 1) / S;
                const int lo = slot * per;
                const int hi = chok ? min(lo + per, total) : lo;
                for (int it = 0; it < per; it += 4) {
                    float4 v[4];
                    float w[4];
                    uint2 e[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        e[u] = pool[min(lo + it + u, POOL + 7)];
                        keep_unconditional(e[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool ok = lo + it + u < hi;
                        e[u].x = ok ? e[u].x : 0u;
                        w[u] = ok ? __uint_as_float(e[u].y) : 0.0f;
                        v[u] = load_row_piece<VEC>(dwf, e[u].x, ci, chl);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc.x = fmaf(w[u], v[u].x, acc.x);
                        acc.y = fmaf(w[u], v[u].y, acc.y);
                        acc.z = fmaf(w[u], v[u].z, acc.z);
                        acc.w = fmaf(w[u], v[u].w, acc.w);
                    }
                }
                wave_lds_sync();
            };
            for (int p0 = beg; p0 < end; p0 += 64) {
                const int p = p0 + lane;
                const bool real = p < end;
                const int pair = real ? t_pairs[p] : 0;
                const int q = pair / h;
                const float nx = sx - q_pts[3 * (int64_t)q + 0];
                const float ny = sy - q_pts[3 * (int64_t)q + 1];
                const float nz = sz - q_pts[3 * (int64_t)q + 2];
                const float* lkp = deformed_kp ? deformed_kp + (int64_t)q * (3 * K) : kernel_points;   // per lane if deformed
                const float* lmod = modulations ? modulations + (int64_t)q * K : nullptr;
                auto kp = [&](int k, int c) { return MODE == 0 ? kpr[MODE == 0 ? 3 * k + c : 0] : lkp[3 * k + c]; };
                auto nomin = [&](int, float) {};
                int total = 0, maxlen = 0;
                if constexpr (MODE == 2) {
                    const float4* kq = g.kp4 + (int64_t)q * K;
                    kp_list_def<K>(nx, ny, nz, real, kq, inv_extent, (unsigned)(q * K), pool, segs, lane, total, maxlen);
                    if (total <= POOL) {
                        flush(total);
                    } else {
                        for (int sub = 0; sub < 4; ++sub) {
                            total = 0; maxlen = 0;
                            kp_list_def<K>(nx, ny, nz, real && (lane >> 4) == sub, kq, inv_extent, (unsigned)(q * K), pool, segs, lane,
                                           total, maxlen);
                            flush(total);
                        }
                    }
                    continue;
                }
                kp_list<K, MODE == 2 ? 1 : MODE>(nx, ny, nz, real, kp, g, inv_extent, (unsigned)(q * K), 1u, lmod, pool, segs, lane, total,
                                 maxlen, nomin);
                if (total <= POOL) {
                    flush(total);
                } else {
                    for (int sub = 0; sub < 4; ++sub) {
                        total = 0; maxlen = 0;
                        kp_list<K, MODE == 2 ? 1 : MODE>(nx, ny, nz, real && (lane >> 4) == sub, kp, g, inv_extent, (unsigned)(q * K), 1u, lmod,
                                         pool, segs, lane, total, maxlen, nomin);
                        flush(total);
                    }
                }
            }
            // sum the S slots
#pragma unroll
            for (int o = G; o < 64; o <<= 1) {
                acc.x += __shfl_xor(acc.x, o, 64);
                acc.y += __shfl_xor(acc.y, o, 64);
                acc.z += __shfl_xor(acc.z, o, 64);
                acc.w += __shfl_xor(acc.w, o, 64);
            }
            if (slot == 0) {
                T* dst = dx + s * ci + ch;
                if (g.gate) {
                    acc.x *= gq.x > 0.0f ? 1.0f : g.gate_slope; acc.y *= gq.y > 0.0f ? 1.0f : g.gate_slope;
                    acc.z *= gq.z > 0.0f ? 1.0f : g.gate_slope; acc.w *= gq.w > 0.0f ? 1.0f : g.gate_slope;
                }
                if (VEC) {
                    if (chok) st4(dst, acc);
                } else {
                    if (ch + 0 < ci) st1(dst + 0, acc.x);
                    if (ch + 1 < ci) st1(dst + 1, acc.y);
                    if (ch + 2 < ci) st1(dst + 2, acc.z);
                    if (ch + 3 < ci) st1(dst + 3, acc.w);
                }
            }
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
template <int K, typename T = float>
__global__ __launch_bounds__(256) void kpconv_gather_bwd_geom_kernel(
    const float* __restrict__ q_pts, int64_t nq, const float* __restrict__ s_pts, int64_t ns,
    const int64_t* __restrict__ inds, int h, const T* __restrict__ x, int ci,
    const T* __restrict__ dwf, const float* __restrict__ deformed_kp,
    const float* __restrict__ modulations, const float* __restrict__ d_min_d2, GeomParams g,
    float* __restrict__ d_kp, float* __restrict__ d_mod, int vec4)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
                    const T* a = dwf + (q * K + k) * ci;
                    const T* b = x + idx * ci;
                    if (vec4) {
                        // 16-byte pieces, four independent partial sums (the per-lane rows are L2 resident)
                        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (int cc = 0; cc < ci; cc += 4) {
                            const float4 av = ld4(a + cc);
                            const float4 bv = ld4(b + cc);
                            s4.x = fmaf(av.x, bv.x, s4.x); s4.y = fmaf(av.y, bv.y, s4.y);
                            s4.z = fmaf(av.z, bv.z, s4.z); s4.w = fmaf(av.w, bv.w, s4.w);
                        }
                        dot = (s4.x + s4.y) + (s4.z + s4.w);
                    } else {
                        for (int cc = 0; cc < ci; ++cc) dot = fmaf(ld1(a + cc), ld1(b + cc), dot);
                    }
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

// ---------------------------------------------------------------------------------------------
// K6 on the matrix core (deformable fast path: linear influence, sum aggregation, packed kernel points).
// What the geometry backward needs per (query, neighbour h, kernel point k) is the scalar
//     dot[k][h] = sum_c dwf[q,k,c] * x[inds[q,h], c]
// -- per query a dense product  dWF_q (15 x Ci) . X_q^T (Ci x H).  The VALU form above evaluates it only where the
// influence is live, one serial Ci-long chain per lane over two scattered rows (24 ms per config-5 level-0 launch).
// Here it runs as v_mfma_f32_16x16x4_f32 over blocks of 16 neighbours:
//   A  lane (i = lane&15, kk = lane>>4): CK = Ci/4 consecutive channels kk*CK .. of dwf[q, i, :], in registers for the
//      whole query (Ci <= 128; wider rows re-load A per block: those levels hold a few hundred points);
//   B  lane (j = lane&15, kk): the same channels of x[inds[q, 16 b + j], :]  (16 .. 64 contiguous bytes per lane);
//   D  lane (j, g = lane>>4) ends with dot[4g + r][16 b + j], r = 0..3 -- exactly the (neighbour, kernel point) pairs
//      whose geometry that lane then evaluates: influence, d influence / d kernel point, running arg-min of d2.
// Per query: reduce the 4 x (3 + 1) sums over the 16 lanes of a group, add the min_d2 path at the arg-min column,
// one float4 store (d x, d y, d z, d modulation) per kernel point.  Sums are per-lane partial sums combined by a fixed
// shuffle tree: deterministic.
// ---------------------------------------------------------------------------------------------
template <int CK, bool AREG, typename T, bool CUT = false>
__global__ __launch_bounds__(256) void kpconv_gather_bwd_geom_def_kernel(
    const float* __restrict__ q_pts, int64_t nq, const float* __restrict__ s_pts, int64_t ns,
    const int64_t* __restrict__ inds, int h, const T* __restrict__ x, int ci, const T* __restrict__ dwf,
    const float4* __restrict__ kp4, const float* __restrict__ d_min_d2, float extent, float4* __restrict__ d_kp4,
    const int32_t* __restrict__ order, int ilv)
{
    constexpr int K = 15;
    constexpr int CB = 4 * CK;                                   // channels per pass of the product
    __shared__ float4 nb_all[4][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;                       // (= i, kk of the A / B operands)
    float4* nb = nb_all[wave];
    const float inv_extent = 1.0f / extent;
    int64_t item0, istep, iend;
    ws_wave_items(nq, order ? ilv : 0, wave, item0, istep, iend);
    for (int64_t item = item0; item < iend; item += istep) {
        const int64_t q = order ? (int64_t)order[item] : item;
        float4 kq[4];
        float cm[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * g + r;
            kq[r] = k < K ? kp4[q * K + k] : make_float4(1.0e9f, 1.0e9f, 1.0e9f, 0.0f);
            cm[r] = kq[r].w * inv_extent;
        }
        const float qx = q_pts[3 * q + 0], qy = q_pts[3 * q + 1], qz = q_pts[3 * q + 2];
        float a[CK];
        auto load_a = [&](int cb) {
            RowLoad<CK, T>::ld(dwf + (q * K + (j < K ? j : 0)) * ci + cb + g * CK, a);
            if (j >= K) {
#pragma unroll
                for (int t = 0; t < CK; ++t) a[t] = 0.0f;
            }
        };
        if (AREG) load_a(0);
        float gk[4][3], gm[4], best[4];
        int bestcol[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gk[r][0] = gk[r][1] = gk[r][2] = 0.0f;
            gm[r] = 0.0f;
            best[r] = 3.4e38f;
            bestcol[r] = 0x7fffffff;
        }
        // index / coordinates of the first 64 columns
        auto fetch = [&](int h0, int& idx, float& px, float& py, float& pz) {
            const int col = h0 + lane;
            idx = -1;
            if (col < h) { const int64_t v = inds[q * h + col]; idx = (v >= 0 && v < ns) ? (int)v : -1; }
            px = py = pz = WS_SHADOW;
            if (idx >= 0) { px = s_pts[3 * (int64_t)idx]; py = s_pts[3 * (int64_t)idx + 1]; pz = s_pts[3 * (int64_t)idx + 2]; }
        };
        int idx; float px, py, pz;
        fetch(0, idx, px, py, pz);
        float cut2 = 3.4e38f;                                     // CUT: see kpconv_gather_fwd_mfma_kernel (sorted rows)
        for (int h0 = 0; h0 < h; h0 += 64) {
            wave_lds_order();                                     // the previous chunk's readers are done
            int ncut = 64;
            {
                const bool incol = h0 + lane < h;                 // past the row: far beyond the shadow point (never the minimum)
                const float ox = px - qx, oy = py - qy, oz = pz - qz;
                nb[lane] = make_float4(incol ? ox : 3.0e18f, incol ? oy : 3.0e18f, incol ? oz : 3.0e18f, __int_as_float(idx));
                if constexpr (CUT) {
                    const float dn2 = (incol && idx >= 0) ? (ox * ox + oy * oy) + oz * oz : 3.4e38f;
                    ncut = __builtin_popcountll(__ballot(dn2 <= cut2));
                    if (ncut == 0 && h0 > 0) break;
                }
            }
            wave_lds_order();
            if (h0 + 64 < h && ncut == 64) fetch(h0 + 64, idx, px, py, pz);     // next chunk's chain under this chunk's work
            const int nblk = min(4, (min(h - h0, ncut) + 15) >> 4);
            float bv[2][CK];
            float4 nv[2];
            auto load_b = [&](int b4, int slot) {
                nv[slot] = nb[16 * b4 + j];
                if (AREG) {
                    const int nidx = __float_as_int(nv[slot].w);
                    RowLoad<CK, T>::ld(x + (size_t)((unsigned)(nidx >= 0 ? nidx : 0) * (unsigned)ci) + g * CK, bv[slot]);
                }
            };
            auto compute = [&](int b4, int slot) {
                f32x4v acc = f32x4v{0.f, 0.f, 0.f, 0.f};
                const float4 n = nv[slot];
                if (AREG) {
#pragma unroll
                    for (int t = 0; t < CK; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bv[slot][t], acc, 0, 0, 0);
                } else {
                    const int nidx = __float_as_int(n.w);
                    const T* xr = x + (size_t)((unsigned)(nidx >= 0 ? nidx : 0) * (unsigned)ci) + g * CK;
                    for (int cb = 0; cb < ci; cb += CB) {
                        load_a(cb);
                        float bb[CK];
                        RowLoad<CK, T>::ld(xr + cb, bb);
#pragma unroll
                        for (int t = 0; t < CK; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bb[t], acc, 0, 0, 0);
                    }
                }
                const int col = h0 + 16 * b4 + j;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dx = n.x - kq[r].x, dy = n.y - kq[r].y, dz = n.z - kq[r].z;
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    const float rs = __builtin_amdgcn_rsqf(d2);
                    const bool nz = d2 > 0.0f;
                    const float sd = nz ? d2 * rs : 0.0f;
                    const float t = 1.0f - sd * inv_extent;
                    const float w = fmaxf(t, 0.0f);
                    const float dot = acc[r];
                    gm[r] = fmaf(w, dot, gm[r]);
                    // w = 1 - |n - kp| / extent  ->  d w / d kp = (n - kp) / (extent |n - kp|), where 0 < w
                    const float coef = (t > 0.0f && nz) ? dot * cm[r] * rs : 0.0f;
                    gk[r][0] = fmaf(coef, dx, gk[r][0]);
                    gk[r][1] = fmaf(coef, dy, gk[r][1]);
                    gk[r][2] = fmaf(coef, dz, gk[r][2]);
                    const bool lower = d2 < best[r];               // columns ascend per lane: the first minimum stays
                    best[r] = lower ? d2 : best[r];
                    bestcol[r] = lower ? col : bestcol[r];
                }
            };
            load_b(0, 0);
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                if (b4 < nblk) {
                    if (b4 + 1 < nblk) load_b(b4 + 1, (b4 + 1) & 1);
                    compute(b4, b4 & 1);
                }
            }
            if constexpr (CUT) {
                if (ncut < 64) break;                             // the row left the cutoff inside this chunk
                if (h0 == 0) {
                    // influence reach and what could still lower a minimum, over this query's 15 kernel points
                    float reach = 0.0f, rq = 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float bm = best[r];
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) bm = fminf(bm, __shfl_xor(bm, o, 64));
                        if (4 * g + r < K) {
                            const float rk = __builtin_sqrtf((kq[r].x * kq[r].x + kq[r].y * kq[r].y) + kq[r].z * kq[r].z);
                            rq = fmaxf(rq, rk);
                            reach = fmaxf(reach, __builtin_sqrtf(bm) + rk);
                        }
                    }
                    float R = fmaxf(rq + extent, reach);
                    R = fmaxf(R, __shfl_xor(R, 16, 64));
                    R = fmaxf(R, __shfl_xor(R, 32, 64));
                    R *= 1.0001f;
                    cut2 = R * R;
                }
            }
        }
        // ---- per kernel point: sums over the 16 lanes of the group, arg-min of d2 (first column wins ties)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                gk[r][0] += __shfl_xor(gk[r][0], o, 64);
                gk[r][1] += __shfl_xor(gk[r][1], o, 64);
                gk[r][2] += __shfl_xor(gk[r][2], o, 64);
                gm[r] += __shfl_xor(gm[r], o, 64);
                const float ob = __shfl_xor(best[r], o, 64);
                const int oc = __shfl_xor(bestcol[r], o, 64);
                const bool take = ob < best[r] || (ob == best[r] && oc < bestcol[r]);
                best[r] = take ? ob : best[r];
                bestcol[r] = take ? oc : bestcol[r];
            }
            const int k = 4 * g + r;
            if (k < K) {
                float gx = gk[r][0], gy = gk[r][1], gz = gk[r][2];
                if (d_min_d2) {
                    // d min_d2[q,k] / d kp = 2 (kp - n) at the arg-min column (shadow columns take part, blocks.py:278-304)
                    const int64_t v = inds[q * h + min(bestcol[r], h - 1)];
                    float mx = WS_SHADOW, my = WS_SHADOW, mz = WS_SHADOW;
                    if (v >= 0 && v < ns) { mx = s_pts[3 * v]; my = s_pts[3 * v + 1]; mz = s_pts[3 * v + 2]; }
                    const float gmin = 2.0f * d_min_d2[q * K + k];
                    gx = fmaf(gmin, kq[r].x - (mx - qx), gx);
                    gy = fmaf(gmin, kq[r].y - (my - qy), gy);
                    gz = fmaf(gmin, kq[r].z - (mz - qz), gz);
                }
                if (j == 0) d_kp4[q * K + k] = make_float4(gx, gy, gz, gm[r]);
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

// ---------------------------------------------------------------------------------------------
// K4 without a transposed table (self-query layers: q_pts == s_pts, the same cloud on both sides).
// The pairs that reach support s are re-derived from the cell grid the neighbour search built:
//   q -> s is a pair  <=>  d2(q, s) < r^2  and  (d2(q, s), s) <= key_last[q],
// key_last[q] being the (distance, index) key of the last neighbour K1 kept in q's row ("infinity" for a
// row that was not truncated).  The distance is symmetric bit for bit (ws_grid.h: ref_d2), so the wave of s
// walks the 3x3x3 cell block exactly as K1 does, keeps the candidates that pass the membership test, orders
// them by index (= ascending pair id, the order of the transposed table, so the sums are the same bits) and
// runs K4's list / flush on them.  No counting-sort atomics, no scattered fill, no per-list sort: the
// table build of the big level-0 matrix (1.6 ms per step) disappears.
// ---------------------------------------------------------------------------------------------
constexpr int GRID_SLAB = 192;      // incoming pairs per support (the search reports rows up to 128)

template <int K, int G, int MODE, bool VEC, typename T = float, bool SORT = true>
__global__ __launch_bounds__(256) void kpconv_gather_bwd_x_grid_kernel(
    const float* __restrict__ s_pts, int64_t ns, const CloudGrid* __restrict__ grids, int nb,
    const int32_t* __restrict__ cell_start, const float4* __restrict__ sorted,
    const unsigned long long* __restrict__ key_last, float r2, const T* __restrict__ dwf, int ci,
    const float* __restrict__ kernel_points, const float* __restrict__ deformed_kp, const float* __restrict__ modulations,
    GeomParams g, T* __restrict__ dx, const int32_t* __restrict__ order, int32_t* __restrict__ overflow)
{
    constexpr int CC = 4 * G;
    constexpr int S = 64 / G;
    constexpr int EPL = GRID_SLAB / 64;          // slab entries per lane
    __shared__ uint2 pool_all[4][POOL_ALLOC];
    __shared__ int segs_all[4][K + 1];
    __shared__ float4 slab_all[4][GRID_SLAB + 64];      // + one dummy slot per lane (unconditional writes)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint2* pool = pool_all[wave];
    int* segs = segs_all[wave];
    float4* slab = slab_all[wave];
    const int j = lane % G;
    const int slot = lane / G;
    const float inv_extent = 1.0f / g.extent;

    float kpr[MODE == 0 ? 3 * K : 1];
    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 3 * K; ++t) kpr[t] = kernel_points[t];
    }

    int64_t item0, istep, iend;
    ws_wave_items(ns, order ? g.ilv : 0, wave, item0, istep, iend);
    CloudGrid gr = grids[0];        // supports come cloud by cloud: the element's grid stays in registers
    for (int64_t item = item0; item < iend; item += istep) {
        const int64_t s = order ? (int64_t)order[item] : item;
        if (s < gr.s_base || s >= gr.s_base + gr.s_len) {
            int b = 0;
            while (b + 1 < nb && s >= grids[b].s_base + grids[b].s_len) ++b;
            gr = grids[b];
        }
        const float sx = s_pts[3 * s + 0], sy = s_pts[3 * s + 1], sz = s_pts[3 * s + 2];
        // ---- candidates: the 9 cell runs around s, membership test, compaction into the slab
        int cnt = 0;
        auto take = [&](const float4& c, bool active) {
            // K1 evaluated this pair with c as the query and s as the support: diff = query - support
            const float d2 = ref_d2(c.x, c.y, c.z, make_float4(sx, sy, sz, 0.0f));
            const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)s;
            bool hit = active && d2 < r2;
            hit = hit && key <= key_last[hit ? __float_as_int(c.w) : 0];          // unconditional gather
            const unsigned long long m = __ballot(hit);
            const int pos = cnt + lane_rank(m);
            slab[(hit && pos < GRID_SLAB) ? pos : GRID_SLAB + lane] = c;      // branch free (dummy slot per lane)
            cnt += __builtin_popcountll(m);
        };
        // a support with an untruncated row of its own: every point within the radius is in that row (the distance is
        // symmetric bit for bit), so the queries that kept s are among its entries -- ~60 candidates instead of the ~700 of
        // the 27-cell walk.  Wave-uniform branch (s is).
        const bool from_row = g.rows != nullptr && key_last[s] == ~0ull;
        if (from_row) {
            for (int h0 = 0; h0 < g.rows_h; h0 += 64) {
                const int col = h0 + lane;
                const int64_t qi = col < g.rows_h ? g.rows[s * g.rows_h + col] : -1;
                const bool active = qi >= 0 && qi < ns;
                const int64_t qq = active ? qi : 0;
                take(make_float4(s_pts[3 * qq + 0], s_pts[3 * qq + 1], s_pts[3 * qq + 2], __int_as_float((int)qq)), active);
            }
        } else if (gr.s_len > 0) {
            const int cx = cell_coord(sx, gr.lo[0], gr.inv_cell);
            const int cy = cell_coord(sy, gr.lo[1], gr.inv_cell);
            const int cz = cell_coord(sz, gr.lo[2], gr.inv_cell);
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, gr.nx - 1);
            int rb[9], re[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
                const bool ok = x0 <= x1 && z >= 0 && z < gr.nz && y >= 0 && y < gr.ny;
                const int row = gr.cell_base + ((ok ? z : 0) * gr.ny + (ok ? y : 0)) * gr.nx;
                rb[r] = ok ? cell_start[row + x0] : 0;
                re[r] = ok ? cell_start[row + x1 + 1] : 0;
            }
            float4 c[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int p = rb[r] + lane;
                c[r] = sorted[p < re[r] ? p : 0];          // unconditional, masked by `active`
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                take(c[r], rb[r] + lane < re[r]);
                for (int p0 = rb[r] + 64; p0 < re[r]; p0 += 64) {
                    const int p = p0 + lane;
                    float4 cc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p < re[r]) cc = sorted[p];
                    take(cc, p < re[r]);
                }
            }
        }
        if (cnt > GRID_SLAB) {      // cannot happen when the search reported rows <= 128; flagged, never silent
            if (lane == 0) atomicMax(overflow, cnt);
            cnt = GRID_SLAB;
        }
        wave_lds_sync();
        // ---- order by index: rank by counting (distinct indices), then permute through registers; only as many
        //      entries per lane as the in-degree needs (1 for <= 64 incoming pairs).  SORT = false keeps the order of the
        //      grid walk (cell run by cell run, compaction order inside a batch of 64): just as deterministic -- it depends
        //      on the grid only -- but not the pair order of the transposed table, so dx then agrees with K4 to fp32
        //      re-association (1e-6) instead of bit for bit; the counting loop it skips is ~ a third of this kernel's
        //      instructions (the kernel is issue-bound: DESIGN.md section 4)
        if (SORT) {
            float4 e[EPL];
            int rk[EPL];
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                e[u] = lane + 64 * u < cnt ? slab[lane + 64 * u] : make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fffffff));
                rk[u] = 0;
            }
            if (cnt <= 64) {
                for (int i = 0; i < cnt; ++i) rk[0] += __float_as_int(slab[i].w) < __float_as_int(e[0].w) ? 1 : 0;
            } else if (cnt <= 128) {
                for (int i = 0; i < cnt; ++i) {
                    const int ki = __float_as_int(slab[i].w);
                    rk[0] += ki < __float_as_int(e[0].w) ? 1 : 0;
                    rk[1] += ki < __float_as_int(e[1].w) ? 1 : 0;
                }
            } else {
                for (int i = 0; i < cnt; ++i) {
                    const int ki = __float_as_int(slab[i].w);
#pragma unroll
                    for (int u = 0; u < EPL; ++u) rk[u] += ki < __float_as_int(e[u].w) ? 1 : 0;
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int u = 0; u < EPL; ++u)
                if (lane + 64 * u < cnt) slab[rk[u]] = e[u];
            wave_lds_sync();
        }
        // ---- K4's list / flush over the slab
        for (int cc0 = 0; cc0 < ci; cc0 += CC) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int ch = cc0 + 4 * j;
            const bool chok = VEC ? (ch + 3 < ci) : (ch < ci);
            const int chl = chok ? ch : 0;
            // the gate row (GeomParams::gate) is fetched here, ahead of the walk: at the store its latency has long passed
            float4 gq = make_float4(1.f, 1.f, 1.f, 1.f);
            if (g.gate && slot == 0) {
                const T* gy = reinterpret_cast<const T*>(g.gate) + s * ci + ch;
                if (VEC) {
                    if (chok) gq = ld4(gy);
                } else {
                    if (ch + 0 < ci) gq.x = ld1(gy + 0);
                    if (ch + 1 < ci) gq.y = ld1(gy + 1);
                    if (ch + 2 < ci) gq.z = ld1(gy + 2);
                    if (ch + 3 < ci) gq.w = ld1(gy + 3);
                }
            }
            auto flush = [&](int total) {
                wave_lds_sync();
                total = min(total, POOL);
                const int per = (total + S - 1) / S;
                const int lo = slot * per;
                const int hi = chok ? min(lo + per, total) : lo;
                for (int it = 0; it < per; it += 4) {
                    float4 v[4];
                    float w[4];
                    uint2 e[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        e[u] = pool[min(lo + it + u, POOL + 7)];
                        keep_unconditional(e[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool ok = lo + it + u < hi;
                        e[u].x = ok ? e[u].x : 0u;
                        w[u] = ok ? __uint_as_float(e[u].y) : 0.0f;
                        v[u] = load_row_piece<VEC>(dwf, e[u].x, ci, chl);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc.x = fmaf(w[u], v[u].x, acc.x);
                        acc.y = fmaf(w[u], v[u].y, acc.y);
                        acc.z = fmaf(w[u], v[u].z, acc.z);
                        acc.w = fmaf(w[u], v[u].w, acc.w);
                    }
                }
                wave_lds_sync();
            };
            for (int p0 = 0; p0 < cnt; p0 += 64) {
                const int p = p0 + lane;
                const bool real = p < cnt;
                const float4 c = real ? slab[p] : make_float4(0.f, 0.f, 0.f, 0.f);
                const int q = real ? __float_as_int(c.w) : 0;
                const float nx = sx - c.x, ny = sy - c.y, nz = sz - c.z;
                const float* lkp = deformed_kp ? deformed_kp + (int64_t)q * (3 * K) : kernel_points;
                const float* lmod = modulations ? modulations + (int64_t)q * K : nullptr;
                auto kp = [&](int k, int cidx) { return MODE == 0 ? kpr[MODE == 0 ? 3 * k + cidx : 0] : lkp[3 * k + cidx]; };
                auto nomin = [&](int, float) {};
                int total = 0, maxlen = 0;
                kp_list<K, MODE>(nx, ny, nz, real, kp, g, inv_extent, (unsigned)(q * K), 1u, lmod, pool, segs, lane, total,
                                 maxlen, nomin);
                if (total <= POOL) {
                    flush(total);
                } else {
                    for (int sub = 0; sub < 4; ++sub) {
                        total = 0; maxlen = 0;
                        kp_list<K, MODE>(nx, ny, nz, real && (lane >> 4) == sub, kp, g, inv_extent, (unsigned)(q * K), 1u, lmod,
                                         pool, segs, lane, total, maxlen, nomin);
                        flush(total);
                    }
                }
            }
#pragma unroll
            for (int o = G; o < 64; o <<= 1) {
                acc.x += __shfl_xor(acc.x, o, 64);
                acc.y += __shfl_xor(acc.y, o, 64);
                acc.z += __shfl_xor(acc.z, o, 64);
                acc.w += __shfl_xor(acc.w, o, 64);
            }
            if (slot == 0) {
                T* dst = dx + s * ci + ch;
                if (g.gate) {
                    acc.x *= gq.x > 0.0f ? 1.0f : g.gate_slope; acc.y *= gq.y > 0.0f ? 1.0f : g.gate_slope;
                    acc.z *= gq.z > 0.0f ? 1.0f : g.gate_slope; acc.w *= gq.w > 0.0f ? 1.0f : g.gate_slope;
                }
                if (VEC) {
                    if (chok) st4(dst, acc);
                } else {
                    if (ch + 0 < ci) st1(dst + 0, acc.x);
                    if (ch + 1 < ci) st1(dst + 1, acc.y);
                    if (ch + 2 < ci) st1(dst + 2, acc.z);
                    if (ch + 3 < ci) st1(dst + 3, acc.w);
                }
            }
        }
        wave_lds_sync();
    }
}


// ---------------------------------------------------------------------------------------------
// K4G for wide rows: in-degrees of several hundred pairs per support (the deformable radius of BASELINE config 5:
// limits 422 / 519 / 472).  Same membership test as kpconv_gather_bwd_x_grid_kernel; what changes:
//   * the slab is a QUEUE, not a bound: whenever it holds 64 candidates more than one list phase consumes, the full
//     64-entry batches are run through the list / flush phases and the remainder moves to the front -- any in-degree,
//     no capacity flag, every list phase but the last on 64 live lanes;
//   * accumulators for up to NCH channel chunks live in registers, so the candidates are walked ONCE per support
//     whatever the row width (one pass for ci <= 4 G NCH = 256 channels);
//   * the candidate runs are walked by a rolled loop (run bounds in LDS): with ~300 candidates per run the first-batch
//     prefetch of the narrow kernel buys nothing and its nine-fold unrolled body would be inlined around every drain;
//   * MODE 2: the deformable fast path (kp_list_def).
// Summation order = order of the walk (a function of the grid alone): deterministic, equal to the transposed-table
// form up to fp32 re-association.
// ---------------------------------------------------------------------------------------------
template <int K, int G, int MODE, bool VEC, int NCH, typename T = float>
__global__ __launch_bounds__(256) void kpconv_gather_bwd_x_gridw_kernel(
    const float* __restrict__ s_pts, int64_t ns, const CloudGrid* __restrict__ grids, int nb,
    const int32_t* __restrict__ cell_start, const float4* __restrict__ sorted,
    const unsigned long long* __restrict__ key_last, float r2, const T* __restrict__ dwf, int ci,
    const float* __restrict__ kernel_points, const float* __restrict__ deformed_kp, const float* __restrict__ modulations,
    GeomParams g, T* __restrict__ dx, const int32_t* __restrict__ order)
{
    constexpr int CC = 4 * G;
    constexpr int S = 64 / G;
    static_assert(NCH == 1 || NCH == 2 || NCH == 4, "channel chunks whose accumulators stay in registers");
    constexpr int SLAB = 192;
    __shared__ uint2 pool_all[4][POOL_ALLOC];
    __shared__ int segs_all[4][K + 1];
    __shared__ float4 slab_all[4][SLAB + 64];      // + one dummy slot per lane (unconditional writes)
    __shared__ int2 runs_all[4][16];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint2* pool = pool_all[wave];
    int* segs = segs_all[wave];
    float4* slab = slab_all[wave];
    int2* runs = runs_all[wave];
    const int j = lane % G;
    const int slot = lane / G;
    const float inv_extent = 1.0f / g.extent;

    float kpr[MODE == 0 ? 3 * K : 1];
    // a pair farther apart than the reach of every kernel point (max |kp| + extent) carries no influence: it is no
    // candidate, and an untruncated row (sorted by distance) is left at the first entry beyond it -- see CUT above
    float cut2 = 3.4e38f;
    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 3 * K; ++t) kpr[t] = kernel_points[t];
        float rr = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) rr = fmaxf(rr, (kpr[3 * k] * kpr[3 * k] + kpr[3 * k + 1] * kpr[3 * k + 1]) + kpr[3 * k + 2] * kpr[3 * k + 2]);
        const float R = (__builtin_sqrtf(rr) + g.extent) * 1.0001f;
        cut2 = R * R;
    } else if (MODE == 2 && g.rmax) {
        const float R = (g.rmax[0] + g.extent) * 1.0001f;
        cut2 = R * R;
    }

    int64_t item0, istep, iend;
    ws_wave_items(ns, order ? g.ilv : 0, wave, item0, istep, iend);
    CloudGrid gr = grids[0];
    for (int64_t item = item0; item < iend; item += istep) {
        const int64_t s = order ? (int64_t)order[item] : item;
        if (s < gr.s_base || s >= gr.s_base + gr.s_len) {
            int b = 0;
            while (b + 1 < nb && s >= grids[b].s_base + grids[b].s_len) ++b;
            gr = grids[b];
        }
        const float sx = s_pts[3 * s + 0], sy = s_pts[3 * s + 1], sz = s_pts[3 * s + 2];
        const bool from_row = g.rows != nullptr && key_last[s] == ~0ull;
        for (int cg0 = 0; cg0 < ci; cg0 += NCH * CC) {
            float4 acc[NCH];
            float4 gq[NCH];
            bool chok[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                gq[c] = make_float4(1.f, 1.f, 1.f, 1.f);
                const int ch = cg0 + c * CC + 4 * j;
                chok[c] = VEC ? (ch + 3 < ci) : (ch < ci);
                if (g.gate && slot == 0 && ch < ci) {
                    const T* gy = reinterpret_cast<const T*>(g.gate) + s * ci + ch;
                    if (VEC) {
                        if (chok[c]) gq[c] = ld4(gy);
                    } else {
                        if (ch + 0 < ci) gq[c].x = ld1(gy + 0);
                        if (ch + 1 < ci) gq[c].y = ld1(gy + 1);
                        if (ch + 2 < ci) gq[c].z = ld1(gy + 2);
                        if (ch + 3 < ci) gq[c].w = ld1(gy + 3);
                    }
                }
            }
            int cnt = 0;
            // flush the pool into the accumulators of every channel chunk (balanced over the S slots)
            auto flush = [&](int total) {
                wave_lds_order();
                total = min(total, POOL);
                const int per = (total + S - 1) / S;
                const int lo = slot * per;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (cg0 + c * CC >= ci) continue;                   // wave-uniform (no `break`: the loop must unroll, or
                                                                        // the accumulators land in scratch memory)
                    const int chl = chok[c] ? cg0 + c * CC + 4 * j : 0;
                    const int hi = chok[c] ? min(lo + per, total) : lo;
                    for (int it = 0; it < per; it += 4) {
                        float4 v[4];
                        float w[4];
                        uint2 e[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            e[u] = pool[min(lo + it + u, POOL + 7)];
                            keep_unconditional(e[u]);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool ok = lo + it + u < hi;
                            e[u].x = ok ? e[u].x : 0u;
                            w[u] = ok ? __uint_as_float(e[u].y) : 0.0f;
                            v[u] = load_row_piece<VEC>(dwf, e[u].x, ci, chl);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            acc[c].x = fmaf(w[u], v[u].x, acc[c].x);
                            acc[c].y = fmaf(w[u], v[u].y, acc[c].y);
                            acc[c].z = fmaf(w[u], v[u].z, acc[c].z);
                            acc[c].w = fmaf(w[u], v[u].w, acc[c].w);
                        }
                    }
                }
                wave_lds_order();
            };
            // membership of candidate c (the query) for this support: inside the radius and not cut off q's row
            auto member = [&](const float4& c, bool active, unsigned long long kl) -> bool {
                const float d2 = ref_d2(c.x, c.y, c.z, make_float4(sx, sy, sz, 0.0f));
                const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)s;
                return active && d2 < r2 && d2 <= cut2 && key <= kl;
            };
            // ONE loop hands 64-pair batches to ONE inlined copy of the list / flush phases (a lambda that is called from
            // several places is outlined, and everything it captures -- the accumulators, the kernel points -- then lives in
            // scratch memory: the first form of this kernel ran 8 x slower than the transposed-table kernel for that reason).
            //   from_row   the support's own (untruncated) row holds every point within the radius: nearly every entry is a
            //              pair, the chunks go to the list phase directly.  The chain  row entry -> coordinates / key_last
            //              of chunk i + 1 and the row entries of chunk i + 2 are in flight under the work of chunk i.
            //   walk       the nine cell runs around the support, 128 candidates per step; hits are queued in the slab
            //              (capacity 2 x 64 + a dummy slot per lane) and leave it 64 at a time.
            const int nchunk = from_row ? (g.rows_h + 63) >> 6 : 0;
            auto load_q = [&](int ch) -> int64_t {
                const int col = 64 * ch + lane;
                return (ch < nchunk && col < g.rows_h) ? g.rows[s * g.rows_h + col] : -1;
            };
            auto load_c = [&](int64_t qi, float4& c, unsigned long long& kl) -> bool {
                const bool active = qi >= 0 && qi < ns;
                const int64_t qq = active ? qi : 0;
                c = make_float4(s_pts[3 * qq + 0], s_pts[3 * qq + 1], s_pts[3 * qq + 2], __int_as_float((int)qq));
                kl = key_last[qq];
                return active;
            };
            float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
            unsigned long long kl0 = 0, kl1 = 0;
            bool a0 = false, a1 = false;
            int64_t q1 = -1;
            int ch = 0;
            if (from_row) {
                a0 = load_c(load_q(0), c0, kl0);
                q1 = load_q(1);
            }
            int run = 0, rp = 0, re = 0;                       // walk state: next run, position / end inside the current run
            if (!from_row && gr.s_len > 0) {
                const int cx = cell_coord(sx, gr.lo[0], gr.inv_cell);
                const int cy = cell_coord(sy, gr.lo[1], gr.inv_cell);
                const int cz = cell_coord(sz, gr.lo[2], gr.inv_cell);
                const int x0 = max(cx - 1, 0), x1 = min(cx + 1, gr.nx - 1);
                if (lane < 9) {
                    const int z = cz + lane / 3 - 1, y = cy + lane % 3 - 1;
                    const bool ok = x0 <= x1 && z >= 0 && z < gr.nz && y >= 0 && y < gr.ny;
                    const int row = gr.cell_base + ((ok ? z : 0) * gr.ny + (ok ? y : 0)) * gr.nx;
                    runs[lane] = make_int2(ok ? cell_start[row + x0] : 0, ok ? cell_start[row + x1 + 1] : 0);
                }
                wave_lds_order();
            } else {
                run = 9;
            }
            for (;;) {
                float4 c;
                bool real;
                if (from_row) {
                    if (ch >= nchunk) break;
                    {   // the row is sorted by distance: past the reach (or at the padding) nothing follows
                        const float d2f = ref_d2(c0.x, c0.y, c0.z, make_float4(sx, sy, sz, 0.0f));
                        const float d2_first = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(d2f)));
                        const int a_first = __builtin_amdgcn_readfirstlane(a0 ? 1 : 0);
                        if (!a_first || d2_first > cut2) break;
                    }
                    if (ch + 1 < nchunk) a1 = load_c(q1, c1, kl1);
                    q1 = load_q(ch + 2);
                    c = c0;
                    real = member(c0, a0, kl0);
                    c0 = c1; kl0 = kl1; a0 = a1; a1 = false;
                    ++ch;
                } else {
                    // fill the queue up to 64 hits (or the end of the walk)
                    while (cnt < 64) {
                        if (rp >= re) {
                            if (run >= 9) break;
                            const int2 rr = runs[run++];
                            rp = __builtin_amdgcn_readfirstlane(rr.x);
                            re = __builtin_amdgcn_readfirstlane(rr.y);
                            continue;
                        }
                        const int pa = rp + lane, pb = rp + 64 + lane;
                        const bool aa = pa < re, ab = pb < re;
                        const float4 ca = sorted[aa ? pa : rp], cb = sorted[ab ? pb : rp];
                        const unsigned long long ka = key_last[__float_as_int(ca.w)], kb = key_last[__float_as_int(cb.w)];
                        const bool ha = member(ca, aa, ka), hb = member(cb, ab, kb);
                        const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
                        const int na = __builtin_popcountll(ma);
                        slab[ha ? cnt + lane_rank(ma) : SLAB + lane] = ca;          // cnt < 64 on entry: both batches fit 192
                        slab[hb ? cnt + na + lane_rank(mb) : SLAB + lane] = cb;
                        cnt += na + __builtin_popcountll(mb);
                        rp += 128;
                    }
                    if (cnt == 0) break;
                    wave_lds_order();
                    const int take_n = min(cnt, 64);
                    real = lane < take_n;
                    c = slab[real ? lane : 0];
                    // the remainder (< 128 entries) moves to the front
                    const int rem = cnt - take_n;
                    const float4 t0 = slab[64 + (lane < rem ? lane : 0)];
                    const float4 t1 = slab[128 + (lane + 64 < rem ? lane : 0)];
                    wave_lds_order();
                    if (lane < rem) slab[lane] = t0;
                    if (lane + 64 < rem) slab[64 + lane] = t1;
                    cnt = rem;
                    wave_lds_order();
                }
                // ---- list + flush of this batch: lane = pair (c = the query's coordinates and index; !real = dead lane)
                const int q = real ? __float_as_int(c.w) : 0;
                const float nx = sx - c.x, ny = sy - c.y, nz = sz - c.z;
                int total = 0, maxlen = 0;
                if constexpr (MODE == 2) {
                    const float4* kq = g.kp4 + (int64_t)q * K;
                    kp_list_def<K>(nx, ny, nz, real, kq, inv_extent, (unsigned)(q * K), pool, segs, lane, total, maxlen);
                    if (total <= POOL) {
                        flush(total);
                    } else {
                        for (int sub = 0; sub < 4; ++sub) {
                            total = 0; maxlen = 0;
                            kp_list_def<K>(nx, ny, nz, real && (lane >> 4) == sub, kq, inv_extent, (unsigned)(q * K), pool, segs, lane,
                                           total, maxlen);
                            flush(total);
                        }
                    }
                } else {
                    const float* lkp = deformed_kp ? deformed_kp + (int64_t)q * (3 * K) : kernel_points;
                    const float* lmod = modulations ? modulations + (int64_t)q * K : nullptr;
                    auto kp = [&](int k, int cidx) { return MODE == 0 ? kpr[MODE == 0 ? 3 * k + cidx : 0] : lkp[3 * k + cidx]; };
                    auto nomin = [&](int, float) {};
                    kp_list<K, MODE == 2 ? 1 : MODE>(nx, ny, nz, real, kp, g, inv_extent, (unsigned)(q * K), 1u, lmod, pool, segs, lane,
                                                     total, maxlen, nomin);
                    if (total <= POOL) {
                        flush(total);
                    } else {
                        for (int sub = 0; sub < 4; ++sub) {
                            total = 0; maxlen = 0;
                            kp_list<K, MODE == 2 ? 1 : MODE>(nx, ny, nz, real && (lane >> 4) == sub, kp, g, inv_extent, (unsigned)(q * K),
                                                             1u, lmod, pool, segs, lane, total, maxlen, nomin);
                            flush(total);
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (cg0 + c * CC >= ci) continue;
#pragma unroll
                for (int o = G; o < 64; o <<= 1) {
                    acc[c].x += __shfl_xor(acc[c].x, o, 64);
                    acc[c].y += __shfl_xor(acc[c].y, o, 64);
                    acc[c].z += __shfl_xor(acc[c].z, o, 64);
                    acc[c].w += __shfl_xor(acc[c].w, o, 64);
                }
                if (slot == 0) {
                    const int ch = cg0 + c * CC + 4 * j;
                    T* dst = dx + s * ci + ch;
                    float4 a = acc[c];
                    if (g.gate) {
                        a.x *= gq[c].x > 0.0f ? 1.0f : g.gate_slope; a.y *= gq[c].y > 0.0f ? 1.0f : g.gate_slope;
                        a.z *= gq[c].z > 0.0f ? 1.0f : g.gate_slope; a.w *= gq[c].w > 0.0f ? 1.0f : g.gate_slope;
                    }
                    if (VEC) {
                        if (chok[c]) st4(dst, a);
                    } else {
                        if (ch + 0 < ci) st1(dst + 0, a.x);
                        if (ch + 1 < ci) st1(dst + 1, a.y);
                        if (ch + 2 < ci) st1(dst + 2, a.z);
                        if (ch + 3 < ci) st1(dst + 3, a.w);
                    }
                }
            }
            wave_lds_order();
        }
    }
}

}  // namespace

// 1 = entry pool + VALU accumulate (kpconv_gather_fwd_kernel), 2 = matrix core (kpconv_gather_fwd_mfma_kernel);
// diagnostic switch (tools / A-B tests), not part of the drop-in surface
extern "C" int ws_kpconv_variant;
int ws_kpconv_variant = 2;
extern "C" int ws_kpconv_ablate;          // diagnostics (GeomParams::ablate); 0 in every product path
int ws_kpconv_ablate = 0;
extern "C" int ws_kpconv_gs;              // diagnostics: > 0 forces the group size of the Ci = 32 matrix-core kernel
int ws_kpconv_gs = 0;
extern "C" int ws_kpconv_grid_rows;       // diagnostics: 0 = K4G always walks the cell grid (WEASAL_K4G_ROWS=0)
int ws_kpconv_grid_rows = 1;
extern "C" int ws_kpconv_split_nt = 4;        // ... with at most this many channels per lane (blocks of 16 x this many channels)
extern "C" int ws_kpconv_split_rows = 4096;   // matrix-core K3 on fewer queries than this: one item per (query, channel block) (0 = never; WEASAL_K3_SPLIT_ROWS)
extern "C" int ws_kpconv_grid_sorted;     // 1: ws_kpconv_gather_bwd_x_grid sums the incoming pairs in index order (the pair order of
extern "C" int ws_kpconv_k6_interleave = 0;       // the same for the geometry backward on the matrix core (WEASAL_K6_INTERLEAVE)
extern "C" int ws_kpconv_gridw_interleave = 512;    // the same for the wide-row K4G of config 5 (WEASAL_K4GW_INTERLEAVE)
extern "C" int ws_kpconv_table_interleave = 0;    // the same for the transposed-table K4 (WEASAL_K4_INTERLEAVE)
extern "C" int ws_kpconv_grid_interleave = 512;   // lab: workgroups per XCD of the interleaved assignment (WEASAL_K4G_INTERLEAVE)
int ws_kpconv_grid_sorted = 0;            //    the transposed table: bit-identical to ws_kpconv_gather_bwd_x); 0: in grid-walk order

namespace {

template <typename T>
int gather_fwd_impl(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                    const int64_t* inds, int32_t h, const T* x, int32_t ci,
                    const float* kernel_points, int32_t k, const float* deformed_kp,
                    const float* modulations, float extent, int32_t influence, int32_t aggregation,
                    const int32_t* order, T* wf, float* min_d2, void* stream, int rows_sorted = 0)
{
    constexpr bool F32 = sizeof(T) == 4;
    int rc = check_common(q_pts, nq, s_pts, ns, h, ci, k, extent, influence, aggregation);
    if (rc) return rc;
    if (nq == 0) return WS_OK;
    WS_REQUIRE(inds && x && wf && (kernel_points || deformed_kp), "NULL argument");
    GeomParams g{extent, influence, aggregation, deformed_kp ? 1 : 0, ws_kpconv_ablate, nullptr, 0.0f, nullptr, 0, nullptr, nullptr,
                 rows_sorted ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    int grid = ws_grid(nq, 4);
    WS_REQUIRE(ns * (int64_t)ci < (1ll << 31), "ns*ci exceeds the 32-bit row offsets of the gather");
    const int vec4 = (ci % 4 == 0) && ws_row_aligned<T>(x) && ws_row_aligned<T>(wf);
    WS_REQUIRE(F32 || vec4, "bf16 feature rows need ci %% 4 == 0 and 8-byte aligned rows (ci=%d)", ci);
    if (ws_kpconv_variant == 2 && ci > 4) {      // (the 3..4-channel input layer: the narrow-row pool form is 10 % faster)
        // matrix-core form (kpconv_gather_fwd_mfma_kernel): NT consecutive channels per lane, 16 NT channels per block
        const bool fastm = !deformed_kp && !modulations && influence == WS_INFLUENCE_LINEAR && aggregation == WS_AGGREGATION_SUM;
        int nt = ci <= 16 ? 1 : (ci <= 32 ? 2 : (ci <= 64 ? 4 : (ci <= 128 ? 8 : 16)));
        const bool al16 = aligned16(x) && aligned16(wf);
        bool vecrow = (ci % nt == 0) && (nt == 1 || al16);
        if (!vecrow) nt = 1;
        if (nt == 1) vecrow = true;       // one channel per lane: any ci, any alignment (lanes past ci are masked)
        if (fastm && !rows_sorted && ws_kpconv_split_rows > 0 && nq < ws_kpconv_split_rows && vecrow && nt > ws_kpconv_split_nt && ws_kpconv_split_nt > 0 &&
            ci % ws_kpconv_split_nt == 0)
            nt = ws_kpconv_split_nt;                              // narrower blocks: more items per query
        if (fastm && !rows_sorted && ws_kpconv_split_rows > 0 && nq < ws_kpconv_split_rows && ci > 16 * nt) {
            g.csplit = (ci + 16 * nt - 1) / (16 * nt);           // one channel block per item
            grid = ws_grid(nq * g.csplit, 4);
        }
#define WS_FWDM2(NTV, MODEV, DEFV)                                                                                  \
    kpconv_gather_fwd_mfma_kernel<NTV, MODEV, DEFV, true, T><<<grid, 256, 0, st>>>(q_pts, nq, s_pts, ns, inds, h, x, ci, \
                                                                                   kernel_points, deformed_kp, modulations, g, wf, min_d2, order)
        if (F32 && nt == 2 && fastm && ws_kpconv_gs > 0) {      // diagnostics: group-size sweep on the dominant shape
            if constexpr (F32) {
#define WS_FWDG(GSX) kpconv_gather_fwd_mfma_kernel<2, 0, false, true, T, GSX><<<grid, 256, 0, st>>>(q_pts, nq, s_pts, ns, inds, h, x, ci, kernel_points, deformed_kp, modulations, g, wf, min_d2, order)
                if (ws_kpconv_gs == 1) WS_FWDG(1);
                else if (ws_kpconv_gs == 2) WS_FWDG(2);
                else if (ws_kpconv_gs == 4) WS_FWDG(4);
                else WS_FWDG(8);
#undef WS_FWDG
            }
            WS_LAUNCH_CHECK();
            return WS_OK;
        }
#define WS_FWDM(NTV)                                    \
    do {                                                \
        if (deformed_kp) WS_FWDM2(NTV, 1, true);        \
        else if (fastm && rows_sorted)                  \
            kpconv_gather_fwd_mfma_kernel<NTV, 0, false, true, T, 0, true><<<grid, 256, 0, st>>>(                      \
                q_pts, nq, s_pts, ns, inds, h, x, ci, kernel_points, deformed_kp, modulations, g, wf, min_d2, order);  \
        else if (fastm) WS_FWDM2(NTV, 0, false);        \
        else WS_FWDM2(NTV, 1, false);                   \
    } while (0)
        if (nt == 1) WS_FWDM(1);
        else if (nt == 2) WS_FWDM(2);
        else if (nt == 4) WS_FWDM(4);
        else if (nt == 8) WS_FWDM(8);
        else WS_FWDM(16);
#undef WS_FWDM
#undef WS_FWDM2
        WS_LAUNCH_CHECK();
        return WS_OK;
    }
#define WS_FWD2(G, MODEV, DEFV, VECV)                                                                              \
    do {                                                                                                           \
        if constexpr (F32 || VECV)                                                                                 \
            kpconv_gather_fwd_kernel<15, G, MODEV, DEFV, VECV, 4, T><<<grid, 256, 0, st>>>(                        \
                q_pts, nq, s_pts, ns, inds, h, x, ci, kernel_points, deformed_kp, modulations, g, wf, min_d2, order); \
    } while (0)
#define WS_FWD(G)                                                                                   \
    do {                                                                                            \
        if (deformed_kp) { if (vec4) WS_FWD2(G, 1, true, true); else WS_FWD2(G, 1, true, false); }  \
        else if (fast) { if (vec4) WS_FWD2(G, 0, false, true); else WS_FWD2(G, 0, false, false); }  \
        else { if (vec4) WS_FWD2(G, 1, false, true); else WS_FWD2(G, 1, false, false); }            \
    } while (0)
    const bool fast = influence == WS_INFLUENCE_LINEAR && aggregation == WS_AGGREGATION_SUM;
    if (ci <= 4 && !vec4) {
        // narrow rows that are not float4 (the 3-channel input layer): 4-byte pieces, 16 slots of 4 lanes
#define WS_FWDN(MODEV, DEFV)                                                                                          \
    do {                                                                                                              \
        if constexpr (F32)                                                                                            \
            kpconv_gather_fwd_kernel<15, 4, MODEV, DEFV, false, 1, T><<<grid, 256, 0, st>>>(                          \
                q_pts, nq, s_pts, ns, inds, h, x, ci, kernel_points, deformed_kp, modulations, g, wf, min_d2, order);  \
    } while (0)
        if (deformed_kp) WS_FWDN(1, true);
        else if (fast) WS_FWDN(0, false);
        else WS_FWDN(1, false);
#undef WS_FWDN
    }
    else if (ci <= 4) WS_FWD(1);
    else if (ci <= 8) WS_FWD(2);
    else if (ci <= 16) WS_FWD(4);
    else if (ci <= 32) WS_FWD(8);
    else WS_FWD(16);
#undef WS_FWD2
#undef WS_FWD
    WS_LAUNCH_CHECK();
    return WS_OK;
}

template <typename T>
int gather_bwd_x_impl(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                      const int64_t* inds, int32_t h, const int32_t* t_offsets, const int32_t* t_pairs,
                      const T* dwf, int32_t ci, const float* kernel_points, int32_t k,
                      const float* deformed_kp, const float* modulations, float extent,
                      int32_t influence, int32_t aggregation, const int32_t* order, T* dx, void* stream,
                      const T* gate = nullptr, float gate_slope = 0.0f)
{
    constexpr bool F32 = sizeof(T) == 4;
    (void)inds;
    int rc = check_common(q_pts, nq, s_pts, ns, h, ci, k, extent, influence, aggregation);
    if (rc) return rc;
    if (ns == 0) return WS_OK;
    WS_REQUIRE(t_offsets && t_pairs && dwf && dx && (kernel_points || deformed_kp), "NULL argument");
    WS_REQUIRE(nq * (int64_t)h < (1ll << 31), "nq*h exceeds int32");
    WS_REQUIRE(nq * (int64_t)k * ci < (1ll << 31), "nq*k*ci exceeds the 32-bit row offsets of the gather");
    GeomParams g{extent, influence, aggregation, deformed_kp ? 1 : 0, ws_kpconv_ablate, gate, gate_slope, nullptr, 0};
    hipStream_t st = (hipStream_t)stream;
    g.ilv = order ? ws_kpconv_table_interleave : 0;
    const int grid = g.ilv > 0 ? 8 * (int)std::max<int64_t>(1, std::min<int64_t>(g.ilv, ws_ceil_div(ns, 32))) : ws_grid(ns, 4);
    const int vec4 = (ci % 4 == 0) && ws_row_aligned<T>(dwf) && ws_row_aligned<T>(dx);
    WS_REQUIRE(F32 || vec4, "bf16 feature rows need ci %% 4 == 0 and 8-byte aligned rows (ci=%d)", ci);
#define WS_BWD2(G, MODEV, VECV)                                                                                       \
    do {                                                                                                              \
        if constexpr (F32 || VECV)                                                                                    \
            kpconv_gather_bwd_x_kernel<15, G, MODEV, VECV, T><<<grid, 256, 0, st>>>(                                   \
                q_pts, nq, s_pts, ns, h, t_offsets, t_pairs, dwf, ci, kernel_points, deformed_kp, modulations, g, dx, order); \
    } while (0)
#define WS_BWD(G)                                                                       \
    do {                                                                                \
        if (fast) { if (vec4) WS_BWD2(G, 0, true); else WS_BWD2(G, 0, false); }          \
        else { if (vec4) WS_BWD2(G, 1, true); else WS_BWD2(G, 1, false); }               \
    } while (0)
    const bool fast = !deformed_kp && !modulations && influence == WS_INFLUENCE_LINEAR && aggregation == WS_AGGREGATION_SUM;
    if (ci <= 4) WS_BWD(1);
    else if (ci <= 8) WS_BWD(2);
    else if (ci <= 16) WS_BWD(4);
    else if (ci <= 32) WS_BWD(8);
    else WS_BWD(16);
#undef WS_BWD2
#undef WS_BWD
    WS_LAUNCH_CHECK();
    return WS_OK;
}

template <typename T>
int gather_bwd_geom_impl(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                         const int64_t* inds, int32_t h, const T* x, int32_t ci, const T* dwf,
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
    GeomParams g{extent, influence, aggregation, 1, 0, nullptr, 0.0f, nullptr, 0};
    hipStream_t st = (hipStream_t)stream;
    kpconv_gather_bwd_geom_kernel<15, T><<<ws_grid(nq, 4), 256, 0, st>>>(q_pts, nq, s_pts, ns, inds, h, x, ci, dwf,
                                                                          deformed_kp, modulations, d_min_d2, g,
                                                                          d_deformed_kp, d_modulations,
                                                                          (ci % 4 == 0) && ws_row_aligned<T>(x) && ws_row_aligned<T>(dwf));
    WS_LAUNCH_CHECK();
    return WS_OK;
}

template <typename T>
int gather_bwd_x_grid_impl(const float* s_pts, int64_t ns, const void* grid_blob, int32_t nb, int64_t cells,
                           const uint64_t* key_last, float radius, const T* dwf, int32_t ci,
                           const float* kernel_points, int32_t k, const float* deformed_kp, const float* modulations,
                           float extent, int32_t influence, int32_t aggregation, const int32_t* order, T* dx,
                           int32_t* overflow, void* stream, const T* gate = nullptr, float gate_slope = 0.0f,
                           const int64_t* rows = nullptr, int32_t rows_h = 0)
{
    constexpr bool F32 = sizeof(T) == 4;
    int rc = check_common(s_pts, ns, s_pts, ns, 1, ci, k, extent, influence, aggregation);
    if (rc) return rc;
    if (ns == 0) return WS_OK;
    WS_REQUIRE(grid_blob && key_last && dwf && dx && overflow && (kernel_points || deformed_kp), "NULL argument");
    WS_REQUIRE(nb >= 1 && cells >= 1, "bad grid nb=%d cells=%lld", nb, (long long)cells);
    WS_REQUIRE(ns * (int64_t)k * ci < (1ll << 31), "ns*k*ci exceeds the 32-bit row offsets of the gather");
    WS_REQUIRE(!rows || rows_h >= 1, "index rows given without their width");
    GeomParams g{extent, influence, aggregation, deformed_kp ? 1 : 0, ws_kpconv_ablate, gate, gate_slope,
                 ws_kpconv_grid_rows ? rows : nullptr, rows_h};
    hipStream_t st = (hipStream_t)stream;
    const char* base = (const char*)grid_blob;
    const CloudGrid* grids = (const CloudGrid*)base;
    const int32_t* cell_start = (const int32_t*)(base + ws_grid_blob_cells_off(nb));
    const float4* sorted = (const float4*)(base + ws_grid_blob_sorted_off(nb, cells));
    const float r2 = radius * radius;                       // neighbors.cpp:226, as in the search
    const unsigned long long* kl = reinterpret_cast<const unsigned long long*>(key_last);
    g.ilv = order ? ws_kpconv_grid_interleave : 0;
    const int grid = g.ilv > 0 ? 8 * (int)std::max<int64_t>(1, std::min<int64_t>(g.ilv, ws_ceil_div(ns, 32))) : ws_grid(ns, 4);
    const int vec4 = (ci % 4 == 0) && ws_row_aligned<T>(dwf) && ws_row_aligned<T>(dx);
    WS_REQUIRE(F32 || vec4, "bf16 feature rows need ci %% 4 == 0 and 8-byte aligned rows (ci=%d)", ci);
#define WS_BWDG2(G, MODEV, VECV)                                                                                     \
    do {                                                                                                             \
        if constexpr (F32 || VECV) {                                                                                 \
            if (ws_kpconv_grid_sorted)                                                                               \
                kpconv_gather_bwd_x_grid_kernel<15, G, MODEV, VECV, T, true><<<grid, 256, 0, st>>>(                   \
                    s_pts, ns, grids, nb, cell_start, sorted, kl, r2, dwf, ci, kernel_points, deformed_kp, modulations, g, dx, order, overflow); \
            else                                                                                                     \
                kpconv_gather_bwd_x_grid_kernel<15, G, MODEV, VECV, T, false><<<grid, 256, 0, st>>>(                  \
                    s_pts, ns, grids, nb, cell_start, sorted, kl, r2, dwf, ci, kernel_points, deformed_kp, modulations, g, dx, order, overflow); \
        }                                                                                                            \
    } while (0)
#define WS_BWDG(G)                                                                      \
    do {                                                                                \
        if (fast) { if (vec4) WS_BWDG2(G, 0, true); else WS_BWDG2(G, 0, false); }        \
        else { if (vec4) WS_BWDG2(G, 1, true); else WS_BWDG2(G, 1, false); }             \
    } while (0)
    const bool fast = !deformed_kp && !modulations && influence == WS_INFLUENCE_LINEAR && aggregation == WS_AGGREGATION_SUM;
    if (ci <= 4) WS_BWDG(1);
    else if (ci <= 8) WS_BWDG(2);
    else if (ci <= 16) WS_BWDG(4);
    else if (ci <= 32) WS_BWDG(8);
    else WS_BWDG(16);
#undef WS_BWDG2
#undef WS_BWDG
    WS_LAUNCH_CHECK();
    return WS_OK;
}

// ---- deformable fast path (MODE 2) and the wide-row grid backward: launchers -------------------------------------------
template <typename T>
int gather_fwd_def_impl(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                        const T* x, int32_t ci, const float4* kp4, int32_t k, float extent, const int32_t* order, T* wf,
                        float* min_d2, void* stream, int rows_sorted)
{
    int rc = check_common(q_pts, nq, s_pts, ns, h, ci, k, extent, WS_INFLUENCE_LINEAR, WS_AGGREGATION_SUM);
    if (rc) return rc;
    if (nq == 0) return WS_OK;
    WS_REQUIRE(inds && x && wf && kp4, "NULL argument");
    WS_REQUIRE(aligned16(kp4), "kp4 must be 16-byte aligned");
    WS_REQUIRE(ns * (int64_t)ci < (1ll << 31), "ns*ci exceeds the 32-bit row offsets of the gather");
    GeomParams g{extent, WS_INFLUENCE_LINEAR, WS_AGGREGATION_SUM, 1, 0, nullptr, 0.0f, nullptr, 0, kp4};
    hipStream_t st = (hipStream_t)stream;
    const int grid = ws_grid(nq, 4);
    int nt = ci <= 16 ? 1 : (ci <= 32 ? 2 : (ci <= 64 ? 4 : (ci <= 128 ? 8 : 16)));
    if (!((ci % nt == 0) && (nt == 1 || (aligned16(x) && aligned16(wf))))) nt = 1;
    if (sizeof(T) == 2 && nt == 1 && (ci % 2)) return ws_fail(WS_ERR_UNSUPPORTED, "bf16 rows need an even channel count (ci=%d)", ci);
#define WS_FWDD(NTV)                                                                                                     \
    do {                                                                                                                 \
        if (rows_sorted)                                                                                                 \
            kpconv_gather_fwd_mfma_kernel<NTV, 2, true, true, T, 0, true><<<grid, 256, 0, st>>>(                          \
                q_pts, nq, s_pts, ns, inds, h, x, ci, nullptr, nullptr, nullptr, g, wf, min_d2, order);                   \
        else                                                                                                             \
            kpconv_gather_fwd_mfma_kernel<NTV, 2, true, true, T><<<grid, 256, 0, st>>>(                                   \
                q_pts, nq, s_pts, ns, inds, h, x, ci, nullptr, nullptr, nullptr, g, wf, min_d2, order);                   \
    } while (0)
    if (nt == 1) WS_FWDD(1);
    else if (nt == 2) WS_FWDD(2);
    else if (nt == 4) WS_FWDD(4);
    else if (nt == 8) WS_FWDD(8);
    else WS_FWDD(16);
#undef WS_FWDD
    WS_LAUNCH_CHECK();
    return WS_OK;
}

template <typename T>
int gather_bwd_x_def_impl(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, int32_t h,
                          const int32_t* t_offsets, const int32_t* t_pairs, const T* dwf, int32_t ci, const float4* kp4,
                          int32_t k, float extent, const int32_t* order, T* dx, void* stream)
{
    constexpr bool F32 = sizeof(T) == 4;
    int rc = check_common(q_pts, nq, s_pts, ns, h, ci, k, extent, WS_INFLUENCE_LINEAR, WS_AGGREGATION_SUM);
    if (rc) return rc;
    if (ns == 0) return WS_OK;
    WS_REQUIRE(t_offsets && t_pairs && dwf && dx && kp4 && aligned16(kp4), "NULL / unaligned argument");
    WS_REQUIRE(nq * (int64_t)h < (1ll << 31), "nq*h exceeds int32");
    WS_REQUIRE(nq * (int64_t)k * ci < (1ll << 31), "nq*k*ci exceeds the 32-bit row offsets of the gather");
    GeomParams g{extent, WS_INFLUENCE_LINEAR, WS_AGGREGATION_SUM, 1, 0, nullptr, 0.0f, nullptr, 0, kp4};
    hipStream_t st = (hipStream_t)stream;
    const int grid = ws_grid(ns, 4);
    const int vec4 = (ci % 4 == 0) && ws_row_aligned<T>(dwf) && ws_row_aligned<T>(dx);
    WS_REQUIRE(F32 || vec4, "bf16 feature rows need ci %% 4 == 0 and 8-byte aligned rows (ci=%d)", ci);
#define WS_BWDD2(G, VECV)                                                                                            \
    do {                                                                                                             \
        if constexpr (F32 || VECV)                                                                                   \
            kpconv_gather_bwd_x_kernel<15, G, 2, VECV, T><<<grid, 256, 0, st>>>(q_pts, nq, s_pts, ns, h, t_offsets, t_pairs, dwf, ci, \
                                                                               nullptr, nullptr, nullptr, g, dx, order); \
    } while (0)
#define WS_BWDD(G) do { if (vec4) WS_BWDD2(G, true); else WS_BWDD2(G, false); } while (0)
    if (ci <= 4) WS_BWDD(1);
    else if (ci <= 8) WS_BWDD(2);
    else if (ci <= 16) WS_BWDD(4);
    else if (ci <= 32) WS_BWDD(8);
    else WS_BWDD(16);
#undef WS_BWDD2
#undef WS_BWDD
    WS_LAUNCH_CHECK();
    return WS_OK;
}

template <typename T>
int gather_bwd_x_gridw_impl(const float* s_pts, int64_t ns, const void* grid_blob, int32_t nb, int64_t cells,
                            const uint64_t* key_last, float radius, const T* dwf, int32_t ci, const float* kernel_points,
                            int32_t k, const float4* kp4, float extent, const int32_t* order, const int64_t* rows, int32_t rows_h,
                            T* dx, void* stream, const float* rmax)
{
    constexpr bool F32 = sizeof(T) == 4;
    int rc = check_common(s_pts, ns, s_pts, ns, 1, ci, k, extent, WS_INFLUENCE_LINEAR, WS_AGGREGATION_SUM);
    if (rc) return rc;
    if (ns == 0) return WS_OK;
    WS_REQUIRE(grid_blob && key_last && dwf && dx && (kernel_points || kp4), "NULL argument");
    WS_REQUIRE(!kp4 || aligned16(kp4), "kp4 must be 16-byte aligned");
    WS_REQUIRE(nb >= 1 && cells >= 1, "bad grid nb=%d cells=%lld", nb, (long long)cells);
    WS_REQUIRE(ns * (int64_t)k * ci < (1ll << 31), "ns*k*ci exceeds the 32-bit row offsets of the gather");
    WS_REQUIRE(!rows || rows_h >= 1, "index rows given without their width");
    GeomParams g{extent, WS_INFLUENCE_LINEAR, WS_AGGREGATION_SUM, kp4 ? 1 : 0, 0, nullptr, 0.0f,
                 ws_kpconv_grid_rows ? rows : nullptr, rows_h, kp4, kp4 ? rmax : nullptr, 0};
    hipStream_t st = (hipStream_t)stream;
    const char* base = (const char*)grid_blob;
    const CloudGrid* grids = (const CloudGrid*)base;
    const int32_t* cell_start = (const int32_t*)(base + ws_grid_blob_cells_off(nb));
    const float4* sorted = (const float4*)(base + ws_grid_blob_sorted_off(nb, cells));
    const float r2 = radius * radius;
    const unsigned long long* kl = reinterpret_cast<const unsigned long long*>(key_last);
    g.ilv = order ? ws_kpconv_gridw_interleave : 0;
    const int grid = g.ilv > 0 ? 8 * (int)std::max<int64_t>(1, std::min<int64_t>(g.ilv, ws_ceil_div(ns, 32))) : ws_grid(ns, 4);
    const int vec4 = (ci % 4 == 0) && ws_row_aligned<T>(dwf) && ws_row_aligned<T>(dx);
    WS_REQUIRE(F32 || vec4, "bf16 feature rows need ci %% 4 == 0 and 8-byte aligned rows (ci=%d)", ci);
#define WS_GW3(G, MODEV, VECV, NCHV)                                                                                  \
    kpconv_gather_bwd_x_gridw_kernel<15, G, MODEV, VECV, NCHV, T><<<grid, 256, 0, st>>>(                              \
        s_pts, ns, grids, nb, cell_start, sorted, kl, r2, dwf, ci, kernel_points, nullptr, nullptr, g, dx, order)
#define WS_GW2(G, MODEV, VECV)                                                                                        \
    do {                                                                                                              \
        if constexpr (F32 || VECV) {                                                                                  \
            if (G < 16 || ci <= 64) WS_GW3(G, MODEV, VECV, 1);                                                        \
            else if constexpr (G == 16) {                                                                             \
                if (ci <= 128) WS_GW3(G, MODEV, VECV, 2);                                                             \
                else WS_GW3(G, MODEV, VECV, 4);                                                                       \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
#define WS_GW(G)                                                                        \
    do {                                                                                \
        if (kp4) { if (vec4) WS_GW2(G, 2, true); else WS_GW2(G, 2, false); }             \
        else { if (vec4) WS_GW2(G, 0, true); else WS_GW2(G, 0, false); }                 \
    } while (0)
    if (ci <= 4) WS_GW(1);
    else if (ci <= 8) WS_GW(2);
    else if (ci <= 16) WS_GW(4);
    else if (ci <= 32) WS_GW(8);
    else WS_GW(16);
#undef WS_GW3
#undef WS_GW2
#undef WS_GW
    WS_LAUNCH_CHECK();
    return WS_OK;
}

template <typename T>
int gather_bwd_geom_def_impl(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                             const T* x, int32_t ci, const T* dwf, const float4* kp4, int32_t k, const float* d_min_d2,
                             float extent, const int32_t* order, float4* d_kp4, void* stream, int rows_sorted)
{
    int rc = check_common(q_pts, nq, s_pts, ns, h, ci, k, extent, WS_INFLUENCE_LINEAR, WS_AGGREGATION_SUM);
    if (rc) return rc;
    if (nq == 0) return WS_OK;
    WS_REQUIRE(inds && x && dwf && kp4 && d_kp4 && aligned16(kp4) && aligned16(d_kp4), "NULL / unaligned argument");
    WS_REQUIRE(ns * (int64_t)ci < (1ll << 31) && nq * (int64_t)k * ci < (1ll << 31), "row offsets exceed 32 bits");
    if (ci % 16 != 0 || !aligned16(x) || !aligned16(dwf))
        return ws_fail(WS_ERR_UNSUPPORTED, "geometry backward on the matrix core needs ci %% 16 == 0 and 16-byte aligned rows (ci=%d)", ci);
    hipStream_t st = (hipStream_t)stream;
    const int ilv = order ? ws_kpconv_k6_interleave : 0;
    const int grid = ilv > 0 ? 8 * (int)std::max<int64_t>(1, std::min<int64_t>(ilv, ws_ceil_div(nq, 32))) : ws_grid(nq, 4);
#define WS_K6(CKV, AREGV)                                                                                                 \
    do {                                                                                                                  \
        if (rows_sorted)                                                                                                  \
            kpconv_gather_bwd_geom_def_kernel<CKV, AREGV, T, true><<<grid, 256, 0, st>>>(q_pts, nq, s_pts, ns, inds, h, x, ci, dwf, kp4, \
                                                                                         d_min_d2, extent, d_kp4, order, ilv); \
        else                                                                                                              \
            kpconv_gather_bwd_geom_def_kernel<CKV, AREGV, T, false><<<grid, 256, 0, st>>>(q_pts, nq, s_pts, ns, inds, h, x, ci, dwf, kp4, \
                                                                                          d_min_d2, extent, d_kp4, order, ilv); \
    } while (0)
    if (ci == 16) WS_K6(4, true);
    else if (ci == 32) WS_K6(8, true);
    else if (ci == 64) WS_K6(16, true);
    else if (ci == 128) WS_K6(32, true);
    else if (ci % 128 == 0) WS_K6(32, false);
    else if (ci % 64 == 0) WS_K6(16, false);
    else if (ci % 32 == 0) WS_K6(8, false);
    else WS_K6(4, false);
#undef WS_K6
    WS_LAUNCH_CHECK();
    return WS_OK;
}

}  // namespace

extern "C" {

// ---- deformable fast path: deformable (+ modulated) KPConv with linear influence and sum aggregation, the per-query kernel
//      points packed as kp4 [nq, 15] float4 (x, y, z, modulation; ws_kpconv_deform_prepare).  rows_bf16: feature rows
//      (x, wf, dwf, dx) are bf16 instead of f32.
int ws_kpconv_gather_fwd_def(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                             const void* x, int32_t ci, const float* kp4, int32_t k, float extent, const int32_t* order,
                             void* wf, float* min_d2, int32_t rows_bf16, int32_t rows_sorted, void* stream)
{
    const float4* kq = reinterpret_cast<const float4*>(kp4);
    if (rows_bf16)
        return gather_fwd_def_impl<bf16_t>(q_pts, nq, s_pts, ns, inds, h, (const bf16_t*)x, ci, kq, k, extent, order, (bf16_t*)wf, min_d2,
                                           stream, rows_sorted);
    return gather_fwd_def_impl<float>(q_pts, nq, s_pts, ns, inds, h, (const float*)x, ci, kq, k, extent, order, (float*)wf, min_d2, stream,
                                      rows_sorted);
}

// ws_kpconv_gather_fwd / _bf16 for index rows that are sorted by distance from their query (rows_sorted != 0: what the radius
// search delivers): the rigid linear / sum forms then stop at the reach of the kernel points (see CUT above); other modes and
// rows_sorted == 0 are the plain entries.
int ws_kpconv_gather_fwd_ex(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                            const void* x, int32_t ci, const float* kernel_points, int32_t k, const float* deformed_kp,
                            const float* modulations, float extent, int32_t influence, int32_t aggregation, const int32_t* order,
                            void* wf, float* min_d2, int32_t rows_bf16, int32_t rows_sorted, void* stream)
{
    if (rows_bf16)
        return gather_fwd_impl<bf16_t>(q_pts, nq, s_pts, ns, inds, h, (const bf16_t*)x, ci, kernel_points, k, deformed_kp, modulations,
                                       extent, influence, aggregation, order, (bf16_t*)wf, min_d2, stream, rows_sorted);
    return gather_fwd_impl<float>(q_pts, nq, s_pts, ns, inds, h, (const float*)x, ci, kernel_points, k, deformed_kp, modulations, extent,
                                  influence, aggregation, order, (float*)wf, min_d2, stream, rows_sorted);
}

int ws_kpconv_gather_bwd_x_def(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, int32_t h,
                               const int32_t* t_offsets, const int32_t* t_pairs, const void* dwf, int32_t ci, const float* kp4,
                               int32_t k, float extent, const int32_t* order, void* dx, int32_t rows_bf16, void* stream)
{
    const float4* kq = reinterpret_cast<const float4*>(kp4);
    if (rows_bf16)
        return gather_bwd_x_def_impl<bf16_t>(q_pts, nq, s_pts, ns, h, t_offsets, t_pairs, (const bf16_t*)dwf, ci, kq, k, extent, order, (bf16_t*)dx, stream);
    return gather_bwd_x_def_impl<float>(q_pts, nq, s_pts, ns, h, t_offsets, t_pairs, (const float*)dwf, ci, kq, k, extent, order, (float*)dx, stream);
}

// the table-free backward for any in-degree (rows wider than 128): rigid (kp4 NULL, kernel_points given) or deformable (kp4)
int ws_kpconv_gather_bwd_x_grid_wide(const float* s_pts, int64_t ns, const void* grid_blob, int32_t nb, int64_t cells,
                                     const uint64_t* key_last, float radius, const void* dwf, int32_t ci,
                                     const float* kernel_points, int32_t k, const float* kp4, const float* kp_rmax, float extent,
                                     const int32_t* order, const int64_t* rows, int32_t rows_h, void* dx, int32_t rows_bf16,
                                     void* stream)
{
    const float4* kq = reinterpret_cast<const float4*>(kp4);
    if (rows_bf16)
        return gather_bwd_x_gridw_impl<bf16_t>(s_pts, ns, grid_blob, nb, cells, key_last, radius, (const bf16_t*)dwf, ci, kernel_points, k,
                                               kq, extent, order, rows, rows_h, (bf16_t*)dx, stream, kp_rmax);
    return gather_bwd_x_gridw_impl<float>(s_pts, ns, grid_blob, nb, cells, key_last, radius, (const float*)dwf, ci, kernel_points, k, kq,
                                          extent, order, rows, rows_h, (float*)dx, stream, kp_rmax);
}

int ws_kpconv_gather_bwd_geom_def(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                                  const void* x, int32_t ci, const void* dwf, const float* kp4, int32_t k, const float* d_min_d2,
                                  float extent, const int32_t* order, float* d_kp4, int32_t rows_bf16, int32_t rows_sorted,
                                  void* stream)
{
    const float4* kq = reinterpret_cast<const float4*>(kp4);
    float4* dk = reinterpret_cast<float4*>(d_kp4);
    if (rows_bf16)
        return gather_bwd_geom_def_impl<bf16_t>(q_pts, nq, s_pts, ns, inds, h, (const bf16_t*)x, ci, (const bf16_t*)dwf, kq, k, d_min_d2,
                                                extent, order, dk, stream, rows_sorted);
    return gather_bwd_geom_def_impl<float>(q_pts, nq, s_pts, ns, inds, h, (const float*)x, ci, (const float*)dwf, kq, k, d_min_d2, extent,
                                           order, dk, stream, rows_sorted);
}

// Forward-only KPConv layer in ONE launch (inference: the testers' forward passes, utils/tester_PseudoLabel.py:164):
// out = act(KPConv(x) + bias) for rigid / linear / sum layers of 32 -> 32 f32 channels -- the level-0 layers of the DALES
// networks, where `wf` is largest (kpconv_gather_fwd_mfma_kernel<..., FUSE>).  Other shapes: WS_ERR_UNSUPPORTED (the caller
// runs gather + contraction as two launches).
int ws_kpconv_layer_fwd_fused(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns, const int64_t* inds, int32_t h,
                              const float* x, int32_t ci, const float* kernel_points, int32_t k, float extent,
                              const int32_t* order, const float* weights, int32_t co, const float* bias, int32_t act, float slope,
                              float* out, void* stream)
{
    int rc = check_common(q_pts, nq, s_pts, ns, h, ci, k, extent, WS_INFLUENCE_LINEAR, WS_AGGREGATION_SUM);
    if (rc) return rc;
    if (ci != 32 || co != 32 || !aligned16(x) || !aligned16(weights) || !aligned16(out))
        return ws_fail(WS_ERR_UNSUPPORTED, "fused forward layer: 32 -> 32 channels, 16-byte aligned rows (got %d -> %d)", ci, co);
    if (nq == 0) return WS_OK;
    WS_REQUIRE(inds && x && kernel_points && weights && out, "NULL argument");
    WS_REQUIRE(ns * (int64_t)ci < (1ll << 31), "ns*ci exceeds the 32-bit row offsets of the gather");
    GeomParams g{extent, WS_INFLUENCE_LINEAR, WS_AGGREGATION_SUM, 0, 0, nullptr, 0.0f, nullptr, 0, nullptr, nullptr, 0};
    FuseArgs fz{weights, bias, slope, act, out};
    kpconv_gather_fwd_mfma_kernel<2, 0, false, true, float, WS_FUSE_GS, false, true><<<ws_grid(nq, 4), 256, 0, (hipStream_t)stream>>>(
        q_pts, nq, s_pts, ns, inds, h, x, ci, kernel_points, nullptr, nullptr, g, nullptr, nullptr, order, fz);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

// Name of the forward gather kernel the dispatchers above launch for a layer (bench.py's roofline.kernel): same selection
// rules as gather_fwd_impl / gather_fwd_def_impl for 16-byte aligned rows.  mode: 0 rigid (kernel_points), 1 deformable
// through the generic entries, 2 deformable fast path (kp4).
int ws_kpconv_gather_fwd_variant(int32_t ci, int32_t mode, int32_t influence, int32_t aggregation, int32_t rows_bf16,
                                 int32_t rows_sorted, char* out, int32_t cap)
{
    WS_REQUIRE(out && cap > 0 && ci >= 1, "bad argument");
    const char* t = rows_bf16 ? "bf16" : "float";
    const bool fast = influence == WS_INFLUENCE_LINEAR && aggregation == WS_AGGREGATION_SUM;
    if (mode == 2 || (ws_kpconv_variant == 2 && ci > 4)) {
        int nt = ci <= 16 ? 1 : (ci <= 32 ? 2 : (ci <= 64 ? 4 : (ci <= 128 ? 8 : 16)));
        if (ci % nt) nt = 1;
        const int m = mode == 2 ? 2 : (mode == 1 ? 1 : (fast ? 0 : 1));
        const bool cut = rows_sorted && (m == 2 || m == 0);
        snprintf(out, (size_t)cap, "kpconv_gather_fwd_mfma_kernel<NT=%d, MODE=%d, DEF=%s, VECROW=true, %s, GS=default, CUT=%s>", nt, m,
                 mode ? "true" : "false", t, cut ? "true" : "false");
        return WS_OK;
    }
    const int vec4 = ci % 4 == 0;
    const int g = (ci <= 4) ? (vec4 ? 1 : 4) : (ci <= 8 ? 2 : (ci <= 16 ? 4 : (ci <= 32 ? 8 : 16)));
    snprintf(out, (size_t)cap, "kpconv_gather_fwd_kernel<K=15, G=%d, MODE=%d, DEF=%s, VEC=%s, PW=%d, %s>%s", g, (mode || !fast) ? 1 : 0,
             mode ? "true" : "false", vec4 ? "true" : "false", (ci <= 4 && !vec4) ? 1 : 4, t,
             (rows_sorted && !mode && fast) ? " (sorted-row cutoff on)" : "");
    return WS_OK;
}

int ws_kpconv_gather_fwd(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                         const int64_t* inds, int32_t h, const float* x, int32_t ci,
                         const float* kernel_points, int32_t k, const float* deformed_kp,
                         const float* modulations, float extent, int32_t influence, int32_t aggregation,
                         const int32_t* order, float* wf, float* min_d2, void* stream)
{
    return gather_fwd_impl<float>(q_pts, nq, s_pts, ns, inds, h, x, ci, kernel_points, k, deformed_kp, modulations, extent,
                                  influence, aggregation, order, wf, min_d2, stream);
}

int ws_kpconv_gather_fwd_bf16(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                              const int64_t* inds, int32_t h, const uint16_t* x, int32_t ci,
                              const float* kernel_points, int32_t k, const float* deformed_kp,
                              const float* modulations, float extent, int32_t influence, int32_t aggregation,
                              const int32_t* order, uint16_t* wf, float* min_d2, void* stream)
{
    return gather_fwd_impl<bf16_t>(q_pts, nq, s_pts, ns, inds, h, reinterpret_cast<const bf16_t*>(x), ci, kernel_points, k,
                                   deformed_kp, modulations, extent, influence, aggregation, order,
                                   reinterpret_cast<bf16_t*>(wf), min_d2, stream);
}

int ws_kpconv_gather_bwd_x(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                           const int64_t* inds, int32_t h, const int32_t* t_offsets, const int32_t* t_pairs,
                           const float* dwf, int32_t ci, const float* kernel_points, int32_t k,
                           const float* deformed_kp, const float* modulations, float extent,
                           int32_t influence, int32_t aggregation, const int32_t* order, float* dx, void* stream)
{
    return gather_bwd_x_impl<float>(q_pts, nq, s_pts, ns, inds, h, t_offsets, t_pairs, dwf, ci, kernel_points, k, deformed_kp,
                                    modulations, extent, influence, aggregation, order, dx, stream);
}

int ws_kpconv_gather_bwd_x_bf16(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                                const int64_t* inds, int32_t h, const int32_t* t_offsets, const int32_t* t_pairs,
                                const uint16_t* dwf, int32_t ci, const float* kernel_points, int32_t k,
                                const float* deformed_kp, const float* modulations, float extent,
                                int32_t influence, int32_t aggregation, const int32_t* order, uint16_t* dx, void* stream)
{
    return gather_bwd_x_impl<bf16_t>(q_pts, nq, s_pts, ns, inds, h, t_offsets, t_pairs, reinterpret_cast<const bf16_t*>(dwf), ci,
                                     kernel_points, k, deformed_kp, modulations, extent, influence, aggregation, order,
                                     reinterpret_cast<bf16_t*>(dx), stream);
}

int ws_kpconv_gather_bwd_geom(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                              const int64_t* inds, int32_t h, const float* x, int32_t ci, const float* dwf,
                              const float* kernel_points, int32_t k, const float* deformed_kp,
                              const float* modulations, const float* d_min_d2, float extent,
                              int32_t influence, int32_t aggregation, float* d_deformed_kp,
                              float* d_modulations, void* stream)
{
    return gather_bwd_geom_impl<float>(q_pts, nq, s_pts, ns, inds, h, x, ci, dwf, kernel_points, k, deformed_kp, modulations,
                                       d_min_d2, extent, influence, aggregation, d_deformed_kp, d_modulations, stream);
}

int ws_kpconv_gather_bwd_geom_bf16(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                                   const int64_t* inds, int32_t h, const uint16_t* x, int32_t ci, const uint16_t* dwf,
                                   const float* kernel_points, int32_t k, const float* deformed_kp,
                                   const float* modulations, const float* d_min_d2, float extent,
                                   int32_t influence, int32_t aggregation, float* d_deformed_kp,
                                   float* d_modulations, void* stream)
{
    return gather_bwd_geom_impl<bf16_t>(q_pts, nq, s_pts, ns, inds, h, reinterpret_cast<const bf16_t*>(x), ci,
                                        reinterpret_cast<const bf16_t*>(dwf), kernel_points, k, deformed_kp, modulations,
                                        d_min_d2, extent, influence, aggregation, d_deformed_kp, d_modulations, stream);
}

int ws_kpconv_gather_bwd_x_grid(const float* s_pts, int64_t ns, const void* grid_blob, int32_t nb, int64_t cells,
                                const uint64_t* key_last, float radius, const float* dwf, int32_t ci,
                                const float* kernel_points, int32_t k, const float* deformed_kp, const float* modulations,
                                float extent, int32_t influence, int32_t aggregation, const int32_t* order, float* dx,
                                int32_t* overflow, void* stream)
{
    return gather_bwd_x_grid_impl<float>(s_pts, ns, grid_blob, nb, cells, key_last, radius, dwf, ci, kernel_points, k, deformed_kp,
                                         modulations, extent, influence, aggregation, order, dx, overflow, stream);
}

int ws_kpconv_gather_bwd_x_grid_bf16(const float* s_pts, int64_t ns, const void* grid_blob, int32_t nb, int64_t cells,
                                     const uint64_t* key_last, float radius, const uint16_t* dwf, int32_t ci,
                                     const float* kernel_points, int32_t k, const float* deformed_kp, const float* modulations,
                                     float extent, int32_t influence, int32_t aggregation, const int32_t* order, uint16_t* dx,
                                     int32_t* overflow, void* stream)
{
    return gather_bwd_x_grid_impl<bf16_t>(s_pts, ns, grid_blob, nb, cells, key_last, radius, reinterpret_cast<const bf16_t*>(dwf), ci,
                                          kernel_points, k, deformed_kp, modulations, extent, influence, aggregation, order,
                                          reinterpret_cast<bf16_t*>(dx), overflow, stream);
}

// K4 / K4G with the activation backward of the preceding unary block folded into the store: dx * LeakyReLU'(gate_y)
// (gate_y [ns, ci] = that block's activated output, i.e. this layer's input x).  gate_y NULL = the plain entries above.
int ws_kpconv_gather_bwd_x_gated(const float* q_pts, int64_t nq, const float* s_pts, int64_t ns,
                                 const int64_t* inds, int32_t h, const int32_t* t_offsets, const int32_t* t_pairs,
                                 const float* dwf, int32_t ci, const float* kernel_points, int32_t k,
                                 const float* deformed_kp, const float* modulations, float extent,
                                 int32_t influence, int32_t aggregation, const int32_t* order, const float* gate_y,
                                 float gate_slope, float* dx, void* stream)
{
    return gather_bwd_x_impl<float>(q_pts, nq, s_pts, ns, inds, h, t_offsets, t_pairs, dwf, ci, kernel_points, k, deformed_kp,
                                    modulations, extent, influence, aggregation, order, dx, stream, gate_y, gate_slope);
}

int ws_kpconv_gather_bwd_x_grid_gated(const float* s_pts, int64_t ns, const void* grid_blob, int32_t nb, int64_t cells,
                                      const uint64_t* key_last, float radius, const float* dwf, int32_t ci,
                                      const float* kernel_points, int32_t k, const float* deformed_kp, const float* modulations,
                                      float extent, int32_t influence, int32_t aggregation, const int32_t* order,
                                      const float* gate_y, float gate_slope, const int64_t* rows, int32_t rows_h, float* dx,
                                      int32_t* overflow, void* stream)
{
    return gather_bwd_x_grid_impl<float>(s_pts, ns, grid_blob, nb, cells, key_last, radius, dwf, ci, kernel_points, k, deformed_kp,
                                         modulations, extent, influence, aggregation, order, dx, overflow, stream, gate_y,
                                         gate_slope, rows, rows_h);
}

}  // extern "C"
