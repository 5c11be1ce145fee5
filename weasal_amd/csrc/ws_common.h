// weasal_amd/csrc/ws_common.h -- shared host/device helpers of libweasal_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/weasal_hip.h"

#define WS_WAVE 64

// thread-local error message (ws_last_error)
char* ws_errbuf();
inline int ws_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ws_errbuf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define WS_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess)                                                           \
            return ws_fail(WS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                           __FILE__, __LINE__);                                          \
    } while (0)

#define WS_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return ws_fail(WS_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// every kernel launch of the library is followed by this check: it also counts them (ws_launch_count: bench.py's launches/step)
extern "C" long long ws_launch_counter;
#define WS_LAUNCH_CHECK()                         \
    do {                                          \
        __atomic_fetch_add(&ws_launch_counter, 1ll, __ATOMIC_RELAXED); \
        WS_HIP(hipGetLastError());                \
    } while (0)

static inline int64_t ws_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// private cross-file entries (not in include/weasal_hip.h): pools.hip for blocks.hip
extern "C" int ws_priv_max_pool_fwd_u8(const float* x, int64_t ns, int32_t c, const int64_t* inds, int64_t nq, int32_t h, float* out,
                                       uint8_t* arg, const int32_t* order_q, void* stream);
extern "C" int ws_priv_max_pool_bwd_u8(const float* dy, const uint8_t* arg, int64_t nq, int32_t h, int32_t c, const int32_t* t_offsets,
                                       const int32_t* t_pairs, int64_t ns, float* dx, const int32_t* order_s, const float* add,
                                       void* stream);        // add: NULL or [ns, c] summed into dx

extern "C" int ws_priv_gemm_xb_ex(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int64_t b_row_stride, int64_t b_col_stride,
                                  int32_t n, const float* bias, const float* residual, int64_t ldr, const int64_t* res_rows,
                                  int64_t res_rows_ld, int64_t res_nrows, int32_t act, float slope, float drop_p, uint64_t drop_seed,
                                  float* y, int64_t ldy, void* scratch, int64_t scratch_bytes, void* stream);   // gemm.hip for blocks.hip

extern "C" int ws_priv_gemm_xty_pitched(const float* x, int64_t m, int32_t k, int64_t ldx, const float* y, int32_t n, int64_t ldy,
                                        float* out, int64_t ldo, void* scratch, void* stream);                        // gemm.hip for blocks.hip

// grid size for wave-per-item / grid-stride kernels: enough workgroups to fill 256 CUs a few
// times over, never more than the work.
static inline int ws_grid(int64_t items, int per_block, int max_blocks = 256 * 16)
{
    int64_t b = ws_ceil_div(items, per_block);
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    if (b > 8) b = (b + 7) / 8 * 8;   // whole rounds over the 8 XCDs (ws_block_range)
    return (int)b;
}

#ifdef __HIPCC__
__device__ __forceinline__ int ws_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ float ws_readlane_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// Contiguous item range of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs
// (observed, not contractual -- only speed depends on it): block b and b+8 share an L2.  The remap
// gives each XCD one contiguous eighth of the (spatially ordered) work list, so that the rows a
// workgroup gathers were just touched by its neighbours on the same L2.
__device__ __forceinline__ void ws_block_range(int64_t n_items, int64_t& beg, int64_t& end)
{
    const int nblk = gridDim.x;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);
    const int64_t per = (n_items + nblk - 1) / nblk;
    beg = (int64_t)b * per;
    end = beg + per < n_items ? beg + per : n_items;
    if (beg > n_items) beg = n_items;
}
// Items of this WAVE (workgroups of 4 waves): i0, i0 + step, ... < end.  ilv = 0: the workgroup's contiguous chunk
// (ws_block_range).  ilv > 0 (grid = 8 x ilv workgroups): the workgroups of an XCD walk ITS eighth of the spatially ordered
// list together -- item = base + 4 * (local workgroup) + wave + t * 4 * ilv -- so that the rows gathered by the workgroups
// resident at one moment overlap and stay in the XCD's L2 (see pools.hip).
__device__ __forceinline__ void ws_wave_items(int64_t n_items, int ilv, int wave, int64_t& i0, int64_t& step, int64_t& end)
{
    if (ilv > 0 && (gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7, lb = blockIdx.x >> 3, nbx = gridDim.x >> 3;
        const int64_t per8 = (n_items + 7) / 8;
        const int64_t b8 = (int64_t)x * per8;
        end = b8 + per8 < n_items ? b8 + per8 : n_items;
        i0 = b8 + (int64_t)lb * 4 + wave;
        step = (int64_t)nbx * 4;
    } else {
        int64_t ibeg, iend;
        ws_block_range(n_items, ibeg, iend);
        i0 = ibeg + wave; step = 4; end = iend;
    }
}
__device__ __forceinline__ float ws_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// nn.Dropout's keep decision as a pure function of (seed, element index): splitmix64 finaliser, upper 32 bits; an element is
// dropped when the hash is below p * 2^32 (loss.hip: ws_dropout_apply; the fused forms in gemm.hip recompute the same bits)
struct WsDrop {
    unsigned long long seed;
    unsigned threshold;
    float scale;           // 1 / (1 - p)
    int on;
};
__device__ __forceinline__ unsigned ws_drop_hash(unsigned long long seed, unsigned long long i)
{
    unsigned long long z = seed + i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (unsigned)(z >> 32);
}
__device__ __forceinline__ float ws_drop1(float v, const WsDrop& d, unsigned long long i)
{
    return ws_drop_hash(d.seed, i) >= d.threshold ? v * d.scale : 0.0f;
}
inline WsDrop ws_drop_args(float p, unsigned long long seed)
{
    const double t = (double)p * 4294967296.0;
    return WsDrop{seed, t >= 4294967295.0 ? 4294967295u : (unsigned)t, 1.0f / (1.0f - p), p > 0.0f ? 1 : 0};
}

#endif
