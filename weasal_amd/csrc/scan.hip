// weasal_amd/csrc/scan.hip -- exclusive prefix sum (see ws_scan.h) and error buffer.
#include "ws_scan.h"

char* ws_errbuf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

namespace {

__device__ __forceinline__ int block_exclusive_scan(int v, int* lds /*[WS_SCAN_BLOCK/64 + 1]*/, int& block_total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < WS_SCAN_BLOCK / 64; ++w) {
        const int s = lds[w];
        if (w < wave) wave_off += s;
        total += s;
    }
    __syncthreads();
    block_total = total;
    return wave_off + inc - v;
}

__global__ __launch_bounds__(WS_SCAN_BLOCK) void scan_tile_sums(const int32_t* __restrict__ in, int64_t n,
                                                                 int32_t* __restrict__ sums)
{
    __shared__ int lds[WS_SCAN_BLOCK / 64 + 1];
    const int64_t base = (int64_t)blockIdx.x * WS_SCAN_TILE;
    int s = 0;
#pragma unroll
    for (int i = 0; i < WS_SCAN_IPT; ++i) {
        const int64_t j = base + (int64_t)i * WS_SCAN_BLOCK + threadIdx.x;
        if (j < n) s += in[j];
    }
    int total;
    block_exclusive_scan(s, lds, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// items of a tile are laid out thread-contiguous: thread t owns [t*IPT, (t+1)*IPT)
__global__ __launch_bounds__(WS_SCAN_BLOCK) void scan_tile_apply(const int32_t* in, int64_t n, int32_t* out,
                                                                  const int32_t* __restrict__ offs)
{
    __shared__ int lds[WS_SCAN_BLOCK / 64 + 1];
    const int64_t base = (int64_t)blockIdx.x * WS_SCAN_TILE + (int64_t)threadIdx.x * WS_SCAN_IPT;
    int v[WS_SCAN_IPT];
    int s = 0;
#pragma unroll
    for (int i = 0; i < WS_SCAN_IPT; ++i) {
        const int64_t j = base + i;
        v[i] = (j < n) ? in[j] : 0;
        s += v[i];
    }
    int total;
    int run = block_exclusive_scan(s, lds, total) + (offs ? offs[blockIdx.x] : 0);
#pragma unroll
    for (int i = 0; i < WS_SCAN_IPT; ++i) {
        const int64_t j = base + i;
        if (j <= n) out[j] = run;   // out[n] = grand total
        run += v[i];
    }
}

}  // namespace

int ws_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* scratch, hipStream_t st)
{
    // scan over n+1 slots (slot n reads as 0) so that out[n] receives the total
    const int64_t slots = n + 1;
    const int64_t tiles = ws_ceil_div(slots, WS_SCAN_TILE);
    if (tiles == 1) {
        scan_tile_apply<<<1, WS_SCAN_BLOCK, 0, st>>>(in, n, out, nullptr);
        WS_LAUNCH_CHECK();
        return WS_OK;
    }
    int32_t* sums = scratch;                 // [tiles] -> becomes [tiles+1] exclusive offsets
    int32_t* rest = scratch + tiles + 1;
    scan_tile_sums<<<(int)tiles, WS_SCAN_BLOCK, 0, st>>>(in, n, sums);
    WS_LAUNCH_CHECK();
    int rc = ws_exclusive_scan_i32(sums, sums, tiles, rest, st);
    if (rc) return rc;
    scan_tile_apply<<<(int)tiles, WS_SCAN_BLOCK, 0, st>>>(in, n, out, sums);
    WS_LAUNCH_CHECK();
    return WS_OK;
}
