// weasal_amd/csrc/ws_scan.h -- device-wide exclusive prefix sum (int32) used by the table builders.
// Three-level reduce-then-scan: 4096 items per 256-thread workgroup, block sums scanned
// recursively.  scratch must hold ws_scan_scratch_items(n) int32.
#pragma once
#include "ws_common.h"

constexpr int WS_SCAN_BLOCK = 256;
constexpr int WS_SCAN_IPT = 16;
constexpr int WS_SCAN_TILE = WS_SCAN_BLOCK * WS_SCAN_IPT;

inline int64_t ws_scan_scratch_items(int64_t n)
{
    int64_t total = 2;
    while (n + 1 > WS_SCAN_TILE) {
        const int64_t t = ws_ceil_div(n + 1, WS_SCAN_TILE);
        total += t + 1;
        n = t;
    }
    return total;
}

// out[i] = sum_{j<i} in[i]; out has n+1 entries (out[n] = total).  in may alias out.
int ws_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* scratch, hipStream_t st);
