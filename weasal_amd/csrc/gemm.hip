// weasal_amd/csrc/gemm.hip -- tall-skinny fp32 GEMMs on the f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// The dense parts of the KPConv hot path are all "M huge, K and N small":
//   the per-point unary MLPs            y = x W^T            (models/blocks.py:490-501, nn.Linear)
//   the kernel contraction               out = wf [N,15Ci] x weights [15Ci,Co]   (blocks.py:370-374)
// and their autograd (dx = dy W, dW = dy^T x).  With M = 400 000 rows and N,K in 32..512 they are
// bound by streaming the tall operand once; rocBLAS's fp32 solutions for these shapes run at a
// few % of that (profiles/r01_*).  Two kernels cover everything:
//
//   gemm_xb  : Y[M,N]  = X[M,K] * B[K,N]        B row-major, small.  (forward and dX)
//   gemm_xty : O[K,N]  = X[M,K]^T * Y[M,N]      reduction over the tall dimension.  (dW)
//
// Both stage 32-deep tiles in LDS with coalesced 16-byte loads (register prefetch of the next
// tile under the MFMAs of the current one) and feed v_mfma_f32_32x32x2_f32: exact fp32 (one
// rounding per product, a k-ordered fma chain), so results differ from rocBLAS only by summation
// order.  A operand lane map: A[i = lane&31][k = lane>>5]; B: B[k = lane>>5][j = lane&31];
// C/D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)  (cdna_hip_programming.md section 3).
#include "ws_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;   // rows of X per workgroup (32 per wave)
constexpr int BK = 32;    // k depth per LDS tile

__device__ __forceinline__ float4 ld4_guard(const float* __restrict__ p, int64_t row, int64_t nrows, int col, int ncols,
                                            int64_t ld, bool vec)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) {
        const float* s = p + row * ld + col;
        if (vec && col + 3 < ncols) {
            v = *reinterpret_cast<const float4*>(s);
        } else {
            if (col + 0 < ncols) v.x = s[0];
            if (col + 1 < ncols) v.y = s[1];
            if (col + 2 < ncols) v.z = s[2];
            if (col + 3 < ncols) v.w = s[3];
        }
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// Y[M,N] = X[M,K] * B[K,N].  Workgroup = 4 waves = 128 rows x (32*NT) columns; wave w owns rows
// 32w..32w+31 and NT accumulator tiles.
// ---------------------------------------------------------------------------------------------
template <int NT, int BKX>
__global__ __launch_bounds__(256) void gemm_xb_kernel(const float* __restrict__ x, int64_t m, int k, int64_t ldx,
                                                       const float* __restrict__ b, int n, int64_t ldb,
                                                       float* __restrict__ y, int64_t ldy, int vecx, int vecb,
                                                       const float* __restrict__ bias, const float* __restrict__ residual,
                                                       int64_t ldr, int act, float slope)
{
    constexpr int BN = 32 * NT;
    __shared__ float Xs[BM][BKX + 1];
    __shared__ __attribute__((aligned(16))) float Bs[BKX][BN];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    constexpr int XCOLS4 = BKX / 4;                      // float4 per X-tile row
    constexpr int XROWS = 256 / XCOLS4;                  // rows covered by one pass of the 256 threads
    constexpr int XPT = BM / XROWS;                      // passes
    const int xr = t / XCOLS4, xc = (t % XCOLS4) * 4;    // X tile: rows xr + XROWS i, cols xc..xc+3
    constexpr int BPT = (BKX * BN / 4 + 255) / 256;      // float4 of the B tile per thread
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    float4 xv[XPT], bv[BPT];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) xv[i] = ld4_guard(x, m0 + xr + XROWS * i, m, k0 + xc, k, ldx, vecx);
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int idx = t + 256 * i;
            const int br = idx / (BN / 4), bc = (idx % (BN / 4)) * 4;
            bv[i] = (br < BKX) ? ld4_guard(b, k0 + br, k, n0 + bc, n, ldb, vecb) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            float* d = &Xs[xr + XROWS * i][xc];
            d[0] = xv[i].x; d[1] = xv[i].y; d[2] = xv[i].z; d[3] = xv[i].w;
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int idx = t + 256 * i;
            const int br = idx / (BN / 4), bc = (idx % (BN / 4)) * 4;
            if (br < BKX) *reinterpret_cast<float4*>(&Bs[br][bc]) = bv[i];
        }
    };

    load_tiles(0);
    const int ai = wave * 32 + (lane & 31), kk = lane >> 5, bj = lane & 31;
    for (int k0 = 0; k0 < k; k0 += BKX) {
        store_tiles();
        __syncthreads();
        if (k0 + BKX < k) load_tiles(k0 + BKX);    // in flight under the MFMAs below
#pragma unroll
        for (int s = 0; s < BKX / 2; ++s) {
            const float a = Xs[ai][2 * s + kk];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const float bb = Bs[2 * s + kk][32 * i + bj];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // epilogue (+ bias[col], + residual[row,col], LeakyReLU): 32 lanes write 128 contiguous bytes of one row
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int col = n0 + 32 * i + (lane & 31);
        const float bv = (bias && col < n) ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < m && col < n) {
                float v = acc[i][r] + bv;
                if (residual) v += residual[row * ldr + col];
                if (act) v = v > 0.0f ? v : v * slope;
                y[row * ldy + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// partial[c][K,N] = X[rows of chunk c, K]^T * Y[rows of chunk c, N].  Workgroup = (32*KT) k-rows x
// (32*NT) columns of the output for one chunk of tall rows.  The 4 waves are KT k-tiles x RG = 4/KT
// row groups: for narrow X (K <= 32 / 64) the waves split the tall rows of every LDS tile instead of
// idling on zero padding, and their accumulators are summed through LDS at the end (fixed order).
// ---------------------------------------------------------------------------------------------
template <int NT, int KT>
__global__ __launch_bounds__(256) void gemm_xty_kernel(const float* __restrict__ x, int64_t m, int k, int64_t ldx,
                                                        const float* __restrict__ yy, int n, int64_t ldy,
                                                        float* __restrict__ partial, int64_t chunk, int vecx, int vecy)
{
    constexpr int BN = 32 * NT;
    constexpr int BKO = 32 * KT;   // output rows (= columns of X) per workgroup
    constexpr int RG = 4 / KT;     // row groups
    constexpr int XS_FLOATS = BK * BKO, YS_FLOATS = BK * BN;
    constexpr int RED_FLOATS = (RG > 1) ? KT * NT * 16 * 64 : 0;
    constexpr int LDS_FLOATS = (XS_FLOATS + YS_FLOATS) > RED_FLOATS ? (XS_FLOATS + YS_FLOATS) : RED_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float (*Xs)[BKO] = reinterpret_cast<float (*)[BKO]>(lds);
    float (*Ys)[BN] = reinterpret_cast<float (*)[BN]>(lds + XS_FLOATS);
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    const int kt = wave % KT, rg = wave / KT;
    const int k0 = blockIdx.y * BKO;
    const int n0 = blockIdx.z * BN;
    const int64_t mbeg = (int64_t)blockIdx.x * chunk;
    const int64_t mend = mbeg + chunk < m ? mbeg + chunk : m;
    constexpr int XPT = (XS_FLOATS / 4 + 255) / 256;
    constexpr int YPT = (YS_FLOATS / 4 + 255) / 256;
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    float4 xv[XPT], yv[YPT];
    auto load_tiles = [&](int64_t r0) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int idx = t + 256 * i;
            const int rr = idx / (BKO / 4), cc = (idx % (BKO / 4)) * 4;
            xv[i] = (rr < BK) ? ld4_guard(x, r0 + rr, mend, k0 + cc, k, ldx, vecx) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < YPT; ++i) {
            const int idx = t + 256 * i;
            const int rr = idx / (BN / 4), cc = (idx % (BN / 4)) * 4;
            yv[i] = (rr < BK) ? ld4_guard(yy, r0 + rr, mend, n0 + cc, n, ldy, vecy) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int idx = t + 256 * i;
            const int rr = idx / (BKO / 4), cc = (idx % (BKO / 4)) * 4;
            if (rr < BK) *reinterpret_cast<float4*>(&Xs[rr][cc]) = xv[i];
        }
#pragma unroll
        for (int i = 0; i < YPT; ++i) {
            const int idx = t + 256 * i;
            const int rr = idx / (BN / 4), cc = (idx % (BN / 4)) * 4;
            if (rr < BK) *reinterpret_cast<float4*>(&Ys[rr][cc]) = yv[i];
        }
    };

    load_tiles(mbeg);
    const int ai = kt * 32 + (lane & 31), kk = lane >> 5, bj = lane & 31;
    for (int64_t r0 = mbeg; r0 < mend; r0 += BK) {
        store_tiles();
        __syncthreads();
        if (r0 + BK < mend) load_tiles(r0 + BK);
#pragma unroll
        for (int ss = 0; ss < BK / 2 / RG; ++ss) {
            const int s = ss * RG + rg;
            const float a = Xs[2 * s + kk][ai];          // A[i = output row][k = tall index]
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const float bb = Ys[2 * s + kk][32 * i + bj];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (RG > 1) {
        // sum the row groups in order rg = 1, 2, 3 onto rg = 0 (layout [kt][i][r][lane])
        float* red = lds;
        for (int src = 1; src < RG; ++src) {
            if (rg == src) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((kt * NT + i) * 16 + r) * 64 + lane] = acc[i][r];
            }
            __syncthreads();
            if (rg == 0) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] += red[((kt * NT + i) * 16 + r) * 64 + lane];
            }
            __syncthreads();
        }
    }
    if (rg == 0) {
        float* out = partial + (int64_t)blockIdx.x * k * n;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int col = n0 + 32 * i + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < k && col < n) out[(int64_t)row * n + col] = acc[i][r];
            }
        }
    }
}

// out[e] = sum_c partial[c][e] in a fixed order (bitwise reproducible): 32 elements x 8 chunk groups
// per workgroup, group g adds chunks g, g+8, ... (coalesced 128-byte reads), then the 8 group sums
// are added in order.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int64_t elems, int chunks,
                                                               float* __restrict__ out)
{
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int64_t e = (int64_t)blockIdx.x * 32 + el;
    float s = 0.0f;
    if (e < elems)
        for (int c = grp; c < chunks; c += 8) s += partial[(int64_t)c * elems + e];
    red[grp][el] = s;
    __syncthreads();
    if (grp == 0 && e < elems) {
        float t = red[0][el];
#pragma unroll
        for (int g2 = 1; g2 < 8; ++g2) t += red[g2][el];
        out[e] = t;
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int64_t xty_chunk(int64_t m, int k, int n)
{
    (void)k; (void)n;
    // <= 256 chunks, a multiple of BK rows each, at least 512 rows (keeps >= 256 workgroups in
    // flight for the tall layers even when K fits one k-tile)
    int64_t c = ws_ceil_div(m, 256);
    if (c < 512) c = 512;
    return ws_ceil_div(c, BK) * BK;
}

}  // namespace

extern "C" {

int ws_gemm_xb_epilogue(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n,
                        const float* bias, const float* residual, int64_t ldr, int32_t act, float slope,
                        float* y, int64_t ldy, void* stream)
{
    WS_REQUIRE(m >= 0 && k >= 1 && n >= 1 && ldx >= k && ldy >= n, "bad sizes m=%lld k=%d n=%d", (long long)m, k, n);
    WS_REQUIRE(!residual || ldr >= n, "residual leading dimension too small");
    WS_REQUIRE(act == 0 || act == 1, "unknown activation %d", act);
    if (m == 0) return WS_OK;
    WS_REQUIRE(x && b && y, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const int vecx = al16(x) && (ldx % 4 == 0);
    const int vecb = al16(b) && (n % 4 == 0);
    const int64_t gx = ws_ceil_div(m, BM);
    WS_REQUIRE(gx < (1ll << 31), "m too large");
    if (n <= 32) {
        gemm_xb_kernel<1, 32><<<dim3((unsigned)gx, 1), 256, 0, st>>>(x, m, k, ldx, b, n, n, y, ldy, vecx, vecb, bias, residual, ldr,
                                                                 act, slope);
    } else if (n <= 64) {
        gemm_xb_kernel<2, 32><<<dim3((unsigned)gx, 1), 256, 0, st>>>(x, m, k, ldx, b, n, n, y, ldy, vecx, vecb, bias, residual, ldr,
                                                                 act, slope);
    } else {
        // measured (tools/gemm_bench.py, M = 400k): shallow K is latency bound and prefers the 64-column
        // tile (more waves per SIMD, L2 serves the X re-read); deep K prefers the 128-column tile (X reuse)
        if (k < 128)
            gemm_xb_kernel<2, 32><<<dim3((unsigned)gx, (unsigned)ws_ceil_div(n, 64)), 256, 0, st>>>(
                x, m, k, ldx, b, n, n, y, ldy, vecx, vecb, bias, residual, ldr, act, slope);
        else
            gemm_xb_kernel<4, 32><<<dim3((unsigned)gx, (unsigned)ws_ceil_div(n, 128)), 256, 0, st>>>(
                x, m, k, ldx, b, n, n, y, ldy, vecx, vecb, bias, residual, ldr, act, slope);
    }
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_gemm_xb(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n, float* y, int64_t ldy,
               void* stream)
{
    return ws_gemm_xb_epilogue(x, m, k, ldx, b, n, nullptr, nullptr, 0, 0, 0.0f, y, ldy, stream);
}

int64_t ws_gemm_xty_scratch_bytes(int64_t m, int32_t k, int32_t n)
{
    const int64_t chunk = xty_chunk(m, k, n);
    return ws_ceil_div(m > 0 ? m : 1, chunk) * (int64_t)k * n * (int64_t)sizeof(float);
}

int ws_gemm_xty(const float* x, int64_t m, int32_t k, int64_t ldx, const float* y, int32_t n, int64_t ldy,
                float* out, void* scratch, void* stream)
{
    WS_REQUIRE(m >= 0 && k >= 1 && n >= 1 && ldx >= k && ldy >= n, "bad sizes m=%lld k=%d n=%d", (long long)m, k, n);
    WS_REQUIRE(out, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        WS_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)k * n, st));
        return WS_OK;
    }
    WS_REQUIRE(x && y && scratch, "NULL argument");
    const int64_t chunk = xty_chunk(m, k, n);
    const int chunks = (int)ws_ceil_div(m, chunk);
    const int vecx = al16(x) && (ldx % 4 == 0);
    const int vecy = al16(y) && (ldy % 4 == 0);
    float* partial = chunks == 1 ? out : (float*)scratch;
#define WS_XTY(NTV, KTV)                                                                                              \
    gemm_xty_kernel<NTV, KTV><<<dim3(chunks, (unsigned)ws_ceil_div(k, 32 * KTV), (unsigned)ws_ceil_div(n, 32 * NTV)), 256, 0, \
                                st>>>(x, m, k, ldx, y, n, ldy, partial, chunk, vecx, vecy)
#define WS_XTY_K(NTV)                    \
    do {                                 \
        if (k <= 32) WS_XTY(NTV, 1);     \
        else if (k <= 64) WS_XTY(NTV, 2); \
        else WS_XTY(NTV, 4);             \
    } while (0)
    if (n <= 32) WS_XTY_K(1);
    else if (n <= 64) WS_XTY_K(2);
    else WS_XTY_K(4);
#undef WS_XTY_K
#undef WS_XTY
    WS_LAUNCH_CHECK();
    if (chunks > 1) {
        const int64_t elems = (int64_t)k * n;
        reduce_partials_kernel<<<(unsigned)ws_ceil_div(elems, 32), 256, 0, st>>>(partial, elems, chunks, out);
        WS_LAUNCH_CHECK();
    }
    return WS_OK;
}

}  // extern "C"
