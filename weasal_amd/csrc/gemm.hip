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
#include "ws_bf16.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;   // rows of X per workgroup (32 per wave)
constexpr int BK = 32;    // k depth per LDS tile

// Optional multiplicative gates of the GEMM epilogues (applied after bias / residual / activation):
//   y    rows [m, n] (pitch ld) of ANOTHER layer's activated output: v *= LeakyReLU'(y) = (y > 0 ? 1 : slope) -- the
//        activation backward of the layer that consumes this product as its gradient (dX = dY W^T gated by that layer's y);
//   mask rows [m, n] of bytes (pitch ldm): v = mask ? v * mscale : 0 -- dropout, forward (on the activated output) and
//        backward (on the gradient) alike.
//   drop (drop.on): nn.Dropout with the keep decision recomputed from (seed, row * dn + col) -- dn = the row length of the
//        contiguous tensor the reference's droplayer sees -- instead of read from a mask: v = keep ? v * scale : 0.
struct XbGate {
    const float* y; int64_t ld; float slope;
    const uint8_t* mask; int64_t ldm; float mscale;
    WsDrop drop; int64_t dn;
    // rrows (NULL = none): the residual row of output row r is residual[rrows[r * rld]] instead of residual[r] -- a row gather
    // (closest_pool / nearest upsampling, blocks.py:80-92) read by the epilogue instead of written out first; indices outside
    // [0, rn) are the shadow row: zeros
    const int64_t* rrows; int64_t rld; int64_t rn;
};
// first float of the residual row of output row `row` (NULL: the shadow row, or no residual)
__device__ __forceinline__ const float* xb_res_row(const float* residual, int64_t ldr, const XbGate& g, int64_t row)
{
    if (!residual) return nullptr;
    if (!g.rrows) return residual + row * ldr;
    const int64_t i = g.rrows[row * g.rld];
    return (i >= 0 && i < g.rn) ? residual + i * ldr : nullptr;
}
// order: dropout, then the LeakyReLU' gate, then the byte mask -- as a backward, dropout_bwd then activation_bwd is the order the
// separate passes run in (d u = dropout_bwd(d x_d); dz = d u * lrelu'(u)); as a forward only one of them is set
__device__ __forceinline__ void xb_gate4(float4& v, const XbGate& g, int64_t row, int col)
{
    if (g.drop.on) {
        const unsigned long long i = (unsigned long long)(row * g.dn + col);
        v.x = ws_drop1(v.x, g.drop, i); v.y = ws_drop1(v.y, g.drop, i + 1);
        v.z = ws_drop1(v.z, g.drop, i + 2); v.w = ws_drop1(v.w, g.drop, i + 3);
    }
    if (g.y) {
        const float4 q = *reinterpret_cast<const float4*>(g.y + row * g.ld + col);
        v.x *= q.x > 0.0f ? 1.0f : g.slope; v.y *= q.y > 0.0f ? 1.0f : g.slope;
        v.z *= q.z > 0.0f ? 1.0f : g.slope; v.w *= q.w > 0.0f ? 1.0f : g.slope;
    }
    if (g.mask) {
        const unsigned mk = *reinterpret_cast<const unsigned*>(g.mask + row * g.ldm + col);
        v.x = (mk & 0xffu) ? v.x * g.mscale : 0.0f;       v.y = (mk & 0xff00u) ? v.y * g.mscale : 0.0f;
        v.z = (mk & 0xff0000u) ? v.z * g.mscale : 0.0f;   v.w = (mk & 0xff000000u) ? v.w * g.mscale : 0.0f;
    }
}
__device__ __forceinline__ float xb_gate1(float v, const XbGate& g, int64_t row, int col)
{
    if (g.drop.on) v = ws_drop1(v, g.drop, (unsigned long long)(row * g.dn + col));
    if (g.y) v *= g.y[row * g.ld + col] > 0.0f ? 1.0f : g.slope;
    if (g.mask) v = g.mask[row * g.ldm + col] ? v * g.mscale : 0.0f;
    return v;
}

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
                                                       int64_t ldr, int act, float slope, const XbGate gate)
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
                if (residual) { const float* rr_ = xb_res_row(residual, ldr, gate, row); if (rr_) v += rr_[col]; }
                if (act) v = v > 0.0f ? v : v * slope;
                y[row * ldy + col] = xb_gate1(v, gate, row, col);
            }
        }
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// shared float4 epilogue of the rows-on-lanes kernels: lane = row, register quad g of tile i = columns
// n0 + 32 i + 8 g + 4 h .. + 3.  Residual quads are loaded with clamped columns (no control flow around
// the loads, so they are all in flight together); only the stores are predicated.
template <int NT>
__device__ __forceinline__ void xb_rows_epilogue(const f32x16 (&acc)[NT], int64_t row, int64_t m, int n0, int n, int h,
                                                 float* __restrict__ y, int64_t ldy, const float* __restrict__ bias,
                                                 const float* __restrict__ residual, int64_t ldr, int act, float slope,
                                                 const XbGate& gate)
{
    const bool live = row < m;
    const int64_t rr = live ? row : m - 1;
    float* yrow = y + rr * ldy;
    const float* rrow = xb_res_row(residual, ldr, gate, rr);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        float4 v[4];
        int col[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            col[g] = n0 + 32 * i + 8 * g + 4 * h;
            v[g] = make_float4(acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
        }
        if (rrow) {
            float4 rq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) rq[g] = *reinterpret_cast<const float4*>(rrow + (col[g] < n ? col[g] : n - 4));
#pragma unroll
            for (int g = 0; g < 4; ++g) { v[g].x += rq[g].x; v[g].y += rq[g].y; v[g].z += rq[g].z; v[g].w += rq[g].w; }
        }
        if (bias) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + (col[g] < n ? col[g] : n - 4));
                v[g].x += bq.x; v[g].y += bq.y; v[g].z += bq.z; v[g].w += bq.w;
            }
        }
        if (act) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v[g].x = v[g].x > 0.0f ? v[g].x : v[g].x * slope;
                v[g].y = v[g].y > 0.0f ? v[g].y : v[g].y * slope;
                v[g].z = v[g].z > 0.0f ? v[g].z : v[g].z * slope;
                v[g].w = v[g].w > 0.0f ? v[g].w : v[g].w * slope;
            }
        }
        if (gate.y || gate.mask || gate.drop.on) {
#pragma unroll
            for (int g = 0; g < 4; ++g) xb_gate4(v[g], gate, rr, col[g] < n ? col[g] : n - 4);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (live && col[g] < n) *reinterpret_cast<float4*>(yrow + col[g]) = v[g];
    }
}

// The same epilogue with the wave's 32 x 32 tile turned through LDS first.  In the rows-on-lanes layout a store instruction
// writes 16 bytes to each of 32 DIFFERENT rows (two column quads): a row's 128-byte line is completed by four separate
// instructions, and residual rows are read the same way -- products whose traffic is their output (K = 32 .. 64) ran at
// 2.5-3.3 TB/s where read-dominated ones reach 4.4-5.2 (tools/gemm_small_k_lab.py).  Staged: lane (r = lane / 8, c = lane % 8)
// owns the float4 at row r + 8 p, columns 4 c .. 4 c + 3 of the tile: one instruction moves 8 whole 128-byte row segments.
// `stage`: wave-private [32][36] floats (the W buffers, free after the main loop).
constexpr int XB_STAGE_LD = 36;
constexpr int XB_STAGE_FLOATS = 32 * XB_STAGE_LD;
template <int NT>
__device__ __forceinline__ void xb_rows_epilogue_staged(const f32x16 (&acc)[NT], int64_t row0, int64_t m, int n0, int n, int lane,
                                                        float* __restrict__ y, int64_t ldy, const float* __restrict__ bias,
                                                        const float* __restrict__ residual, int64_t ldr, int act, float slope,
                                                        const XbGate& gate, float* __restrict__ stage)
{
    const int idx = lane & 31, h = lane >> 5;
    const int r8 = lane >> 3, c4 = lane & 7;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(&stage[idx * XB_STAGE_LD + 8 * g + 4 * h]) =
                make_float4(acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int col = n0 + 32 * i + 4 * c4;
        const int colc = col < n ? col : n - 4;
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bq = *reinterpret_cast<const float4*>(bias + colc);
        float4 v[4], rq[4];
        int64_t rr[4];
        bool live[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int rl = r8 + 8 * p;
            const int64_t row = row0 + rl;
            live[p] = row < m && col < n;
            rr[p] = row < m ? row : m - 1;
            v[p] = *reinterpret_cast<const float4*>(&stage[rl * XB_STAGE_LD + 4 * c4]);
            rq[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (residual) {
                const float* rrow = xb_res_row(residual, ldr, gate, rr[p]);
                if (rrow) rq[p] = *reinterpret_cast<const float4*>(rrow + colc);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            v[p].x = (v[p].x + rq[p].x) + bq.x; v[p].y = (v[p].y + rq[p].y) + bq.y;      // residual first, then bias: the order of
            v[p].z = (v[p].z + rq[p].z) + bq.z; v[p].w = (v[p].w + rq[p].w) + bq.w;      // xb_rows_epilogue (bit-identical results)
            if (act) {
                v[p].x = v[p].x > 0.0f ? v[p].x : v[p].x * slope;
                v[p].y = v[p].y > 0.0f ? v[p].y : v[p].y * slope;
                v[p].z = v[p].z > 0.0f ? v[p].z : v[p].z * slope;
                v[p].w = v[p].w > 0.0f ? v[p].w : v[p].w * slope;
            }
            if (gate.y || gate.mask || gate.drop.on) xb_gate4(v[p], gate, rr[p], colc);
            if (live[p]) *reinterpret_cast<float4*>(y + rr[p] * ldy + col) = v[p];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// gemm_xb2: same product, "rows on the lanes".  The MFMA is fed transposed: A operand = the small
// matrix (i = output column), B operand = X (j = row of X), so that
//   * X never goes through LDS: lane (idx, h) reads float4 X[row idx][k0 + 8j + 4h ..] straight into
//     the B-operand registers (a row's 128-byte line is consumed by 4 consecutive loads of one wave);
//     MFMA step (j, e) contracts k = k0 + 8j + 4h + e, the two wave halves supplying two different k;
//   * the small matrix chunk [32 x BN] is staged in LDS already interleaved as Ws[t][j][h][idx][e], so
//     one conflict-free ds_read_b128 feeds four MFMAs; double buffered: ONE barrier per 32-deep chunk;
//   * a lane ends up with 4 consecutive output columns of its own row per register quad: the epilogue
//     is float4 (bias, residual, LeakyReLU, store) instead of 16 scalar stores per tile.
// Workgroup = 4 waves, wave w owns RT row tiles of 32 rows, all waves share the W chunk.
// ---------------------------------------------------------------------------------------------
// Preconditions (checked by the launcher; everything else takes gemm_xb_kernel): k % 32 == 0, n % 4 == 0,
// x / y / residual rows 16-byte aligned.  Rows past m and columns past n are clamped on the loads
// (valid memory, results never stored), so the main loop has no divergent control flow at all.
template <int NT, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void gemm_xb2_kernel(
    const float* __restrict__ x, int64_t m, int k, int64_t ldx, const float* __restrict__ b, int n, int ldb, int bcs,
    float* __restrict__ y, int64_t ldy, const float* __restrict__ bias, const float* __restrict__ residual, int64_t ldr,
    int act, float slope, int csplit, float* __restrict__ partial, const XbGate gate, int staged)
{
    // split-K (gridDim.z > 1; few rows, deep k): workgroup z contracts chunks [z*csplit, (z+1)*csplit) and
    // writes its raw sums to partial[z][m][n]; splitk_epilogue_kernel adds them in a fixed order
    constexpr int WM = 4 / WN;                            // wave grid WM x WN: rows x column groups
    constexpr int CT = NT * WN;                           // 32-column tiles per workgroup
    constexpr int BN = 32 * CT;
    constexpr int KC = 32;
    constexpr int WBUF = CT * 1024;                       // floats per W chunk buffer
    constexpr int WS_FLOATS = 2 * WBUF > 4 * XB_STAGE_FLOATS ? 2 * WBUF : 4 * XB_STAGE_FLOATS;
    __shared__ __attribute__((aligned(16))) float Ws[WS_FLOATS];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lane = t & 63;
    const int idx = lane & 31, h = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * (32 * WM) + wm * 32 + idx;
    const int n0 = blockIdx.y * BN;
    const int cbeg = blockIdx.z * csplit;
    const int cend = (cbeg + csplit) * KC < k ? cbeg + csplit : k / KC;

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // W staging: group g = t + 256 i -> column c = g % BN, row quad q = g / BN (rows 4q .. 4q+3 of the
    // chunk) -> one float4 at Ws[tile = c/32][j = q/2][h = q%2][idx = c%32][e = 0..3]
    // buffer addressing (wave-uniform descriptors from kernel arguments / blockIdx, per-lane constant byte
    // offsets, the chunk advance in the scalar offset): no vector address arithmetic in the loop; rows of
    // X past m read as 0 through the bounds check
    // B[kk][col] = b[kk * ldb + col * bcs]: bcs = 1 is the row-major [K,N] matrix, ldb = 1 with bcs = K' reads a
    // row-major [N,K'] matrix as its transpose (nn.Linear's weight, or a weight's transpose in a backward) in place
    const auto wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0,
                                                        (int)(((int64_t)(k - 1) * ldb + (int64_t)(n - 1) * bcs + 1) * 4), 0x00020000);
    const int64_t brow0 = (int64_t)blockIdx.x * (32 * WM);
    const int64_t brows = m - brow0 < 32 * WM ? m - brow0 : 32 * WM;
    const auto xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + brow0 * ldx), 0,
                                                        (int)(((brows - 1) * ldx + k) * 4), 0x00020000);
    int woff[CT];
    int wlds[CT];
    // transposed-in-place small matrix (ldb == 1): the four k of a row quad are 16 contiguous bytes and the 8 quads of a
    // chunk one 128-byte line per column -> lanes run along k (8 lanes per column, one dwordx4 each: 8 whole lines per wave
    // instruction); row-major: lanes run along the columns, four dword loads one row apart
    const bool wtr = ldb == 1;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int g = t + 256 * i;
        const int c = wtr ? g / 8 : g % BN, q = wtr ? g % 8 : g / BN;
        int col = n0 + c;
        col = col < n ? col : n - 1;
        woff[i] = (4 * q * ldb + col * bcs) * 4;
        wlds[i] = ((((c >> 5) * 4 + (q >> 1)) * 2 + (q & 1)) * 32 + (c & 31)) * 4;
    }
    float4 wv[CT];
    auto load_w = [&](int chunk) {
        const int so = chunk * KC * ldb * 4;                       // uniform byte offset of the chunk
        if (wtr) {
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wsrd, woff[i], so, 0);
                wv[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            wv[i].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wsrd, woff[i], so, 0));
            wv[i].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wsrd, woff[i], so + ldb * 4, 0));
            wv[i].z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wsrd, woff[i], so + ldb * 8, 0));
            wv[i].w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wsrd, woff[i], so + ldb * 12, 0));
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CT; ++i) *reinterpret_cast<float4*>(&Ws[buf * WBUF + wlds[i]]) = wv[i];
    };
    const int xoff = (int)((wm * 32 + idx) * ldx + 4 * h) * 4;
    float4 xn[4], xc[4];
    auto load_x = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xsrd, xoff + 32 * j, chunk * (KC * 4), 0);
            xn[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    };

    const int last = cend - 1;
    load_w(cbeg);
    load_x(cbeg);
    {   // chunk 1 of W goes out before anything waits on chunk 0
        float4 w0[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) w0[i] = wv[i];
        load_w(last < cbeg + 1 ? last : cbeg + 1);
#pragma unroll
        for (int i = 0; i < CT; ++i) *reinterpret_cast<float4*>(&Ws[wlds[i]]) = w0[i];
    }
    __syncthreads();
    for (int c = cbeg; c < cend; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xc[j] = xn[j];
        store_w((c - cbeg + 1) & 1);                            // chunk c+1 (a harmless repeat after the last one)
        load_x(c + 1 < last ? c + 1 : last);
        load_w(c + 2 < last ? c + 2 : last);
        const float* wb = &Ws[((c - cbeg) & 1) * WBUF + wn * (NT * 1024) + (h * 32 + idx) * 4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads of step j+1 out of step j (registers)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const float4 a4 = *reinterpret_cast<const float4*>(wb + (i * 4 + j) * 256);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, xc[j].x, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, xc[j].y, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, xc[j].z, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, xc[j].w, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (staged) {
        // (the main loop ended on a barrier: nobody reads the W buffers any more)
        const int64_t row0 = (int64_t)blockIdx.x * (32 * WM) + wm * 32;
        float* stage = &Ws[wave * XB_STAGE_FLOATS];
        if (partial)
            xb_rows_epilogue_staged<NT>(acc, row0, m, n0 + wn * (32 * NT), n, lane, partial + (int64_t)blockIdx.z * m * n, n, nullptr,
                                        nullptr, 0, 0, 0.0f, XbGate{}, stage);
        else
            xb_rows_epilogue_staged<NT>(acc, row0, m, n0 + wn * (32 * NT), n, lane, y, ldy, bias, residual, ldr, act, slope, gate, stage);
        return;
    }
    if (partial)
        xb_rows_epilogue<NT>(acc, row, m, n0 + wn * (32 * NT), n, h, partial + (int64_t)blockIdx.z * m * n, n, nullptr, nullptr, 0,
                             0, 0.0f, XbGate{});
    else
        xb_rows_epilogue<NT>(acc, row, m, n0 + wn * (32 * NT), n, h, y, ldy, bias, residual, ldr, act, slope, gate);
}

// ---------------------------------------------------------------------------------------------
// gemm_xb_shallow: Y = act(X [m, k] @ B [k, n] + bias + residual) for contractions too shallow and ragged for the MFMA tiles
// (k = 9: the logits' dX; k = 45: the 3-channel input layer's contraction).  The arithmetic is nothing (k FMAs per output); what
// matters is that the output -- all of the traffic -- leaves as whole 128-byte row segments and that the epilogue menu
// (residual, bias, LeakyReLU, gates) is the float4 one.  Thread = 4 consecutive columns of one row: the k values of its row
// come from L1 (the n / 4 threads of a row read the same addresses), the 4 columns of B from LDS (one ds_read_b128 per k,
// the same address for all rows of a wave's column); sequential fmaf over k (fixed order).  gemm_xb_kernel's scalar
// epilogue took 90-96 us on these shapes at M = 400 000, 3-4 x their streaming time.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_xb_shallow_kernel(const float* __restrict__ x, int64_t m, int k, int64_t ldx,
                                                               const float* __restrict__ b, int n, float* __restrict__ y, int64_t ldy,
                                                               const float* __restrict__ bias, const float* __restrict__ residual,
                                                               int64_t ldr, int act, float slope, const XbGate gate, int rows_per_wg)
{
    // LDS: B as [kp][n] (kp = k rounded up to 4, the extra rows zero) and the workgroup's rows of X as [rows_per_wg][kp] (each
    // element of X loaded once, coalesced; a thread then takes four k of its row per ds_read_b128 -- through global loads the
    // n / 4 threads of a row each issued the same k dword loads: address-unit bound, 121 us on the 45-deep shape)
    extern __shared__ __attribute__((aligned(16))) float shallow_lds[];
    const int kp = (k + 3) & ~3;
    float* bs = shallow_lds;
    float* xs = shallow_lds + kp * n;
    const int t = threadIdx.x;
    for (int e = t; e < kp * n / 4; e += 256)
        reinterpret_cast<float4*>(bs)[e] = e < k * n / 4 ? reinterpret_cast<const float4*>(b)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = r0 + rows_per_wg < m ? r0 + rows_per_wg : m;
    const int nrows = (int)(r1 - r0);
    {   // (row, k) of element e = t + 256 i, advanced without a division per element
        int r = t / kp, kk = t - r * kp;
        const int dr = 256 / kp, dk = 256 - dr * kp;
        for (int e = t; e < nrows * kp; e += 256) {
            xs[e] = kk < k ? x[(r0 + r) * ldx + kk] : 0.0f;
            r += dr; kk += dk;
            if (kk >= kp) { kk -= kp; ++r; }
        }
    }
    __syncthreads();
    const int n4 = n >> 2;
    const int per = 256 / n4;                              // rows per pass (n4 <= 256)
    const int c = (t % n4) * 4, rl = t / n4;
    if (rl >= per) return;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bq = *reinterpret_cast<const float4*>(bias + c);
    // four rows per thread and trip (rows r, r + per, r + 2 per, r + 3 per): one read of the B quad serves four rows -- with one
    // row per trip the kernel was bound by LDS bandwidth (5 ds_read_b128 per 16 FMAs)
    for (int rb = rl; rb < nrows; rb += 4 * per) {
        float4 v[4];
        const float* xr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int r = rb + u * per;
            xr[u] = xs + (r < nrows ? r : rb) * kp;              // (rows past the end: recomputed, never stored)
        }
        for (int kk = 0; kk < kp; kk += 4) {
            const float4 w0 = *reinterpret_cast<const float4*>(&bs[kk * n + c]);
            const float4 w1 = *reinterpret_cast<const float4*>(&bs[(kk + 1) * n + c]);
            const float4 w2 = *reinterpret_cast<const float4*>(&bs[(kk + 2) * n + c]);
            const float4 w3 = *reinterpret_cast<const float4*>(&bs[(kk + 3) * n + c]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 a = *reinterpret_cast<const float4*>(xr[u] + kk);
                v[u].x = fmaf(a.x, w0.x, v[u].x); v[u].y = fmaf(a.x, w0.y, v[u].y); v[u].z = fmaf(a.x, w0.z, v[u].z); v[u].w = fmaf(a.x, w0.w, v[u].w);
                v[u].x = fmaf(a.y, w1.x, v[u].x); v[u].y = fmaf(a.y, w1.y, v[u].y); v[u].z = fmaf(a.y, w1.z, v[u].z); v[u].w = fmaf(a.y, w1.w, v[u].w);
                v[u].x = fmaf(a.z, w2.x, v[u].x); v[u].y = fmaf(a.z, w2.y, v[u].y); v[u].z = fmaf(a.z, w2.z, v[u].z); v[u].w = fmaf(a.z, w2.w, v[u].w);
                v[u].x = fmaf(a.w, w3.x, v[u].x); v[u].y = fmaf(a.w, w3.y, v[u].y); v[u].z = fmaf(a.w, w3.z, v[u].z); v[u].w = fmaf(a.w, w3.w, v[u].w);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb + u * per;
            if (r >= nrows) break;
            const int64_t row = r0 + r;
            float4 o = v[u];
            if (residual) {
                const float* rrow = xb_res_row(residual, ldr, gate, row);
                if (rrow) { const float4 rq = *reinterpret_cast<const float4*>(rrow + c); o.x += rq.x; o.y += rq.y; o.z += rq.z; o.w += rq.w; }
            }
            o.x += bq.x; o.y += bq.y; o.z += bq.z; o.w += bq.w;
            if (act) {
                o.x = o.x > 0.0f ? o.x : o.x * slope; o.y = o.y > 0.0f ? o.y : o.y * slope;
                o.z = o.z > 0.0f ? o.z : o.z * slope; o.w = o.w > 0.0f ? o.w : o.w * slope;
            }
            if (gate.y || gate.mask || gate.drop.on) xb_gate4(o, gate, row, c);
            *reinterpret_cast<float4*>(y + row * ldy + c) = o;
        }
    }
}

// The three-way bf16 split products (gemm_xb3, gemm_xty2<..., SPLIT>) are a LAB build only (-DWS_LAB_SPLIT_GEMM; tools/
// split_gemm_lab.py): measured in round 2 (15 % per product, 2.3 % per step, fp32-accurate), not the reference's
// arithmetic behind a line that says f32 -- they are not compiled into the product library.
#ifdef WS_LAB_SPLIT_GEMM
// ---------------------------------------------------------------------------------------------
// gemm_xb3: the same product, fp32 in and out, with every fp32 PRODUCT evaluated on the bf16 MFMA from exact three-way
// splits of both operands:  x = xh + xm + xl,  w = wh + wm + wl  (bf16 pieces: xh = bf16(x), xm = bf16(x - xh),
// xl = bf16(x - xh - xm); the subtractions are exact in fp32, so the three pieces carry the full 24-bit mantissa), and
//     x w  =  xh wh + xh wm + xm wh + xh wl + xl wh + xm wm   +  (xm wl + xl wm + xl wl  <  2^-23 |x w|)
// six v_mfma_f32_32x32x16_bf16 with fp32 accumulation per 16-deep step instead of eight v_mfma_f32_32x32x2_f32:
// bf16 products are exact in fp32, the accumulation is the MFMA's fp32 adder either way, and the dropped cross terms
// are below the rounding of one fp32 product.  What changes is the cost: 6 x 8 passes instead of 8 x 16 for the same
// 32 x 32 x 16 block (2.7 x the matrix-core throughput), which is what bounds this network in fp32: ~ 0.7 TFLOP per
// DALES step against the ~ 100 TFLOP/s the f32-input MFMA sustains.  Measured error against float64 is that of the
// f32-input kernels (tests/test_kpconv_gpu.py::test_split_bf16_products_are_fp32_accurate).
// Layout: rows on the lanes as in gemm_xb2 (A operand = the small matrix, B operand = X): lane (j, h) of a 32x32x16
// step holds 8 consecutive k of row j -- two 16-byte loads of X, split in registers; the 32-deep chunk of the small
// matrix is loaded as fp32 (any strides), split by the staging threads and kept in LDS as three bf16 pieces,
// k-contiguous rows of 64 + 16 bytes (conflict-free 16-byte fragment reads), double buffered: one barrier per chunk.
// Preconditions as gemm_xb2 (k % 32 == 0, n % 4 == 0, 16-byte aligned rows); same epilogue, same split-K.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& ph, unsigned& pm, unsigned& pl)
{
    ph = ws_pack_bf2(a, b);
    const float ra = a - ws_bf_lo(ph), rb = b - ws_bf_hi(ph);
    pm = ws_pack_bf2(ra, rb);
    pl = ws_pack_bf2(ra - ws_bf_lo(pm), rb - ws_bf_hi(pm));
}

template <int NT>
__global__ __launch_bounds__(256) void gemm_xb3_kernel(
    const float* __restrict__ x, int64_t m, int k, int64_t ldx, const float* __restrict__ b, int n, int ldb, int bcs,
    float* __restrict__ y, int64_t ldy, const float* __restrict__ bias, const float* __restrict__ residual, int64_t ldr,
    int act, float slope, int csplit, float* __restrict__ partial, const XbGate gate)
{
    constexpr int BN = 32 * NT;
    constexpr int KC = 32;
    constexpr int ROWB = 80;                        // bytes per staged row of a piece: 64 + 16 of padding
    constexpr int PB = BN * 16 / 256;               // (column, k pair) items of the chunk per thread
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][3][BN * ROWB];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    const int j = lane & 31, h = lane >> 5;
    const int64_t brow0 = (int64_t)blockIdx.x * 128;
    const int64_t brows = m - brow0 < 128 ? m - brow0 : 128;
    const int n0 = blockIdx.y * BN;
    const int cbeg = blockIdx.z * csplit;
    const int cend = (cbeg + csplit) * KC < k ? cbeg + csplit : k / KC;

    const auto wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0,
                                                        (int)(((int64_t)(k - 1) * ldb + (int64_t)(n - 1) * bcs + 1) * 4), 0x00020000);
    const auto xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + brow0 * ldx), 0,
                                                        (int)(((brows - 1) * ldx + k) * 4), 0x00020000);
    // staging items: row-major small matrix -> lanes run along the columns (coalesced rows); transposed in place
    // (ldb == 1) -> lanes run along k (the 16 pairs of a column are one 128-byte line)
    const bool wtr = ldb == 1;
    int woff[PB], wlds[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int g = t + 256 * i;
        const int c = wtr ? g / 16 : g % BN, kp = wtr ? g % 16 : g / BN;
        int col = n0 + c;
        col = col < n ? col : n - 1;                // clamped: valid memory, the column is never stored
        woff[i] = (2 * kp * ldb + col * bcs) * 4;
        wlds[i] = c * ROWB + 4 * kp;
    }
    float w0[PB], w1[PB];
    auto load_w = [&](int chunk) {
        const int so = chunk * KC * ldb * 4;
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            w0[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wsrd, woff[i], so, 0));
            w1[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wsrd, woff[i], so + ldb * 4, 0));
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            unsigned ph, pm, pl;
            split3_pair(w0[i], w1[i], ph, pm, pl);
            *reinterpret_cast<unsigned*>(&Bs[buf][0][wlds[i]]) = ph;
            *reinterpret_cast<unsigned*>(&Bs[buf][1][wlds[i]]) = pm;
            *reinterpret_cast<unsigned*>(&Bs[buf][2][wlds[i]]) = pl;
        }
    };
    const int xoff = (int)(((wave * 32 + j) * ldx + 8 * h) * 4);
    // X runs two chunks ahead of the MFMAs (xn = chunk c + 1, xf = chunk c + 2 in flight): with 2 waves per SIMD the
    // bytes in flight, not the matrix cores, bound the tall shapes
    u32x4 xn[4], xf[4];
    auto load_x = [&](u32x4 (&dst)[4], int chunk) {
#pragma unroll
        for (int q = 0; q < 4; ++q)                  // step s = q / 2 (16 k each), half q % 2 of the lane's 8 k
            dst[q] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, xoff + 64 * (q >> 1) + 16 * (q & 1), chunk * (KC * 4), 0);
    };

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int last = cend - 1;
    auto split_x = [&](const u32x4 (&raw)[4], u32x4 (&ph)[2], u32x4 (&pm)[2], u32x4 (&pl)[2]) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const u32x4 a = raw[2 * s2], bq = raw[2 * s2 + 1];
            unsigned qh[4], qm[4], ql[4];
            split3_pair(__uint_as_float(a.x), __uint_as_float(a.y), qh[0], qm[0], ql[0]);
            split3_pair(__uint_as_float(a.z), __uint_as_float(a.w), qh[1], qm[1], ql[1]);
            split3_pair(__uint_as_float(bq.x), __uint_as_float(bq.y), qh[2], qm[2], ql[2]);
            split3_pair(__uint_as_float(bq.z), __uint_as_float(bq.w), qh[3], qm[3], ql[3]);
            ph[s2] = u32x4{qh[0], qh[1], qh[2], qh[3]};
            pm[s2] = u32x4{qm[0], qm[1], qm[2], qm[3]};
            pl[s2] = u32x4{ql[0], ql[1], ql[2], ql[3]};
        }
    };
    // software pipeline: while the matrix cores work on chunk c, the vector ALU splits chunk c + 1 (X: registers -> the
    // pieces of the next iteration; small matrix: registers -> LDS buffer of the next iteration) and chunk c + 2 of X is
    // in flight.  The bf16 MFMA leaves the issue port free for 7 of its 8 passes: the splits cost no time of their own.
    u32x4 xh[2], xm[2], xl[2];
    load_w(cbeg);
    load_x(xn, cbeg);
    load_x(xf, cbeg + 1 < last ? cbeg + 1 : last);
    store_w(0);
    split_x(xn, xh, xm, xl);
    __syncthreads();
    for (int c = cbeg; c < cend; ++c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xn[q] = xf[q];
        // the small matrix first: its registers are consumed at the end of THIS iteration, and loads return in order --
        // waiting for them must not also wait for the X chunk requested after them (which is not needed before the next one)
        load_w(c + 1 < last ? c + 1 : last);
        __builtin_amdgcn_sched_barrier(0);
        load_x(xf, c + 2 < last ? c + 2 : last);
        __builtin_amdgcn_sched_barrier(0);
        const int buf = (c - cbeg) & 1;
        const unsigned char* bh = &Bs[buf][0][j * ROWB + 16 * h];
        const unsigned char* bm = &Bs[buf][1][j * ROWB + 16 * h];
        const unsigned char* bl = &Bs[buf][2][j * ROWB + 16 * h];
        u32x4 nh[2], nm[2], nl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8_t fxh = *reinterpret_cast<const bf16x8_t*>(&xh[s2]);
            const bf16x8_t fxm = *reinterpret_cast<const bf16x8_t*>(&xm[s2]);
            const bf16x8_t fxl = *reinterpret_cast<const bf16x8_t*>(&xl[s2]);
            bf16x8_t fwh[NT], fwm[NT], fwl[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                fwh[i] = *reinterpret_cast<const bf16x8_t*>(bh + (32 * i) * ROWB + 32 * s2);
                fwm[i] = *reinterpret_cast<const bf16x8_t*>(bm + (32 * i) * ROWB + 32 * s2);
                fwl[i] = *reinterpret_cast<const bf16x8_t*>(bl + (32 * i) * ROWB + 32 * s2);
            }
            // smallest terms first; the NT tiles interleave so that no MFMA waits for the previous one's result
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwm[i], fxm, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwh[i], fxl, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwl[i], fxh, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwh[i], fxm, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwm[i], fxh, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwh[i], fxh, acc[i], 0, 0, 0);
            if (s2 == 0) split_x(xn, nh, nm, nl);     // under the MFMAs of step 0 ...
            else store_w(buf ^ 1);                    // ... and of step 1 (chunk c + 1; a harmless repeat after the last one)
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) { xh[s2] = nh[s2]; xm[s2] = nm[s2]; xl[s2] = nl[s2]; }
        __syncthreads();
    }
    const int64_t row = brow0 + wave * 32 + j;
    if (partial)
        xb_rows_epilogue<NT>(acc, row, m, n0, n, h, partial + (int64_t)blockIdx.z * m * n, n, nullptr, nullptr, 0, 0, 0.0f,
                             XbGate{});
    else
        xb_rows_epilogue<NT>(acc, row, m, n0, n, h, y, ldy, bias, residual, ldr, act, slope, gate);
}

#endif  // WS_LAB_SPLIT_GEMM

// y = act(sum_z partial[z] + bias + residual): the epilogue of a split-K gemm_xb2 (fixed order over z)
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ partial, int splits, int64_t m, int n,
                                                               float* __restrict__ y, int64_t ldy, const float* __restrict__ bias,
                                                               const float* __restrict__ residual, int64_t ldr, int act, float slope,
                                                               const XbGate gate)
{
    const int n4 = n >> 2;
    const int64_t total = m * n4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / n4;
        const int c = (int)(e % n4) * 4;
        float4 v = *reinterpret_cast<const float4*>(partial + r * n + c);
        for (int z = 1; z < splits; ++z) {
            const float4 p = *reinterpret_cast<const float4*>(partial + ((int64_t)z * m + r) * n + c);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if (bias) { const float4 bq = *reinterpret_cast<const float4*>(bias + c); v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w; }
        if (residual) {
            const float* rrow = xb_res_row(residual, ldr, gate, r);
            if (rrow) {
                const float4 rq = *reinterpret_cast<const float4*>(rrow + c);
                v.x += rq.x; v.y += rq.y; v.z += rq.z; v.w += rq.w;
            }
        }
        if (act) {
            v.x = v.x > 0.0f ? v.x : v.x * slope; v.y = v.y > 0.0f ? v.y : v.y * slope;
            v.z = v.z > 0.0f ? v.z : v.z * slope; v.w = v.w > 0.0f ? v.w : v.w * slope;
        }
        xb_gate4(v, gate, r, c);
        *reinterpret_cast<float4*>(y + r * ldy + c) = v;
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

// ---------------------------------------------------------------------------------------------
// gemm_xty2: the same reduction with NO LDS staging.  For O = X^T Y both MFMA operands index the
// tall dimension with k: lane (idx, h) of a step supplies row r + h of X (A operand, i = X column)
// and of Y (B operand, j = Y column), so a wave can load its operands straight from global memory in
// MFMA layout -- lanes 0-31 read 32*KT contiguous floats of one row, lanes 32-63 of the next.
// A lane loads KT (NT) adjacent columns at once; column c0 + KT*idx + tk belongs to the interleaved
// tile tk, so one dwordx2 load feeds two tiles.  A wave owns (32 KT) x (32 NT) outputs; the 4 waves
// of a workgroup are WK x WN tiles x WR row groups (row groups are summed through LDS at the end,
// fixed order).  Loads run U steps ahead of the MFMAs; no barrier in the main loop.
// ---------------------------------------------------------------------------------------------
// TI = float, or bf16_t for the bf16-feature path: 2-byte operand loads widened exactly to f32 in registers (the
// products and sums stay fp32: dW is the fp32 master gradient).
template <int KT, typename TI>
__device__ __forceinline__ void xty_load_cols(__amdgpu_buffer_rsrc_t srd, int off, int soff, float (&out)[KT])
{
    if constexpr (sizeof(TI) == 4) {
        if constexpr (KT == 2) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(srd, off, soff, 0);
            out[0] = __uint_as_float(v.x);
            out[1] = __uint_as_float(v.y);
        } else {
            out[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srd, off, soff, 0));
        }
    } else {
        if constexpr (KT == 2) {
            const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(srd, off, soff, 0);
            out[0] = ws_bf_lo(v);
            out[1] = ws_bf_hi(v);
        } else {
            out[0] = ws_bf_lo((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(srd, off, soff, 0));
        }
    }
}

// SPLIT (float operands only; opt-in, ws_gemm_split): the 16 rows of a block are one v_mfma_f32_32x32x16_bf16 step per
// bf16 piece pair -- both operands split three ways in registers as in gemm_xb3 (lane group h holds rows h, 2 + h, ..,
// 14 + h of the block as its 8 k; A and B use the same assignment, so the contraction is the same set of products)
template <int KT, int NT, int WK, int WN, typename TI = float, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_xty2_kernel(const TI* __restrict__ x, int64_t m, int k, int64_t ldx,
                                                         const TI* __restrict__ yy, int n, int64_t ldy,
                                                         float* __restrict__ partial, int64_t chunk)
{
    constexpr int ES = (int)sizeof(TI);
    constexpr int WR = 4 / (WK * WN);
    constexpr int U = 8;                                   // steps (row pairs) per block
    __shared__ float red[(WR > 1) ? WK * WN * KT * NT * 16 * 64 : 1];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    const int idx = lane & 31, h = lane >> 5;
    const int wr = wave / (WK * WN), wk = (wave / WN) % WK, wn = wave % WN;
    const int c0 = (blockIdx.y * WK + wk) * (32 * KT);
    const int n0 = (blockIdx.z * WN + wn) * (32 * NT);
    const int64_t mbeg = (int64_t)blockIdx.x * chunk;
    const int64_t mend = mbeg + chunk < m ? mbeg + chunk : m;

    f32x16 acc[KT][NT];
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.0f;

    // columns past the end are clamped (their outputs are never stored); rows past the end are clamped
    // and the A operand zeroed
    int xc = c0 + KT * idx;
    xc = xc + KT <= k ? xc : (k - KT > 0 ? k - KT : 0);
    int yc = n0 + NT * idx;
    yc = yc + NT <= n ? yc : (n - NT > 0 ? n - NT : 0);
    // buffer addressing: one descriptor per operand covering this workgroup's rows (wave-uniform: built
    // from kernel arguments and blockIdx only), per-lane constant byte offset, the row advance in the
    // scalar offset -> no vector address arithmetic next to the MFMAs, and rows past the end of the
    // chunk read as 0 through the descriptor's bounds check (no tail code)
    const int64_t nrows = mend - mbeg;
    const auto xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<TI*>(x + mbeg * ldx), 0,
                                                        (int)(((nrows - 1) * ldx + k) * ES), 0x00020000);
    const auto ysrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<TI*>(yy + mbeg * ldy), 0,
                                                        (int)(((nrows - 1) * ldy + n) * ES), 0x00020000);
    const int xoff = (int)(h * ldx + xc) * ES, yoff = (int)(h * ldy + yc) * ES;
    const int xstep = (int)ldx * 2 * ES, ystep = (int)ldy * 2 * ES;      // bytes per step (two rows)
    float xa[2][U][KT], ya[2][U][NT];
    auto load_block = [&](int blk, int slot) {                 // blk = index of the 2U-row block in the chunk (uniform)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int step = blk * U + u;
            xty_load_cols<KT, TI>(xsrd, xoff, step * xstep, xa[slot][u]);
            xty_load_cols<NT, TI>(ysrd, yoff, step * ystep, ya[slot][u]);
        }
    };
    auto compute = [&](int slot) {
        if constexpr (SPLIT) {
            static_assert(U == 8, "one bf16 MFMA step per block of 16 rows");
            u32x4 xp[KT][3], yp[NT][3];
#pragma unroll
            for (int a = 0; a < KT; ++a) {
                unsigned q[3][4];
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) split3_pair(xa[slot][2 * pr][a], xa[slot][2 * pr + 1][a], q[0][pr], q[1][pr], q[2][pr]);
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) xp[a][pc] = u32x4{q[pc][0], q[pc][1], q[pc][2], q[pc][3]};
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                unsigned q[3][4];
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) split3_pair(ya[slot][2 * pr][i], ya[slot][2 * pr + 1][i], q[0][pr], q[1][pr], q[2][pr]);
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) yp[i][pc] = u32x4{q[pc][0], q[pc][1], q[pc][2], q[pc][3]};
            }
            // (piece of X, piece of Y): smallest products first -- (m,m) (h,l) (l,h) (h,m) (m,h) (h,h)
            constexpr int PX[6] = {1, 0, 2, 0, 1, 0}, PY[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                for (int a = 0; a < KT; ++a)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        acc[a][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&xp[a][PX[t6]]),
                                                                            *reinterpret_cast<const bf16x8_t*>(&yp[i][PY[t6]]),
                                                                            acc[a][i], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int a = 0; a < KT; ++a)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        acc[a][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[slot][u][a], ya[slot][u][i], acc[a][i], 0, 0, 0);
        }
    };
    // row group wr takes blocks wr, wr + WR, ... of 2U rows; two blocks per trip (static register slots)
    const int nblk = (int)((nrows + 2 * U - 1) / (2 * U));
    int blk = wr;
    if (blk < nblk) load_block(blk, 0);
    for (; blk < nblk; blk += 2 * WR) {
        if (blk + WR < nblk) load_block(blk + WR, 1);
        compute(0);
        if (blk + WR < nblk) {
            if (blk + 2 * WR < nblk) load_block(blk + 2 * WR, 0);
            compute(1);
        }
    }
    if (WR > 1) {
        // sum the row groups in order 1, 2, 3 onto group 0 (layout [wk][wn][a][i][r][lane])
        const int base = ((wk * WN + wn) * KT * NT) * 16 * 64;
        for (int src = 1; src < WR; ++src) {
            if (wr == src) {
#pragma unroll
                for (int a = 0; a < KT; ++a)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) red[base + ((a * NT + i) * 16 + r) * 64 + lane] = acc[a][i][r];
            }
            __syncthreads();
            if (wr == 0) {
#pragma unroll
                for (int a = 0; a < KT; ++a)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][i][r] += red[base + ((a * NT + i) * 16 + r) * 64 + lane];
            }
            __syncthreads();
        }
    }
    if (wr == 0) {
        float* out = partial + (int64_t)blockIdx.x * k * n;
#pragma unroll
        for (int a = 0; a < KT; ++a)
            if constexpr (NT == 2) {
                // the two interleaved column tiles of a lane are adjacent columns: one 8-byte store per row (a wave
                // instruction writes two full 256-byte row segments) when the row pitch allows it
                const int col = n0 + 2 * idx;
                if ((n & 1) == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = c0 + KT * ((r & 3) + 8 * (r >> 2) + 4 * h) + a;
                        if (row < k && col + 1 < n)
                            *reinterpret_cast<float2*>(out + (int64_t)row * n + col) = make_float2(acc[a][0][r], acc[a][1][r]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = c0 + KT * ((r & 3) + 8 * (r >> 2) + 4 * h) + a;
                            if (row < k && col + i < n) out[(int64_t)row * n + col + i] = acc[a][i][r];
                        }
                }
            } else {
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int col = n0 + NT * idx + i;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = c0 + KT * ((r & 3) + 8 * (r >> 2) + 4 * h) + a;
                        if (row < k && col < n) out[(int64_t)row * n + col] = acc[a][i][r];
                    }
                }
            }
    }
}

// out[e] = sum_c partial[c][e] in a fixed order (bitwise reproducible): 32 elements x 8 chunk groups
// per workgroup, group g adds chunks g, g+8, ... (coalesced 128-byte reads), then the 8 group sums
// are added in order.
// (n, ldo: the output is [elems / n, n] with row pitch ldo; ldo == n = flat)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int64_t elems, int chunks,
                                                               float* __restrict__ out, int n = 0, int64_t ldo = 0)
{
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int64_t e = (int64_t)blockIdx.x * 32 + el;
    float s = 0.0f;
    if (e < elems) {
        int c = grp;
        for (; c + 24 < chunks; c += 32) {                 // four loads in flight, added in order
            const float v0 = partial[(int64_t)c * elems + e], v1 = partial[(int64_t)(c + 8) * elems + e];
            const float v2 = partial[(int64_t)(c + 16) * elems + e], v3 = partial[(int64_t)(c + 24) * elems + e];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; c < chunks; c += 8) s += partial[(int64_t)c * elems + e];
    }
    red[grp][el] = s;
    __syncthreads();
    if (grp == 0 && e < elems) {
        float t = red[0][el];
#pragma unroll
        for (int g2 = 1; g2 < 8; ++g2) t += red[g2][el];
        out[ldo > n ? (e / n) * ldo + e % n : e] = t;
    }
}

// the same sum for LARGE outputs of few chunks (dW of the deep levels: k n up to 4 M elements, <= ~16 chunks): a thread owns
// four consecutive elements and adds the chunks in order 0, 1, 2, ... (four 16-byte loads in flight)
__global__ __launch_bounds__(256) void reduce_partials_wide_kernel(const float* __restrict__ partial, int64_t elems, int chunks,
                                                                    float* __restrict__ out, int n = 0, int64_t ldo = 0)
{
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= elems) return;
    const float4* p = reinterpret_cast<const float4*>(partial + e);
    const int64_t st = elems / 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int c = 0;
    for (; c + 3 < chunks; c += 4) {
        const float4 v0 = p[(int64_t)c * st], v1 = p[(int64_t)(c + 1) * st], v2 = p[(int64_t)(c + 2) * st], v3 = p[(int64_t)(c + 3) * st];
        s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
        s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
        s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
        s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
    }
    for (; c < chunks; ++c) {
        const float4 v = p[(int64_t)c * st];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (ldo > n ? (e / n) * ldo + e % n : e)) = s;      // (n % 4 == 0: a quad stays in its row)
}

// ---------------------------------------------------------------------------------------------
// dz = LeakyReLU'(y) * dy and the column sums of dz (the bias gradient of a BatchNormBlock,
// models/blocks.py:465) in ONE pass over [m, n]: partial column sums per row chunk (threads = column
// groups x row lanes, row lanes added through LDS in a fixed order), chunks added by
// reduce_partials_kernel.  V = 4: float4 columns; V = 1: any n.
// ---------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void act_bwd_colsum_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                                              int64_t m, int n, int64_t lddy, int64_t ldy, float slope,
                                                              float* __restrict__ dz, int64_t lddz,
                                                              float* __restrict__ partial, int64_t chunk, const WsDrop drop)
{
    // drop.on: dy is the gradient of a DROPPED tensor (nn.Dropout applied to the activated output y, fused into the producing
    // epilogue): g = keep(row * n + col) ? g * scale : 0 first, exactly the separate dropout backward, then the activation
    __shared__ float red[256 * V];
    const int t = threadIdx.x;
    const int ncg = (n + V - 1) / V;                    // column groups
    const int per = ncg < 256 ? ncg : 256;              // column groups per pass
    const int R = 256 / per;                            // row lanes
    const int rl = t / per, cgl = t % per;
    const int64_t mbeg = (int64_t)blockIdx.x * chunk;
    const int64_t mend = mbeg + chunk < m ? mbeg + chunk : m;
    {   // one pass of `per` column groups per workgroup row of the grid (blockIdx.y)
        const int cg = (int)blockIdx.y * per + cgl;
        const int col = cg * V;
        float s[V];
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] = 0.0f;
        if (rl < R && cg < ncg) {
            typedef float vec_t __attribute__((ext_vector_type(V)));
            const int64_t sdy = (int64_t)R * lddy, sy = (int64_t)R * ldy, sdz = (int64_t)R * lddz;
            const float* pg = dy + (mbeg + rl) * lddy + col;
            const float* pa = yact ? yact + (mbeg + rl) * ldy + col : nullptr;
            float* pz = yact ? dz + (mbeg + rl) * lddz + col : nullptr;
            auto one = [&](vec_t g, vec_t a, float* out, int64_t row) {
                if (drop.on) {
                    const unsigned long long i0 = (unsigned long long)(row * n + col);
#pragma unroll
                    for (int e = 0; e < V; ++e) g[e] = ws_drop1(g[e], drop, i0 + e);
                }
                if (pa) {
#pragma unroll
                    for (int e = 0; e < V; ++e) g[e] = a[e] > 0.0f ? g[e] : g[e] * slope;
                    *reinterpret_cast<vec_t*>(out) = g;
                }
#pragma unroll
                for (int e = 0; e < V; ++e) s[e] += g[e];
            };
            int64_t r = mbeg + rl;
            for (; r + 3 * R < mend; r += 4 * R) {        // four independent rows in flight per thread
                vec_t g0 = *reinterpret_cast<const vec_t*>(pg), g1 = *reinterpret_cast<const vec_t*>(pg + sdy);
                vec_t g2 = *reinterpret_cast<const vec_t*>(pg + 2 * sdy), g3 = *reinterpret_cast<const vec_t*>(pg + 3 * sdy);
                vec_t a0 = g0, a1 = g1, a2 = g2, a3 = g3;
                if (pa) {
                    a0 = *reinterpret_cast<const vec_t*>(pa); a1 = *reinterpret_cast<const vec_t*>(pa + sy);
                    a2 = *reinterpret_cast<const vec_t*>(pa + 2 * sy); a3 = *reinterpret_cast<const vec_t*>(pa + 3 * sy);
                }
                one(g0, a0, pz, r); one(g1, a1, pz + sdz, r + R); one(g2, a2, pz + 2 * sdz, r + 2 * R); one(g3, a3, pz + 3 * sdz, r + 3 * R);
                pg += 4 * sdy;
                if (pa) { pa += 4 * sy; pz += 4 * sdz; }
            }
            for (; r < mend; r += R) {
                vec_t g0 = *reinterpret_cast<const vec_t*>(pg);
                vec_t a0 = pa ? *reinterpret_cast<const vec_t*>(pa) : g0;
                one(g0, a0, pz, r);
                pg += sdy;
                if (pa) { pa += sy; pz += sdz; }
            }
        }
        if (partial) {
#pragma unroll
            for (int e = 0; e < V; ++e) red[t * V + e] = s[e];
            __syncthreads();
            if (rl == 0 && cg < ncg) {
                for (int q = 1; q < R; ++q)
#pragma unroll
                    for (int e = 0; e < V; ++e) s[e] += red[(q * per + cgl) * V + e];
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (col + e < n) partial[(int64_t)blockIdx.x * n + col + e] = s[e];
            }
            __syncthreads();
        }
    }
}

int64_t colsum_chunk(int64_t m)
{
    int64_t c = ws_ceil_div(m, 768);
    return c < 16 ? 16 : c;                             // <= 768 chunks (three workgroups per CU) of >= 16 rows
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// chunk sums -> out: the wide form for large outputs of few chunks, the grouped form otherwise
void launch_reduce_partials(const float* partial, int64_t elems, int chunks, float* out, hipStream_t st, int n = 0, int64_t ldo = 0)
{
    if (elems >= 65536 && chunks <= 64 && elems % 4 == 0 && al16(partial) && al16(out) && (ldo <= n || (n % 4 == 0 && ldo % 4 == 0)))
        reduce_partials_wide_kernel<<<(unsigned)ws_ceil_div(elems, 1024), 256, 0, st>>>(partial, elems, chunks, out, n, ldo);
    else
        reduce_partials_kernel<<<(unsigned)ws_ceil_div(elems, 32), 256, 0, st>>>(partial, elems, chunks, out, n, ldo);
}

int64_t xty_chunk(int64_t m, int k, int n)
{
    if (m >= 32768) {
        // tall operands: <= 256 chunks, a multiple of BK rows each, at least 512 rows (keeps >= 256 workgroups in
        // flight even when K fits one k-tile)
        int64_t c = ws_ceil_div(m, 256);
        if (c < 512) c = 512;
        return ws_ceil_div(c, BK) * BK;
    }
    // short operands (the deep pyramid levels): one chunk is a serial chain of m / 2 MFMA steps per wave, so the rows are
    // split until the chain (A / chunks) balances the write + re-read of the partial outputs (B * chunks):
    //   A = m/2 steps x 4 tiles x 64 cycles at 2.4 GHz,  B = 2 x 4 k n bytes at ~2 TB/s (measured: scattered partial stores),  chunks = sqrt(A / B),
    // at most 256 workgroups (tiles of 128 x 128 outputs x chunks: one per CU -- beyond that the MFMA pipes are shared and
    // more chunks only add partial traffic; 257..511 workgroups would double the time) and at least 32 rows per chunk.
    const double a_us = (double)m * 0.5 * 4.0 * 64.0 / 2400.0;
    const double b_us = 16.0 * (double)k * (double)n / 4.0e6;
    const int64_t tiles = ws_ceil_div(k, 128) * ws_ceil_div(n, 128);
    int64_t chunks = (int64_t)(__builtin_sqrt(a_us / b_us) + 0.5);
    if (chunks > 256 / tiles) chunks = 256 / tiles;
    if (chunks > m / 32) chunks = m / 32;
    if (chunks < 1) chunks = 1;
    return ws_ceil_div(ws_ceil_div(m, chunks), BK) * BK;
}

}  // namespace

extern "C" {

// diagnostic switches of tools/gemm_lab.cpp (not part of the drop-in surface of include/weasal_hip.h)
int ws_gemm_wave_cols = 0;  // forced wave grid of gemm_xb2 (column groups 1 / 2), 0 = automatic
int ws_gemm_thin_k = 64;    // products with k <= this take 64-column tiles (more, lighter workgroups: they are all prologue and epilogue) instead of 128
int ws_gemm_shallow = 1;     // products with k <= 64 that the MFMA tiles do not take (k % 32 != 0), n % 4 == 0, many rows: 1 = gemm_xb_shallow_kernel, 0 = gemm_xb_kernel
int ws_gemm_staged = 1;     // gemm_xb2 epilogue: 1 = the tile turned through LDS (whole 128-byte row segments per store), 0 = per-lane rows
int ws_gemm_variant = 2;    // 1 = LDS-staged tiles (gemm_xb / gemm_xty), 2 = operands straight from global memory
#ifdef WS_LAB_SPLIT_GEMM
int ws_gemm_split = 0;      // 0 = the f32-input MFMA (default: the benchmark's fp32 numbers are measured on it);
                            // 1 = y = x b with every fp32 product as six bf16 MFMA partial products of exact three-way splits
                            // (gemm_xb3; WEASAL_GEMM_SPLIT=1): error against float64 not above the f32-input kernels, 15 % less
                            // time per product at the tall shapes, 2.3 % per DALES training step (12.26 -> 11.98 ms).  Opt-in:
                            // the gain does not pay for a second arithmetic behind the label "fp32".  2 = the same with 64-column tiles
#endif

int64_t ws_gemm_xb_scratch_bytes(int64_t m, int32_t k, int32_t n)
{
    // room for up to 16 split-K partial outputs; only short, deep products use it
    if (m <= 0 || m >= 32768 || k < 512) return 0;
    return 16 * m * (int64_t)n * (int64_t)sizeof(float);
}

// lab switch (tools/gemm_shapes.py): every product of a step timed on its own (events + synchronisation) and printed
extern "C" int ws_gemm_log = 0;

static int gemm_xb_core(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n,
                        const float* bias, const float* residual, int64_t ldr, int32_t act, float slope,
                        float* y, int64_t ldy, void* scratch, int64_t scratch_bytes, void* stream,
                        int64_t brs, int64_t bcs, XbGate gate);

static int gemm_xb_impl(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n,
                        const float* bias, const float* residual, int64_t ldr, int32_t act, float slope,
                        float* y, int64_t ldy, void* scratch, int64_t scratch_bytes, void* stream,
                        int64_t brs = -1, int64_t bcs = 1, XbGate gate = XbGate{})
{
    if (!ws_gemm_log)
        return gemm_xb_core(x, m, k, ldx, b, n, bias, residual, ldr, act, slope, y, ldy, scratch, scratch_bytes, stream, brs, bcs, gate);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    WS_HIP(hipEventCreate(&e0)); WS_HIP(hipEventCreate(&e1));
    WS_HIP(hipEventRecord(e0, st));
    const int rc = gemm_xb_core(x, m, k, ldx, b, n, bias, residual, ldr, act, slope, y, ldy, scratch, scratch_bytes, stream, brs, bcs, gate);
    WS_HIP(hipEventRecord(e1, st));
    WS_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    WS_HIP(hipEventElapsedTime(&ms, e0, e1));
    fprintf(stderr, "GEMMLOG xb m=%lld k=%d n=%d us=%.1f tflops=%.1f\n", (long long)m, k, n, ms * 1e3, 2.0 * m * k * n / (ms * 1e-3) / 1e12);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

static int gemm_xb_core(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n,
                        const float* bias, const float* residual, int64_t ldr, int32_t act, float slope,
                        float* y, int64_t ldy, void* scratch, int64_t scratch_bytes, void* stream,
                        int64_t brs, int64_t bcs, XbGate gate)
{
    if (brs < 0) brs = n;                       // row-major [K,N]
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
    if (ws_gemm_variant == 2 && vecx && k % 32 == 0 && n % 4 == 0 && al16(y) && ldy % 4 == 0 &&
        (!residual || (al16(residual) && ldr % 4 == 0)) && (!bias || al16(bias)) &&
        (int64_t)(k - 1) * brs + (int64_t)(n - 1) * bcs < (1ll << 29) && 128 * ldx < (1ll << 29) && al16(b) &&
        (bcs == 1 ? true : (brs == 1 && bcs % 4 == 0)) &&
        (!gate.y || (al16(gate.y) && gate.ld % 4 == 0)) && (!gate.mask || ((reinterpret_cast<uintptr_t>(gate.mask) & 3u) == 0 && gate.ldm % 4 == 0))) {
#define WS_XB2(NTV, WNV)                                                                                        \
    do {                                                                                                        \
        const unsigned gx2 = (unsigned)ws_ceil_div(m, 32 * (4 / WNV)), gy2 = (unsigned)ws_ceil_div(n, 32 * NTV * WNV); \
        const int nch = k / 32;                                                                                 \
        int splits = 1;                                                                                         \
        if (scratch && (int64_t)gx2 * gy2 < 256 && nch >= 16) {                                                 \
            splits = (int)ws_ceil_div(768, (int64_t)gx2 * gy2);                                                 \
            if (splits > nch / 8) splits = nch / 8;                                                             \
            if (splits > 16) splits = 16;                                                                       \
            if ((int64_t)splits * m * n * 4 > scratch_bytes) splits = (int)(scratch_bytes / (m * n * 4));       \
            if (splits < 2) splits = 1;                                                                         \
        }                                                                                                       \
        const int csplit = (int)ws_ceil_div(nch, splits);                                                       \
        splits = (int)ws_ceil_div(nch, csplit);                                                                 \
        float* part = splits > 1 ? (float*)scratch : nullptr;                                                   \
        gemm_xb2_kernel<NTV, WNV><<<dim3(gx2, gy2, (unsigned)splits), 256, 0, st>>>(x, m, k, ldx, b, n, (int)brs, (int)bcs, y, ldy, bias, residual, \
                                                                                    ldr, act, slope, csplit, part, gate, ws_gemm_staged);   \
        if (splits > 1)                                                                                         \
            splitk_epilogue_kernel<<<ws_grid(m * (n / 4), 256), 256, 0, st>>>(part, splits, m, n, y, ldy, bias, residual, ldr, \
                                                                              act, slope, gate);                \
    } while (0)
#ifdef WS_LAB_SPLIT_GEMM
        if (ws_gemm_split) {
            // fp32 products as six bf16 MFMA partial products (gemm_xb3): 128 rows x 32 NT columns per workgroup
#define WS_XB3(NTV)                                                                                             \
    do {                                                                                                        \
        const unsigned gx3 = (unsigned)ws_ceil_div(m, 128), gy3 = (unsigned)ws_ceil_div(n, 32 * NTV);           \
        const int nch = k / 32;                                                                                 \
        int splits = 1;                                                                                         \
        if (scratch && (int64_t)gx3 * gy3 < 256 && nch >= 8) {                                                  \
            splits = (int)ws_ceil_div(512, (int64_t)gx3 * gy3);                                                 \
            if (splits > nch / 4) splits = nch / 4;                                                             \
            if (splits > 16) splits = 16;                                                                       \
            if ((int64_t)splits * m * n * 4 > scratch_bytes) splits = (int)(scratch_bytes / (m * n * 4));       \
            if (splits < 2) splits = 1;                                                                         \
        }                                                                                                       \
        const int csplit = (int)ws_ceil_div(nch, splits);                                                       \
        splits = (int)ws_ceil_div(nch, csplit);                                                                 \
        float* part = splits > 1 ? (float*)scratch : nullptr;                                                   \
        gemm_xb3_kernel<NTV><<<dim3(gx3, gy3, (unsigned)splits), 256, 0, st>>>(x, m, k, ldx, b, n, (int)brs, (int)bcs, y, ldy, bias, \
                                                                               residual, ldr, act, slope, csplit, part, gate); \
        if (splits > 1)                                                                                         \
            splitk_epilogue_kernel<<<ws_grid(m * (n / 4), 256), 256, 0, st>>>(part, splits, m, n, y, ldy, bias, residual, ldr, \
                                                                              act, slope, gate);                \
    } while (0)
            if (n <= 32) WS_XB3(1);
            else if (n <= 64 || ws_gemm_split == 2) WS_XB3(2);
            else WS_XB3(4);
#undef WS_XB3
            WS_LAUNCH_CHECK();
            return WS_OK;
        }
#endif  // WS_LAB_SPLIT_GEMM
        const int64_t tiles = ws_ceil_div(m, 32);
        int wn = tiles >= 2048 ? 1 : 2;
        if (ws_gemm_wave_cols) wn = ws_gemm_wave_cols;
        if (n <= 32) wn = 1;
        if (wn == 1) {
            if (n <= 32) WS_XB2(1, 1);
            else if (n <= 64 || (ws_gemm_thin_k > 0 && k <= ws_gemm_thin_k)) WS_XB2(2, 1);
            else WS_XB2(4, 1);
        } else {
            if (n <= 64) WS_XB2(1, 2);
            else WS_XB2(2, 2);
        }
#undef WS_XB2
        WS_LAUNCH_CHECK();
        return WS_OK;
    }
    WS_REQUIRE(bcs == 1 && brs == n, "a strided small matrix needs k %% 32 == 0, n %% 4 == 0 and 16-byte aligned rows "
                                     "(k=%d n=%d): pass a row-major [K,N] copy for this shape", k, n);
    if (ws_gemm_shallow && k <= 64 && n % 4 == 0 && n <= 1024 && (int64_t)k * n <= 8192 && m >= 4096 && al16(b) && al16(y) && ldy % 4 == 0 &&
        (!residual || (al16(residual) && ldr % 4 == 0)) && (!bias || al16(bias)) && (!gate.y || (al16(gate.y) && gate.ld % 4 == 0)) &&
        (!gate.mask || ((reinterpret_cast<uintptr_t>(gate.mask) & 3u) == 0 && gate.ldm % 4 == 0))) {
        // shallow, ragged contraction over many rows: the streaming VALU form (gemm_xb_shallow_kernel)
        const int per = 256 / (n / 4);
        const int kp = (k + 3) & ~3;
        int rows = per * 8;                                             // 8 row passes per workgroup
        gemm_xb_shallow_kernel<<<(unsigned)ws_ceil_div(m, rows), 256, sizeof(float) * ((size_t)kp * n + (size_t)rows * kp), st>>>(
            x, m, k, ldx, b, n, y, ldy, bias, residual, ldr, act, slope, gate, rows);
        WS_LAUNCH_CHECK();
        return WS_OK;
    }
    if (n <= 32) {
        gemm_xb_kernel<1, 32><<<dim3((unsigned)gx, 1), 256, 0, st>>>(x, m, k, ldx, b, n, n, y, ldy, vecx, vecb, bias, residual, ldr,
                                                                 act, slope, gate);
    } else if (n <= 64) {
        gemm_xb_kernel<2, 32><<<dim3((unsigned)gx, 1), 256, 0, st>>>(x, m, k, ldx, b, n, n, y, ldy, vecx, vecb, bias, residual, ldr,
                                                                 act, slope, gate);
    } else {
        // measured (tools/gemm_bench.py, M = 400k): shallow K is latency bound and prefers the 64-column
        // tile (more waves per SIMD, L2 serves the X re-read); deep K prefers the 128-column tile (X reuse)
        if (k < 128)
            gemm_xb_kernel<2, 32><<<dim3((unsigned)gx, (unsigned)ws_ceil_div(n, 64)), 256, 0, st>>>(
                x, m, k, ldx, b, n, n, y, ldy, vecx, vecb, bias, residual, ldr, act, slope, gate);
        else
            gemm_xb_kernel<4, 32><<<dim3((unsigned)gx, (unsigned)ws_ceil_div(n, 128)), 256, 0, st>>>(
                x, m, k, ldx, b, n, n, y, ldy, vecx, vecb, bias, residual, ldr, act, slope, gate);
    }
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int64_t ws_act_bwd_colsum_scratch_bytes(int64_t m, int32_t n)
{
    return ws_ceil_div(m > 0 ? m : 1, colsum_chunk(m)) * (int64_t)n * (int64_t)sizeof(float);
}

static int act_bwd_colsum_impl(const float* dy, int64_t m, int32_t n, int64_t lddy, const float* y, int64_t ldy, float slope,
                               float* dz, int64_t lddz, float* colsum, void* scratch, void* stream, WsDrop drop);

int ws_act_bwd_colsum(const float* dy, int64_t m, int32_t n, int64_t lddy, const float* y, int64_t ldy, float slope,
                      float* dz, int64_t lddz, float* colsum, void* scratch, void* stream)
{
    return act_bwd_colsum_impl(dy, m, n, lddy, y, ldy, slope, dz, lddz, colsum, scratch, stream, WsDrop{});
}

int ws_act_bwd_colsum_dropout(const float* dy, int64_t m, int32_t n, int64_t lddy, const float* y, int64_t ldy, float slope,
                              float drop_p, uint64_t drop_seed, float* dz, int64_t lddz, float* colsum, void* scratch, void* stream)
{
    WS_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "bad drop probability %g", (double)drop_p);
    WS_REQUIRE(y && dz, "the dropout form writes dz (it needs y and dz)");
    return act_bwd_colsum_impl(dy, m, n, lddy, y, ldy, slope, dz, lddz, colsum, scratch, stream, ws_drop_args(drop_p, drop_seed));
}

static int act_bwd_colsum_impl(const float* dy, int64_t m, int32_t n, int64_t lddy, const float* y, int64_t ldy, float slope,
                               float* dz, int64_t lddz, float* colsum, void* scratch, void* stream, WsDrop drop)
{
    WS_REQUIRE(m >= 0 && n >= 1 && lddy >= n, "bad sizes m=%lld n=%d", (long long)m, n);
    WS_REQUIRE(!y || (dz && ldy >= n && lddz >= n), "activation backward needs y and dz");
    WS_REQUIRE(!colsum || scratch, "column sums need scratch");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        if (colsum) WS_HIP(hipMemsetAsync(colsum, 0, sizeof(float) * (size_t)n, st));
        return WS_OK;
    }
    WS_REQUIRE(dy && (y || colsum), "NULL argument");
    const int64_t chunk = colsum_chunk(m);
    const int chunks = (int)ws_ceil_div(m, chunk);
    float* partial = colsum ? (chunks == 1 ? colsum : (float*)scratch) : nullptr;
    const bool vec = n % 4 == 0 && al16(dy) && lddy % 4 == 0 && (!y || (al16(y) && ldy % 4 == 0 && al16(dz) && lddz % 4 == 0));
    if (vec)
        act_bwd_colsum_kernel<4><<<dim3(chunks, (unsigned)ws_ceil_div(ws_ceil_div(n, 4), 256)), 256, 0, st>>>(
            dy, y, m, n, lddy, ldy, slope, dz, lddz, partial, chunk, drop);
    else
        act_bwd_colsum_kernel<1><<<dim3(chunks, (unsigned)ws_ceil_div(n, 256)), 256, 0, st>>>(dy, y, m, n, lddy, ldy, slope, dz, lddz,
                                                                                              partial, chunk, drop);
    WS_LAUNCH_CHECK();
    if (colsum && chunks > 1) {
        reduce_partials_kernel<<<(unsigned)ws_ceil_div(n, 32), 256, 0, st>>>(partial, n, chunks, colsum);
        WS_LAUNCH_CHECK();
    }
    return WS_OK;
}

int ws_gemm_xb_epilogue(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n,
                        const float* bias, const float* residual, int64_t ldr, int32_t act, float slope,
                        float* y, int64_t ldy, void* stream)
{
    return gemm_xb_impl(x, m, k, ldx, b, n, bias, residual, ldr, act, slope, y, ldy, nullptr, 0, stream);
}

int ws_gemm_xb_epilogue_splitk(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n,
                               const float* bias, const float* residual, int64_t ldr, int32_t act, float slope,
                               float* y, int64_t ldy, void* scratch, int64_t scratch_bytes, void* stream)
{
    return gemm_xb_impl(x, m, k, ldx, b, n, bias, residual, ldr, act, slope, y, ldy, scratch, scratch_bytes, stream);
}

int ws_gemm_xb_epilogue_strided(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int64_t b_row_stride,
                                int64_t b_col_stride, int32_t n, const float* bias, const float* residual, int64_t ldr,
                                int32_t act, float slope, float* y, int64_t ldy, void* scratch, int64_t scratch_bytes,
                                void* stream)
{
    return gemm_xb_impl(x, m, k, ldx, b, n, bias, residual, ldr, act, slope, y, ldy, scratch, scratch_bytes, stream,
                        b_row_stride, b_col_stride);
}

int ws_gemm_xb_gated_strided(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int64_t b_row_stride,
                             int64_t b_col_stride, int32_t n, const float* bias, const float* residual, int64_t ldr,
                             int32_t act, float slope, const float* gate_y, int64_t ldg, float gate_slope, const uint8_t* mask,
                             int64_t ldm, float mask_scale, float* y, int64_t ldy, void* scratch, int64_t scratch_bytes,
                             void* stream)
{
    WS_REQUIRE(!gate_y || ldg >= n, "gate leading dimension too small");
    WS_REQUIRE(!mask || ldm >= n, "mask leading dimension too small");
    return gemm_xb_impl(x, m, k, ldx, b, n, bias, residual, ldr, act, slope, y, ldy, scratch, scratch_bytes, stream,
                        b_row_stride, b_col_stride, XbGate{gate_y, ldg, gate_slope, mask, ldm, mask_scale, WsDrop{}, 0, nullptr, 0, 0});
}

int ws_gemm_xb_dropout_strided(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int64_t b_row_stride,
                               int64_t b_col_stride, int32_t n, const float* bias, const float* residual, int64_t ldr,
                               int32_t act, float slope, float drop_p, uint64_t drop_seed, float* y, int64_t ldy, void* scratch,
                               int64_t scratch_bytes, void* stream)
{
    WS_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "bad drop probability %g", (double)drop_p);
    XbGate gate{};
    gate.drop = ws_drop_args(drop_p, drop_seed);
    gate.dn = n;
    return gemm_xb_impl(x, m, k, ldx, b, n, bias, residual, ldr, act, slope, y, ldy, scratch, scratch_bytes, stream,
                        b_row_stride, b_col_stride, gate);
}

int ws_gemm_xb_gate_dropout(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n, const float* gate_y,
                            int64_t ldg, float gate_slope, float drop_p, uint64_t drop_seed, float* y, int64_t ldy, void* scratch,
                            int64_t scratch_bytes, void* stream)
{
    WS_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "bad drop probability %g", (double)drop_p);
    WS_REQUIRE(!gate_y || ldg >= n, "gate leading dimension too small");
    XbGate gate{};
    gate.y = gate_y; gate.ld = ldg; gate.slope = gate_slope;
    gate.drop = ws_drop_args(drop_p, drop_seed);
    gate.dn = n;
    return gemm_xb_impl(x, m, k, ldx, b, n, nullptr, nullptr, 0, 0, 0.0f, y, ldy, scratch, scratch_bytes, stream, -1, 1, gate);
}

// private to the library (ws_common.h): the strided product with the whole epilogue menu -- dropout (drop_p > 0) and / or a
// gathered residual (res_rows != NULL: row r adds residual[res_rows[r * res_rows_ld]], indices outside [0, res_nrows) add nothing)
int ws_priv_gemm_xb_ex(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int64_t b_row_stride, int64_t b_col_stride,
                       int32_t n, const float* bias, const float* residual, int64_t ldr, const int64_t* res_rows, int64_t res_rows_ld,
                       int64_t res_nrows, int32_t act, float slope, float drop_p, uint64_t drop_seed, float* y, int64_t ldy,
                       void* scratch, int64_t scratch_bytes, void* stream)
{
    WS_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "bad drop probability %g", (double)drop_p);
    WS_REQUIRE(!res_rows || (residual && res_rows_ld >= 1 && res_nrows >= 0), "gathered residual: NULL residual / bad sizes");
    XbGate gate{};
    gate.drop = ws_drop_args(drop_p, drop_seed);
    gate.dn = n;
    gate.rrows = res_rows; gate.rld = res_rows_ld; gate.rn = res_nrows;
    return gemm_xb_impl(x, m, k, ldx, b, n, bias, residual, ldr, act, slope, y, ldy, scratch, scratch_bytes, stream,
                        b_row_stride, b_col_stride, gate);
}

int ws_gemm_xb(const float* x, int64_t m, int32_t k, int64_t ldx, const float* b, int32_t n, float* y, int64_t ldy,
               void* stream)
{
    return gemm_xb_impl(x, m, k, ldx, b, n, nullptr, nullptr, 0, 0, 0.0f, y, ldy, nullptr, 0, stream);
}

int64_t ws_gemm_xty_scratch_bytes(int64_t m, int32_t k, int32_t n)
{
    const int64_t chunk = xty_chunk(m, k, n);
    return ws_ceil_div(m > 0 ? m : 1, chunk) * (int64_t)k * n * (int64_t)sizeof(float);
}

static int gemm_xty_core(const float* x, int64_t m, int32_t k, int64_t ldx, const float* y, int32_t n, int64_t ldy,
                         float* out, void* scratch, void* stream, int64_t ldo = 0);

int ws_gemm_xty(const float* x, int64_t m, int32_t k, int64_t ldx, const float* y, int32_t n, int64_t ldy,
                float* out, void* scratch, void* stream)
{
    if (!ws_gemm_log) return gemm_xty_core(x, m, k, ldx, y, n, ldy, out, scratch, stream);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    WS_HIP(hipEventCreate(&e0)); WS_HIP(hipEventCreate(&e1));
    WS_HIP(hipEventRecord(e0, st));
    const int rc = gemm_xty_core(x, m, k, ldx, y, n, ldy, out, scratch, stream);
    WS_HIP(hipEventRecord(e1, st));
    WS_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    WS_HIP(hipEventElapsedTime(&ms, e0, e1));
    fprintf(stderr, "GEMMLOG xty m=%lld k=%d n=%d us=%.1f tflops=%.1f\n", (long long)m, k, n, ms * 1e3, 2.0 * m * k * n / (ms * 1e-3) / 1e12);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

static int gemm_xty_core(const float* x, int64_t m, int32_t k, int64_t ldx, const float* y, int32_t n, int64_t ldy,
                         float* out, void* scratch, void* stream, int64_t ldo)
{
    // ldo > n: `out` is a column block of a wider matrix (row pitch ldo): the chunk sums go through the scratch buffer and the
    // reduction writes the pitched rows (also for a single chunk, where it is the copy a caller would otherwise make)
    const bool pitched = ldo > n;
    WS_REQUIRE(m >= 0 && k >= 1 && n >= 1 && ldx >= k && ldy >= n, "bad sizes m=%lld k=%d n=%d", (long long)m, k, n);
    WS_REQUIRE(out, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        if (pitched) WS_HIP(hipMemset2DAsync(out, sizeof(float) * (size_t)ldo, 0, sizeof(float) * (size_t)n, (size_t)k, st));
        else WS_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)k * n, st));
        return WS_OK;
    }
    WS_REQUIRE(x && y && scratch, "NULL argument");
    const int64_t chunk = xty_chunk(m, k, n);
    const int chunks = (int)ws_ceil_div(m, chunk);
    const int vecx = al16(x) && (ldx % 4 == 0);
    const int vecy = al16(y) && (ldy % 4 == 0);
    float* partial = (chunks == 1 && !pitched) ? out : (float*)scratch;
    const bool al8x = (reinterpret_cast<uintptr_t>(x) & 7u) == 0 && ldx % 2 == 0;
    const bool al8y = (reinterpret_cast<uintptr_t>(y) & 7u) == 0 && ldy % 2 == 0;
    if (ws_gemm_variant == 2 && chunk * (ldx > ldy ? ldx : ldy) * 4 < (1ll << 31)) {
        // per-wave tile (32 KT) x (32 NT); waves WK x WN over the output, the rest split the rows
        const int kt = (k > 32 && al8x) ? 2 : 1, nt = (n > 32 && al8y) ? 2 : 1;
        const int wk = k > 32 * kt ? 2 : 1;
        const int wn = (n > 32 * nt && wk == 1) || (n > 32 * nt && k > 32 * kt) ? 2 : 1;
#ifdef WS_LAB_SPLIT_GEMM
#define WS_XTY2_SPLIT(KTV, NTV, WKV, WNV)                                                                             \
        if (ws_gemm_split)                                                                                            \
            gemm_xty2_kernel<KTV, NTV, WKV, WNV, float, true><<<g3, 256, 0, st>>>(x, m, k, ldx, y, n, ldy, partial, chunk); \
        else
#else
#define WS_XTY2_SPLIT(KTV, NTV, WKV, WNV)
#endif
#define WS_XTY2(KTV, NTV, WKV, WNV)                                                                                   \
    do {                                                                                                              \
        const dim3 g3(chunks, (unsigned)ws_ceil_div(k, 32 * KTV * WKV), (unsigned)ws_ceil_div(n, 32 * NTV * WNV));    \
        WS_XTY2_SPLIT(KTV, NTV, WKV, WNV)                                                                             \
            gemm_xty2_kernel<KTV, NTV, WKV, WNV><<<g3, 256, 0, st>>>(x, m, k, ldx, y, n, ldy, partial, chunk);        \
    } while (0)
#define WS_XTY2_W(KTV, NTV)                          \
    do {                                             \
        if (wk == 2 && wn == 2) WS_XTY2(KTV, NTV, 2, 2); \
        else if (wk == 2) WS_XTY2(KTV, NTV, 2, 1);   \
        else if (wn == 2) WS_XTY2(KTV, NTV, 1, 2);   \
        else WS_XTY2(KTV, NTV, 1, 1);                \
    } while (0)
        if (kt == 2 && nt == 2) WS_XTY2_W(2, 2);
        else if (kt == 2) WS_XTY2_W(2, 1);
        else if (nt == 2) WS_XTY2_W(1, 2);
        else WS_XTY2_W(1, 1);
#undef WS_XTY2_W
#undef WS_XTY2
    } else {
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
    }
    WS_LAUNCH_CHECK();
    if (chunks > 1 || pitched) {
        const int64_t elems = (int64_t)k * n;
        launch_reduce_partials(partial, elems, chunks, out, st, n, pitched ? ldo : 0);
        WS_LAUNCH_CHECK();
    }
    return WS_OK;
}

// private to the library (ws_common.h): dW written as a column block of a wider matrix (row pitch ldo >= n)
int ws_priv_gemm_xty_pitched(const float* x, int64_t m, int32_t k, int64_t ldx, const float* y, int32_t n, int64_t ldy, float* out,
                             int64_t ldo, void* scratch, void* stream)
{
    WS_REQUIRE(ldo >= n, "output pitch smaller than a row");
    return gemm_xty_core(x, m, k, ldx, y, n, ldy, out, scratch, stream, ldo);
}


// dW = X^T dY with bf16 rows (BASELINE config 5): the LDS-free reduction above with 2-byte operand loads.
// Requirements: k, n even or <= 32 handled by the one-column form; every shape of the bf16 path qualifies.
int ws_gemm_xty_bf16(const uint16_t* x, int64_t m, int32_t k, int64_t ldx, const uint16_t* y, int32_t n, int64_t ldy,
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
    WS_REQUIRE(chunk * (ldx > ldy ? ldx : ldy) * 2 < (1ll << 31), "operand exceeds the 32-bit buffer offsets");
    float* partial = chunks == 1 ? out : (float*)scratch;
    const bf16_t* xb = reinterpret_cast<const bf16_t*>(x);
    const bf16_t* yb = reinterpret_cast<const bf16_t*>(y);
    const bool al4x = (reinterpret_cast<uintptr_t>(x) & 3u) == 0 && ldx % 2 == 0 && k % 2 == 0;
    const bool al4y = (reinterpret_cast<uintptr_t>(y) & 3u) == 0 && ldy % 2 == 0 && n % 2 == 0;
    const int kt = (k > 32 && al4x) ? 2 : 1, nt = (n > 32 && al4y) ? 2 : 1;
    const int wk = k > 32 * kt ? 2 : 1;
    const int wn = (n > 32 * nt && wk == 1) || (n > 32 * nt && k > 32 * kt) ? 2 : 1;
#define WS_XTY2B(KTV, NTV, WKV, WNV)                                                                                  \
    gemm_xty2_kernel<KTV, NTV, WKV, WNV, bf16_t><<<dim3(chunks, (unsigned)ws_ceil_div(k, 32 * KTV * WKV),             \
                                                        (unsigned)ws_ceil_div(n, 32 * NTV * WNV)), 256, 0, st>>>(     \
        xb, m, k, ldx, yb, n, ldy, partial, chunk)
#define WS_XTY2B_W(KTV, NTV)                          \
    do {                                              \
        if (wk == 2 && wn == 2) WS_XTY2B(KTV, NTV, 2, 2); \
        else if (wk == 2) WS_XTY2B(KTV, NTV, 2, 1);   \
        else if (wn == 2) WS_XTY2B(KTV, NTV, 1, 2);   \
        else WS_XTY2B(KTV, NTV, 1, 1);                \
    } while (0)
    if (kt == 2 && nt == 2) WS_XTY2B_W(2, 2);
    else if (kt == 2) WS_XTY2B_W(2, 1);
    else if (nt == 2) WS_XTY2B_W(1, 2);
    else WS_XTY2B_W(1, 1);
#undef WS_XTY2B_W
#undef WS_XTY2B
    WS_LAUNCH_CHECK();
    if (chunks > 1) {
        const int64_t elems = (int64_t)k * n;
        launch_reduce_partials(partial, elems, chunks, out, st);
        WS_LAUNCH_CHECK();
    }
    return WS_OK;
}

}  // extern "C"
