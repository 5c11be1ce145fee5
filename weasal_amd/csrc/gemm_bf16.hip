// weasal_amd/csrc/gemm_bf16.hip -- the dense parts of the bf16-feature path (BASELINE config 5) on the bf16 MFMA.
//
// Same products as gemm.hip (unary 1x1 MLPs, models/blocks.py:490-501; the kernel contraction
// wf [N,15Ci] x weights [15Ci,Co], blocks.py:370-374; their dX), with the tall operand stored as bf16 rows in HBM,
// the small matrix as bf16, fp32 accumulation in v_mfma_f32_32x32x16_bf16 and an fp32 epilogue (bias, residual,
// LeakyReLU) that rounds once to bf16 (or keeps fp32: logits, the offsets of deformable KPConv).
//
//   gemm_xbt_bf16 :  Y[M,N] = act( X[M,K] * Bt[N,K]^T + bias + residual )
//
// Both operands are K-contiguous, which is the MFMA's own fragment layout: lane (j, h) of a 32x32x16 step holds 8
// consecutive k of one row, so an X fragment is ONE 16-byte global load per lane and never goes through LDS.  The
// MFMA is fed "rows on the lanes" (A operand = Bt, B operand = X) as in gemm_xb2: a lane ends with 4 consecutive
// output columns of its own row per register quad -> 8-byte bf16 (16-byte f32) stores.  Only the 32-deep chunk of
// Bt is staged through LDS (shared by the 4 waves, double buffered, one barrier per chunk, rows padded to 80 bytes
// so that the 16-byte fragment reads are bank-conflict free).  With bf16 MFMA at 16x the f32 rate these products are
// bound by streaming X once: 2*M*K bytes.
//
// dW = X^T dY stays on the exact-f32 MFMA (gemm.hip: gemm_xty2 with 2-byte operand loads; bf16 -> f32 is exact, the
// products and the running sums are fp32, the result is the fp32 master gradient).
#include "ws_common.h"
#include "ws_bf16.h"

extern "C" int ws_gemm_staged;      // gemm.hip: 1 = epilogues turned through LDS

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool ws_bf16_staged_flag(int s) { return s != 0; }

template <int NT, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_xbt_bf16_kernel(
    const bf16_t* __restrict__ x, int64_t m, int k, int64_t ldx, const bf16_t* __restrict__ bt, int n, int64_t ldbt,
    void* __restrict__ yv, int64_t ldy, const float* __restrict__ bias, const bf16_t* __restrict__ residual, int64_t ldr,
    int act, float slope, int vecout, int staged)
{
    constexpr int BN = 32 * NT;
    constexpr int KC = 32;                         // k per chunk (two MFMA steps of 16)
    constexpr int ROWB = 80;                       // bytes per staged Bt row: 64 + 16 of padding
    constexpr int BP = (BN * 4 + 255) / 256;       // 16-byte pieces of the Bt chunk per thread
    constexpr int STAGE_LD = 36, STAGE_BYTES = 4 * 32 * STAGE_LD * 4;        // epilogue: one [32][36] float tile per wave
    constexpr int BS_BYTES = 2 * BN * ROWB > STAGE_BYTES ? 2 * BN * ROWB : STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char Bs_raw[BS_BYTES];
    unsigned char (*Bs)[BN * ROWB] = reinterpret_cast<unsigned char (*)[BN * ROWB]>(Bs_raw);
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    const int j = lane & 31, h = lane >> 5;
    const int64_t brow0 = (int64_t)blockIdx.x * 128;
    const int64_t brows = m - brow0 < 128 ? m - brow0 : 128;
    const int n0 = blockIdx.y * BN;

    // wave-uniform buffer descriptors; rows of X past the end of this workgroup's range read as 0
    const auto bsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(bt), 0, (int)(((int64_t)(n - 1) * ldbt + k) * 2), 0x00020000);
    const auto xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x + brow0 * ldx), 0,
                                                        (int)(((brows - 1) * ldx + k) * 2), 0x00020000);
    int boff[BP], blds[BP];
    bool bok[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int g = t + 256 * i;
        const int c = g >> 2, p = g & 3;
        bok[i] = c < BN;
        int col = n0 + c;
        col = col < n ? col : n - 1;               // clamped: valid memory, the column is never stored
        boff[i] = (int)((col * ldbt + 8 * p) * 2);
        blds[i] = c * ROWB + 16 * p;
    }
    u32x4 bv[BP];
    auto load_b = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < BP; ++i) bv[i] = __builtin_amdgcn_raw_buffer_load_b128(bsrd, boff[i], chunk * (KC * 2), 0);
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BP; ++i)
            if (bok[i]) *reinterpret_cast<u32x4*>(&Bs[buf][blds[i]]) = bv[i];
    };
    const int xoff = (int)(((wave * 32 + j) * ldx + 8 * h) * 2);
    u32x4 xn[2], xc[2];
    auto load_x = [&](int chunk) {
#pragma unroll
        for (int s = 0; s < 2; ++s) xn[s] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, xoff + 32 * s, chunk * (KC * 2), 0);
    };

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int nch = k / KC;
    load_b(0);
    load_x(0);
    store_b(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        xc[0] = xn[0]; xc[1] = xn[1];
        const int nx = c + 1 < nch ? c + 1 : c;
        load_x(nx);
        load_b(nx);
        const unsigned char* bb = &Bs[c & 1][j * ROWB + 16 * h];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(&xc[s]);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const bf16x8_t wfrag = *reinterpret_cast<const bf16x8_t*>(bb + (32 * i) * ROWB + 32 * s);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag, xf, acc[i], 0, 0, 0);
            }
        }
        store_b((c + 1) & 1);                      // (a harmless repeat after the last chunk)
        __syncthreads();
    }

    if (vecout && (n & 3) == 0 && ws_bf16_staged_flag(staged)) {
        // the wave's 32 x 32 tile turned through LDS (see gemm.hip, xb_rows_epilogue_staged): lane (r = lane / 8, c = lane % 8)
        // stores columns 4 c .. 4 c + 3 of rows r + 8 p -- 8 row segments of 64 (bf16) / 128 (f32) contiguous bytes per
        // instruction instead of 8 / 16 bytes to each of 32 rows.  Same arithmetic in the same order.
        float* stage = reinterpret_cast<float*>(Bs_raw) + wave * (32 * STAGE_LD);
        const int r8 = lane >> 3, c4 = lane & 7;
        const int64_t row0 = brow0 + wave * 32;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&stage[j * STAGE_LD + 8 * g + 4 * h]) =
                    make_float4(acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int col = n0 + 32 * i + 4 * c4;
            if (col < n) {
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias) b4 = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int rl = r8 + 8 * p;
                    const int64_t row = row0 + rl;
                    if (row >= m) continue;
                    const float4 a = *reinterpret_cast<const float4*>(&stage[rl * STAGE_LD + 4 * c4]);
                    float v[4] = {a.x + b4.x, a.y + b4.y, a.z + b4.z, a.w + b4.w};
                    if (residual) { const float4 r4 = ld4(residual + row * ldr + col); v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
                    if (act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : v[e] * slope;
                    }
                    if (OUT_F32) st4(reinterpret_cast<float*>(yv) + row * ldy + col, make_float4(v[0], v[1], v[2], v[3]));
                    else st4(reinterpret_cast<bf16_t*>(yv) + row * ldy + col, make_float4(v[0], v[1], v[2], v[3]));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    // epilogue: lane = row, register quad g of tile i = columns n0 + 32 i + 8 g + 4 h .. + 3
    const int64_t row = brow0 + wave * 32 + j;
    if (row >= m) return;
    float* yf = reinterpret_cast<float*>(yv) + row * ldy;
    bf16_t* yb = reinterpret_cast<bf16_t*>(yv) + row * ldy;
    const bf16_t* rrow = residual ? residual + row * ldr : nullptr;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = n0 + 32 * i + 8 * g + 4 * h;
            if (col >= n) continue;
            float v[4] = {acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
            if (vecout && col + 3 < n) {
                if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + col); v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
                if (rrow) { const float4 r4 = ld4(rrow + col); v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
                if (act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : v[e] * slope;
                }
                if (OUT_F32) st4(yf + col, make_float4(v[0], v[1], v[2], v[3]));
                else st4(yb + col, make_float4(v[0], v[1], v[2], v[3]));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (col + e >= n) continue;
                    float w = v[e];
                    if (bias) w += bias[col + e];
                    if (rrow) w += ld1(rrow + col + e);
                    if (act) w = w > 0.0f ? w : w * slope;
                    if (OUT_F32) yf[col + e] = w;
                    else st1(yb + col + e, w);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dz = LeakyReLU'(y) * dy (bf16 rows) and the fp32 column sums of dz (the bias gradient), one pass: the bf16 form of
// gemm.hip's act_bwd_colsum_kernel<4>.  dy may be f32 (the logits' gradient) or bf16.
// ---------------------------------------------------------------------------------------------------------------
template <typename TG>
__global__ __launch_bounds__(256) void act_bwd_colsum_bf16_kernel(const TG* __restrict__ dy, const bf16_t* __restrict__ yact,
                                                                   int64_t m, int n, int64_t lddy, int64_t ldy, float slope,
                                                                   bf16_t* __restrict__ dz, int64_t lddz,
                                                                   float* __restrict__ partial, int64_t chunk)
{
    __shared__ float red[256 * 4];
    const int t = threadIdx.x;
    const int ncg = n / 4;
    const int per = ncg < 256 ? ncg : 256;
    const int R = 256 / per;
    const int rl = t / per, cgl = t % per;
    const int64_t mbeg = (int64_t)blockIdx.x * chunk;
    const int64_t mend = mbeg + chunk < m ? mbeg + chunk : m;
    for (int cg0 = 0; cg0 < ncg; cg0 += per) {
        const int cg = cg0 + cgl;
        const int col = cg * 4;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        if (rl < R && cg < ncg) {
            for (int64_t r = mbeg + rl; r < mend; r += R) {
                float4 g = ld4(dy + r * lddy + col);
                if (yact) {
                    const float4 a = ld4(yact + r * ldy + col);
                    g.x = a.x > 0.0f ? g.x : g.x * slope; g.y = a.y > 0.0f ? g.y : g.y * slope;
                    g.z = a.z > 0.0f ? g.z : g.z * slope; g.w = a.w > 0.0f ? g.w : g.w * slope;
                }
                if (dz) {
                    st4(dz + r * lddz + col, g);
                    g = ld4(dz + r * lddz + col);          // the sums are those of the values the GEMMs will read
                }
                s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
            }
        }
        if (partial) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[t * 4 + e] = s[e];
            __syncthreads();
            if (rl == 0 && cg < ncg) {
                for (int q = 1; q < R; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] += red[(q * per + cgl) * 4 + e];
#pragma unroll
                for (int e = 0; e < 4; ++e) partial[(int64_t)blockIdx.x * n + col + e] = s[e];
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void reduce_partials_bf_kernel(const float* __restrict__ partial, int64_t elems, int chunks,
                                                                  float* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    float s = 0.0f;
    for (int c = 0; c < chunks; ++c) s += partial[(int64_t)c * elems + e];      // fixed order
    out[e] = s;
}

int64_t colsum_chunk_bf(int64_t m)
{
    int64_t c = ws_ceil_div(m, 768);
    return c < 128 ? 128 : c;
}

bool al16b(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool al8b(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

}  // namespace

extern "C" {

int ws_gemm_xbt_bf16(const uint16_t* x, int64_t m, int32_t k, int64_t ldx, const uint16_t* bt, int32_t n, int64_t ldbt,
                     const float* bias, const uint16_t* residual, int64_t ldr, int32_t act, float slope,
                     void* y, int64_t ldy, int32_t out_f32, void* stream)
{
    WS_REQUIRE(m >= 0 && k >= 32 && n >= 1 && ldx >= k && ldy >= n && ldbt >= k, "bad sizes m=%lld k=%d n=%d", (long long)m, k, n);
    WS_REQUIRE(k % 32 == 0, "k=%d: the bf16 product needs k %% 32 == 0 (callers pad or take the fp32 kernel)", k);
    WS_REQUIRE(ldx % 8 == 0 && ldbt % 8 == 0, "rows of x and bt must be 16-byte multiples (ldx=%lld ldbt=%lld)", (long long)ldx, (long long)ldbt);
    WS_REQUIRE(!residual || ldr >= n, "residual leading dimension too small");
    WS_REQUIRE(act == 0 || act == 1, "unknown activation %d", act);
    if (m == 0) return WS_OK;
    WS_REQUIRE(x && bt && y, "NULL argument");
    WS_REQUIRE(al16b(x) && al16b(bt), "x and bt must be 16-byte aligned");
    WS_REQUIRE(128 * ldx * 2 + (int64_t)k * 2 < (1ll << 31) && ((int64_t)n * ldbt) * 2 < (1ll << 31), "operand exceeds the 32-bit buffer offsets");
    hipStream_t st = (hipStream_t)stream;
    const int64_t gx = ws_ceil_div(m, 128);
    WS_REQUIRE(gx < (1ll << 31), "m too large");
    const int vecout = (n % 4 == 0) && (ldy % 4 == 0) && (out_f32 ? al16b(y) : al8b(y)) && (!bias || al16b(bias)) &&
                       (!residual || (al8b(residual) && ldr % 4 == 0));
    const bf16_t* xb = reinterpret_cast<const bf16_t*>(x);
    const bf16_t* bb = reinterpret_cast<const bf16_t*>(bt);
    const bf16_t* rb = reinterpret_cast<const bf16_t*>(residual);
#define WS_XBT(NTV)                                                                                                        \
    do {                                                                                                                   \
        const dim3 grid((unsigned)gx, (unsigned)ws_ceil_div(n, 32 * NTV));                                                 \
        if (out_f32) gemm_xbt_bf16_kernel<NTV, true><<<grid, 256, 0, st>>>(xb, m, k, ldx, bb, n, ldbt, y, ldy, bias, rb, ldr, act, slope, vecout, ws_gemm_staged); \
        else gemm_xbt_bf16_kernel<NTV, false><<<grid, 256, 0, st>>>(xb, m, k, ldx, bb, n, ldbt, y, ldy, bias, rb, ldr, act, slope, vecout, ws_gemm_staged);       \
    } while (0)
    if (n <= 32) WS_XBT(1);
    else if (n <= 64) WS_XBT(2);
    else WS_XBT(4);
#undef WS_XBT
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int64_t ws_act_bwd_colsum_bf16_scratch_bytes(int64_t m, int32_t n)
{
    return ws_ceil_div(m > 0 ? m : 1, colsum_chunk_bf(m)) * (int64_t)n * (int64_t)sizeof(float);
}

int ws_act_bwd_colsum_bf16(const void* dy, int32_t dy_f32, int64_t m, int32_t n, int64_t lddy, const uint16_t* y, int64_t ldy,
                           float slope, uint16_t* dz, int64_t lddz, float* colsum, void* scratch, void* stream)
{
    WS_REQUIRE(m >= 0 && n >= 4 && n % 4 == 0 && lddy >= n && lddy % 4 == 0, "bad sizes m=%lld n=%d (n %% 4 == 0 required)", (long long)m, n);
    WS_REQUIRE(!y || (ldy >= n && ldy % 4 == 0), "bad y leading dimension");
    WS_REQUIRE(!dz || (lddz >= n && lddz % 4 == 0), "bad dz leading dimension");
    WS_REQUIRE(!y || dz, "activation backward needs dz");
    WS_REQUIRE(!colsum || scratch, "column sums need scratch");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        if (colsum) WS_HIP(hipMemsetAsync(colsum, 0, sizeof(float) * (size_t)n, st));
        return WS_OK;
    }
    WS_REQUIRE(dy && (dz || colsum), "NULL argument");
    WS_REQUIRE(al8b(dy) && (!y || al8b(y)) && (!dz || al8b(dz)), "rows must be 8-byte aligned");
    const int64_t chunk = colsum_chunk_bf(m);
    const int chunks = (int)ws_ceil_div(m, chunk);
    float* partial = colsum ? (chunks == 1 ? colsum : (float*)scratch) : nullptr;
    const bf16_t* yb = reinterpret_cast<const bf16_t*>(y);
    bf16_t* dzb = reinterpret_cast<bf16_t*>(dz);
    if (dy_f32) {
        WS_REQUIRE(al16b(dy), "f32 dy must be 16-byte aligned");
        act_bwd_colsum_bf16_kernel<float><<<chunks, 256, 0, st>>>((const float*)dy, yb, m, n, lddy, ldy, slope, dzb, lddz, partial, chunk);
    } else {
        act_bwd_colsum_bf16_kernel<bf16_t><<<chunks, 256, 0, st>>>((const bf16_t*)dy, yb, m, n, lddy, ldy, slope, dzb, lddz, partial, chunk);
    }
    WS_LAUNCH_CHECK();
    if (colsum && chunks > 1) {
        reduce_partials_bf_kernel<<<(unsigned)ws_ceil_div(n, 256), 256, 0, st>>>(partial, n, chunks, colsum);
        WS_LAUNCH_CHECK();
    }
    return WS_OK;
}

}  // extern "C"
