// weasal_amd/csrc/loss.hip -- the segmentation loss of the training step as two passes over the logits.
//
//   ws_softmax_ce_fwd / _bwd   models/architectures.py:362-373 (KPFCNN.loss): labels -> class positions (-1 = ignored), then
//                              torch.nn.CrossEntropyLoss(weight, ignore_index = -1) over [1, C, N]:
//                                  loss = sum_i w[t_i] (logsumexp(x_i) - x_i[t_i]) / sum_i w[t_i]      (valid rows only)
//                              d loss / d x_i[j] = g w[t_i] / sum w * (softmax(x_i)[j] - [j == t_i]),  0 for ignored rows
// The stock path is a label mapping (4 element-wise kernels), log-softmax and nll_loss2d forward / backward (0.45 ms per
// DALES step for 400 000 x 9 logits); here a thread owns one row (C <= 64 floats, adjacent rows adjacent in memory), the
// row sums are added per workgroup and then in a fixed order by one workgroup: the result is reproducible bit for bit.
#include "ws_common.h"

namespace {

constexpr int CE_MAX_C = 64;
constexpr int CE_BLOCKS = 1024;

// class position of a raw label: lut [lut_n] maps label values to positions (its last entry is the spare -1 that
// out-of-table labels take, weasal_amd/architectures.py KPFCNN._targets); lut == NULL: labels are positions already
__device__ __forceinline__ int ce_target(const int64_t* __restrict__ labels, int64_t i, const int64_t* __restrict__ lut, int lut_n,
                                         int c)
{
    int64_t t = labels[i];
    if (lut) t = lut[(t >= 0 && t < lut_n - 1) ? t : lut_n - 1];
    return (t >= 0 && t < c) ? (int)t : -1;
}

__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const float* __restrict__ logits, int64_t n, int c, int64_t ldl,
                                                             const int64_t* __restrict__ labels, const int64_t* __restrict__ lut,
                                                             int lut_n, const float* __restrict__ class_w,
                                                             float* __restrict__ part /*[2][gridDim.x]*/)
{
    __shared__ float red[2][4];
    float ls = 0.0f, ws = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int t = ce_target(labels, i, lut, lut_n, c);
        if (t < 0) continue;
        const float* x = logits + i * ldl;
        float m = x[0];
        for (int j = 1; j < c; ++j) m = fmaxf(m, x[j]);
        float s = 0.0f;
        for (int j = 0; j < c; ++j) s += expf(x[j] - m);
        const float w = class_w ? class_w[t] : 1.0f;
        ls += w * ((m + logf(s)) - x[t]);
        ws += w;
    }
    ls = ws_wave_sum(ls);
    ws = ws_wave_sum(ws);
    if (ws_lane() == 0) { red[0][threadIdx.x >> 6] = ls; red[1][threadIdx.x >> 6] = ws; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        part[gridDim.x + blockIdx.x] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
}

__global__ __launch_bounds__(256) void softmax_ce_final_kernel(const float* __restrict__ part, int blocks, float* __restrict__ loss,
                                                               float* __restrict__ wsum)
{
    __shared__ double red[2][256];
    double ls = 0.0, ws = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 256) { ls += (double)part[b]; ws += (double)part[blocks + b]; }
    red[0][threadIdx.x] = ls;
    red[1][threadIdx.x] = ws;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *wsum = (float)red[1][0];
        *loss = (float)(red[0][0] / red[1][0]);          // no valid row: 0 / 0 = nan, like the stock loss
    }
}

__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const float* __restrict__ logits, int64_t n, int c, int64_t ldl,
                                                             const int64_t* __restrict__ labels, const int64_t* __restrict__ lut,
                                                             int lut_n, const float* __restrict__ class_w,
                                                             const float* __restrict__ grad_loss, const float* __restrict__ wsum,
                                                             float* __restrict__ dlogits, int64_t ldd)
{
    const float scale = grad_loss[0] / wsum[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int t = ce_target(labels, i, lut, lut_n, c);
        float* d = dlogits + i * ldd;
        if (t < 0) {
            for (int j = 0; j < c; ++j) d[j] = 0.0f;
            continue;
        }
        const float* x = logits + i * ldl;
        float m = x[0];
        for (int j = 1; j < c; ++j) m = fmaxf(m, x[j]);
        float s = 0.0f;
        for (int j = 0; j < c; ++j) s += expf(x[j] - m);
        const float g = scale * (class_w ? class_w[t] : 1.0f);
        const float inv = 1.0f / s;
        for (int j = 0; j < c; ++j) d[j] = g * (expf(x[j] - m) * inv - (j == t ? 1.0f : 0.0f));
    }
}

}  // namespace

extern "C" {

int64_t ws_softmax_ce_scratch_bytes(int64_t n)
{
    (void)n;
    return 2 * CE_BLOCKS * (int64_t)sizeof(float);
}

int ws_softmax_ce_fwd(const float* logits, int64_t n, int32_t c, int64_t ldl, const int64_t* labels, const int64_t* lut,
                      int32_t lut_n, const float* class_w, float* loss, float* wsum, void* scratch, void* stream)
{
    WS_REQUIRE(n >= 0 && c >= 1 && c <= CE_MAX_C && ldl >= c, "bad sizes n=%lld c=%d (c <= %d)", (long long)n, c, CE_MAX_C);
    WS_REQUIRE(loss && wsum && scratch && (n == 0 || (logits && labels)), "NULL argument");
    WS_REQUIRE(!lut || lut_n >= 2, "the label table needs at least one value and the spare entry");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = ws_grid(n, 256, CE_BLOCKS);
    softmax_ce_fwd_kernel<<<blocks, 256, 0, st>>>(logits, n, c, ldl, labels, lut, lut_n, class_w, (float*)scratch);
    WS_LAUNCH_CHECK();
    softmax_ce_final_kernel<<<1, 256, 0, st>>>((const float*)scratch, blocks, loss, wsum);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

int ws_softmax_ce_bwd(const float* logits, int64_t n, int32_t c, int64_t ldl, const int64_t* labels, const int64_t* lut,
                      int32_t lut_n, const float* class_w, const float* grad_loss, const float* wsum, float* dlogits, int64_t ldd,
                      void* stream)
{
    WS_REQUIRE(n >= 0 && c >= 1 && c <= CE_MAX_C && ldl >= c && ldd >= c, "bad sizes n=%lld c=%d (c <= %d)", (long long)n, c, CE_MAX_C);
    if (n == 0) return WS_OK;
    WS_REQUIRE(logits && labels && grad_loss && wsum && dlogits, "NULL argument");
    WS_REQUIRE(!lut || lut_n >= 2, "the label table needs at least one value and the spare entry");
    softmax_ce_bwd_kernel<<<ws_grid(n, 256), 256, 0, (hipStream_t)stream>>>(logits, n, c, ldl, labels, lut, lut_n, class_w, grad_loss,
                                                                          wsum, dlogits, ldd);
    WS_LAUNCH_CHECK();
    return WS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// nn.Dropout on the decoder output in front of the head (models/architectures.py:345-349: x = droplayer(x); x = head_mlp(x)).
// out = in * keep / (1 - p) with keep ~ Bernoulli(1 - p) from a counter-based generator: keep(i) is a pure function of
// (seed, i), so the backward recomputes it (the same call on the incoming gradient) and no mask is stored.  The framework's
// pair (fused_dropout + masked_scale) moves the mask through memory and spends ~30 instructions per element on Philox:
// 74 + 67 us per DALES step; this one reads and writes each element once.
// ---------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ in, int64_t n, unsigned threshold, float scale,
                                                             unsigned long long seed, float* __restrict__ out)
{
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(in)[i];
        v.x = ws_drop_hash(seed, 4 * i + 0) >= threshold ? v.x * scale : 0.0f;
        v.y = ws_drop_hash(seed, 4 * i + 1) >= threshold ? v.y * scale : 0.0f;
        v.z = ws_drop_hash(seed, 4 * i + 2) >= threshold ? v.z * scale : 0.0f;
        v.w = ws_drop_hash(seed, 4 * i + 3) >= threshold ? v.w * scale : 0.0f;
        reinterpret_cast<float4*>(out)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        out[i] = ws_drop_hash(seed, i) >= threshold ? in[i] * scale : 0.0f;
    }
}

}  // namespace

extern "C" int ws_dropout_apply(const float* in, int64_t n, float p, uint64_t seed, float* out, void* stream)
{
    WS_REQUIRE(n >= 0 && p >= 0.0f && p < 1.0f, "bad sizes / drop probability %g", (double)p);
    if (n == 0) return WS_OK;
    WS_REQUIRE(in && out, "NULL argument");
    WS_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0, "16-byte aligned tensors expected");
    const WsDrop d = ws_drop_args(p, (unsigned long long)seed);                     // drop when hash < p * 2^32
    dropout_apply_kernel<<<ws_grid(n / 4 + 1, 256), 256, 0, (hipStream_t)stream>>>(in, n, d.threshold, d.scale, d.seed, out);
    WS_LAUNCH_CHECK();
    return WS_OK;
}
