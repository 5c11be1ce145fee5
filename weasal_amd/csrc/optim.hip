// weasal_amd/csrc/optim.hip -- the parameter update of the training step in one pass over the parameters.
//
//   ws_sgd_step   utils/trainer_PseudoLabel.py:216-218 (torch.nn.utils.clip_grad_value_ then optimizer.step()) with the
//                 optimizer of :72-82 (torch.optim.SGD, momentum, weight decay, no dampening / Nesterov):
//                     g   = clamp(grad, -clip, clip)                  (clip > 0)
//                     g  += weight_decay * p                          (weight_decay != 0)
//                     buf = first ? g : momentum * buf + g            (momentum != 0)
//                     p  -= lr * buf
// The stock path is one clamp, and three to four `foreach` passes over ~230 tensors (0.36 ms and ~7 launches of
// multi_tensor_apply per DALES step); here every parameter element is read and written once: p, grad, buf in, p, buf out.
// A launch takes up to WS_SGD_MAX tensors as kernel arguments (pointer table + block prefix): no device-side table,
// no host-to-device copy, ~230 tensors = 3 launches.
#include "ws_common.h"

namespace {

constexpr int SGD_MAX = 96;            // tensors per launch: 96 x (3 pointers + size + block prefix) = 3.5 KB of kernel arguments
constexpr int SGD_CHUNK = 4096;        // elements per workgroup

struct SgdArgs {
    float* p[SGD_MAX];
    const float* g[SGD_MAX];
    float* buf[SGD_MAX];
    int n[SGD_MAX];
    int first_block[SGD_MAX + 1];
    int count;
};

__device__ __forceinline__ float sgd_one(float p, float g, float& b, float lr, float momentum, float wd, float clip, int first)
{
    if (clip > 0.0f) g = fminf(fmaxf(g, -clip), clip);
    if (wd != 0.0f) g = __fadd_rn(g, __fmul_rn(wd, p));
    if (momentum != 0.0f) {
        b = first ? g : __fadd_rn(__fmul_rn(b, momentum), g);
        g = b;
    }
    return __fadd_rn(p, __fmul_rn(-lr, g));
}

__global__ __launch_bounds__(256) void sgd_step_kernel(const SgdArgs a, float lr, float momentum, float wd, float clip, int first)
{
    // tensor of this workgroup: the prefix is ascending, <= 96 entries -> binary search on scalars
    int lo = 0, hi = a.count;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    float* __restrict__ p = a.p[lo];
    const float* __restrict__ g = a.g[lo];
    float* __restrict__ buf = a.buf[lo];
    const int n = a.n[lo];
    const int base = ((int)blockIdx.x - a.first_block[lo]) * SGD_CHUNK;
    const int end = base + SGD_CHUNK < n ? base + SGD_CHUNK : n;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(buf)) & 15u) == 0;
    int i = base + 4 * threadIdx.x;
    if (vec) {
        for (; i + 3 < end; i += 1024) {
            float4 pv = *reinterpret_cast<const float4*>(p + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            float4 bv = (buf && !first) ? *reinterpret_cast<const float4*>(buf + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            pv.x = sgd_one(pv.x, gv.x, bv.x, lr, momentum, wd, clip, first);
            pv.y = sgd_one(pv.y, gv.y, bv.y, lr, momentum, wd, clip, first);
            pv.z = sgd_one(pv.z, gv.z, bv.z, lr, momentum, wd, clip, first);
            pv.w = sgd_one(pv.w, gv.w, bv.w, lr, momentum, wd, clip, first);
            *reinterpret_cast<float4*>(p + i) = pv;
            if (buf) *reinterpret_cast<float4*>(buf + i) = bv;
        }
        // the (at most 3) elements past the last whole quad of the tensor: the thread whose quad straddles `end`
        if (i < end) {
            for (int e = i; e < end; ++e) {
                float b = (buf && !first) ? buf[e] : 0.0f;
                p[e] = sgd_one(p[e], g[e], b, lr, momentum, wd, clip, first);
                if (buf) buf[e] = b;
            }
        }
    } else {
        for (int e = base + threadIdx.x; e < end; e += 256) {
            float b = (buf && !first) ? buf[e] : 0.0f;
            p[e] = sgd_one(p[e], g[e], b, lr, momentum, wd, clip, first);
            if (buf) buf[e] = b;
        }
    }
}

}  // namespace

extern "C" {

int ws_sgd_step(float* const* h_params, const float* const* h_grads, float* const* h_bufs, const int64_t* h_sizes, int32_t count,
                float lr, float momentum, float weight_decay, float clip_value, int32_t first, void* stream)
{
    WS_REQUIRE(count >= 0, "bad tensor count %d", count);
    if (count == 0) return WS_OK;
    WS_REQUIRE(h_params && h_grads && h_sizes, "NULL argument");
    WS_REQUIRE(momentum == 0.0f || h_bufs, "momentum needs the momentum buffers");
    hipStream_t st = (hipStream_t)stream;
    int t = 0;
    while (t < count) {
        SgdArgs a;
        a.count = 0;
        int blocks = 0;
        while (t < count && a.count < SGD_MAX) {
            const int64_t n = h_sizes[t];
            WS_REQUIRE(n >= 0 && n < (1ll << 31) - SGD_CHUNK, "tensor %d too large (%lld elements)", t, (long long)n);
            if (n > 0) {
                WS_REQUIRE(h_params[t] && h_grads[t] && (momentum == 0.0f || h_bufs[t]), "NULL tensor %d", t);
                const int j = a.count++;
                a.p[j] = h_params[t];
                a.g[j] = h_grads[t];
                a.buf[j] = momentum != 0.0f ? h_bufs[t] : nullptr;
                a.n[j] = (int)n;
                a.first_block[j] = blocks;
                blocks += (int)ws_ceil_div(n, SGD_CHUNK);
            }
            ++t;
        }
        if (a.count == 0) break;
        a.first_block[a.count] = blocks;
        sgd_step_kernel<<<blocks, 256, 0, st>>>(a, lr, momentum, weight_decay, clip_value, first);
        WS_LAUNCH_CHECK();
    }
    return WS_OK;
}

}  // extern "C"
