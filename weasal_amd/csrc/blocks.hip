// weasal_amd/csrc/blocks.hip -- whole network blocks behind one C call (host code: launch sequences over the kernels
// of kpconv.hip / gemm.hip / pools.hip on one stream).
//
// Reference units: models/blocks.py:510-564 (SimpleBlock), :624-709 (ResnetBottleneckBlock) and their autograd;
// models/architectures.py:339-343 + blocks.py:473-507 (nearest_upsample -> concat -> unary).  The reference runs each
// of them as ~10 torch ops plus autograd nodes driven from Python; here one call launches the 6-7 (forward) or 10-14
// (backward) kernels back to back, every gradient accumulation is the residual operand of a GEMM epilogue and the
// scratch comes from one caller-owned arena (no allocation, no host synchronisation).
#include <vector>

#include "ws_common.h"

// diagnostics switch (WEASAL_BLOCK_GATES=0): activation backward as separate passes instead of epilogue / store gates
extern "C" int ws_block_gates = 1;
extern "C" int ws_block_gather_residual = 1;   // decoder step: 1 = the upsampled rows are gathered by the last epilogue, 0 = written out first (A/B)
// Diagnostics (WEASAL_BLOCK_SIDE_ROWS=<rows>): blocks with fewer query rows than this run their weight-gradient products
// (dW = X^T dZ: leaves of the backward, nothing on the chain to dX waits for them) on a side stream next to the dX chain.
// Default 0 = off: measured on the DALES step the deep levels' products are bound by the matrix cores and by their
// fixed launch / prologue latency, not by idle CUs -- 12.21 ms per step with the side stream, 12.11 without.
extern "C" int64_t ws_block_side_rows = 0;
// A/B switch (WEASAL_FUSED_INFER=0): forward-only blocks run gather + contraction as two launches like the training path
extern "C" int ws_block_fused_infer = 1;
// A/B switch (WEASAL_POOL_ORDER=0): the strided blocks' max-pool walks its rows by index instead of in cell order
extern "C" int ws_block_pool_order = 1;

namespace {

// One side stream and a few events per device, created on first use and kept for the life of the process.
struct Side {
    hipStream_t st = nullptr;
    hipEvent_t fork[3] = {nullptr, nullptr, nullptr};
    hipEvent_t join = nullptr;
};
int side_for_current_device(Side** out)
{
    static Side sides[16];
    int dev = 0;
    WS_HIP(hipGetDevice(&dev));
    WS_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
    Side& sd = sides[dev];
    if (!sd.st) {
        hipStream_t st;
        WS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (auto& e : sd.fork) WS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        WS_HIP(hipEventCreateWithFlags(&sd.join, hipEventDisableTiming));
        sd.st = st;
    }
    *out = &sd;
    return WS_OK;
}

struct Arena {
    char* base;
    int64_t cap, off;
    Arena(void* b, int64_t c) : base((char*)b), cap(c), off(0) {}
    template <typename T> T* take(int64_t count)
    {
        off = (off + 255) & ~(int64_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * (int64_t)sizeof(T);
        return p;
    }
    bool fits() const { return off <= cap; }
};

inline int64_t max3(int64_t a, int64_t b, int64_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

__global__ __launch_bounds__(256) void transpose_small_kernel(const float* __restrict__ w, int rows, int cols, int64_t ld,
                                                               float* __restrict__ out)
{
    // out[c][r] = w[r*ld + c]   (weights: a few hundred KB at most)
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)rows * cols) return;
    const int c = (int)(e / rows), r = (int)(e % rows);
    out[e] = w[(int64_t)r * ld + c];
}

__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t n4)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        float4 x = reinterpret_cast<float4*>(a)[e];
        const float4 y = reinterpret_cast<const float4*>(b)[e];
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        reinterpret_cast<float4*>(a)[e] = x;
    }
}

// y = act(x [m,k] @ w^T + bias + residual), w = nn.Linear weight [n,k] with leading dimension ldw, read in place as
// its transpose when the strided MFMA path takes the shape, through an explicit transpose in `tr` otherwise
struct Lin {
    const float* w; int n, k; int64_t ldw;
    bool in_place() const { return k % 32 == 0 && n % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)w & 15u) == 0; }
    int64_t tr_floats() const { return in_place() ? 0 : (int64_t)k * n; }
};

int linear_fwd(const Lin& l, const float* x, int64_t m, int64_t ldx, const float* bias, const float* residual, int64_t ldr,
               int act, float slope, float* y, float* tr, void* tmp, int64_t tmp_bytes, hipStream_t st)
{
    if (l.in_place())
        return ws_gemm_xb_epilogue_strided(x, m, l.k, ldx, l.w, 1, l.ldw, l.n, bias, residual, ldr, act, slope, y, l.n, tmp, tmp_bytes, st);
    transpose_small_kernel<<<(unsigned)ws_ceil_div((int64_t)l.k * l.n, 256), 256, 0, st>>>(l.w, l.n, l.k, l.ldw, tr);
    return ws_gemm_xb_epilogue_strided(x, m, l.k, ldx, tr, l.n, 1, l.n, bias, residual, ldr, act, slope, y, l.n, tmp, tmp_bytes, st);
}

#define WS_TRY(call)            \
    do {                        \
        int rc__ = (call);      \
        if (rc__) return rc__;  \
    } while (0)

// ---- launch timer (bench.py: HIP events around the K3 launch, on the launch stream) --------------------------
struct TimerRec { hipEvent_t a, b, c; int64_t nq; int32_t h, ci; };      // a .. b: the K3 launch, a .. c: K3 + the contraction
std::vector<TimerRec>& timer_recs() { static std::vector<TimerRec> v; return v; }

// the strided block's arg-max record lives in the block's own arena slot (only ws_kpblock_bwd reads it): bytes when the
// neighbour columns fit one (4x fewer bytes for the backward's per-pair reads), the forward and the backward decide alike
bool arg_bytes(const ws_kpblock* d)
{
    return d->h <= 255 && d->in_dim % 4 == 0 && ((uintptr_t)d->feat & 15u) == 0 && ((uintptr_t)d->pooled & 15u) == 0 &&
           ((uintptr_t)d->arg & 3u) == 0;
}

int check_kpblock(const ws_kpblock* d)
{
    WS_REQUIRE(d, "NULL descriptor");
    WS_REQUIRE(d->nq >= 0 && d->ns >= 0 && d->h >= 1, "bad sizes nq=%lld ns=%lld h=%d", (long long)d->nq, (long long)d->ns, d->h);
    if (d->k != 15) return ws_fail(WS_ERR_UNSUPPORTED, "num_kernel_points=%d: K=15 only", d->k);
    WS_REQUIRE(d->in_dim >= 1 && d->conv_in >= 1 && d->conv_out >= 1 && d->out_dim >= 1, "bad widths");
    WS_REQUIRE(d->q_pts && d->s_pts && d->inds && d->kernel_points && d->feat && d->wk && (d->wf || d->infer) && d->out, "NULL argument");
    WS_REQUIRE(d->w1 ? (d->x1 != nullptr) : (d->conv_in == d->in_dim), "unary1: x1 buffer missing or conv_in != in_dim");
    WS_REQUIRE(d->w2 ? (d->x2 != nullptr) : (d->out_dim == d->conv_out), "unary2: x2 buffer missing or out_dim != conv_out");
    WS_REQUIRE(!d->w2 || d->ws || d->in_dim == d->out_dim, "shortcut needs a projection (ws) when in_dim != out_dim");
    WS_REQUIRE(!d->strided || !d->w2 || (d->pooled && d->arg), "strided block: pooled / arg buffers missing");
    WS_REQUIRE(d->strided || d->nq == d->ns, "non-strided block: nq must equal ns");
    if (d->conv_out % 4 || d->out_dim % 4 || (d->w1 && d->conv_in % 4) || (d->w2 && d->in_dim % 4))
        return ws_fail(WS_ERR_UNSUPPORTED, "block widths must be multiples of 4 (in=%d conv=%d->%d out=%d)", d->in_dim, d->conv_in,
                       d->conv_out, d->out_dim);
    return WS_OK;
}

int64_t kpblock_tmp_fwd(const ws_kpblock* d)
{
    int64_t t = ws_gemm_xb_scratch_bytes(d->nq, d->k * d->conv_in, d->conv_out);
    if (d->w1) t = max3(t, ws_gemm_xb_scratch_bytes(d->ns, d->in_dim, d->conv_in), 0);
    if (d->w2) t = max3(t, ws_gemm_xb_scratch_bytes(d->nq, d->conv_out, d->out_dim), d->ws ? ws_gemm_xb_scratch_bytes(d->nq, d->in_dim, d->out_dim) : 0);
    return t;
}

int kpblock_fwd(const ws_kpblock* d, Arena& ar, hipStream_t st, bool run)
{
    const int64_t nq = d->nq, ns = d->ns;
    const Lin l1{d->w1, d->conv_in, d->in_dim, d->in_dim}, l2{d->w2, d->out_dim, d->conv_out, d->conv_out},
        lsc{d->ws, d->out_dim, d->in_dim, d->in_dim};
    float* tr1 = d->w1 ? ar.take<float>(l1.tr_floats()) : nullptr;
    float* tr2 = d->w2 ? ar.take<float>(l2.tr_floats()) : nullptr;
    float* trs = d->ws ? ar.take<float>(lsc.tr_floats()) : nullptr;
    float* sc = (d->w2 && d->ws) ? ar.take<float>(nq * d->out_dim) : nullptr;
    const int64_t tmp_bytes = kpblock_tmp_fwd(d);
    void* tmp = ar.take<char>(tmp_bytes > 16 ? tmp_bytes : 16);
    if (!run) return WS_OK;
    if (nq == 0) return WS_OK;
    const float* x1 = d->feat;
    if (d->w1) {
        WS_TRY(linear_fwd(l1, d->feat, ns, d->in_dim, d->b1, nullptr, 0, 1, d->slope, d->x1, tr1, tmp, tmp_bytes, st));
        x1 = d->x1;
    }
    TimerRec rec{};
    if (d->timed) {
        WS_HIP(hipEventCreate(&rec.a));
        WS_HIP(hipEventCreate(&rec.b));
        rec.nq = nq; rec.h = d->h; rec.ci = d->conv_in;
        WS_HIP(hipEventRecord(rec.a, st));
    }
    float* x2 = d->w2 ? d->x2 : d->out;
    // forward only and a layer the one-launch kernel covers: the contraction runs inside the gather, no `wf`
    const bool one_launch = d->infer && d->conv_in == 32 && d->conv_out == 32 && !d->rows_sorted && ((uintptr_t)x1 & 15u) == 0 &&
                            ((uintptr_t)d->wk & 15u) == 0 && ((uintptr_t)x2 & 15u) == 0 && ws_block_fused_infer;
    if (one_launch) {
        WS_TRY(ws_kpconv_layer_fwd_fused(d->q_pts, nq, d->s_pts, ns, d->inds, d->h, x1, d->conv_in, d->kernel_points, d->k, d->extent,
                                         d->order_q, d->wk, d->conv_out, d->bk, 1, d->slope, x2, st));
        if (d->timed) WS_HIP(hipEventRecord(rec.b, st));
    } else {
        WS_REQUIRE(d->wf, "wf buffer missing");
        WS_TRY(ws_kpconv_gather_fwd_ex(d->q_pts, nq, d->s_pts, ns, d->inds, d->h, x1, d->conv_in, d->kernel_points, d->k, nullptr,
                                       nullptr, d->extent, WS_INFLUENCE_LINEAR, WS_AGGREGATION_SUM, d->order_q, d->wf, nullptr, 0,
                                       d->rows_sorted, st));
        if (d->timed) WS_HIP(hipEventRecord(rec.b, st));
        WS_TRY(ws_gemm_xb_epilogue_strided(d->wf, nq, d->k * d->conv_in, (int64_t)d->k * d->conv_in, d->wk, d->conv_out, 1, d->conv_out,
                                           d->bk, nullptr, 0, 1, d->slope, x2, d->conv_out, tmp, tmp_bytes, st));
    }
    if (d->timed) {
        WS_HIP(hipEventCreate(&rec.c));
        WS_HIP(hipEventRecord(rec.c, st));
        timer_recs().push_back(rec);
    }
    if (!d->w2) return WS_OK;
    const float* sc_in = d->feat;
    if (d->strided) {
        if (arg_bytes(d))
            WS_TRY(ws_priv_max_pool_fwd_u8(d->feat, ns, d->in_dim, d->inds, nq, d->h, d->pooled, reinterpret_cast<uint8_t*>(d->arg),
                                           ws_block_pool_order ? d->order_q : nullptr, st));
        else
            WS_TRY(ws_max_pool_fwd(d->feat, ns, d->in_dim, d->inds, nq, d->h, d->pooled, d->arg, st));
        sc_in = d->pooled;
    }
    const float* res = sc_in;
    int64_t ldr = d->in_dim;
    if (d->ws) {
        WS_TRY(linear_fwd(lsc, sc_in, nq, d->in_dim, d->bs, nullptr, 0, 0, 0.0f, sc, trs, tmp, tmp_bytes, st));
        res = sc;
        ldr = d->out_dim;
    }
    return linear_fwd(l2, x2, nq, d->conv_out, d->b2, res, ldr, 1, d->slope, d->out, tr2, tmp, tmp_bytes, st);
}

int kpblock_bwd(const ws_kpblock* d, Arena& ar, hipStream_t st, bool run)
{
    const int64_t nq = d->nq, ns = d->ns;
    const int kc = d->k * d->conv_in;
    const bool need_dx1 = d->dfeat != nullptr || d->w1 != nullptr;
    const bool want_sc = d->w2 && d->dfeat;                       // the shortcut's share of dfeat
    // ---- arena plan (identical in the size query and in the run)
    float* dz = d->w2 ? ar.take<float>(nq * d->out_dim) : nullptr;
    float* g2 = ar.take<float>(nq * d->conv_out);                  // dx2 -> dz2 (simple block: dz2 of dout)
    float* dwf = need_dx1 ? ar.take<float>(nq * (int64_t)kc) : nullptr;
    float* dx1 = (need_dx1 && d->w1) ? ar.take<float>(ns * d->conv_in) : nullptr;
    float* dscin = (want_sc && d->ws) ? ar.take<float>(nq * d->in_dim) : nullptr;
    float* dfsc = (want_sc && d->strided) ? ar.take<float>(ns * d->in_dim) : nullptr;
    const Lin lk{d->wk, kc, d->conv_out, d->conv_out};           // dwf = dz2 @ wk^T: wk [kc, conv_out] read as a "Linear" weight
    float* trk = need_dx1 ? ar.take<float>(lk.tr_floats()) : nullptr;
    int64_t tmp_bytes = max3(ws_act_bwd_colsum_scratch_bytes(nq, d->out_dim > d->conv_out ? d->out_dim : d->conv_out),
                             ws_act_bwd_colsum_scratch_bytes(ns, d->conv_in), ws_gemm_xty_scratch_bytes(nq, kc, d->conv_out));
    if (d->w2) tmp_bytes = max3(tmp_bytes, ws_gemm_xty_scratch_bytes(nq, d->out_dim, d->conv_out),
                                d->ws ? ws_gemm_xty_scratch_bytes(nq, d->out_dim, d->in_dim) : 0);
    if (d->w1) tmp_bytes = max3(tmp_bytes, ws_gemm_xty_scratch_bytes(ns, d->conv_in, d->in_dim), 0);
    tmp_bytes = max3(tmp_bytes, ws_gemm_xb_scratch_bytes(nq, d->conv_out, kc), ws_gemm_xb_scratch_bytes(nq, d->out_dim, d->conv_out));
    tmp_bytes = max3(tmp_bytes, ws_gemm_xb_scratch_bytes(nq, d->out_dim, d->in_dim), ws_gemm_xb_scratch_bytes(ns, d->conv_in, d->in_dim));
    void* tmp = ar.take<char>(tmp_bytes > 16 ? tmp_bytes : 16);
    // the weight-gradient products of a short block run on the side stream with a scratch of their own
    const bool use_side = ws_block_side_rows > 0 && nq > 0 && ns > 0 && nq < ws_block_side_rows;
    int64_t side_bytes = ws_gemm_xty_scratch_bytes(nq, kc, d->conv_out);
    if (d->w2) side_bytes = max3(side_bytes, ws_gemm_xty_scratch_bytes(nq, d->out_dim, d->conv_out),
                                 d->ws ? ws_gemm_xty_scratch_bytes(nq, d->out_dim, d->in_dim) : 0);
    if (d->w1) side_bytes = max3(side_bytes, ws_gemm_xty_scratch_bytes(ns, d->conv_in, d->in_dim), 0);
    void* tmp_w = use_side ? (void*)ar.take<char>(side_bytes > 16 ? side_bytes : 16) : tmp;
    if (!run) return WS_OK;
    WS_REQUIRE(d->dout && d->dwk, "NULL gradient buffer");
    WS_REQUIRE(!d->w1 || d->dw1, "dw1 missing");
    WS_REQUIRE(!d->w2 || d->dw2, "dw2 missing");
    WS_REQUIRE(!d->ws || !d->w2 || d->dws, "dws missing");
    if (nq == 0 || ns == 0) {
        // no rows: every gradient is zero
        WS_HIP(hipMemsetAsync(d->dwk, 0, sizeof(float) * (size_t)kc * d->conv_out, st));
        if (d->dbk) WS_HIP(hipMemsetAsync(d->dbk, 0, sizeof(float) * d->conv_out, st));
        if (d->w1) { WS_HIP(hipMemsetAsync(d->dw1, 0, sizeof(float) * (size_t)d->conv_in * d->in_dim, st)); if (d->db1) WS_HIP(hipMemsetAsync(d->db1, 0, sizeof(float) * d->conv_in, st)); }
        if (d->w2) { WS_HIP(hipMemsetAsync(d->dw2, 0, sizeof(float) * (size_t)d->out_dim * d->conv_out, st)); if (d->db2) WS_HIP(hipMemsetAsync(d->db2, 0, sizeof(float) * d->out_dim, st)); }
        if (d->w2 && d->ws) WS_HIP(hipMemsetAsync(d->dws, 0, sizeof(float) * (size_t)d->out_dim * d->in_dim, st));
        if (d->dfeat && ns > 0) {
            if (d->dfeat_add) WS_HIP(hipMemcpyAsync(d->dfeat, d->dfeat_add, sizeof(float) * (size_t)ns * d->in_dim, hipMemcpyDeviceToDevice, st));
            else WS_HIP(hipMemsetAsync(d->dfeat, 0, sizeof(float) * (size_t)ns * d->in_dim, st));
        }
        return WS_OK;
    }
    bool gated2 = false;
    // stw: the stream of the dW products; fork(i) makes it wait for what the main stream has produced so far
    Side* side = nullptr;
    if (use_side) WS_TRY(side_for_current_device(&side));
    hipStream_t stw = side ? side->st : st;
    auto fork = [&](int i) -> int {
        if (!side) return WS_OK;
        WS_HIP(hipEventRecord(side->fork[i], st));
        WS_HIP(hipStreamWaitEvent(stw, side->fork[i], 0));
        return WS_OK;
    };
    const float* sc_res = nullptr;          // the shortcut's gradient w.r.t. feat rows [ns,in_dim]
    bool added = false;                     // dfeat_add already summed in (by the pool backward's store)
    const float* gin = d->dout;             // gradient entering the convolution's activation
    const float* yconv = d->out;
    WS_REQUIRE(!(d->gate_dfeat && d->dfeat_add && !(d->strided && want_sc)),
               "gate_dfeat with dfeat_add needs the strided shortcut (the sum has to be complete before the gate)");
    if (d->w2) {
        // dz = dout * lrelu'(out), db2 = column sums (dout_pregated: the consumer has done the multiplication)
        if (d->dout_pregated) {
            if (d->db2) WS_TRY(ws_act_bwd_colsum(d->dout, nq, d->out_dim, d->out_dim, nullptr, 0, 0.0f, nullptr, 0, d->db2, tmp, st));
            dz = const_cast<float*>(d->dout);          // (read only from here on)
        } else
        WS_TRY(ws_act_bwd_colsum(d->dout, nq, d->out_dim, d->out_dim, d->out, d->out_dim, d->slope, dz, d->out_dim, d->db2, tmp, st));
        WS_TRY(fork(0));
        WS_TRY(ws_gemm_xty(dz, nq, d->out_dim, d->out_dim, d->x2, d->conv_out, d->conv_out, d->dw2, tmp_w, stw));
        if (d->ws) {
            const float* sc_in = d->strided ? d->pooled : d->feat;
            WS_TRY(ws_gemm_xty(dz, nq, d->out_dim, d->out_dim, sc_in, d->in_dim, d->in_dim, d->dws, tmp_w, stw));
        }
        if (want_sc) {
            const float* dsc = dz;          // [nq, in_dim] when there is no projection (in_dim == out_dim)
            if (d->ws) {
                WS_TRY(ws_gemm_xb_epilogue_strided(dz, nq, d->out_dim, d->out_dim, d->ws, d->in_dim, 1, d->in_dim, nullptr, nullptr, 0, 0,
                                                   0.0f, dscin, d->in_dim, tmp, tmp_bytes, st));
                dsc = dscin;
            }
            if (d->strided) {
                WS_REQUIRE(d->t_offsets && d->t_pairs, "strided block backward needs the transposed table");
                if (arg_bytes(d))
                {
                    WS_TRY(ws_priv_max_pool_bwd_u8(dsc, reinterpret_cast<const uint8_t*>(d->arg), nq, d->h, d->in_dim, d->t_offsets,
                                                   d->t_pairs, ns, dfsc, ws_block_pool_order ? d->order_s : nullptr, d->dfeat_add, st));
                    added = d->dfeat_add != nullptr;
                }
                else {
                    WS_TRY(ws_max_pool_bwd(dsc, d->arg, nq, d->h, d->in_dim, d->t_offsets, d->t_pairs, ns, dfsc, st));
                    if (d->dfeat_add) {
                        WS_REQUIRE(d->in_dim % 4 == 0, "dfeat_add needs in_dim %% 4 == 0");
                        const int64_t n4 = ns * d->in_dim / 4;
                        add_rows_kernel<<<ws_grid(n4, 256), 256, 0, st>>>(dfsc, d->dfeat_add, n4);
                        WS_LAUNCH_CHECK();
                        added = true;
                    }
                }
                sc_res = dfsc;
            } else {
                sc_res = dsc;
            }
        }
        // dx2 = dz @ w2; without a bias gradient to sum (use_bn: BatchNormBlock is an identity, blocks.py:453-463) the
        // activation backward of the convolution's LeakyReLU rides on the epilogue: dz2 = dx2 * lrelu'(x2)
        gated2 = !d->dbk && ws_block_gates;
        WS_TRY(ws_gemm_xb_gated_strided(dz, nq, d->out_dim, d->out_dim, d->w2, d->conv_out, 1, d->conv_out, nullptr, nullptr, 0, 0, 0.0f,
                                        gated2 ? d->x2 : nullptr, d->conv_out, d->slope, nullptr, 0, 0.0f, g2, d->conv_out, tmp, tmp_bytes,
                                        st));
        gin = g2;
        yconv = d->x2;
    }
    // dz2 = g * lrelu'(x2), dbk
    if (!d->w2 && d->dout_pregated) {          // simple block whose consumer has applied the LeakyReLU' already
        if (d->dbk) WS_TRY(ws_act_bwd_colsum(gin, nq, d->conv_out, d->conv_out, nullptr, 0, 0.0f, nullptr, 0, d->dbk, tmp, st));
        g2 = const_cast<float*>(gin);              // (read only from here on)
    } else if (!gated2)
        WS_TRY(ws_act_bwd_colsum(gin, nq, d->conv_out, d->conv_out, yconv, d->conv_out, d->slope, g2, d->conv_out, d->dbk, tmp, st));
    WS_TRY(fork(1));
    WS_TRY(ws_gemm_xty(d->wf, nq, kc, kc, g2, d->conv_out, d->conv_out, d->dwk, tmp_w, stw));
    auto join = [&]() -> int {              // the caller's stream continues after the side products
        if (!side) return WS_OK;
        WS_HIP(hipEventRecord(side->join, stw));
        WS_HIP(hipStreamWaitEvent(st, side->join, 0));
        return WS_OK;
    };
    if (!need_dx1) return join();
    // dwf = dz2 @ wk^T
    WS_TRY(linear_fwd(lk, g2, nq, d->conv_out, nullptr, nullptr, 0, 0, 0.0f, dwf, trk, tmp, tmp_bytes, st));
    float* dx1_out = d->w1 ? dx1 : d->dfeat;
    // unary1's LeakyReLU backward rides on the store of K4 / K4G when no bias gradient has to be summed from dz1
    const float* gate1 = (d->w1 && !d->db1 && ws_block_gates) ? d->x1 : nullptr;
    if (!d->w1 && d->gate_dfeat && d->dfeat) gate1 = d->feat;   // no unary1: K4 / K4G writes dfeat itself -- times LeakyReLU'(feat)
    if (d->grid_blob) {
        WS_REQUIRE(d->key_last && d->grid_overflow && nq == ns, "grid backward needs key_last / overflow and a self-query layer");
        WS_TRY(ws_kpconv_gather_bwd_x_grid_gated(d->s_pts, ns, d->grid_blob, d->grid_nb, d->grid_cells, d->key_last, d->grid_radius, dwf,
                                                 d->conv_in, d->kernel_points, d->k, nullptr, nullptr, d->extent, WS_INFLUENCE_LINEAR,
                                                 WS_AGGREGATION_SUM, d->order_s, gate1, d->slope, d->inds, d->h, dx1_out,
                                                 d->grid_overflow, st));
    } else {
        WS_REQUIRE(d->t_offsets && d->t_pairs, "KPConv backward needs the search grid or the transposed table");
        WS_TRY(ws_kpconv_gather_bwd_x_gated(d->q_pts, nq, d->s_pts, ns, d->inds, d->h, d->t_offsets, d->t_pairs, dwf, d->conv_in,
                                            d->kernel_points, d->k, nullptr, nullptr, d->extent, WS_INFLUENCE_LINEAR,
                                            WS_AGGREGATION_SUM, d->order_s, gate1, d->slope, dx1_out, st));
    }
    if (d->w1) {
        if (!gate1)
            WS_TRY(ws_act_bwd_colsum(dx1, ns, d->conv_in, d->conv_in, d->x1, d->conv_in, d->slope, dx1, d->conv_in, d->db1, tmp, st));
        WS_TRY(fork(2));
        WS_TRY(ws_gemm_xty(dx1, ns, d->conv_in, d->conv_in, d->feat, d->in_dim, d->in_dim, d->dw1, tmp_w, stw));
        if (d->dfeat)
            WS_TRY(ws_gemm_xb_gated_strided(dx1, ns, d->conv_in, d->conv_in, d->w1, d->in_dim, 1, d->in_dim, nullptr, sc_res, d->in_dim,
                                            0, 0.0f, d->gate_dfeat ? d->feat : nullptr, d->in_dim, d->slope, nullptr, 0, 0.0f, d->dfeat,
                                            d->in_dim, tmp, tmp_bytes, st));
    } else if (d->dfeat && sc_res) {
        WS_REQUIRE(!d->gate_dfeat, "gate_dfeat: a block without unary1 has no shortcut to add after the gate");
        const int64_t n4 = ns * d->in_dim / 4;
        add_rows_kernel<<<ws_grid(n4, 256), 256, 0, st>>>(d->dfeat, sc_res, n4);
        WS_LAUNCH_CHECK();
    }
    if (d->dfeat && d->dfeat_add && !added) {       // (block shapes without a strided shortcut: a pass of its own)
        WS_REQUIRE(d->in_dim % 4 == 0, "dfeat_add needs in_dim %% 4 == 0");
        const int64_t n4 = ns * d->in_dim / 4;
        add_rows_kernel<<<ws_grid(n4, 256), 256, 0, st>>>(d->dfeat, d->dfeat_add, n4);
        WS_LAUNCH_CHECK();
    }
    return join();
}

// ---- decoder step ------------------------------------------------------------------------------------------------
int check_upunary(const ws_upunary* d)
{
    WS_REQUIRE(d, "NULL descriptor");
    WS_REQUIRE(d->nc >= 0 && d->nf >= 0 && d->c_up >= 1 && d->c_skip >= 1 && d->out_dim >= 1 && d->h_up >= 1, "bad sizes");
    WS_REQUIRE(d->xc && d->skip && d->ups && d->w && d->out && d->ldw >= d->c_up + d->c_skip, "NULL argument / bad ldw");
    if (d->c_up % 32 || d->c_skip % 32 || d->out_dim % 32 || d->ldw % 4 || ((uintptr_t)d->w & 15u))
        return ws_fail(WS_ERR_UNSUPPORTED, "decoder step: widths must be multiples of 32 (c_up=%d c_skip=%d out=%d)", d->c_up, d->c_skip,
                       d->out_dim);
    WS_REQUIRE(d->drop_p >= 0.0f && d->drop_p < 1.0f, "bad drop probability %g", (double)d->drop_p);
    if (d->drop_p > 0.0f && !d->relu) return ws_fail(WS_ERR_UNSUPPORTED, "decoder step: the fused dropout follows a LeakyReLU (relu = 0)");
    WS_REQUIRE(!(d->dout_pregated && !d->relu), "dout_pregated needs this step's LeakyReLU");
    return WS_OK;
}

int upunary_fwd(const ws_upunary* d, Arena& ar, hipStream_t st, bool run)
{
    float* up = ar.take<float>(d->nf * d->out_dim);
    const int64_t tmp_bytes = max3(ws_gemm_xb_scratch_bytes(d->nc, d->c_up, d->out_dim), ws_gemm_xb_scratch_bytes(d->nf, d->c_skip, d->out_dim), 16);
    void* tmp = ar.take<char>(tmp_bytes);
    if (!run || d->nf == 0) return WS_OK;
    WS_REQUIRE(d->yc, "yc buffer missing");
    if (d->nc > 0)
        WS_TRY(ws_gemm_xb_epilogue_strided(d->xc, d->nc, d->c_up, d->c_up, d->w, 1, d->ldw, d->out_dim, nullptr, nullptr, 0, 0, 0.0f, d->yc,
                                           d->out_dim, tmp, tmp_bytes, st));
    if (ws_block_gather_residual && d->nc > 0)
        // nearest upsampling (closest_pool) read by the epilogue: out row r adds yc[ups[r, 0]] (nothing for the shadow index);
        // the droplayer in front of the head (drop_p > 0) rides on the same epilogue
        return ws_priv_gemm_xb_ex(d->skip, d->nf, d->c_skip, d->c_skip, d->w + d->c_up, 1, d->ldw, d->out_dim, d->b, d->yc, d->out_dim,
                                  d->ups, d->h_up, d->nc, d->relu ? 1 : 0, d->slope, d->drop_p, d->drop_seed, d->out, d->out_dim, tmp,
                                  tmp_bytes, st);
    WS_TRY(ws_closest_pool_fwd(d->yc, d->nc, d->out_dim, d->ups, d->nf, d->h_up, up, st));
    if (d->drop_p > 0.0f)
        return ws_gemm_xb_dropout_strided(d->skip, d->nf, d->c_skip, d->c_skip, d->w + d->c_up, 1, d->ldw, d->out_dim, d->b, up, d->out_dim,
                                          1, d->slope, d->drop_p, d->drop_seed, d->out, d->out_dim, tmp, tmp_bytes, st);
    return ws_gemm_xb_epilogue_strided(d->skip, d->nf, d->c_skip, d->c_skip, d->w + d->c_up, 1, d->ldw, d->out_dim, d->b, up, d->out_dim,
                                       d->relu ? 1 : 0, d->slope, d->out, d->out_dim, tmp, tmp_bytes, st);
}

int upunary_bwd(const ws_upunary* d, Arena& ar, hipStream_t st, bool run)
{
    float* dz = ar.take<float>(d->nf * d->out_dim);
    float* dyc = ar.take<float>(d->nc * d->out_dim);
    int64_t tmp_bytes = max3(ws_act_bwd_colsum_scratch_bytes(d->nf, d->out_dim), ws_gemm_xty_scratch_bytes(d->nf, d->out_dim, d->c_skip),
                             ws_gemm_xty_scratch_bytes(d->nc, d->out_dim, d->c_up));
    tmp_bytes = max3(tmp_bytes, ws_gemm_xb_scratch_bytes(d->nf, d->out_dim, d->c_skip), ws_gemm_xb_scratch_bytes(d->nc, d->out_dim, d->c_up));
    void* tmp = ar.take<char>(tmp_bytes > 16 ? tmp_bytes : 16);
    if (!run) return WS_OK;
    WS_REQUIRE(d->dout && d->dw && d->dxc && d->dskip && d->t_offsets && d->t_pairs, "NULL gradient buffer / table");
    const int64_t cw = d->c_up + d->c_skip;
    if (d->nf == 0 || d->nc == 0) {
        WS_HIP(hipMemsetAsync(d->dw, 0, sizeof(float) * (size_t)d->out_dim * cw, st));
        if (d->db) WS_HIP(hipMemsetAsync(d->db, 0, sizeof(float) * d->out_dim, st));
        if (d->nc > 0) WS_HIP(hipMemsetAsync(d->dxc, 0, sizeof(float) * (size_t)d->nc * d->c_up, st));
        if (d->nf > 0) WS_HIP(hipMemsetAsync(d->dskip, 0, sizeof(float) * (size_t)d->nf * d->c_skip, st));
        return WS_OK;
    }
    // dz = dout * lrelu'(out) (identity when !relu), db
    const float* g = d->dout;
    if (d->relu && d->dout_pregated) {             // the consumer has applied this step's LeakyReLU' (and dropout backward) already
        if (d->db) WS_TRY(ws_act_bwd_colsum(d->dout, d->nf, d->out_dim, d->out_dim, nullptr, 0, 0.0f, nullptr, 0, d->db, tmp, st));
    } else if (d->drop_p > 0.0f) {
        WS_TRY(ws_act_bwd_colsum_dropout(d->dout, d->nf, d->out_dim, d->out_dim, d->out, d->out_dim, d->slope, d->drop_p, d->drop_seed, dz,
                                         d->out_dim, d->db, tmp, st));
        g = dz;
    } else if (d->relu) {
        WS_TRY(ws_act_bwd_colsum(d->dout, d->nf, d->out_dim, d->out_dim, d->out, d->out_dim, d->slope, dz, d->out_dim, d->db, tmp, st));
        g = dz;
    } else if (d->db) {
        WS_TRY(ws_act_bwd_colsum(d->dout, d->nf, d->out_dim, d->out_dim, nullptr, 0, 0.0f, nullptr, 0, d->db, tmp, st));
    }
    // skip side
    // (the two halves of dw [out, c_up + c_skip] are written in place: column blocks of pitch cw)
    WS_TRY(ws_priv_gemm_xty_pitched(g, d->nf, d->out_dim, d->out_dim, d->skip, d->c_skip, d->c_skip, d->dw + d->c_up, cw, tmp, st));
    WS_TRY(ws_gemm_xb_epilogue_strided(g, d->nf, d->out_dim, d->out_dim, d->w + d->c_up, d->ldw, 1, d->c_skip, nullptr, nullptr, 0, 0, 0.0f,
                                       d->dskip, d->c_skip, tmp, tmp_bytes, st));
    // coarse side: nearest upsampling backward, then the x-part of the unary
    WS_TRY(ws_closest_pool_bwd(g, d->nf, 1, d->out_dim, d->t_offsets, d->t_pairs, d->nc, dyc, st));
    WS_TRY(ws_priv_gemm_xty_pitched(dyc, d->nc, d->out_dim, d->out_dim, d->xc, d->c_up, d->c_up, d->dw, cw, tmp, st));
    return ws_gemm_xb_gated_strided(dyc, d->nc, d->out_dim, d->out_dim, d->w, d->ldw, 1, d->c_up, nullptr, nullptr, 0, 0, 0.0f,
                                    d->gate_dxc ? d->xc : nullptr, d->c_up, d->slope, nullptr, 0, 0.0f, d->dxc, d->c_up, tmp, tmp_bytes, st);
}

}  // namespace

extern "C" {

int64_t ws_kpblock_fwd_scratch_bytes(const ws_kpblock* d)
{
    if (check_kpblock(d)) return -1;
    Arena ar(nullptr, 0);
    kpblock_fwd(d, ar, nullptr, false);
    return ar.off + 256;
}

int64_t ws_kpblock_bwd_scratch_bytes(const ws_kpblock* d)
{
    if (check_kpblock(d)) return -1;
    Arena ar(nullptr, 0);
    kpblock_bwd(d, ar, nullptr, false);
    return ar.off + 256;
}

int ws_kpblock_fwd(const ws_kpblock* d, void* scratch, int64_t scratch_bytes, void* stream)
{
    WS_TRY(check_kpblock(d));
    WS_REQUIRE(scratch && ((uintptr_t)scratch & 15u) == 0, "scratch must be a 16-byte aligned device buffer");
    Arena plan(nullptr, 0);
    kpblock_fwd(d, plan, nullptr, false);
    if (plan.off > scratch_bytes) return ws_fail(WS_ERR_CAPACITY, "scratch too small: %lld < %lld bytes", (long long)scratch_bytes, (long long)plan.off);
    Arena ar(scratch, scratch_bytes);
    return kpblock_fwd(d, ar, (hipStream_t)stream, true);
}

int ws_kpblock_bwd(const ws_kpblock* d, void* scratch, int64_t scratch_bytes, void* stream)
{
    WS_TRY(check_kpblock(d));
    WS_REQUIRE(scratch && ((uintptr_t)scratch & 15u) == 0, "scratch must be a 16-byte aligned device buffer");
    Arena plan(nullptr, 0);
    kpblock_bwd(d, plan, nullptr, false);
    if (plan.off > scratch_bytes) return ws_fail(WS_ERR_CAPACITY, "scratch too small: %lld < %lld bytes", (long long)scratch_bytes, (long long)plan.off);
    Arena ar(scratch, scratch_bytes);
    return kpblock_bwd(d, ar, (hipStream_t)stream, true);
}

int64_t ws_upunary_fwd_scratch_bytes(const ws_upunary* d)
{
    if (check_upunary(d)) return -1;
    Arena ar(nullptr, 0);
    upunary_fwd(d, ar, nullptr, false);
    return ar.off + 256;
}

int64_t ws_upunary_bwd_scratch_bytes(const ws_upunary* d)
{
    if (check_upunary(d)) return -1;
    Arena ar(nullptr, 0);
    upunary_bwd(d, ar, nullptr, false);
    return ar.off + 256;
}

int ws_upunary_fwd(const ws_upunary* d, void* scratch, int64_t scratch_bytes, void* stream)
{
    WS_TRY(check_upunary(d));
    WS_REQUIRE(scratch && ((uintptr_t)scratch & 15u) == 0, "scratch must be a 16-byte aligned device buffer");
    Arena plan(nullptr, 0);
    upunary_fwd(d, plan, nullptr, false);
    if (plan.off > scratch_bytes) return ws_fail(WS_ERR_CAPACITY, "scratch too small: %lld < %lld bytes", (long long)scratch_bytes, (long long)plan.off);
    Arena ar(scratch, scratch_bytes);
    return upunary_fwd(d, ar, (hipStream_t)stream, true);
}

int ws_upunary_bwd(const ws_upunary* d, void* scratch, int64_t scratch_bytes, void* stream)
{
    WS_TRY(check_upunary(d));
    WS_REQUIRE(scratch && ((uintptr_t)scratch & 15u) == 0, "scratch must be a 16-byte aligned device buffer");
    Arena plan(nullptr, 0);
    upunary_bwd(d, plan, nullptr, false);
    if (plan.off > scratch_bytes) return ws_fail(WS_ERR_CAPACITY, "scratch too small: %lld < %lld bytes", (long long)scratch_bytes, (long long)plan.off);
    Arena ar(scratch, scratch_bytes);
    return upunary_bwd(d, ar, (hipStream_t)stream, true);
}

int ws_timer_reset(void)
{
    for (auto& r : timer_recs()) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
        (void)hipEventDestroy(r.c);
    }
    timer_recs().clear();
    return WS_OK;
}

int ws_timer_count(void) { return (int)timer_recs().size(); }

int ws_timer_read(int32_t i, int64_t* nq, int32_t* h, int32_t* ci, float* ms)
{
    WS_REQUIRE(i >= 0 && i < (int)timer_recs().size() && nq && h && ci && ms, "bad timer record index %d", i);
    const TimerRec& r = timer_recs()[i];
    WS_HIP(hipEventSynchronize(r.b));
    WS_HIP(hipEventElapsedTime(ms, r.a, r.b));
    *nq = r.nq; *h = r.h; *ci = r.ci;
    return WS_OK;
}

int ws_timer_read_layer(int32_t i, float* ms)
{
    WS_REQUIRE(i >= 0 && i < (int)timer_recs().size() && ms, "bad timer record index %d", i);
    const TimerRec& r = timer_recs()[i];
    WS_HIP(hipEventSynchronize(r.c));
    WS_HIP(hipEventElapsedTime(ms, r.a, r.c));
    return WS_OK;
}

}  // extern "C"
