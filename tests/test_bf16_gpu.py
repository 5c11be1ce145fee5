"""GPU parity of the bf16-feature path (BASELINE config 5: "DALES deformable-KPConv bf16"; SURVEY.md section 8d C5:
feature rows / weights bf16 in HBM, fp32 accumulate, geometry fp32).

The reference has no reduced-precision path, so the oracle is the fp32 restatement of the reference's op sequence
(oracle/kpconv_ref.py, pinned by goldens g4/g5/g6) evaluated on the SAME bf16-rounded inputs and bf16-rounded
weights.  What then differs is only what the bf16 path rounds on the way: the weighted features wf, each layer's
output, and in the backward dwf / dx / dz -- one round-to-nearest-even to 8 significant bits each
(unit roundoff u = 2^-9 = 1.95e-3); every sum runs in fp32.

Tolerances held (max|a-b| / max|ref| per tensor), and why:
  * dense products alone (one rounding of the output):                 4e-3   (= 2u: the element of largest
    magnitude is off by at most u, others relative to the max less)
  * one rigid KPConv layer, activations (wf and out rounded):           1e-2
  * its gradients dx / dW (dz, dwf, dx rounded; dW is an fp32 sum of products of rounded rows):   2e-2
  * deformable + modulated layer (the offsets come from a bf16-row convolution and move the kernel points,
    so rounding enters the geometry):   out 2e-2; parameter gradients (sums over all points) 5e-2.  The gradient with
    respect to the kernel-point positions is DISCONTINUOUS where a neighbour crosses the influence extent
    (d w / d kp jumps from -(kp - n)/(extent |kp - n|) to 0, models/blocks.py:337), so a 1e-2 perturbation of the
    offsets (what the bf16-row offset convolution delivers) flips the fraction 3 * 1e-2 * max|kp| / extent ~ 4 % of the
    boundary terms: the gradients that flow through the offsets (dx, d offset weights / bias) are held at 0.12 with
    `linear` influence (measured 0.05-0.07) -- and at 3e-2 with the smooth `gaussian` influence, which has no such
    jump and therefore pins the bf16 arithmetic of that chain itself
  * pooling: bit-exact (max / copy of bf16 values).
Integer outputs (neighbours, subsampling) are untouched by the feature dtype.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return ((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def rel2(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return ((a - ref).norm() / ref.norm().clamp_min(1e-30)).item()


def rbf(t):
    """round to bf16 and back (the values the bf16 path sees)"""
    return t.to(BF).to(torch.float32)


@pytest.mark.parametrize("m,k,n,with_bias,with_res,slope,out_f32",
                         [(5000, 64, 32, True, False, 0.1, False), (70001, 480, 32, True, False, 0.1, False),
                          (20000, 32, 128, False, True, 0.1, False), (9000, 128, 9, True, False, 0.1, True),
                          (6000, 960, 64, True, True, None, False), (300, 3840, 256, False, False, None, False),
                          (33, 512, 60, False, False, None, True), (400000, 128, 128, True, True, 0.1, False)])
def test_bf16_dense_products_vs_float64(gpu, m, k, n, with_bias, with_res, slope, out_f32):
    """ws_gemm_xbt_bf16 / ws_gemm_xty_bf16 / ws_act_bwd_colsum_bf16 through ops.matmul_epilogue (forward, dx, dW, bias
    gradient, residual gradient) against float64 on the same bf16-rounded operands"""
    from weasal_amd import ops
    torch.manual_seed(m + n)
    x = torch.randn(m, k, device=gpu).to(BF)
    b = torch.randn(k, n, device=gpu) / k ** 0.5
    bias = torch.randn(n, device=gpu) if with_bias else None
    res = torch.randn(m, n, device=gpu).to(BF) if with_res else None
    dy = torch.randn(m, n, device=gpu)
    dy = dy if out_f32 else dy.to(BF)
    leaves = [t.clone().requires_grad_(True) for t in (x, b) + ((bias,) if with_bias else ()) + ((res,) if with_res else ())]
    it = iter(leaves)
    xx, bb = next(it), next(it)
    bi = next(it) if with_bias else None
    rr = next(it) if with_res else None
    y = ops.matmul_epilogue(xx, bb, bias=bi, residual=rr, slope=slope, out_f32=out_f32)
    assert y.dtype == (torch.float32 if out_f32 else BF)
    y.backward(dy)
    # float64 on the rounded operands
    l64 = [t.detach().double().requires_grad_(True) for t in (x, rbf(b)) + ((bias,) if with_bias else ()) + ((res,) if with_res else ())]
    it = iter(l64)
    x6, b6 = next(it), next(it)
    y6 = x6 @ b6
    if with_bias:
        y6 = y6 + next(it)
    if with_res:
        y6 = y6 + next(it)
    if slope is not None:
        y6 = torch.nn.functional.leaky_relu(y6, slope)
    y6.backward(dy.double())
    assert rel(y, y6) < (1e-5 if out_f32 else 4e-3)
    assert xx.grad.dtype == BF and bb.grad.dtype == torch.float32
    assert rel(xx.grad, x6.grad) < 8e-3             # dz rounded, then dx rounded
    assert rel(bb.grad, b6.grad) < 8e-3             # fp32 sum over rounded dz
    for a, r in zip(leaves[2:], l64[2:]):
        assert rel(a.grad, r.grad) < 8e-3


def _geometry(gpu, n=6000, radius=0.9, seed=0):
    from weasal_amd import ops
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
    lens = np.array([n // 2, n - n // 2], np.int32)
    P = torch.from_numpy(pts).to(gpu)
    inds = ops.radius_neighbors(P, P, lens, lens, radius, dtype=torch.int64)
    return P, inds


def _layer_pair(gpu, ci, co, extent, radius, **kw):
    """the GPU module (fp32 masters; the bf16 path rounds them itself) and its CPU twin with bf16-rounded weights"""
    from weasal_amd.blocks import KPConv
    np.random.seed(1)
    torch.manual_seed(1)
    conv = KPConv(15, 3, ci, co, extent, radius, **kw)
    if kw.get("deformable"):
        with torch.no_grad():                   # offsets of a useful size (zero-mean init gives ~0.01 extents)
            conv.offset_conv.weights.mul_(4.0)
            conv.offset_bias.normal_(0.0, 0.05)
    twin = copy.deepcopy(conv)
    with torch.no_grad():
        twin.weights.copy_(rbf(twin.weights))
        if kw.get("deformable"):
            twin.offset_conv.weights.copy_(rbf(twin.offset_conv.weights))
    return conv.to(gpu), twin


@pytest.mark.parametrize("ci,co", [(32, 32), (64, 64), (128, 64)])
def test_rigid_kpconv_bf16_vs_oracle(gpu, ci, co):
    from oracle import kpconv_ref
    P, inds = _geometry(gpu)
    conv, twin = _layer_pair(gpu, ci, co, 0.36, 0.9)
    torch.manual_seed(2)
    x = torch.randn(P.shape[0], ci, device=gpu).to(BF)
    dy = torch.randn(P.shape[0], co, device=gpu).to(BF)
    xg = x.clone().requires_grad_(True)
    out = conv(P, P, inds, xg)
    assert out.dtype == BF
    out.backward(dy)
    xc = x.float().cpu().requires_grad_(True)
    with kpconv_ref.cpu_reference_mode():
        ref = twin(P.cpu(), P.cpu(), inds.cpu(), xc)
    ref.backward(dy.float().cpu())
    assert rel(out, ref) < 1e-2
    assert xg.grad.dtype == BF and conv.weights.grad.dtype == torch.float32
    assert rel(xg.grad, xc.grad) < 2e-2
    assert rel(conv.weights.grad, twin.weights.grad) < 2e-2


@pytest.mark.parametrize("modulated,influence", [(False, "linear"), (True, "linear"), (True, "gaussian")])
def test_deformable_kpconv_bf16_vs_oracle(gpu, modulated, influence):
    """deformable (+ modulated) KPConv, models/blocks.py:244-325,366-367: bf16 rows, f32 offsets / geometry"""
    import types
    from oracle import kpconv_ref
    from weasal_amd.architectures import p2p_fitting_regularizer
    ci, co = 32, 64
    P, inds = _geometry(gpu, radius=1.2, seed=3)
    conv, twin = _layer_pair(gpu, ci, co, 0.36, 0.9, deformable=True, modulated=modulated, KP_influence=influence)
    torch.manual_seed(4)
    x = torch.randn(P.shape[0], ci, device=gpu).to(BF)
    dy = torch.randn(P.shape[0], co, device=gpu).to(BF)
    xg = x.clone().requires_grad_(True)
    out = conv(P, P, inds, xg)
    assert out.dtype == BF and conv.offset_features.dtype == torch.float32 and conv.min_d2.dtype == torch.float32
    mk = lambda c: types.SimpleNamespace(modules=lambda: [c], l1=torch.nn.L1Loss(), K=15, repulse_extent=1.2,
                                         deform_fitting_power=1.0)
    reg = p2p_fitting_regularizer(mk(conv))
    ((out.float() * dy.float()).sum() + reg).backward()
    xc = x.float().cpu().requires_grad_(True)
    with kpconv_ref.cpu_reference_mode():
        ref = twin(P.cpu(), P.cpu(), inds.cpu(), xc)
        reg_c = p2p_fitting_regularizer(mk(twin))
    ((ref * dy.float().cpu()).sum() + reg_c).backward()
    assert rel(conv.offset_features, twin.offset_features) < 1e-2
    assert rel(conv.deformed_KP, twin.deformed_KP) < 1e-2
    assert rel(conv.min_d2, twin.min_d2) < 2e-2
    assert rel(out, ref) < 2e-2
    assert abs(float(reg) - float(reg_c)) < 2e-2 * abs(float(reg_c))
    errs = {"dx_l2": rel2(xg.grad, xc.grad), "dx_max": rel(xg.grad, xc.grad),
            "dW": rel(conv.weights.grad, twin.weights.grad),
            "dW_off": rel(conv.offset_conv.weights.grad, twin.offset_conv.weights.grad),
            "db_off": rel(conv.offset_bias.grad, twin.offset_bias.grad)}
    print("bf16 deformable gradient errors:", errs)
    tol = 0.12 if influence == "linear" else 3e-2
    assert errs["dx_l2"] < tol and errs["dx_max"] < 2.5 * tol, errs
    assert errs["dW"] < 3e-2 and errs["dW_off"] < tol and errs["db_off"] < tol, errs


def test_pools_bf16_bit_exact(gpu):
    from oracle import kpconv_ref
    from weasal_amd import blocks, ops
    rng = np.random.default_rng(7)
    ns, nq, h, c = 5000, 1700, 23, 64
    inds = torch.from_numpy(rng.integers(0, ns + 1, size=(nq, h))).to(gpu)
    x = torch.randn(ns, c, device=gpu).to(BF).requires_grad_(True)
    dy = torch.randn(nq, c, device=gpu).to(BF)
    mp = blocks.max_pool(x, inds)
    assert mp.dtype == BF
    assert torch.equal(mp.detach().float().cpu(), kpconv_ref.max_pool_ref(x.detach().float().cpu(), inds.cpu()))
    mp.backward(dy)
    xc = x.detach().float().cpu().requires_grad_(True)
    kpconv_ref.max_pool_ref(xc, inds.cpu()).backward(dy.float().cpu())
    assert rel(x.grad, xc.grad) < 8e-3                                    # fp32 sum of bf16 rows, rounded once
    up = torch.from_numpy(rng.integers(0, nq + 1, size=(ns, 5))).to(gpu)
    xq = torch.randn(nq, c, device=gpu).to(BF).requires_grad_(True)
    cp = blocks.closest_pool(xq, up)
    assert torch.equal(cp.detach().float().cpu(), kpconv_ref.closest_pool_ref(xq.detach().float().cpu(), up.cpu()))
    g = torch.randn(ns, c, device=gpu).to(BF)
    cp.backward(g)
    xqc = xq.detach().float().cpu().requires_grad_(True)
    kpconv_ref.closest_pool_ref(xqc, up.cpu()).backward(g.float().cpu())
    assert rel(xq.grad, xqc.grad) < 8e-3


@pytest.mark.timeout(900)
def test_config5_network_bf16_steps_close_to_f32(gpu):
    """BASELINE config 5 as specified (every resnetb -> resnetb_deformable, modulated, deform_radius 5.0, bf16 rows) on
    small spheres: the bf16 network's loss follows the f32 network's (same masters, same batch) within 2 %, its
    parameters stay finite and the loss decreases over SGD steps"""
    from weasal_amd import config as wcfg, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.trainer import make_optimizer, train_step
    losses = {}
    for name in ("DALESDeformConfig", "DALESDeformF32Config"):
        cfg = getattr(wcfg, name)()
        cfg.dropout = 0.0
        cfg.first_subsampling_dl = 0.4
        assert all('deformable' in b for b in cfg.architecture if b.startswith('resnetb')) and cfg.modulated
        np.random.seed(3)
        torch.manual_seed(3)
        net = KPFCNN(cfg, np.arange(9), []).to(gpu).train()
        opt = make_optimizer(net, cfg)
        pts, feats, labels, lens = synthetic.make_inputs(11, 2, 6000, 5.0, cfg.in_features_dim)
        np.random.seed(5)
        batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu),
                                    torch.from_numpy(labels).to(gpu), lens, [80, 90, 100, 100, 80])
        ls = []
        for _ in range(4):
            loss, out = train_step(net, opt, batch, cfg)
            ls.append(loss.item())
        assert out.dtype == torch.float32
        assert all(np.isfinite(ls)) and all(torch.isfinite(p).all() for p in net.parameters())
        assert ls[-1] < ls[0]
        losses[name] = ls
    a, b = losses["DALESDeformConfig"], losses["DALESDeformF32Config"]
    assert abs(a[0] - b[0]) < 2e-2 * abs(b[0]), (a, b)
