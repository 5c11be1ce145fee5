"""GPU: the deformable fast path (packed kernel points kp4, MODE-2 gather kernels, geometry backward on the matrix core,
fused prepare / regulariser kernels, queue form of the grid backward) against the generic kernels that tests/
test_kpconv_gpu.py pins on the reference's goldens g5, on the same inputs: f32 rows, agreement to fp32 re-association.
The oracle comparisons at real widths are in tests/test_config5_wide_gpu.py; this file isolates the A/B."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _geometry(gpu, n, radius, seed, strided):
    from weasal_amd import ops
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
    lens = np.array([n // 2, n - n // 2], np.int32)
    P = torch.from_numpy(pts).to(gpu)
    if not strided:
        return P, P, ops.radius_neighbors(P, P, lens, lens, radius, dtype=torch.int64)
    Q = P[::4].contiguous()
    ql = np.array([len(range(0, int(lens[0]), 4)), 0], np.int32)
    ql[1] = Q.shape[0] - ql[0]
    return Q, P, ops.radius_neighbors(Q, P, ql, lens, radius, dtype=torch.int64)


@pytest.mark.parametrize("ci,co,radius,strided,modulated", [(16, 16, 0.9, False, False), (32, 64, 1.3, False, True),
                                                            (48, 32, 1.1, True, True), (64, 64, 1.6, False, True),
                                                            (128, 32, 1.2, True, True), (256, 64, 1.0, False, True)])
def test_fast_path_equals_generic_kernels(gpu, ci, co, radius, strided, modulated):
    from weasal_amd import blocks
    from weasal_amd.architectures import p2p_fitting_regularizer
    from weasal_amd.blocks import KPConv
    import weasal_amd.architectures as arch
    q, s, inds = _geometry(gpu, 5000, radius, ci, strided)
    np.random.seed(1)
    torch.manual_seed(1)
    conv = KPConv(15, 3, ci, co, 0.4 * radius, radius, deformable=True, modulated=modulated).to(gpu)
    with torch.no_grad():
        conv.offset_conv.weights.mul_(4.0 * (32.0 / ci) ** 0.5)
        conv.offset_bias.normal_(0.0, 0.05)
    x = torch.randn(s.shape[0], ci, device=gpu)
    dy = torch.randn(q.shape[0], co, device=gpu)
    mk = lambda c: types.SimpleNamespace(modules=lambda: [c], l1=torch.nn.L1Loss(), K=15, repulse_extent=1.2, deform_fitting_power=1.0)
    res = {}
    for fast in (True, False):
        blocks.DEFORM_FAST_PATH = fast
        arch.REGULARIZER_KERNEL = fast
        try:
            conv.zero_grad()
            xg = x.clone().requires_grad_(True)
            out = conv(q, s, inds, xg)
            reg = p2p_fitting_regularizer(mk(conv))
            ((out * dy).sum() + reg).backward()
            res[fast] = dict(out=out.detach(), reg=reg.detach(), min_d2=conv.min_d2.detach(), dkp=conv.deformed_KP.detach(),
                             dx=xg.grad, dW=conv.weights.grad.clone(), dWo=conv.offset_conv.weights.grad.clone(),
                             dbo=conv.offset_bias.grad.clone())
        finally:
            blocks.DEFORM_FAST_PATH = True
            arch.REGULARIZER_KERNEL = True
    a, b = res[True], res[False]
    assert rel(a["dkp"], b["dkp"]) < 1e-6
    for key, tol in (("out", 2e-5), ("min_d2", 1e-5), ("reg", 1e-5), ("dW", 2e-5), ("dx", 1e-4), ("dWo", 1e-4), ("dbo", 1e-4)):
        assert rel(a[key], b[key]) < tol, (key, rel(a[key], b[key]))


def test_regularizer_kernel_vs_torch_ops(gpu):
    """ws_p2p_regularizer_fwd / _bwd against the torch-op restatement of models/architectures.py:36-51"""
    import weasal_amd.architectures as arch
    from weasal_amd.architectures import p2p_fitting_regularizer
    torch.manual_seed(3)
    n, ext = 7001, 0.8
    res = {}
    for fast in (True, False):
        arch.REGULARIZER_KERNEL = fast
        try:
            dkp = (torch.randn(n, 15, 3, device=gpu) * 0.5 * ext).requires_grad_(True)
            md = (torch.rand(n, 15, device=gpu) * ext * ext).requires_grad_(True)
            m = types.SimpleNamespace(min_d2=md, deformed_KP=dkp, KP_extent=ext)
            net = types.SimpleNamespace(_deformable_layers=[m], l1=torch.nn.L1Loss(), K=15, repulse_extent=1.2, deform_fitting_power=1.0)
            torch.manual_seed(3)
            with torch.no_grad():
                dkp.copy_(torch.randn(n, 15, 3, device=gpu) * 0.5 * ext)
                md.copy_(torch.rand(n, 15, device=gpu) * ext * ext)
            reg = p2p_fitting_regularizer(net)
            (3.0 * reg).backward()
            res[fast] = (reg.detach(), dkp.grad.clone(), md.grad.clone())
        finally:
            arch.REGULARIZER_KERNEL = True
    assert rel(res[True][0], res[False][0]) < 1e-6
    assert rel(res[True][1], res[False][1]) < 1e-5 and rel(res[True][2], res[False][2]) < 1e-6


@pytest.mark.parametrize("ci", [32, 64, 256])
def test_grid_backward_queue_form_equals_transposed_table(gpu, ci):
    """rigid KPConv on rows of ~300 neighbours: dx through the search grid (ws_kpconv_gather_bwd_x_grid_wide, candidates
    from the supports' own rows and from the grid walk) against dx through the transposed table, fp32 re-association"""
    from weasal_amd import config as wcfg, ops, pyramid
    from weasal_amd.kernel_points import load_kernels
    cfg = wcfg.DALESDeformF32Config()
    rng = np.random.default_rng(9)
    pts = rng.uniform(-4, 4, size=(6000, 3)).astype(np.float32)
    P = torch.from_numpy(pts).to(gpu)
    np.random.seed(0)
    batch = pyramid.build_batch(cfg, P, torch.ones(6000, 3, device=gpu), torch.zeros(6000, dtype=torch.int64, device=gpu),
                                np.array([6000], np.int32), [260, 519, 472, 193, 34])
    batch.activate()
    inds = batch.neighbors[0]
    grid = ops._grid_for(inds)
    assert inds.shape[1] == 260 and grid is not None and grid.max_count > 260        # truncated rows: key_last at work
    kp = torch.from_numpy(load_kernels(1.0, 15, dimension=3, fixed="center").astype(np.float32)).to(gpu)
    torch.manual_seed(ci)
    x = torch.randn(6000, ci, device=gpu, requires_grad=True)
    wf, _ = ops.kpconv_gather(x, P, P, inds, kp, 0.8)
    g = torch.randn_like(wf)
    dx_grid, = torch.autograd.grad(wf, x, g, retain_graph=True)
    ops.GRID_BACKWARD = False
    try:
        dx_tab, = torch.autograd.grad(wf, x, g)
    finally:
        ops.GRID_BACKWARD = True
    assert rel(dx_grid, dx_tab) < 2e-6


@pytest.mark.parametrize("rows", ["f32", "bf16"])
def test_sorted_row_cutoff_changes_nothing(gpu, rows):
    """Rows that come from the radius search are sorted by distance; with the deformable search radius (2 r) most of a row
    lies beyond the reach of every kernel point (0.69 r + extent) and the kernels stop there.  The skipped influences are
    exact zeros and the skipped columns cannot hold a minimum: forward, min_d2 and every gradient must be BIT-identical to
    the full walk (ops.SORTED_ROW_CUTOFF = False), rigid and deformable, self-query (grid backward) and strided (table)."""
    from weasal_amd import config as wcfg, ops, pyramid
    from weasal_amd.blocks import KPConv
    cfg = wcfg.DALESDeformF32Config()
    rng = np.random.default_rng(11)
    pts = rng.uniform(-4, 4, size=(6000, 3)).astype(np.float32)
    P = torch.from_numpy(pts).to(gpu)
    np.random.seed(0)
    batch = pyramid.build_batch(cfg, P, torch.ones(6000, 3, device=gpu), torch.zeros(6000, dtype=torch.int64, device=gpu),
                                np.array([6000], np.int32), [300, 519, 472, 193, 34])
    batch.activate()
    dt = torch.bfloat16 if rows == "bf16" else torch.float32
    for q, s, inds in ((batch.points[0], batch.points[0], batch.neighbors[0]), (batch.points[1], batch.points[0], batch.pools[0])):
        assert ops.rows_cutoff_pays(inds, 1.0) and inds.shape[1] >= 250
        for deformable in (False, True):
            np.random.seed(1)
            torch.manual_seed(1)
            conv = KPConv(15, 3, 32, 32, 0.4, 1.0, deformable=deformable, modulated=deformable).to(gpu)
            if deformable:
                with torch.no_grad():
                    conv.offset_conv.weights.mul_(4.0)
                    conv.offset_bias.normal_(0.0, 0.05)
            x = torch.randn(s.shape[0], 32, device=gpu).to(dt)
            dy = torch.randn(q.shape[0], 32, device=gpu).to(dt)
            res = {}
            for cut in (True, False):
                ops.SORTED_ROW_CUTOFF = cut
                try:
                    conv.zero_grad()
                    xg = x.clone().requires_grad_(True)
                    out = conv(q, s, inds, xg)
                    loss = (out.float() * dy.float()).sum()
                    if deformable:
                        loss = loss + 0.01 * conv.min_d2.sum() + 0.01 * (conv.deformed_KP ** 2).sum()
                    loss.backward()
                    res[cut] = {"out": out.detach(), "dW": conv.weights.grad.clone(), "dx": xg.grad.clone()}
                    if deformable:
                        res[cut].update(min_d2=conv.min_d2.detach(), dWo=conv.offset_conv.weights.grad.clone(),
                                        dbo=conv.offset_bias.grad.clone())
                finally:
                    ops.SORTED_ROW_CUTOFF = True
            a, b = res[True], res[False]
            # the forward, min_d2 and what depends on them alone: the same sums over the same non-zero terms in the same order
            for key in ("out", "dW", "min_d2"):
                if key in a:
                    assert torch.equal(a[key], b[key]), (key, deformable, tuple(inds.shape))
            # dx of a grid-walk support groups its (fewer) surviving candidates into other batches of 64: fp32 re-association,
            # nothing else; the gradients behind it inherit that
            tol = 2e-6 if rows == "f32" else 1e-2
            for key in ("dx", "dWo", "dbo"):
                if key in a:
                    assert rel(a[key], b[key]) < tol, (key, rel(a[key], b[key]), deformable, tuple(inds.shape))
