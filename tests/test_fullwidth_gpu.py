"""GPU: the networks at their REAL widths through the REAL device pyramid, against the CPU oracle.

Every golden fixture is a <= 64-channel scale model with H ~ 20-40; the paths the benchmark runs in the deep
DALES levels (G = 16 lanes per row, Ci 128-512, rows of 65-128 neighbours on two columns per lane, split-K GEMMs,
the short-operand branch, the table-free backward K4G that only a batch built by pyramid.build_batch takes) were
covered by random-index edge tests and self-consistency only ("What's weak").  Here:

  * BASELINE config 3 (DALES_PseudoLabel, first_features_dim 128, limits 59/73/81/77/56): 2 x 50 000-point spheres,
    the batch from pyramid.build_batch (search grids exported, cell orders registered, tables pre-built), dropout 0;
    logits / loss / every parameter gradient against oracle.kpconv_ref.cpu_reference_mode() evaluated on the SAME
    index matrices (reference unit: models/architectures.py:328-384, utils/trainer_PseudoLabel.py:199-219);
  * BASELINE config 2 (Vaihingen3D_PseudoLabel, first_features_dim 64, in_features_dim 4, no neighbour limits):
    4 x 3 000-point spheres; the device pyramid against oracle.pyramid_ref (bit-exact up to exact-distance ties),
    then one whole SGD step against the oracle.

Tolerances (north_star: 1e-4 relative on fp32 activations): logits 1e-4 of max|ref|, loss 1e-5 relative, parameter
gradients 1e-3 of max|ref| per tensor (30 layers of fp32 re-association), parameters after the step 1e-4.
Parameter gradients at full width are ILL-CONDITIONED tensor by tensor: the network is piecewise linear with kinks
(LeakyReLU, max-pool arg-max, the clamp of the linear influence), so its gradient is discontinuous in the activations,
and many tensors are sums with heavy cancellation (|grad| ~ 1e-9 from terms of 1e-6).  Measured on the fp32 CPU oracle
itself: multiplying the input features by (1 + 1e-7 * noise) moves single parameter-gradient tensors by up to 5e-3 of
their maximum while the logits move by 1e-6; an fp32 re-association in any kernel is a perturbation of that kind.  The
per-tensor 1e-3 bound is therefore held where it is well defined -- the golden network g8 (tests/test_pyramid_gpu.py) --
and at full width the gradient is compared as ONE vector, with the tolerance calibrated inside the test: its relative
L2 error over all parameters must be within 3 x (+ 1e-4) the change of the ORACLE's own gradient vector when the input
features are perturbed by 1e-6 (measured per block: the GPU's gradients of the block outputs deviate from the oracle's by
1e-3 .. 9e-2 in max norm, the perturbed oracle's by 3e-3 .. 9e-2, while the activations agree to 1e-6:
tools/blockgrad_diag.py), and no tensor may be off by more than 5e-2 of its maximum.
"""
import copy

import numpy as np
import pytest
import torch

from conftest import assert_neighbors_equal

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = np.asarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _cpu_copy(batch):
    """the same batch (same index matrices) as plain CPU tensors"""
    from weasal_amd.pyramid import PyramidBatch
    flat = (batch.points + batch.neighbors + batch.pools + batch.upsamples + batch.lengths
            + [batch.features, batch.labels])
    return PyramidBatch([t.detach().cpu() for t in flat])


def _grad_vector_error(grads_a, grads_b):
    """(relative L2 error of the concatenated gradient, {name: max-norm error relative to the tensor's maximum})"""
    num = den = 0.0
    errs = {}
    for name, b in grads_b.items():
        a = grads_a[name].detach().double().cpu()
        b = b.detach().double().cpu()
        num += float(((a - b) ** 2).sum())
        den += float((b ** 2).sum())
        errs[name] = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    return (num / den) ** 0.5, errs


def _perturbed_oracle_grads(net_cpu, batch_cpu, cfg, eps=1e-6):
    """the fp32 CPU oracle's parameter gradients with the input features multiplied by (1 + eps * noise)"""
    from oracle import kpconv_ref
    from weasal_amd.architectures import KPFCNN
    net2 = KPFCNN(cfg, np.arange(9), [])          # (a module that has run holds graph tensors: no deepcopy)
    net2.load_state_dict(net_cpu.state_dict())
    net2.train()
    b2 = _cpu_copy(batch_cpu)
    gen = torch.Generator().manual_seed(123)
    b2.features = b2.features * (1 + eps * torch.randn(b2.features.shape, generator=gen))
    with kpconv_ref.cpu_reference_mode():
        out = net2(b2, cfg)
        net2.loss(out, b2.labels).backward()
    return {k: p.grad for k, p in net2.named_parameters() if p.grad is not None}


def _f64_truth_grads(net_cpu, batch_cpu, cfg, clip=None):
    """the same network, batch and loss evaluated in float64 by the oracle's op sequence: the gradient every fp32
    evaluation (the oracle's own included) is an approximation of"""
    from oracle import kpconv_ref
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.pyramid import PyramidBatch
    net2 = KPFCNN(cfg, np.arange(9), [])
    net2.load_state_dict(net_cpu.state_dict())
    net2.double().train()
    flat = (batch_cpu.points + batch_cpu.neighbors + batch_cpu.pools + batch_cpu.upsamples + batch_cpu.lengths
            + [batch_cpu.features, batch_cpu.labels])
    b2 = PyramidBatch([t.detach().double() if t.is_floating_point() else t.detach() for t in flat])
    with kpconv_ref.cpu_reference_mode():
        out = net2(b2, cfg)
        net2.loss(out, b2.labels).backward()
    g = {k: p.grad for k, p in net2.named_parameters() if p.grad is not None}
    return {k: (v.clamp(-clip, clip) if clip else v) for k, v in g.items()}


def _check_grads(net, ref_grads, truth_grads, tag):
    """Every parameter gradient against the float64 truth: the GPU's fp32 gradient may be no farther from it than 3 x the
    fp32 ORACLE's own distance (+ 1e-6 of the tensor's maximum): 'no worse than the fp32 reference against the truth'.
    ref_grads: the fp32 oracle's gradients; truth_grads: the float64 evaluation (both treated as the GPU's were: clipped
    or not).  The measured per-tensor numbers are written to gpurun_out/fullwidth_gradients_<tag>.json."""
    import json
    import os
    rows = {}
    num_g = num_o = den = 0.0
    for name, p in net.named_parameters():
        assert (p.grad is None) == (name not in ref_grads), name
        if p.grad is None:
            continue
        t = truth_grads[name].double()
        g = p.grad.detach().double().cpu()
        o = ref_grads[name].detach().double()
        scale = float(t.abs().max().clamp_min(1e-300))
        rows[name] = {"gpu_vs_f64": float((g - t).abs().max()) / scale, "oracle_f32_vs_f64": float((o - t).abs().max()) / scale,
                      "gpu_vs_oracle_f32": float((g - o).abs().max()) / scale, "max_abs": scale, "numel": t.numel()}
        num_g += float(((g - t) ** 2).sum())
        num_o += float(((o - t) ** 2).sum())
        den += float((t ** 2).sum())
    summary = {"global_rel_l2_gpu_vs_f64": (num_g / den) ** 0.5, "global_rel_l2_oracle_f32_vs_f64": (num_o / den) ** 0.5,
               "worst_gpu_vs_f64": max(r["gpu_vs_f64"] for r in rows.values()),
               "worst_oracle_f32_vs_f64": max(r["oracle_f32_vs_f64"] for r in rows.values())}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fullwidth_gradients_%s.json" % tag)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        json.dump({"summary": summary, "tensors": rows}, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    bad = {k: r for k, r in rows.items() if r["gpu_vs_f64"] > 3 * r["oracle_f32_vs_f64"] + 1e-6}
    assert not bad, (summary, bad)
    assert summary["global_rel_l2_gpu_vs_f64"] <= 3 * summary["global_rel_l2_oracle_f32_vs_f64"] + 1e-6, summary
    return len(rows)


def _oracle_step(net_cpu, batch_cpu, cfg):
    from oracle import kpconv_ref
    with kpconv_ref.cpu_reference_mode():
        out = net_cpu(batch_cpu, cfg)
        loss = net_cpu.loss(out, batch_cpu.labels)
        loss.backward()
    return out, loss


@pytest.mark.timeout(3000)
def test_dales_full_width_network_vs_oracle(gpu):
    from weasal_amd import config as wcfg, ops, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    wl = synthetic.WORKLOADS["dales"]
    cfg = wcfg.DALESPLConfig()
    cfg.dropout = 0.0
    np.random.seed(3)
    torch.manual_seed(3)
    net = KPFCNN(cfg, np.arange(9), [])
    net_cpu = copy.deepcopy(net)
    net.to(gpu).train()
    net_cpu.train()
    pts, feats, labels, lens = synthetic.make_inputs(4242, 2, wl["points"], wl["radius"], cfg.in_features_dim)
    np.random.seed(9)
    batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu),
                                torch.from_numpy(labels).to(gpu), lens, wl["limits"])
    # the paths this test exists for are live
    assert ops.GRID_BACKWARD and len(batch.search_grids) >= 4            # K4G on the self-query levels
    widths = [m.shape[1] for m in batch.neighbors]
    assert widths[0] == 59 and max(widths) > 64                           # two-columns-per-lane rows
    assert batch.points[0].shape[0] > 65536 and batch.points[2].shape[0] < 4096    # tall products and the short, split-K ones
    out = net(batch, cfg)
    loss = net.loss(out, batch.labels)
    loss.backward()
    torch.cuda.synchronize()
    for _, grid in batch.search_grids:
        assert int(grid.overflow.item()) == 0

    batch_cpu = _cpu_copy(batch)
    out_c, loss_c = _oracle_step(net_cpu, batch_cpu, cfg)
    assert _rel(out, out_c) < 1e-4
    assert abs(loss.item() - loss_c.item()) < 1e-5 * abs(loss_c.item())
    ref_grads = {k: p.grad for k, p in net_cpu.named_parameters() if p.grad is not None}
    assert _check_grads(net, ref_grads, _f64_truth_grads(net_cpu, batch_cpu, cfg), "dales") >= 40


@pytest.mark.timeout(1500)
def test_vaihingen_real_widths_pyramid_and_step_vs_oracle(gpu):
    from oracle import pyramid_ref
    from weasal_amd import config as wcfg, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.trainer import make_optimizer, train_step
    wl = synthetic.WORKLOADS["vaihingen"]
    cfg = wcfg.Vaihingen3DPLConfig()
    cfg.dropout = 0.0
    pts, feats, labels, lens = synthetic.make_inputs(777, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
    # ---- pyramid: device vs CPU oracle, same np.random stream for the grid orientations
    np.random.seed(21)
    batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu),
                                torch.from_numpy(labels).to(gpu), lens, wl["limits"])
    np.random.seed(21)
    li = pyramid_ref.segmentation_inputs(cfg, pts, feats, labels, lens, wl["limits"])
    L = cfg.num_layers
    assert len(batch.points) == L == 5
    for l in range(L):
        p_l = li[l]
        assert np.array_equal(batch.points[l].cpu().numpy(), p_l), l
        assert np.array_equal(batch.lengths[l].cpu().numpy(), li[4 * L + l]), l
        assert_neighbors_equal(p_l, p_l, batch.neighbors[l].cpu().numpy(), li[L + l], False)
        if l < L - 1:
            nxt = li[l + 1]
            assert_neighbors_equal(nxt, p_l, batch.pools[l].cpu().numpy(), li[2 * L + l], False)
            assert_neighbors_equal(p_l, nxt, batch.upsamples[l].cpu().numpy(), li[3 * L + l], False)
    # ---- one SGD step at the real widths (64 -> 1024 channels)
    np.random.seed(5)
    torch.manual_seed(5)
    net = KPFCNN(cfg, np.arange(9), [])
    assert net.encoder_blocks[0].KPConv.out_channels == 32 and net.encoder_blocks[-1].out_dim == 1024
    net_cpu = copy.deepcopy(net)
    net.to(gpu).train()
    net_cpu.train()
    opt = make_optimizer(net, cfg)
    loss, out = train_step(net, opt, batch, cfg)
    torch.cuda.synchronize()

    from oracle import kpconv_ref
    opt_c = make_optimizer(net_cpu, cfg)
    batch_cpu = _cpu_copy(batch)
    net_before = copy.deepcopy(net_cpu)          # the parameters the gradients were taken at
    with kpconv_ref.cpu_reference_mode():
        loss_c, out_c = train_step(net_cpu, opt_c, batch_cpu, cfg)
    assert _rel(out, out_c) < 1e-4
    assert abs(loss.item() - loss_c.item()) < 1e-5 * abs(loss_c.item())
    # gradients: the stock step on the CPU clipped them in place (clip_grad_value_), the fused update on the GPU clips in
    # registers and leaves .grad as the backward produced it: clip the GPU side the same way, then compare
    for p in net.parameters():
        if p.grad is not None:
            p.grad.clamp_(-cfg.grad_clip_norm, cfg.grad_clip_norm)
    ref_grads = {k: p.grad for k, p in net_cpu.named_parameters() if p.grad is not None}
    assert _check_grads(net, ref_grads, _f64_truth_grads(net_before, batch_cpu, cfg, clip=cfg.grad_clip_norm), "vaihingen") >= 40
    ref = dict(net_cpu.named_parameters())
    for name, p in net.named_parameters():
        assert _rel(p, ref[name]) < 1e-4, name


def test_full_size_gather_is_adjoint_and_linear(gpu):
    """BASELINE config 3 at its FULL size (8 x 50 000-point spheres, limits 59/73/...): size-independent properties of the
    gather kernels, where the oracle is too slow to follow.  The KPConv weighted-feature map x -> wf is linear, and its
    backward must be its adjoint:  <wf(x), g> = <x, dx(g)>  for every level's self-query layer (K3 against K4G -- candidates
    from the supports' own rows and from the grid walk, truncated rows included) and for the strided layer of level 0
    (K3 against the table form K4).  float64 inner products; 2e-5 relative (fp32 sums in different orders)."""
    from weasal_amd import config as wcfg, ops, pyramid, synthetic
    from weasal_amd.kernel_points import load_kernels
    cfg = wcfg.DALESPLConfig()
    wl = synthetic.WORKLOADS["dales"]
    pts, feats, labels, lens = synthetic.make_inputs(77, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
    np.random.seed(3)
    batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu), torch.from_numpy(labels).to(gpu),
                                lens, wl["limits"])
    batch.activate()
    assert batch.points[0].shape[0] == 400000 and len(batch.search_grids) >= 4
    torch.manual_seed(5)
    checked = 0
    for lvl, ci in ((0, 32), (1, 64), (2, 128), (3, 256)):
        r = cfg.first_subsampling_dl * cfg.conv_radius * 2 ** lvl
        extent = r * cfg.KP_extent / cfg.conv_radius
        kp = torch.from_numpy(load_kernels(r, 15, dimension=3, fixed="center").astype(np.float32)).to(gpu)
        cases = [(batch.points[lvl], batch.points[lvl], batch.neighbors[lvl])]
        if lvl == 0:
            cases.append((batch.points[1], batch.points[0], batch.pools[0]))          # strided: queries of level 1, table form
        for q_pts, s_pts, inds in cases:
            x = torch.randn(s_pts.shape[0], ci, device=gpu, requires_grad=True)
            y = torch.randn(s_pts.shape[0], ci, device=gpu)
            wf, _ = ops.kpconv_gather(x, q_pts, s_pts, inds, kp, extent)
            g = torch.randn_like(wf)
            dx, = torch.autograd.grad(wf, x, g)
            lhs = float((wf.detach().double() * g.double()).sum())
            rhs = float((x.detach().double() * dx.double()).sum())
            # both sides are sums of n random-sign terms: their size, and the size of an fp32 error of 1e-6 per element, scale
            # with |wf| |g| / sqrt(n)
            scale = float(wf.detach().double().norm() * g.double().norm()) / wf.numel() ** 0.5
            assert abs(lhs - rhs) <= 2e-5 * scale, (lvl, lhs, rhs, scale)
            # linearity: wf(2 x - 3 y) = 2 wf(x) - 3 wf(y)
            wy, _ = ops.kpconv_gather(y, q_pts, s_pts, inds, kp, extent)
            wz, _ = ops.kpconv_gather(2.0 * x.detach() - 3.0 * y, q_pts, s_pts, inds, kp, extent)
            ref = 2.0 * wf.detach() - 3.0 * wy
            assert float((wz - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), lvl
            checked += 1
    assert checked == 5
    for _, grid in batch.search_grids:
        assert int(grid.overflow.item()) == 0


def test_full_size_pools_and_products_properties(gpu):
    """the same full-size batch: properties of the pooling kernels and the dense products that need no oracle --
    closest_pool is linear with its backward as adjoint; every max_pool output is attained by one of its row's sources
    and its backward routes each gradient entry to exactly that source (column sums preserved); the three products of a
    layer are mutually adjoint, <x W, g> = <W, x^T g> = <x, g W^T>; the loss gradient rows sum to zero."""
    from weasal_amd import config as wcfg, ops, pyramid, synthetic
    cfg = wcfg.DALESPLConfig()
    wl = synthetic.WORKLOADS["dales"]
    pts, feats, labels, lens = synthetic.make_inputs(78, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
    np.random.seed(4)
    batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu), torch.from_numpy(labels).to(gpu),
                                lens, wl["limits"])
    batch.activate()
    torch.manual_seed(6)
    n0, n1 = batch.points[0].shape[0], batch.points[1].shape[0]
    # ---- nearest upsampling (level 1 -> 0): linear, adjoint backward
    xc = torch.randn(n1, 128, device=gpu, requires_grad=True)
    up = ops.closest_pool(xc, batch.upsamples[0])
    g = torch.randn_like(up)
    dxc, = torch.autograd.grad(up, xc, g)
    lhs, rhs = float((up.detach().double() * g.double()).sum()), float((xc.detach().double() * dxc.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * float(up.detach().double().norm() * g.double().norm()) / up.numel() ** 0.5
    # ---- max pooling (level 0 -> 1)
    x = torch.randn(n0, 128, device=gpu, requires_grad=True)
    pooled = ops.max_pool(x, batch.pools[0])
    xs = torch.cat([x.detach(), torch.zeros(1, 128, device=gpu)])          # the shadow row (blocks.py:104)
    probe = torch.arange(0, n1, 97, device=gpu)                            # a sample of rows: every output is the max of its sources
    gathered = xs[batch.pools[0][probe]]
    assert torch.equal(pooled.detach()[probe], gathered.max(dim=1).values)
    gp = torch.randn_like(pooled)
    dx, = torch.autograd.grad(pooled, x, gp)
    routed = float(dx.double().sum())                                      # entries whose maximum is the shadow row are dropped
    hit_shadow = pooled.detach() == 0.0
    kept = float((gp.double() * (~hit_shadow)).sum())
    assert abs(routed - kept) <= 1e-6 * float(gp.double().abs().sum())
    # ---- the three products of a 400 000-row layer are mutually adjoint
    xw = torch.randn(n0, 128, device=gpu)
    w = torch.randn(128, 128, device=gpu) / 128 ** 0.5
    gy = torch.randn(n0, 128, device=gpu)
    y = ops._gemm_xb(xw, w)
    dw = ops._gemm_xty(ops._lib.lib(), xw, gy)
    dxw = ops._gemm_xb(gy, w.t().contiguous())
    a = float((y.double() * gy.double()).sum())
    b = float((w.double() * dw.double()).sum())
    c = float((xw.double() * dxw.double()).sum())
    scale = float(y.double().norm() * gy.double().norm()) / y.numel() ** 0.5
    assert abs(a - b) <= 2e-5 * scale and abs(a - c) <= 2e-5 * scale
    # ---- loss: rows of the gradient sum to zero (softmax - one-hot), ignored rows are zero
    logits = torch.randn(n0, 9, device=gpu, requires_grad=True)
    lab = torch.randint(-1, 9, (n0,), device=gpu)
    loss = ops.cross_entropy(logits, lab)
    dl, = torch.autograd.grad(loss, logits)
    assert float(loss.detach()) > 0 and float(dl.sum(dim=1).abs().max()) <= 1e-9
    assert float(dl[lab < 0].abs().sum()) == 0.0
