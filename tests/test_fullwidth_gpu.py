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


def _check_grads(net, ref_grads, pert_grads):
    """ref_grads: the oracle's gradients (as the GPU's were taken: clipped or not); pert_grads: the oracle's gradients
    under a 1e-6 relative perturbation of the input features (same treatment)"""
    gpu = {}
    for name, p in net.named_parameters():
        assert (p.grad is None) == (name not in ref_grads), name
        if p.grad is not None:
            gpu[name] = p.grad
    glob, errs = _grad_vector_error(gpu, ref_grads)
    glob_p, errs_p = _grad_vector_error(pert_grads, ref_grads)
    worst = max(errs, key=errs.get)
    print("gradient check: global rel-L2 gpu %.2e, oracle under a 1e-6 input perturbation %.2e; worst tensor %s %.2e (oracle: %.2e)"
          % (glob, glob_p, worst, errs[worst], errs_p[worst]))
    assert glob <= 3 * glob_p + 1e-4, (glob, glob_p)
    assert max(errs.values()) < 5e-2, (worst, errs[worst])
    return len(errs)


def _oracle_step(net_cpu, batch_cpu, cfg):
    from oracle import kpconv_ref
    with kpconv_ref.cpu_reference_mode():
        out = net_cpu(batch_cpu, cfg)
        loss = net_cpu.loss(out, batch_cpu.labels)
        loss.backward()
    return out, loss


@pytest.mark.timeout(1500)
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
    assert _check_grads(net, ref_grads, _perturbed_oracle_grads(net_cpu, batch_cpu, cfg)) >= 40


@pytest.mark.timeout(900)
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
    pert = {k: g.clamp(-cfg.grad_clip_norm, cfg.grad_clip_norm) for k, g in _perturbed_oracle_grads(net_before, batch_cpu, cfg).items()}
    assert _check_grads(net, ref_grads, pert) >= 40
    ref = dict(net_cpu.named_parameters())
    for name, p in net.named_parameters():
        assert _rel(p, ref[name]) < 1e-4, name
