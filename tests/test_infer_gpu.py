"""GPU: the forward-only path (the testers' passes, utils/tester_PseudoLabel.py:164: `net(batch, config)` under no_grad).
32 -> 32 rigid layers then run as ONE launch -- gather and kernel contraction in the same kernel, `wf` never stored
(ws_kpconv_layer_fwd_fused) --; everything else as in training.  Pinned against the two-launch form on the same inputs
(same gather arithmetic; the 480-deep contraction is summed as four quarters instead of one chain: fp32 re-association)
and, through the network, against golden g8 (the reference's own KPFCNN logits) and the full-width training-mode forward."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("n,nq,h,act,with_bias", [(5003, 5003, 41, 1, True), (9000, 2207, 59, 0, False), (700, 13, 70, 1, False),
                                                  (64, 64, 5, 1, True)])
def test_fused_forward_layer_vs_two_launches(gpu, n, nq, h, act, with_bias):
    from weasal_amd import _lib, ops
    from weasal_amd._lib import check, current_stream, ptr
    from weasal_amd.kernel_points import load_kernels
    lib = _lib.lib()
    rng = np.random.default_rng(n + h)
    s_pts = torch.from_numpy(rng.uniform(-2, 2, size=(n, 3)).astype(np.float32)).to(gpu)
    q_pts = s_pts if nq == n else s_pts[rng.choice(n, nq, replace=False)].contiguous()
    inds = torch.from_numpy(rng.integers(0, n + 1, size=(nq, h))).to(gpu)          # shadow indices (n) included
    d = (s_pts[inds.clamp(max=n - 1)] - q_pts[:, None, :]).norm(dim=2)
    inds = torch.where(d < 1.1, inds, torch.full_like(inds, n))                     # far "neighbours" -> shadow: a realistic density of zeros
    x = torch.randn(n, 32, device=gpu)
    kp = torch.from_numpy(load_kernels(1.0, 15, dimension=3, fixed="center").astype(np.float32)).to(gpu)
    w = torch.randn(15, 32, 32, device=gpu) / 22.0
    bias = torch.randn(32, device=gpu) if with_bias else None
    order = torch.randperm(nq, device=gpu).to(torch.int32)
    out = torch.full((nq, 32), float("nan"), device=gpu)
    check(lib.ws_kpconv_layer_fwd_fused(ptr(q_pts), nq, ptr(s_pts), n, ptr(inds), h, ptr(x), 32, ptr(kp), 15, 0.4, ptr(order), ptr(w), 32,
                                        ptr(bias), act, 0.1, ptr(out), current_stream()))
    wf, _ = ops.kpconv_gather(x, q_pts, s_pts, inds, kp, 0.4)
    ref = ops.matmul_epilogue(wf.reshape(nq, -1), w.reshape(480, 32), bias=bias, slope=0.1 if act else None)
    assert bool(torch.isfinite(out).all())
    assert rel(out, ref) < 2e-6
    # shapes the one-launch kernel does not cover are refused, not mangled
    x64 = torch.randn(n, 64, device=gpu)
    rc = lib.ws_kpconv_layer_fwd_fused(ptr(q_pts), nq, ptr(s_pts), n, ptr(inds), h, ptr(x64), 64, ptr(kp), 15, 0.4, None, ptr(w), 32,
                                       None, 0, 0.0, ptr(out), current_stream())
    assert rc != 0 and b"32 -> 32" in lib.ws_last_error()


def test_network_forward_without_grad_takes_the_one_launch_layers(gpu):
    """DALES KP-FCNN at full width, 2 x 20 000 points: logits under no_grad (one-launch level-0 layers, no activations kept)
    against the same network's training-style forward; the block calls are told `infer` exactly when grad mode is off"""
    from weasal_amd import config as wcfg, fused, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    cfg = wcfg.DALESPLConfig()
    cfg.dropout = 0.0
    np.random.seed(2)
    torch.manual_seed(2)
    net = KPFCNN(cfg, np.arange(9), []).to(gpu).eval()
    pts, feats, labels, lens = synthetic.make_inputs(99, 2, 20000, 6.5, cfg.in_features_dim)
    np.random.seed(1)
    batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu), torch.from_numpy(labels).to(gpu),
                                lens, synthetic.WORKLOADS["dales"]["limits"])
    seen = []
    orig = fused._KPBlockFn.apply

    def spy(*a):
        seen.append((a[-1].infer, a[-1].conv_in, a[-1].conv_out))
        return orig(*a)
    fused._KPBlockFn.apply = spy
    try:
        ref = net(batch, cfg)
        n_train = len(seen)
        with torch.no_grad():
            out = net(batch, cfg)
    finally:
        fused._KPBlockFn.apply = orig
    assert all(not s[0] for s in seen[:n_train]) and all(s[0] for s in seen[n_train:])
    assert sum(1 for s in seen[n_train:] if s[1:] == (32, 32)) == 2            # enc1 and enc2: the level-0 32 -> 32 layers
    assert rel(out, ref) < 1e-5


def test_golden_network_logits_without_grad(gpu):
    """golden g8 (the reference's own KPFCNN forward on the g7 pyramid): the same logits from the forward-only path"""
    from test_pyramid_gpu import _cfg
    from weasal_amd import pyramid
    from weasal_amd.architectures import KPFCNN
    g8, g7 = golden("g8_kpfcnn.npz"), golden("g7_pyramid.npz")
    cfg = _cfg()
    np.random.seed(0)
    net = KPFCNN(cfg, np.arange(9), [])
    net.load_state_dict({k[4:]: torch.from_numpy(g8[k]) for k in g8.files if k.startswith("sd0/")}, strict=False)
    net.to(gpu).eval()
    L = 5
    flat = ([g7["points_%d" % l] for l in range(L)] + [g7["neighbors_%d" % l] for l in range(L)]
            + [g7["pools_%d" % l] for l in range(L)] + [g7["upsamples_%d" % l] for l in range(L)]
            + [g7["lengths_%d" % l] for l in range(L)] + [g7["features"], g7["labels"]])
    batch = pyramid.PyramidBatch([torch.from_numpy(np.ascontiguousarray(a)) for a in flat]).to(gpu)
    with torch.no_grad():
        out = net(batch, cfg)
    assert rel(out, torch.from_numpy(g8["logits"])) < 1e-4
