"""CPU: the plain-torch KPConv restatement (oracle/kpconv_ref.py) and the module classes of
weasal_amd evaluated through it, against the golden vectors generated from the reference
(g4 rigid, g5 deformable, g6 pools, g8 network, g9 kernel points)."""
import types

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import kpconv_ref

TOL = 1e-5   # same op sequence on the same CPU: only bmm blocking may differ


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def t(a):
    return torch.from_numpy(np.asarray(a))


CASES = [("g4_kpconv_3_64.npz", "linear", "sum"), ("g4_kpconv_32_32.npz", "linear", "sum"),
         ("g4_kpconv_32_32_strided.npz", "linear", "sum"), ("g4_kpconv_64_64.npz", "linear", "sum"),
         ("g4_kpconv_16_16_gaussian.npz", "gaussian", "sum"), ("g4_kpconv_16_16_constant.npz", "constant", "sum"),
         ("g4_kpconv_16_16_closest.npz", "linear", "closest")]


@pytest.mark.parametrize("name,infl,aggr", CASES)
def test_rigid_restatement(name, infl, aggr):
    g = golden(name)
    x = t(g["x"]).requires_grad_(True)
    w = t(g["weights"]).requires_grad_(True)
    out = kpconv_ref.kpconv_forward(t(g["q_pts"]), t(g["s_pts"]), t(g["inds"]), x, w, t(g["kernel_points"]),
                                    float(g["KP_extent"]), infl, aggr)
    assert rel(out.detach().numpy(), g["out"]) < TOL
    (out * t(g["dy"])).sum().backward()
    assert rel(x.grad.numpy(), g["grad_x"]) < TOL
    assert rel(w.grad.numpy(), g["grad_weights"]) < TOL


@pytest.mark.parametrize("name,modulated", [("g5_kpconv_deform_16_16.npz", False),
                                            ("g5_kpconv_deform_mod_16_32.npz", True),
                                            ("g5_kpconv_deform_strided_16_16.npz", False)])
def test_deformable_module_through_restatement(name, modulated):
    from weasal_amd.architectures import p2p_fitting_regularizer
    from weasal_amd.blocks import KPConv
    g = golden(name)
    ci, co = g["x"].shape[1], g["out"].shape[1]
    np.random.seed(0)
    conv = KPConv(15, 3, ci, co, float(g["KP_extent"]), float(g["radius"]), deformable=True, modulated=modulated)
    with torch.no_grad():
        conv.weights.copy_(t(g["weights"]))
        conv.kernel_points.copy_(t(g["kernel_points"]))
        conv.offset_conv.weights.copy_(t(g["offset_weights"]))
        conv.offset_conv.kernel_points.copy_(t(g["offset_kernel_points"]))
        conv.offset_bias.copy_(t(g["offset_bias"]))
    x = t(g["x"]).requires_grad_(True)
    with kpconv_ref.cpu_reference_mode():
        out = conv(t(g["q_pts"]), t(g["s_pts"]), t(g["inds"]), x)
    assert rel(out.detach().numpy(), g["out"]) < TOL
    assert rel(conv.min_d2.detach().numpy(), g["min_d2"]) < TOL
    assert rel(conv.deformed_KP.detach().numpy(), g["deformed_KP"]) < TOL
    net = types.SimpleNamespace(modules=lambda: [conv], l1=torch.nn.L1Loss(), K=15, repulse_extent=1.2,
                                deform_fitting_power=1.0)
    reg = p2p_fitting_regularizer(net)
    assert abs(float(reg) - float(g["reg_loss"])) <= TOL * abs(float(g["reg_loss"]))
    ((out * t(g["dy"])).sum() + reg).backward()
    assert rel(x.grad.numpy(), g["grad_x"]) < 10 * TOL
    assert rel(conv.offset_conv.weights.grad.numpy(), g["grad_offset_weights"]) < 10 * TOL
    assert rel(conv.offset_bias.grad.numpy(), g["grad_offset_bias"]) < 10 * TOL
    # state_dict keys of the reference
    assert set(conv.state_dict().keys()) == {"weights", "kernel_points", "offset_bias", "offset_conv.weights",
                                             "offset_conv.kernel_points"}


def test_pools_restatement():
    g = golden("g6_pools.npz")
    assert np.array_equal(kpconv_ref.max_pool_ref(t(g["x"]), t(g["inds"])).numpy(), g["max_pool"])
    assert np.array_equal(kpconv_ref.closest_pool_ref(t(g["xc"]), t(g["up"])).numpy(), g["closest_pool"])


def test_kernel_points_table_and_replay():
    from weasal_amd import kernel_points as kpm
    g = golden("g9_kernel_points.npz")
    assert np.array_equal(kpm._disposition(15, 3, "center"), g["kernel_points"])
    r = golden("g9_load_kernels.npz")
    np.random.seed(int(r["seed"]))
    assert np.array_equal(kpm.load_kernels(float(r["radius"]), 15, 3, "center"), r["kernel_points"])
    with pytest.raises(NotImplementedError):
        kpm.load_kernels(1.0, 20, 3, "center")


def _small_config():
    from weasal_amd.config import Config

    class Cfg(Config):
        architecture = ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
                        'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
                        'nearest_upsample', 'unary', 'nearest_upsample', 'unary',
                        'nearest_upsample', 'unary', 'nearest_upsample', 'unary']
        first_subsampling_dl = 0.24
        deform_radius = 6.0
        first_features_dim = 16
        in_features_dim = 4
        batch_norm_momentum = 0.02
        repulse_extent = 1.2
        learning_rate = 0.01
        momentum = 0.98
        dropout = 0
        saving = False
    return Cfg()


def test_kpfcnn_vs_golden_cpu():
    """G8: logits, loss, grads and one SGD step of the reference network, reproduced by
    weasal_amd.architectures.KPFCNN (CPU, through the restatement)."""
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.trainer import make_optimizer
    g8, g7 = golden("g8_kpfcnn.npz"), golden("g7_pyramid.npz")
    cfg = _small_config()
    np.random.seed(0)
    net = KPFCNN(cfg, np.arange(9), [])
    sd = {k[4:]: t(g8[k]) for k in g8.files if k.startswith("sd0/")}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all("num_batches_tracked" in m for m in missing), (missing, unexpected)
    net.train()
    b = types.SimpleNamespace()
    b.points = [t(g7["points_%d" % l]) for l in range(5)]
    b.neighbors = [t(g7["neighbors_%d" % l]) for l in range(5)]
    b.pools = [t(g7["pools_%d" % l]) for l in range(5)]
    b.upsamples = [t(g7["upsamples_%d" % l]) for l in range(5)]
    b.lengths = [t(g7["lengths_%d" % l]) for l in range(5)]
    b.features, b.labels = t(g7["features"]), t(g7["labels"])
    opt = make_optimizer(net, cfg)
    opt.zero_grad()
    with kpconv_ref.cpu_reference_mode():
        out = net(b, cfg)
        loss = net.loss(out, b.labels)
        loss.backward()
    assert rel(out.detach().numpy(), g8["logits"]) < 1e-4
    assert abs(loss.item() - float(g8["loss"])) < 1e-5
    assert abs(net.accuracy(out, b.labels) - float(g8["acc"])) < 1e-6
    grads = dict(net.named_parameters())
    for k in g8.files:
        if k.startswith("grad/"):
            assert rel(grads[k[5:]].grad.numpy(), g8[k]) < 1e-3, k
    torch.nn.utils.clip_grad_value_(net.parameters(), cfg.grad_clip_norm)
    opt.step()
    sd1 = net.state_dict()
    for k in g8.files:
        if k.startswith("sd1/"):
            assert rel(sd1[k[4:]].numpy(), g8[k]) < 1e-4, k


def test_kpfcnn_mprm_vs_golden_cpu():
    """G10: the weak-label network (attention blocks, class logits, CAMs, both losses, gradient norms) of the
    reference, reproduced by weasal_amd.architectures.KPFCNN_mprm on the CPU through the restatement -- the host
    logic and module wiring, without the HIP operators (those are checked by the GPU twin of this test)."""
    from weasal_amd.architectures import KPFCNN_mprm
    from weasal_amd.config import Config
    g = golden("g10_mprm.npz")

    class Cfg(Config):
        num_classes = 6
        architecture = ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
                        'nearest_upsample', 'nearest_upsample']
        first_subsampling_dl = 0.3
        deform_radius = 1.0
        first_features_dim = 16
        in_features_dim = 4
        batch_norm_momentum = 0.02
        repulse_extent = 1.2
        saving = False
    cfg = Cfg()
    np.random.seed(0)
    net = KPFCNN_mprm(cfg, np.arange(6), [])
    sd = {k[4:]: t(g[k]) for k in g.files if k.startswith("sd0/")}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all("num_batches_tracked" in m for m in missing), (missing, unexpected)
    net.train()
    b = types.SimpleNamespace()
    b.points = [t(g["points_%d" % l]) for l in range(3)]
    b.neighbors = [t(g["neighbors_%d" % l]) for l in range(3)]
    b.pools = [t(g["pools_%d" % l]) for l in range(3)]
    b.upsamples = [t(g["upsamples_%d" % l]) for l in range(3)]
    b.lengths = [t(g["lengths_%d" % l]) for l in range(3)]
    b.features, b.labels, b.center_pts = t(g["features"]), t(g["labels"]), t(g["center_pts"])
    with kpconv_ref.cpu_reference_mode():
        x, cla, cam = net(b, cfg)
        loss_cls = net.class_logits_loss(cla, t(g["cloud_lb"]))
        sizes, flat = g["region_sizes"], g["regions_flat"]
        regions = [[flat[:sizes[0]], flat[sizes[0]:sizes[0] + sizes[1]]], []]
        loss_reg = net.region_mprm_loss(cam, regions, [[g["regions_lb"][0], g["regions_lb"][1]], []], b.lengths[0])
        (loss_cls + loss_reg).backward()
    assert rel(x.detach().numpy(), g["x"]) < 1e-4
    for i in range(4):
        assert rel(cla[i].detach().numpy(), g["cla_logits_%d" % i]) < 1e-4
        assert rel(cam[i].detach().numpy(), g["cam_%d" % i]) < 1e-4
    assert abs(loss_cls.item() - float(g["loss_cls"])) < 1e-5 and abs(loss_reg.item() - float(g["loss_reg"])) < 1e-5
    grads = {k: v.grad for k, v in net.named_parameters() if v.grad is not None}
    assert sorted(grads) == [str(n) for n in g["grad_names"]]
    for n, ref_norm in zip(g["grad_names"], g["grad_norms"]):
        assert abs(float(grads[str(n)].double().norm()) - float(ref_norm)) <= 1e-3 * max(float(ref_norm), 1e-8), n
