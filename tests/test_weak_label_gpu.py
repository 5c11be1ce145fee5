"""GPU: the weak-label training step (BASELINE config 1: Vaihingen3D_WeakLabel KPFCNN_mprm, in_radius = 4 m, batch = 2).

  * weasal_amd.trainer.train_step_weak against golden g12_wl_step.npz: ONE step of the reference's own KPFCNN_mprm through
    the step body of utils/trainer_WeakLabel.py:199-216 (three-output forward, region_mprm_loss, clip_grad_NORM_, SGD with
    the 'offset' parameter group) on the g10 batch and initial state, with the configuration's clip (1: not reached) and
    with a clip below the gradient norm (the scaling branch).  Loss 1e-4, total gradient norm 1e-3, parameters after the
    step 1e-4 of their own maximum, the parameter DELTA of every tensor 1e-3 (that is the gradient, through the update).
  * config 1 at its real shape -- Vaihingen3DWLConfig (3 layers, 64 first features, dl0 0.24), 2 spheres of 3 000 points,
    R = 4 m, synthetic region labels -- pyramid + one step on the GPU against the same classes evaluated by the CPU oracle
    (oracle.kpconv_ref.cpu_reference_mode; the class itself is pinned by g10 / g12): logits, class logits, CAMs 1e-4,
    loss 1e-5, gradient norm 1e-3, parameters after the step 1e-4.
"""
import copy

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("prefix,clip", [("c1/", 1.0), ("c002/", 0.02)])
def test_weak_label_step_vs_golden(gpu, prefix, clip):
    from test_pyramid_gpu import golden as _g  # noqa: F401  (same fixture directory)
    from weasal_amd import config as wcfg
    from weasal_amd.architectures import KPFCNN_mprm
    from weasal_amd.pyramid import PyramidBatch
    from weasal_amd.trainer import make_optimizer, train_step_weak
    g, s = golden("g10_mprm.npz"), golden("g12_wl_step.npz")

    class Cfg(wcfg.Vaihingen3DWLConfig):
        dataset = "GoldenWL"
        num_classes = 6
        first_subsampling_dl = 0.3
        first_features_dim = 16
        class_w = []
        weight_decay = 1e-3
    cfg = Cfg()
    cfg.grad_clip_norm = clip
    assert float(s["lr"]) == np.float32(cfg.learning_rate) and float(s["momentum"]) == np.float32(cfg.momentum)
    L = 3
    li = [torch.from_numpy(g["points_%d" % l]).to(gpu) for l in range(L)]
    li += [torch.from_numpy(g["neighbors_%d" % l].astype(np.int64)).to(gpu) for l in range(L)]
    li += [torch.from_numpy(g["pools_%d" % l].astype(np.int64)).to(gpu) for l in range(L)]
    li += [torch.from_numpy(g["upsamples_%d" % l].astype(np.int64)).to(gpu) for l in range(L)]
    li += [torch.from_numpy(g["lengths_%d" % l].astype(np.int32)).to(gpu) for l in range(L)]
    li += [torch.from_numpy(g["features"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)]
    batch = PyramidBatch(li)
    batch.center_pts = torch.from_numpy(g["center_pts"]).to(gpu)
    sizes, flat = g["region_sizes"], g["regions_flat"]
    batch.region = [[flat[:sizes[0]], flat[sizes[0]:sizes[0] + sizes[1]]], []]
    batch.region_lb = [[g["regions_lb"][0], g["regions_lb"][1]], []]
    np.random.seed(0)
    torch.manual_seed(0)
    net = KPFCNN_mprm(cfg, np.arange(6), []).to(gpu).train()
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd0/")}
    net.load_state_dict(sd, strict=False)
    opt = make_optimizer(net, cfg)
    loss, (logits, cla, cam) = train_step_weak(net, opt, batch, cfg)
    assert abs(float(loss) - float(s[prefix + "loss"])) <= 1e-4 * abs(float(s[prefix + "loss"]))
    assert abs(float(net.grad_norm) - float(s[prefix + "total_norm"])) <= 1e-3 * float(s[prefix + "total_norm"])
    assert abs(net.accuracy(logits, batch.labels) - float(s[prefix + "acc"])) < 1e-6
    names = [str(n) for n in s["stepped_names"]]
    params = dict(net.named_parameters())
    assert sorted(k for k, v in params.items() if v.grad is not None) == names
    for n, dref in zip(names, s[prefix + "delta_norms"]):
        delta = float((params[n].detach().double().cpu() - sd[n].double()).norm())
        assert abs(delta - float(dref)) <= 1e-3 * max(float(dref), 1e-12), (n, delta, float(dref))
        key = prefix + "after/" + n
        if key in s.files:
            assert _rel(params[n], torch.from_numpy(s[key])) < 1e-4, n
    # a batch without any sub-region label is skipped (trainer_WeakLabel.py:181-184)
    batch.region = [[], []]
    assert train_step_weak(net, opt, batch, cfg) == (None, None)


@pytest.mark.timeout(900)
def test_config1_real_shape_step_vs_oracle(gpu):
    from oracle import kpconv_ref
    from test_fullwidth_gpu import _cpu_copy
    from weasal_amd import config as wcfg, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN_mprm
    from weasal_amd.trainer import make_optimizer, train_step_weak
    wl = synthetic.WORKLOADS["vaihingen_wl"]
    cfg = wcfg.Vaihingen3DWLConfig()
    assert cfg.num_layers == 3 and cfg.first_features_dim == 64 and cfg.grad_clip_norm == 1
    pts, feats, labels, lens = synthetic.make_inputs(5150, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
    region, region_lb, cloud_lb, centers = synthetic.make_weak_labels(7, pts, labels, lens, num_classes=cfg.num_classes)
    assert all(len(r) > 0 for r in region)
    np.random.seed(31)
    batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu),
                                torch.from_numpy(labels).to(gpu), lens, wl["limits"])
    batch.region, batch.region_lb = region, region_lb
    batch.cloud_lb = torch.from_numpy(cloud_lb).to(gpu)
    batch.center_pts = torch.from_numpy(centers).to(gpu)
    assert batch.points[0].shape[0] == 6000 and len(batch.points) == 3
    np.random.seed(5)
    torch.manual_seed(5)
    net = KPFCNN_mprm(cfg, np.arange(cfg.num_classes), [])
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("gamma"):                # the attention paths start switched off (gamma = 0, blocks.py:784,850,982)
                p.fill_(0.37)
    net_cpu = copy.deepcopy(net)
    net.to(gpu).train()
    net_cpu.train()
    assert sum(p.numel() for p in net.parameters() if p.requires_grad) > 5_000_000      # 5.45 M parameters
    opt = make_optimizer(net, cfg)
    loss, (logits, cla, cam) = train_step_weak(net, opt, batch, cfg)
    torch.cuda.synchronize()
    # ---- the same classes on the CPU oracle
    batch_cpu = _cpu_copy(batch)
    batch_cpu.region, batch_cpu.region_lb = region, region_lb
    batch_cpu.cloud_lb = torch.from_numpy(cloud_lb)
    batch_cpu.center_pts = torch.from_numpy(centers)
    opt_c = make_optimizer(net_cpu, cfg)
    with kpconv_ref.cpu_reference_mode():
        loss_c, (logits_c, cla_c, cam_c) = train_step_weak(net_cpu, opt_c, batch_cpu, cfg)
    assert _rel(logits, logits_c) < 1e-4
    for a, b in zip(cla, cla_c):
        assert _rel(a, b) < 1e-4
    for a, b in zip(cam, cam_c):
        assert _rel(a, b) < 1e-4
    assert abs(float(loss) - float(loss_c)) <= 1e-5 * abs(float(loss_c))
    assert abs(float(net.grad_norm) - float(net_cpu.grad_norm)) <= 1e-3 * float(net_cpu.grad_norm)
    ref = dict(net_cpu.named_parameters())
    for name, p in net.named_parameters():
        assert _rel(p, ref[name]) < 1e-4, name
