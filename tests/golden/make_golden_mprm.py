#!/usr/bin/env python3
"""tests/golden/make_golden_mprm.py -- golden vectors of the weak-label network:
  g10_mprm.npz   KPFCNN_mprm forward (x, class logits, class-activation maps), class_logits_loss,
                 region_mprm_loss, gradients           (models/architectures.py:507-807, models/blocks.py:758-1011)

RUNS ONLY IN THE BUILD CONTAINER, with the import aids of make_golden.py (imported as a module: its generators are
behind a __main__ guard).  One more aid, local to this script: the reference's attention blocks and losses call
`.cuda()` on fresh tensors (blocks.py:796,863,989; architectures.py:770,779); the build container has no GPU, so
`torch.Tensor.cuda` is made the identity for the duration of this script -- the arithmetic is untouched, every
tensor simply stays on the CPU."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402  (registers the import aids, chdir's to the reference)
import numpy as np                # noqa: E402
import torch                      # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self

from models.architectures import KPFCNN_mprm   # noqa: E402
from datasets.common import PointCloudDataset  # noqa: E402
from utils.config import Config                # noqa: E402


class SmallWLConfig(Config):
    """DALES weak-label architecture (train_DALES_WeakLabel.py:54-61) at reduced width"""
    dataset = "GoldenWL"
    num_classes = 6
    architecture = ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
                    'nearest_upsample', 'nearest_upsample']
    num_kernel_points = 15
    first_subsampling_dl = 0.3
    conv_radius = 2.5
    deform_radius = 1.0
    KP_extent = 1.0
    KP_influence = 'linear'
    aggregation_mode = 'sum'
    first_features_dim = 16
    in_features_dim = 4
    modulated = False
    use_batch_norm = True
    batch_norm_momentum = 0.02
    deform_fitting_mode = 'point2point'
    deform_fitting_power = 1.0
    deform_lr_factor = 0.1
    repulse_extent = 1.2
    class_w = []
    saving = False


def main():
    cfg = SmallWLConfig()
    rng = np.random.default_rng(77)
    pts = np.concatenate([mg.sphere(rng, 1400, 3.0), mg.sphere(rng, 1100, 3.0)])
    lens = np.array([1400, 1100], np.int32)
    feats = np.concatenate([np.ones((2500, 1), np.float32), rng.random((2500, 1)).astype(np.float32),
                            pts[:, 2:], pts[:, 2:]], axis=1).astype(np.float32)
    labels = rng.integers(0, 6, size=2500).astype(np.int64)
    dsobj = PointCloudDataset("goldenwl")
    dsobj.config = cfg
    limits = [22, 26, 28]
    dsobj.neighborhood_limits = limits
    np.random.seed(4321)
    li = dsobj.segmentation_inputs(pts, feats, labels, lens)
    L = 3

    class Batch:
        pass
    b = Batch()
    b.points = [torch.from_numpy(a) for a in li[0:L]]
    b.neighbors = [torch.from_numpy(a) for a in li[L:2 * L]]
    b.pools = [torch.from_numpy(a) for a in li[2 * L:3 * L]]
    b.upsamples = [torch.from_numpy(a) for a in li[3 * L:4 * L]]
    b.lengths = [torch.from_numpy(a) for a in li[4 * L:5 * L]]
    b.features = torch.from_numpy(li[5 * L])
    b.labels = torch.from_numpy(li[5 * L + 1])
    center_pts = np.array([[1.0, -2.0, 37.5], [4.0, 0.5, 12.25]], np.float32)
    b.center_pts = torch.from_numpy(center_pts)

    np.random.seed(55)
    torch.manual_seed(55)
    net = KPFCNN_mprm(cfg, np.arange(6), [])
    net.train()
    # the attention gammas start at 0 (blocks.py:784,850,982): give them values so that the attention paths matter
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("gamma"):
                p.fill_(0.37)
    sd0 = {k: v.detach().clone().numpy() for k, v in net.state_dict().items() if "num_batches_tracked" not in k}

    x, cla_logits, cam = net(b, cfg)
    cloud_lb = torch.from_numpy((rng.random((2, 6)) > 0.5).astype(np.float32))
    loss_cls = net.class_logits_loss(cla_logits, cloud_lb)
    # overlap regions: index sets per input sphere (DALES_WeakLabel.py regions), one sphere without regions
    regions_all = [[np.sort(rng.choice(1400, size=300, replace=False)), np.sort(rng.choice(1400, size=150, replace=False))], []]
    regions_lb = [[(rng.random(6) > 0.5).astype(np.float32), (rng.random(6) > 0.5).astype(np.float32)], []]
    loss_reg = net.region_mprm_loss(cam, regions_all, regions_lb, b.lengths[0])
    acc = net.accuracy(x, b.labels)
    loss = loss_cls + loss_reg
    loss.backward()
    grads = {k: v.grad.detach().clone().numpy() for k, v in net.named_parameters() if v.grad is not None}

    arrs = dict(points=pts, lens=lens, features=feats, labels=labels, limits=np.array(limits, np.int32),
                np_seed=np.int64(4321), center_pts=center_pts, cloud_lb=cloud_lb.numpy(),
                x=x.detach().numpy(), loss_cls=np.float32(loss_cls.item()), loss_reg=np.float32(loss_reg.item()),
                acc=np.float32(acc), region_sizes=np.array([len(r) for r in regions_all[0]], np.int64),
                regions_flat=np.concatenate(regions_all[0]).astype(np.int64), regions_lb=np.stack(regions_lb[0]))
    for i in range(4):
        arrs["cla_logits_%d" % i] = cla_logits[i].detach().numpy()
        arrs["cam_%d" % i] = cam[i].detach().numpy()
    for l in range(L):
        arrs["points_%d" % l] = li[l]
        arrs["neighbors_%d" % l] = li[L + l]
        arrs["pools_%d" % l] = li[2 * L + l]
        arrs["upsamples_%d" % l] = li[3 * L + l]
        arrs["lengths_%d" % l] = li[4 * L + l]
    for k, v in sd0.items():
        arrs["sd0/" + k] = v
    names = sorted(grads.keys())
    arrs["grad_names"] = np.array(names)
    arrs["grad_norms"] = np.array([np.linalg.norm(grads[k].astype(np.float64)) for k in names])
    for k in names:
        if grads[k].size <= 4096:
            arrs["grad/" + k] = grads[k]
    mg.save("g10_mprm.npz", **arrs)
    print("params", len(sd0), "grads", len(names), "loss", float(loss_cls), float(loss_reg), "acc", acc)


if __name__ == "__main__":
    assert mg.geom.have_ref(), "run `make -C oracle ref` first"
    main()
