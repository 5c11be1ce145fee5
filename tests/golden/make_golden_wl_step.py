#!/usr/bin/env python3
"""tests/golden/make_golden_wl_step.py -- golden vector of ONE weak-label training step (BASELINE config 1 plumbing):
  g12_wl_step.npz   the reference's KPFCNN_mprm through the step body of utils/trainer_WeakLabel.py:199-216 --
                    SGD with the reference's two parameter groups (:80-87), zero_grad, three-output forward,
                    region_mprm_loss(cam, batch.region, batch.region_lb, batch.lengths[0]), backward,
                    clip_grad_norm_(net.parameters(), grad_clip_norm = 1), optimizer.step() -- on the golden batch and
                    initial state of g10_mprm.npz (inputs are read from there; nothing is duplicated).
RUNS ONLY IN THE BUILD CONTAINER, with the import aids of make_golden.py / make_golden_mprm.py (`.cuda()` made the
identity: the container has no GPU).  The step lines are torch's own calls in the trainer's order; the model, its
losses and its parameter naming ('offset' -> second group) are the reference's."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402
import numpy as np                # noqa: E402
import torch                      # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self

from make_golden_mprm import SmallWLConfig      # noqa: E402
from models.architectures import KPFCNN_mprm   # noqa: E402


def one_step(clip, arrs, prefix):
    g = np.load(os.path.join(HERE, "g10_mprm.npz"))
    cfg = SmallWLConfig()
    cfg.learning_rate = 0.01            # train_Vaihingen3D_WeakLabel.py:143-146
    cfg.momentum = 0.98
    cfg.weight_decay = 1e-3             # utils/config.py default
    cfg.grad_clip_norm = clip
    cfg.loss_type = 'region_mprm_loss'
    L = 3

    class Batch:
        pass
    b = Batch()
    b.points = [torch.from_numpy(g["points_%d" % l]) for l in range(L)]
    b.neighbors = [torch.from_numpy(g["neighbors_%d" % l]) for l in range(L)]
    b.pools = [torch.from_numpy(g["pools_%d" % l]) for l in range(L)]
    b.upsamples = [torch.from_numpy(g["upsamples_%d" % l]) for l in range(L)]
    b.lengths = [torch.from_numpy(g["lengths_%d" % l]) for l in range(L)]
    b.features = torch.from_numpy(g["features"])
    b.labels = torch.from_numpy(g["labels"])
    b.center_pts = torch.from_numpy(g["center_pts"])
    sizes, flat = g["region_sizes"], g["regions_flat"]
    b.region = [[flat[:sizes[0]], flat[sizes[0]:sizes[0] + sizes[1]]], []]
    b.region_lb = [[g["regions_lb"][0], g["regions_lb"][1]], []]

    np.random.seed(0)
    torch.manual_seed(0)
    net = KPFCNN_mprm(cfg, np.arange(6), [])
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd0/")}
    net.load_state_dict(sd, strict=False)
    net.train()
    # utils/trainer_WeakLabel.py:80-87
    deform_params = [v for k, v in net.named_parameters() if 'offset' in k]
    other_params = [v for k, v in net.named_parameters() if 'offset' not in k]
    optimizer = torch.optim.SGD([{'params': other_params}, {'params': deform_params, 'lr': cfg.learning_rate * cfg.deform_lr_factor}],
                                lr=cfg.learning_rate, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
    # :181-184, :199-216
    assert any(b.region)
    optimizer.zero_grad()
    logits, class_logits, cam = net(b, cfg)
    loss = net.region_mprm_loss(cam, b.region, b.region_lb, b.lengths[0])
    acc = net.accuracy(logits, b.labels)
    loss.backward()
    total_norm = torch.nn.utils.clip_grad_norm_(net.parameters(), cfg.grad_clip_norm)
    optimizer.step()

    arrs.update({prefix + "loss": np.float32(loss.item()), prefix + "acc": np.float32(acc),
                 prefix + "total_norm": np.float32(float(total_norm)), prefix + "clip": np.float32(clip),
                 "lr": np.float32(cfg.learning_rate), "momentum": np.float32(cfg.momentum),
                 "weight_decay": np.float32(cfg.weight_decay)})
    names = sorted(k for k, v in net.named_parameters() if v.grad is not None)
    arrs["stepped_names"] = np.array(names)
    params = dict(net.named_parameters())
    arrs[prefix + "delta_norms"] = np.array([float((params[k].detach().double() - sd[k].double()).norm()) for k in names])
    for k in names:
        if params[k].numel() <= 4096:
            arrs[prefix + "after/" + k] = params[k].detach().numpy()
    print("clip", clip, "loss", float(loss), "acc", acc, "total grad norm", float(total_norm), "stepped", len(names))


def main():
    arrs = {}
    one_step(1.0, arrs, "c1/")          # the configuration's value (train_Vaihingen3D_WeakLabel.py:146): not reached by this batch
    one_step(0.02, arrs, "c002/")       # a bound below the gradient's norm: the scaling branch of clip_grad_norm_
    mg.save("g12_wl_step.npz", **arrs)


if __name__ == "__main__":
    assert mg.geom.have_ref(), "run `make -C oracle ref` first"
    main()
