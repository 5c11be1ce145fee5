"""Diagnostic: three Vaihingen training steps twice in one process: are the final weights bit-identical?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import config as wcfg, pyramid, synthetic
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer, train_step
dev = torch.device("cuda:0")
def run(nsteps=3):
    cfg = wcfg.Vaihingen3DPLConfig(); cfg.dropout = 0.0
    np.random.seed(1); torch.manual_seed(1)
    net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
    opt = make_optimizer(net, cfg)
    wl = synthetic.WORKLOADS["vaihingen"]
    outs = []
    for step in range(nsteps):
        pts, feats, labels, lens = synthetic.make_inputs(40 + step, 2, wl["points"], wl["radius"], cfg.in_features_dim)
        np.random.seed(step)
        batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev), lens, wl["limits"])
        loss, out = train_step(net, opt, batch, cfg)
        outs.append((float(loss), [m.clone() for m in batch.neighbors + batch.pools + batch.upsamples], out.detach().clone()))
    torch.cuda.synchronize()
    return {k: v.detach().cpu() for k, v in net.state_dict().items()}, outs
a, oa = run(); b, ob = run()
bad = [k for k in a if not torch.equal(a[k], b[k])]
print("weights differing:", len(bad), "of", len(a), bad[:5])
for s, (x, y) in enumerate(zip(oa, ob)):
    print("step", s, "loss", x[0], y[0], "matrices equal", all(torch.equal(p, q) for p, q in zip(x[1], y[1])), "outputs equal", torch.equal(x[2], y[2]))
