"""Diagnostic: Vaihingen training steps with and without the single-rank gradient exchange: where do the weights part?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import config as wcfg, pyramid, synthetic, dp
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer, train_step
import torch.distributed as dist
dev = torch.device("cuda:0")
def run(use_dp, nsteps):
    sync = None
    if use_dp:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dp.init_from_env(backend="nccl", force=True)
        sync = dp.GradSync(buckets=1, single_rank_exchange=True)
    cfg = wcfg.Vaihingen3DPLConfig(); cfg.dropout = 0.0
    np.random.seed(1); torch.manual_seed(1)
    net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
    opt = make_optimizer(net, cfg)
    wl = synthetic.WORKLOADS["vaihingen"]
    grads = []
    for step in range(nsteps):
        pts, feats, labels, lens = synthetic.make_inputs(40 + step, 2, wl["points"], wl["radius"], cfg.in_features_dim)
        np.random.seed(step)
        batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev), lens, wl["limits"])
        loss, out = train_step(net, opt, batch, cfg, grad_sync=sync)
        grads.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    torch.cuda.synchronize()
    if use_dp: dist.destroy_process_group()
    return {k: v.detach().cpu() for k, v in net.state_dict().items()}, grads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
a, ga = run(False, n); b, gb = run(True, n)
for s in range(n):
    bad = [(k, float((ga[s][k] - gb[s][k]).abs().max()), float(ga[s][k].abs().max())) for k in ga[s] if not torch.equal(ga[s][k], gb[s][k])]
    print("step", s, "gradients differing:", len(bad), "of", len(ga[s]), bad[:4])
bad = [k for k in a if not torch.equal(a[k], b[k])]
print("weights differing:", len(bad), "of", len(a), bad[:5])
