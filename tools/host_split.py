"""Diagnostic: host time of the two threads of a step, each alone: the training step on a prebuilt batch, and the pyramid build.
usage: python3 tools/host_split.py <workload> [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import config as wcfg, pyramid, synthetic
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer, train_step, freeze_gc
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "vaihingen"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
wl = synthetic.WORKLOADS[name]; cfg = getattr(wcfg, wl["config"])()
np.random.seed(1); torch.manual_seed(1)
net = KPFCNN(cfg, np.arange(9), []).to(dev).train(); opt = make_optimizer(net, cfg)
p, f, l, le = synthetic.make_inputs(0, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
P, F, Lb = torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev)
b = pyramid.build_batch(cfg, P, F, Lb, le, wl["limits"])
for _ in range(10): train_step(net, opt, b, cfg, epoch=0)
freeze_gc(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    train_step(net, opt, b, cfg, epoch=0)
    if i % 4 == 3: torch.cuda.synchronize()
torch.cuda.synchronize()
t_train = (time.perf_counter() - t0) / steps * 1e3
t0 = time.perf_counter()
for i in range(steps):
    train_step(net, opt, b, cfg, epoch=0)
t_issue = (time.perf_counter() - t0) / steps * 1e3
torch.cuda.synchronize()
for _ in range(5): pyramid.build_batch(cfg, P, F, Lb, le, wl["limits"])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    bb = pyramid.build_batch(cfg, P, F, Lb, le, wl["limits"])
torch.cuda.synchronize()
t_pyr = (time.perf_counter() - t0) / steps * 1e3
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for i in range(10):
    bb = pyramid.build_batch(cfg, P, F, Lb, le, wl["limits"])
pr.disable()
print("%s: training step alone %.2f ms wall (sync every 4) / %.2f ms issue only; pyramid build alone %.2f ms wall per batch" % (name, t_train, t_issue, t_pyr))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:5000])
