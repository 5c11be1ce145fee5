"""Diagnostic: cProfile of the Vaihingen (BASELINE config 2) training step on a prebuilt batch: the host side of a launch-bound step."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import config as wcfg, pyramid, synthetic
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer, train_step, freeze_gc
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS["vaihingen"]; cfg = wcfg.Vaihingen3DPLConfig()
np.random.seed(1); torch.manual_seed(1)
net = KPFCNN(cfg, np.arange(9), []).to(dev).train(); opt = make_optimizer(net, cfg)
p, f, l, le = synthetic.make_inputs(0, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
b = pyramid.build_batch(cfg, torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev), le, wl["limits"])
for _ in range(10): train_step(net, opt, b, cfg, epoch=0)
freeze_gc(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50): train_step(net, opt, b, cfg, epoch=0)
t_issue = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
print("training step alone: host issue %.2f ms / step" % t_issue)
pr = cProfile.Profile(); pr.enable()
for i in range(20): train_step(net, opt, b, cfg, epoch=0)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:7000])
