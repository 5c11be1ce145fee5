"""Diagnostic: cProfile of the training thread alone (prebuilt batches): where the ~4 ms of host work per step go."""
import cProfile, pstats, sys, io, numpy as np, torch
sys.path.insert(0, '.')
from weasal_amd import config as wcfg, pyramid, synthetic
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer, train_step, freeze_gc
dev = torch.device('cuda:0')
wl = synthetic.WORKLOADS['dales']; cfg = wcfg.DALESPLConfig()
np.random.seed(1); torch.manual_seed(1)
net = KPFCNN(cfg, np.arange(9), []).to(dev).train(); opt = make_optimizer(net, cfg)
p, f, l, le = synthetic.make_inputs(0, 8, 50000, 10.0, 3)
b = pyramid.build_batch(cfg, torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev), le, wl['limits'])
for _ in range(8): train_step(net, opt, b, cfg)
freeze_gc(); torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    train_step(net, opt, b, cfg)
    if _ % 4 == 3: torch.cuda.synchronize()
print("un-profiled host time per step (with a sync every 4 steps): %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    train_step(net, opt, b, cfg)
    torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30); print(s.getvalue()[:7000])
