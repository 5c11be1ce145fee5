"""Diagnostic: cProfile of the weak-label training step (BASELINE config 1) on a prebuilt batch: where the host time goes."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import config as wcfg, pyramid, synthetic
from weasal_amd.architectures import KPFCNN_mprm
from weasal_amd.trainer import make_optimizer, train_step_weak, freeze_gc
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS["vaihingen_wl"]; cfg = wcfg.Vaihingen3DWLConfig()
np.random.seed(1); torch.manual_seed(1)
net = KPFCNN_mprm(cfg, np.arange(cfg.num_classes), []).to(dev).train(); opt = make_optimizer(net, cfg)
p, f, l, le = synthetic.make_inputs(0, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
b = pyramid.build_batch(cfg, torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev), le, wl["limits"])
region, region_lb, cloud_lb, centers = synthetic.make_weak_labels(0, p, l, le, num_classes=cfg.num_classes)
b.region, b.region_lb = region, region_lb
b.cloud_lb, b.center_pts = torch.from_numpy(cloud_lb).to(dev), torch.from_numpy(centers).to(dev)
for _ in range(10): train_step_weak(net, opt, b, cfg)
freeze_gc(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40):
    train_step_weak(net, opt, b, cfg)
t_issue = (time.perf_counter() - t0) / 40 * 1e3
torch.cuda.synchronize()
print("weak-label step alone: host issue %.2f ms / step, wall %.2f" % (t_issue, (time.perf_counter() - t0) / 40 * 1e3))
pr = cProfile.Profile(); pr.enable()
for i in range(10): train_step_weak(net, opt, b, cfg)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
