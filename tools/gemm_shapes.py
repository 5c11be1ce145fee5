"""Diagnostic: every dense product of one DALES training step (with the contrastive term), timed on its own by the library
(WEASAL_GEMM_LOG=1: events + synchronisation around each product), aggregated per shape."""
import os, sys, subprocess, collections, re
if os.environ.get("WEASAL_GEMM_LOG") != "1":
    env = dict(os.environ, WEASAL_GEMM_LOG="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
    agg = collections.OrderedDict()
    lines = [l for l in r.stderr.splitlines() if l.startswith("GEMMLOG")]
    # the last step only: the child prints a marker before it
    txt = r.stderr.split("GEMMLOG-LAST-STEP")[-1]
    for l in txt.splitlines():
        m = re.match(r"GEMMLOG (\w+) m=(\d+) k=(\d+) n=(\d+) us=([\d.]+) tflops=([\d.]+)", l)
        if m:
            key = (m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)))
            a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(m.group(5))
    tot = sum(a[1] for a in agg.values()); fl = sum(2.0 * k[1] * k[2] * k[3] * a[0] for k, a in agg.items())
    print("%d products, %.2f ms, %.1f GFLOP -> %.1f TFLOP/s" % (sum(a[0] for a in agg.values()), tot / 1e3, fl / 1e9, fl / (tot * 1e-6) / 1e12))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-4s m=%7d k=%5d n=%5d  x%d  %8.1f us total  %6.1f us each  %6.1f TFLOP/s" % (k[0], k[1], k[2], k[3], a[0], a[1], a[1] / a[0], 2.0 * k[1] * k[2] * k[3] / (a[1] / a[0] * 1e-6) / 1e12))
    sys.exit(0)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import config as wcfg, pyramid, synthetic
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer, train_step
dev = torch.device("cuda:0")
cfg = wcfg.DALESPLConfig(); wl = synthetic.WORKLOADS["dales"]
torch.manual_seed(0); np.random.seed(0)
net = KPFCNN(cfg, np.arange(9), []).to(dev).train(); opt = make_optimizer(net, cfg)
p, f, l, le = synthetic.make_inputs(0, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
b = pyramid.build_batch(cfg, torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev), le, wl["limits"])
for _ in range(3): train_step(net, opt, b, cfg, epoch=0)
torch.cuda.synchronize()
print("GEMMLOG-LAST-STEP", file=sys.stderr, flush=True)
train_step(net, opt, b, cfg, epoch=0)
torch.cuda.synchronize()
