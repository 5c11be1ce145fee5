"""Diagnostic: the training step alone (forward + loss + backward + SGD) on prebuilt DALES batches -- no pyramid work on the
GPU, nothing on a second stream: the clean per-step GPU time of the training stream."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from weasal_amd import config as wcfg, pyramid, synthetic
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer, train_step, freeze_gc, InFlightLimiter
dev = torch.device('cuda:0')
cfg = wcfg.DALESPLConfig()
torch.manual_seed(0); np.random.seed(0)
net = KPFCNN(cfg, np.arange(9), []).to(dev); net.train()
opt = make_optimizer(net, cfg)
wl = synthetic.WORKLOADS['dales']
batches = []
for s in range(3):
    pts, feats, labels, lens = synthetic.make_inputs(s, wl['spheres'], wl['points'], wl['radius'], cfg.in_features_dim)
    batches.append(pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev), lens, wl['limits']))
EP = 0 if (len(sys.argv) > 2 and sys.argv[2] == 'contrast') else None
for i in range(6): train_step(net, opt, batches[i % 3], cfg, epoch=EP)
freeze_gc()
torch.cuda.synchronize()
lim = InFlightLimiter(4)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t0 = time.perf_counter()
wait = 0.0
for i in range(steps):
    train_step(net, opt, batches[i % 3], cfg, epoch=EP)
    tw = time.perf_counter()
    lim.tick()
    wait += time.perf_counter() - tw
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
from weasal_amd import fused
import ctypes
print("training step alone: %.3f ms/step (host issue %.3f ms/step, of which waiting for the GPU %.3f); gates=%d"
      % (1e3 * dt / steps, 1e3 * t_issue / steps, 1e3 * wait / steps, ctypes.c_int.in_dll(fused._bind(), "ws_block_gates").value))
