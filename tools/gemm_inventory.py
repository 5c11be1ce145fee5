"""Diagnostic: every MFMA GEMM launch of one DALES training step with its shape, time and streaming floor."""
import sys, collections
import numpy as np, torch
sys.path.insert(0, '.')
from weasal_amd import ops, _lib, config as wcfg, pyramid, synthetic
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer, train_step

dev = torch.device('cuda:0')
lib = _lib.lib()
cfg = wcfg.DALESPLConfig()
torch.manual_seed(0); np.random.seed(0)
net = KPFCNN(cfg, np.arange(9), []).to(dev); net.train()
opt = make_optimizer(net, cfg)
wl = synthetic.WORKLOADS['dales']
pts, feats, labels, lens = synthetic.make_inputs(0, wl['spheres'], wl['points'], wl['radius'], cfg.in_features_dim)
pts, feats, labels = (torch.from_numpy(a).to(dev) for a in (pts, feats, labels))
def step():
    b = pyramid.build_batch(cfg, pts, feats, labels, lens)
    return train_step(net, opt, b, cfg)
for _ in range(4): step()
torch.cuda.synchronize()
rec = []
def wrap(name, shape_of):
    f = getattr(lib, name)
    def g(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(*a); e1.record()
        rec.append((name, shape_of(a), e0, e1))
        return r
    setattr(lib, name, g)
wrap('ws_gemm_xb', lambda a: (a[1], a[2], a[5], 0))
wrap('ws_gemm_xb_epilogue', lambda a: (a[1], a[2], a[5], (1 if a[6] else 0) + (2 if a[7] else 0) + (4 if a[9] else 0)))
wrap('ws_gemm_xty', lambda a: (a[1], a[2], a[5], 0))
step(); torch.cuda.synchronize()
tot = collections.OrderedDict()
for name, sh, e0, e1 in rec:
    t = e0.elapsed_time(e1) * 1e3
    k = (name, sh)
    c = tot.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += t
grand = 0.0
rows = []
for (name, (m, k, n, ep)), (cnt, t) in tot.items():
    byt = 4.0 * m * (k + n) + (4.0 * m * n if ep & 2 else 0)
    flop = 2.0 * m * k * n
    floor = max(byt / 5.0e12, flop / 150e12) * 1e6 * cnt
    rows.append((t, name, m, k, n, ep, cnt, floor))
    grand += t
rows.sort(reverse=True)
for t, name, m, k, n, ep, cnt, floor in rows:
    print("%-20s M=%7d K=%4d N=%4d ep=%d x%d  %8.1f us  floor %7.1f us  eff %.2f" % (name, m, k, n, ep, cnt, t, floor, floor / t))
print("total %.1f us over %d launches" % (grand, len(rec)))
