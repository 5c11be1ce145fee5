"""Diagnostic: cProfile of the host side of a few training steps (where does the Python/launch time go)."""
import cProfile, pstats, sys, io, numpy as np, torch
sys.path.insert(0, '.')
from weasal_amd import config as wcfg, pyramid, synthetic
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer, train_step
dev = torch.device('cuda:0')
wl = synthetic.WORKLOADS['dales']; cfg = wcfg.DALESPLConfig()
np.random.seed(1); torch.manual_seed(1)
net = KPFCNN(cfg, np.arange(9), []).to(dev).train(); opt = make_optimizer(net, cfg)
p, f, l, le = synthetic.make_inputs(0, 8, 50000, 10.0, 3)
inp = (torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev), le)
def step():
    b = pyramid.build_batch(cfg, inp[0], inp[1], inp[2], inp[3], wl['limits'])
    train_step(net, opt, b, cfg)
for _ in range(12): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
