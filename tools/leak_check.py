import gc, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from weasal_amd import config as wcfg, pyramid, synthetic
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer, train_step
dev = torch.device("cuda:0")
cfg = wcfg.Vaihingen3DPLConfig(); wl = synthetic.WORKLOADS["vaihingen"]
torch.manual_seed(0); np.random.seed(0)
net = KPFCNN(cfg, np.arange(9), []).to(dev).train(); opt = make_optimizer(net, cfg)
p, f, l, le = synthetic.make_inputs(0, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
gc.collect(); gc.disable()
mem = []
for step in range(60):
    b = pyramid.build_batch(cfg, torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev), le, wl["limits"])
    train_step(net, opt, b, cfg, epoch=0)
    del b
    if step % 10 == 9:
        torch.cuda.synchronize(); mem.append(torch.cuda.memory_allocated() >> 20)
print("MiB allocated every 10 steps with the cyclic collector off:", mem)
assert mem[-1] <= mem[1] + 8, mem
print("no growth")
