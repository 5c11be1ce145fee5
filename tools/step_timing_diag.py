"""Diagnostic (not part of the product): per-step wall times of pyramid / forward / backward / optimizer
with a synchronisation after each phase."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from weasal_amd import config as wcfg, ops, pyramid, synthetic
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer
dev = torch.device('cuda:0')
wl = synthetic.WORKLOADS['dales']; cfg = wcfg.DALESPLConfig()
np.random.seed(1); torch.manual_seed(1)
net = KPFCNN(cfg, np.arange(9), []).to(dev).train(); opt = make_optimizer(net, cfg)
inputs = []
for i in range(4):
    p, f, l, le = synthetic.make_inputs(i, 8, 50000, 10.0, 3)
    inputs.append((torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev), le))
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 14):
    t0 = T()
    p, f, l, le = inputs[i % 4]
    b = pyramid.build_batch(cfg, p, f, l, le, wl['limits'])
    t1 = T()
    opt.zero_grad(set_to_none=False)
    out = net(b, cfg); t2 = T()
    loss = net.loss(out, b.labels); t3 = T()
    loss.backward(); t4 = T()
    torch.nn.utils.clip_grad_value_(net.parameters(), cfg.grad_clip_norm); t5 = T()
    opt.step(); t6 = T()
    print("%2d pyr %.1f fwd %.1f loss %.1f bwd %.1f clip %.1f opt %.1f  reserved %.1f GB" % (
        i, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t5-t4), 1e3*(t6-t5), torch.cuda.memory_reserved()/1e9))
