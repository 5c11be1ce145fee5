"""Diagnostic (not part of the product): un-synchronised training steps as bench.py issues them; prints, per step,
the host time at which the step was issued and the GPU time at which it finished (event), to locate stalls."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from weasal_amd import config as wcfg, ops, pyramid, synthetic
from weasal_amd.architectures import KPFCNN
from weasal_amd.trainer import make_optimizer, train_step, InFlightLimiter
from weasal_amd.prefetch import PyramidPrefetcher
dev = torch.device('cuda:0')
wl = synthetic.WORKLOADS['dales']; cfg = wcfg.DALESPLConfig()
np.random.seed(1); torch.manual_seed(1)
net = KPFCNN(cfg, np.arange(9), []).to(dev).train(); opt = make_optimizer(net, cfg)
inputs = []
for i in range(4):
    p, f, l, le = synthetic.make_inputs(i, 8, 50000, 10.0, 3)
    inputs.append((torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev), le))
use_prefetch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
def src():
    i = 0
    while True:
        yield inputs[i % 4]; i += 1
pf = PyramidPrefetcher(cfg, src(), wl['limits'], depth=2) if use_prefetch else None
def step(i):
    if pf is not None:
        b = next(pf)
    else:
        p, f, l, le = inputs[i % 4]
        b = pyramid.build_batch(cfg, p, f, l, le, wl['limits'])
    train_step(net, opt, b, cfg)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
import gc
if len(sys.argv) > 4 and sys.argv[4] == 'nogc':
    gc.collect(); gc.freeze(); gc.disable()
lim = InFlightLimiter(int(sys.argv[3])) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else None
for phase in range(2):
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
    t0 = time.perf_counter()
    host, evs = [], []
    for i in range(n):
        step(i)
        if lim is not None: lim.tick()
        host.append(time.perf_counter() - t0)
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    torch.cuda.synchronize()
    gpu = [ev0.elapsed_time(e) for e in evs]
    print("phase", phase, "total %.1f ms for %d steps" % (gpu[-1], n))
    prev_h = prev_g = 0.0
    for i in range(n):
        dh, dg = 1e3 * host[i] - prev_h, gpu[i] - prev_g
        flag = "  <<<" if dg > 30 or dh > 30 else ""
        print("  step %2d host +%6.1f ms  gpu +%6.1f ms%s" % (i, dh, dg, flag))
        prev_h, prev_g = 1e3 * host[i], gpu[i]
if pf is not None: pf.close()
