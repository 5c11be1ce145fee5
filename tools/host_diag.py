"""Diagnostic (GPU box): where the host time of a training step goes.
 1. raw launch cost: N tiny kernels through the C ABI from Python (ctypes) back to back;
 2. cProfile of the main thread's step with the pyramid prefetched on the side stream;
 3. kernel-launch count per step (torch profiler is not used: rocprofv3 --stats gives it)."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import _lib, config as wcfg, fused, ops, synthetic  # noqa: E402
from weasal_amd.architectures import KPFCNN  # noqa: E402
from weasal_amd.prefetch import PyramidPrefetcher  # noqa: E402
from weasal_amd.trainer import freeze_gc, make_optimizer, train_step  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
# ---- 1. launch cost
x = torch.zeros(64, 3, device=dev)
out = torch.empty_like(x)
lens = np.array([64], np.int32)
rot = np.eye(3, dtype=np.float32)[None]
for _ in range(200):
    ops.rotate_clouds_host(x, lens, rot)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3000
for _ in range(n):
    ops.rotate_clouds_host(x, lens, rot)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("tiny launch through ops (python+ctypes+hip): %.2f us issue, %.2f us incl. drain" % (1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
a = torch.zeros(1024, device=dev)
t0 = time.perf_counter()
for _ in range(n):
    a.add_(1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("torch add_ launch: %.2f us issue" % (1e6 * (t1 - t0) / n))

# ---- 2. step profile
wl = synthetic.WORKLOADS["dales"]
cfg = wcfg.DALESPLConfig()
np.random.seed(1)
torch.manual_seed(1)
net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
opt = make_optimizer(net, cfg)
inputs = []
for i in range(4):
    p, f, l, le = synthetic.make_inputs(i, 8, 50000, 10.0, 3)
    inputs.append((torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev), le))


def endless():
    i = 0
    while True:
        yield inputs[i % 4]
        i += 1


for mode in (sys.argv[1:] or ["fused", "ops"]):
    fused.FUSED_BLOCKS = mode == "fused"
    pf = PyramidPrefetcher(cfg, endless(), wl["limits"], depth=2, device=dev)
    for _ in range(12):
        train_step(net, opt, next(pf), cfg)
    torch.cuda.synchronize()
    freeze_gc()
    t0 = time.perf_counter()
    for _ in range(20):
        train_step(net, opt, next(pf), cfg)
    ti = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print("[%s] 20 steps: issue %.2f ms/step, total %.2f ms/step" % (mode, 50 * ti, 50 * tt), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        train_step(net, opt, next(pf), cfg)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
    print(s.getvalue()[:3500], flush=True)
    pf.close()
