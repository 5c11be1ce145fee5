"""Diagnostic: K1 (radius search) launches of the DALES pyramid's level 0 timed alone: conv search (400 000 self queries),
pool search (71 000 queries in 400 000 supports), upsample search (400 000 queries in 71 000 supports, 2 r)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import config as wcfg, ops, pyramid, synthetic
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS["dales"]
cfg = wcfg.DALESPLConfig()
pts, feats, labels, lens = synthetic.make_inputs(1, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
np.random.seed(0)
batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev), lens, wl["limits"])
P0, P1 = batch.points[0], batch.points[1]
l0, l1 = [int(v) for v in batch.lengths[0].cpu()], [int(v) for v in batch.lengths[1].cpu()]
r = cfg.first_subsampling_dl * cfg.conv_radius
def timeit(fn, rep=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rep): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / rep)
    return best
def search(q, s, ql, sl, radius, limit):
    d = ops.DeferredSearches(dev)
    d.add(q, s, ql, sl, radius, limit)
    return d
for name, q, s, ql, sl, radius, limit in (("conv  L0", P0, P0, l0, l0, r, wl["limits"][0]), ("pool  L0", P1, P0, l1, l0, r, wl["limits"][0]),
                                          ("upsmp L0", P0, P1, l0, l1, 2 * r, wl["limits"][1])):
    t = timeit(lambda: search(q, s, ql, sl, radius, limit))
    print("%s: %d queries in %d supports, limit %d: %.3f ms per search (grid build + fill, asynchronous form)" % (name, q.shape[0], s.shape[0], limit, t), flush=True)
