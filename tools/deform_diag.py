#!/usr/bin/env python3
"""tools/deform_diag.py -- A/B of the deformable fast path on the config-5 level-0 self-query layer (7 000 points, H = 422):
gradients of the offset features and of x with the fast path / generic kernels and with / without the grid backward."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import sphere  # noqa: E402
from weasal_amd import blocks, config as wcfg, ops, pyramid  # noqa: E402
from weasal_amd.blocks import KPConv  # noqa: E402

gpu = torch.device("cuda:0")
cfg = wcfg.DALESDeformF32Config()
rng = np.random.default_rng(5)
pts = sphere(rng, 7000, 5.2)
np.random.seed(2)
batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.ones(7000, 3, device=gpu), torch.zeros(7000, dtype=torch.int64, device=gpu),
                            np.array([7000], np.int32), [422, 519, 472, 193, 34])
batch.activate()
P, inds = batch.points[0], batch.neighbors[0]
print("grid", ops._grid_for(inds) is not None, "max_count", ops._grid_for(inds).max_count if ops._grid_for(inds) else None)
np.random.seed(1)
torch.manual_seed(1)
conv = KPConv(15, 3, 32, 32, 0.4, 1.0, deformable=True, modulated=True).to(gpu)
with torch.no_grad():
    conv.offset_conv.weights.mul_(4.0)
    conv.offset_bias.normal_(0.0, 0.05)
torch.manual_seed(4)
x = torch.randn(7000, 32, device=gpu)
dy = torch.randn(7000, 32, device=gpu)
res = {}
for fast in (1, 0):
    for gridb in (1, 0):
        blocks.DEFORM_FAST_PATH = bool(fast)
        ops.GRID_BACKWARD = bool(gridb)
        conv.zero_grad()
        xg = x.clone().requires_grad_(True)
        out = conv(P, P, inds, xg)
        keep = {}
        conv.offset_features.register_hook(lambda g, keep=keep: keep.__setitem__("doff", g.clone()))
        import types
        from weasal_amd.architectures import p2p_fitting_regularizer
        net = types.SimpleNamespace(modules=lambda: [conv], l1=torch.nn.L1Loss(), K=15, repulse_extent=1.2, deform_fitting_power=1.0)
        mode = os.environ.get("DIAG_LOSS", "reg")
        if mode == "reg":
            extra = p2p_fitting_regularizer(net)
        elif mode == "rep":
            extra = (conv.deformed_KP ** 2).sum() * 1e-3
        else:
            extra = 0.01 * conv.min_d2.sum()
        ((out * dy).sum() + extra).backward()
        res[(fast, gridb)] = dict(dx=xg.grad.clone(), doff=keep["doff"], dWo=conv.offset_conv.weights.grad.clone(), out=out.detach())
ref = res[(0, 0)]


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


for key, r in res.items():
    print(key, {k: "%.2e" % rel(v, ref[k]) for k, v in r.items()})
d = (res[(1, 0)]["doff"] - ref["doff"]).abs().amax(dim=1)
bad = torch.nonzero(d > 1e-4 * ref["doff"].abs().max()).flatten()
print("fast, table: rows of d_offset_features that differ:", bad.numel(), bad[:20].tolist())
if bad.numel():
    cnt = (inds[bad] < 7000).sum(dim=1)
    print("their neighbour counts:", cnt[:20].tolist())
    cols = torch.nonzero((res[(1, 0)]["doff"][bad[0]] - ref["doff"][bad[0]]).abs() > 1e-6).flatten()
    print("columns of the first bad row:", cols.tolist())
