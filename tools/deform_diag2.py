#!/usr/bin/env python3
"""tools/deform_diag2.py -- the body of tests/test_config5_wide_gpu.py::test_deformable_kpconv_real_width_vs_oracle[self-0-32-f32]
run with the fast path and with the generic kernels in ONE process; GPU tensors compared directly and against the oracle."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_config5_wide_gpu as T  # noqa: E402
from oracle import kpconv_ref  # noqa: E402
from weasal_amd import blocks  # noqa: E402
from weasal_amd.architectures import p2p_fitting_regularizer  # noqa: E402

gpu = torch.device("cuda:0")
cfg, batch = T._small_dense_batch(gpu)
batch.activate()
lvl, ci = 0, 32
r = cfg.first_subsampling_dl * cfg.conv_radius * 2 ** lvl
extent = r * cfg.KP_extent / cfg.conv_radius
P, inds = batch.points[lvl], batch.neighbors[lvl]
mk = lambda c: types.SimpleNamespace(modules=lambda: [c], l1=torch.nn.L1Loss(), K=15, repulse_extent=1.2, deform_fitting_power=1.0)
res = {}
order = [int(v) for v in os.environ.get("DIAG_ORDER", "1,0").split(",")]
for fast in order:
    blocks.DEFORM_FAST_PATH = bool(fast)
    conv, twin = T._layer_pair(gpu, ci, ci, extent, r, False)
    torch.manual_seed(4 + lvl)
    x = torch.randn(P.shape[0], ci, device=gpu)
    dy = torch.randn(P.shape[0], ci, device=gpu)
    xg = x.clone().requires_grad_(True)
    out = conv(P, P, inds, xg)
    keep = {}
    conv.offset_features.register_hook(lambda g, keep=keep: keep.__setitem__("doff", g.clone()))
    conv.min_d2.register_hook(lambda g, keep=keep: keep.__setitem__("dmin", g.clone()))
    conv.deformed_KP.register_hook(lambda g, keep=keep: keep.__setitem__("ddkp", g.clone()))
    reg = p2p_fitting_regularizer(mk(conv))
    ((out.float() * dy.float()).sum() + reg).backward()
    torch.cuda.synchronize()
    res[fast] = dict(dx=xg.grad.clone(), dWo=conv.offset_conv.weights.grad.clone(), out=out.detach(), **keep)
xc = x.float().cpu().requires_grad_(True)
with kpconv_ref.cpu_reference_mode():
    ref = twin(P.cpu(), P.cpu(), inds.cpu(), xc)
    keep = {}
    twin.offset_features.register_hook(lambda g, keep=keep: keep.__setitem__("doff", g.clone()))
    twin.min_d2.register_hook(lambda g, keep=keep: keep.__setitem__("dmin", g.clone()))
    twin.deformed_KP.register_hook(lambda g, keep=keep: keep.__setitem__("ddkp", g.clone()))
    reg_c = p2p_fitting_regularizer(mk(twin))
((ref * dy.float().cpu()).sum() + reg_c).backward()
orc = dict(dx=xc.grad, dWo=twin.offset_conv.weights.grad, out=ref.detach(), **keep)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())


for fast in order:
    print("fast" if fast else "generic", "vs oracle:", {k: "%.2e" % rel(v, orc[k]) for k, v in res[fast].items()})
if len(order) == 2:
    print("fast vs generic:", {k: "%.2e" % rel(v, res[0][k]) for k, v in res[1].items()})
d = (res[order[0]]["doff"].cpu() - orc["doff"]).abs().amax(dim=1)
bad = torch.nonzero(d > 1e-4 * orc["doff"].abs().max()).flatten()
print("rows of d_offset_features off vs oracle:", bad.numel(), bad[:10].tolist())
if bad.numel():
    b0 = int(bad[0])
    diff = (res[order[0]]["doff"][b0].cpu() - orc["doff"][b0])
    print("row", b0, "columns off:", torch.nonzero(diff.abs() > 1e-5 * orc["doff"].abs().max()).flatten().tolist())
    print("count of real neighbours:", int((inds[b0] < P.shape[0]).sum()))
    print("gpu ", res[order[0]]["doff"][b0][:9].tolist())
    print("orc ", orc["doff"][b0][:9].tolist())
