"""Diagnostic: per-parameter gradient error of the full-width DALES network against the fp32 / fp64 CPU oracle, with
the K3 forward variant and the block calls switched (which component moves the error)."""
import copy
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from test_fullwidth_gpu import _cpu_copy, _oracle64_grads, _oracle_step, _rel  # noqa: E402
from weasal_amd import _lib, config as wcfg, fused, pyramid, synthetic  # noqa: E402
from weasal_amd.architectures import KPFCNN  # noqa: E402

gpu = torch.device("cuda:0")
wl = synthetic.WORKLOADS["dales"]
cfg = wcfg.DALESPLConfig()
cfg.dropout = 0.0
np.random.seed(3)
torch.manual_seed(3)
net0 = KPFCNN(cfg, np.arange(9), [])
pts, feats, labels, lens = synthetic.make_inputs(4242, 2, wl["points"], wl["radius"], cfg.in_features_dim)
np.random.seed(9)
batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu),
                            torch.from_numpy(labels).to(gpu), lens, wl["limits"])
batch_cpu = _cpu_copy(batch)
net_cpu = copy.deepcopy(net0).train()
_oracle_step(net_cpu, batch_cpu, cfg)
g64 = _oracle64_grads(net_cpu, batch_cpu, cfg)
ref = {k: p.grad for k, p in net_cpu.named_parameters()}
variant = C.c_int.in_dll(_lib.lib(), "ws_kpconv_variant")
for v, fb in ((2, True), (1, True), (2, False), (1, False)):
    variant.value = v
    fused.FUSED_BLOCKS = fb
    net = copy.deepcopy(net0).to(gpu).train()
    out = net(batch, cfg)
    net.loss(out, batch.labels).backward()
    torch.cuda.synchronize()
    rows = []
    for k, p in net.named_parameters():
        if p.grad is None:
            continue
        rows.append((k, _rel(p.grad, ref[k]), _rel(p.grad, g64[k]), _rel(ref[k], g64[k])))
    worst = sorted(rows, key=lambda r: -r[2] / max(r[3], 1e-12))[:6]
    print("K3 variant %d, block calls %s: worst gradient-error ratios (name, vs f32 oracle, vs f64, f32 oracle vs f64)" % (v, fb))
    for r in worst:
        print("   %-40s %.2e %.2e %.2e  ratio %.1f" % (r[0], r[1], r[2], r[3], r[2] / max(r[3], 1e-12)), flush=True)
variant.value = 2
fused.FUSED_BLOCKS = True
