"""Diagnostic: gradient w.r.t. every block's OUTPUT (GPU vs fp32 CPU oracle vs oracle under a 1e-6 input perturbation) on
the full-width DALES network: where along the backward chain a discrepancy enters."""
import copy, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from test_fullwidth_gpu import _cpu_copy, _rel  # noqa: E402
from oracle import kpconv_ref  # noqa: E402
from weasal_amd import config as wcfg, pyramid, synthetic  # noqa: E402
from weasal_amd.architectures import KPFCNN  # noqa: E402

gpu = torch.device("cuda:0")
wl = synthetic.WORKLOADS["dales"]; cfg = wcfg.DALESPLConfig(); cfg.dropout = 0.0
np.random.seed(3); torch.manual_seed(3)
net0 = KPFCNN(cfg, np.arange(9), [])
pts, feats, labels, lens = synthetic.make_inputs(4242, 2, wl["points"], wl["radius"], cfg.in_features_dim)
np.random.seed(9)
batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu), torch.from_numpy(labels).to(gpu), lens, wl["limits"])


def run(net, b, ctx=None, eps=0.0):
    grads, outs = {}, {}
    hooks = []
    blocks = [("enc%d" % i, m) for i, m in enumerate(net.encoder_blocks)] + [("dec%d" % i, m) for i, m in enumerate(net.decoder_blocks)]
    for name, m in blocks:
        def fh(mod, inp, out, name=name):
            outs[name] = out.detach()
            if out.requires_grad:
                out.register_hook(lambda g, name=name: grads.__setitem__(name, g.detach().clone()))
        hooks.append(m.register_forward_hook(fh))
    if eps:
        gen = torch.Generator().manual_seed(123)
        b.features = b.features * (1 + eps * torch.randn(b.features.shape, generator=gen))
    if ctx is not None:
        with ctx:
            out = net(b, cfg); net.loss(out, b.labels).backward()
    else:
        out = net(b, cfg); net.loss(out, b.labels).backward()
    for h in hooks: h.remove()
    return outs, grads


net_g = copy.deepcopy(net0).to(gpu).train()
og, gg = run(net_g, batch)
torch.cuda.synchronize()
net_c = copy.deepcopy(net0).train()
oc, gc = run(net_c, _cpu_copy(batch), kpconv_ref.cpu_reference_mode())
net_p = copy.deepcopy(net0).train()
op_, gp = run(net_p, _cpu_copy(batch), kpconv_ref.cpu_reference_mode(), eps=1e-6)
print("%-6s %10s %10s | %10s %10s   (activation: gpu vs oracle, oracle-perturbed vs oracle | grad of output: same)" % ("block", "act gpu", "act pert", "grad gpu", "grad pert"))
for k in oc:
    if k in gc and k in gg:
        print("%-6s %10.2e %10.2e | %10.2e %10.2e   rows %d" % (k, _rel(og[k], oc[k]), _rel(op_[k], oc[k]), _rel(gg[k], gc[k]), _rel(gp[k], gc[k]), oc[k].shape[0]), flush=True)
