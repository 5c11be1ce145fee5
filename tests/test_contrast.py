"""KPFCNN.contrast_loss (fused HIP operator ops.contrast_rows) against oracle/contrast_ref.py --
PARITY UNPINNED (torch_scatter is absent; see the oracle header).  Bar: 1e-4 relative on the loss and on the
gradient of the logits (fp32, different summation order)."""
import numpy as np
import pytest
import torch

from oracle.contrast_ref import contrast_loss_ref
from weasal_amd import config as wcfg
from weasal_amd.architectures import KPFCNN


def _case(seed, n, n_labeled, sharp):
    g = torch.Generator().manual_seed(seed)
    outputs = torch.randn(n, 9, generator=g) * sharp
    labels = torch.full((n,), 100, dtype=torch.int64)            # unlabeled (> 10)
    idx = torch.randperm(n, generator=g)[:n_labeled]
    labels[idx] = torch.randint(0, 9, (n_labeled,), generator=g)
    return outputs, labels


def _run(dev, seed, n, n_labeled, sharp, thd):
    cfg = wcfg.DALESPLConfig()
    cfg.contrast_thd = thd
    outputs, labels = _case(seed, n, n_labeled, sharp)
    with torch.no_grad():
        prob = torch.softmax(outputs, 1)
        valid = int((((prob.max(1)[0] > thd / 100) | (labels < 10))).sum())
    if valid == 0:
        draw = torch.zeros(0, dtype=torch.int64)
    else:
        g = torch.Generator().manual_seed(seed + 1)
        draw = torch.randint(0, valid, (1000 if valid >= 1000 else 1000 - valid,), generator=g)
    o_ref = outputs.clone().requires_grad_(True)
    ref = contrast_loss_ref(o_ref, labels, thd, draw)
    net = KPFCNN.__new__(KPFCNN)                                  # the method needs no weights
    o = outputs.clone().to(dev).requires_grad_(True)
    got = KPFCNN.contrast_loss(net, o, labels.to(dev), cfg, slice_draw=draw)
    assert got.device.type == dev.type
    if valid == 0:
        assert float(got) == 0.0 and float(ref) == 0.0
        return
    g_, r_ = float(got.detach()), float(ref.detach())
    assert abs(g_ - r_) <= 1e-4 * max(1.0, abs(r_)), (g_, r_)
    if ref.requires_grad and got.requires_grad:
        ref.backward(); got.backward()
        scale = float(o_ref.grad.abs().max())
        assert float((o.grad.cpu() - o_ref.grad).abs().max()) <= 1e-4 * max(scale, 1e-12)


CASES = [(0, 3000, 300, 2.0, 10),      # num_valid >= slc_con
         (1, 1500, 40, 0.05, 60),      # few valid points: the padded-slice branch (:450-454)
         (2, 800, 0, 0.01, 99),        # nothing valid: returns 0
         (3, 2500, 2500, 1.0, 10)]     # fully labeled


def test_contrast_oracle_cpu_sanity():
    """the restatement itself: nothing valid -> 0; otherwise a finite positive loss with a finite gradient"""
    outputs, labels = _case(2, 800, 0, 0.01)
    assert float(contrast_loss_ref(outputs, labels, 99, torch.zeros(0, dtype=torch.int64))) == 0.0
    outputs, labels = _case(0, 1200, 300, 2.0)
    o = outputs.clone().requires_grad_(True)
    valid = int(((torch.softmax(outputs, 1).max(1)[0] > 0.1) | (labels < 10)).sum())
    draw = torch.randint(0, valid, (1000 if valid >= 1000 else 1000 - valid,), generator=torch.Generator().manual_seed(5))
    loss = contrast_loss_ref(o, labels, 10, draw)
    loss.backward()
    assert np.isfinite(float(loss.detach())) and float(loss.detach()) > 0 and bool(torch.isfinite(o.grad).all())


def test_contrast_product_rejects_cpu_tensors():
    """no CPU path in the product: the fused operator refuses host tensors"""
    cfg = wcfg.DALESPLConfig()
    outputs, labels = _case(0, 1200, 300, 2.0)
    with pytest.raises(RuntimeError):
        KPFCNN.contrast_loss(KPFCNN.__new__(KPFCNN), outputs, labels, cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_contrast_loss_gpu(case):
    _run(torch.device("cuda:0"), *case)
