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


def _both_forms(dev, outputs, labels, thd, slice_draw=None, seed=None):
    """contrast_loss through the head / tail kernels and through the torch-op form: (loss, grad, slice) of each"""
    from weasal_amd import architectures
    cfg = wcfg.DALESPLConfig()
    cfg.contrast_thd = thd
    res = []
    for kernels in (True, False):
        architectures.CONTRAST_KERNELS = kernels
        try:
            if seed is not None:
                torch.manual_seed(seed)
                torch.cuda.manual_seed(seed)
            net = KPFCNN.__new__(KPFCNN)
            o = outputs.clone().to(dev).requires_grad_(True)
            loss = KPFCNN.contrast_loss(net, o, labels.to(dev), cfg, slice_draw=slice_draw)
            loss.backward()
            res.append((float(loss.detach()), o.grad.cpu(), net.pts_loss.detach().cpu(),
                        net.contrast_slice.cpu() if kernels else None))
        finally:
            architectures.CONTRAST_KERNELS = True
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_contrast_head_tail_kernels_match_torch_form(case):
    """the fused head / tail (ws_contrast_head_* / _tail_*) against the same loss written with torch ops around
    ops.contrast_rows: same slice, loss and logits gradient to fp32 rounding"""
    seed, n, n_labeled, sharp, thd = case
    dev = torch.device("cuda:0")
    outputs, labels = _case(seed, n, n_labeled, sharp)
    with torch.no_grad():
        valid = int((((torch.softmax(outputs, 1).max(1)[0] > thd / 100) | (labels < 10))).sum())
    g = torch.Generator().manual_seed(seed + 1)
    draw = torch.zeros(0, dtype=torch.int64) if valid == 0 else \
        torch.randint(0, valid, (1000 if valid >= 1000 else 1000 - valid,), generator=g)
    (lk, gk, pk, slc), (lt, gt, pt, _) = _both_forms(dev, outputs, labels, thd, slice_draw=draw)
    assert abs(lk - lt) <= 2e-6 * max(1.0, abs(lt)), (lk, lt)
    assert float((pk - pt).abs().max()) <= 2e-6 * max(1.0, float(pt.abs().max()))
    assert float((gk - gt).abs().max()) <= 2e-5 * max(float(gt.abs().max()), 1e-12)
    if valid > 0:
        # the slice: position r in the list of valid points
        with torch.no_grad():
            cert = ((torch.softmax(outputs, 1).max(1)[0] > thd / 100) | (labels < 10))
            lst = torch.where(cert)[0]
            r = draw if valid >= 1000 else torch.cat((torch.arange(valid), draw))
        assert torch.equal(slc, lst[r])


@pytest.mark.gpu
def test_contrast_device_draw_full_size():
    """400 000 points (DALES batch): the device-side draw from uniforms picks the same points as the prefix-sum formula
    on the same uniforms, duplicates of a point in the slice included; loss and gradient agree with the torch-op form"""
    dev = torch.device("cuda:0")
    n = 400_000
    outputs, labels = _case(11, n, 2000, 1.5)
    (lk, gk, pk, slc), (lt, gt, pt, _) = _both_forms(dev, outputs, labels, 40, seed=123)
    assert abs(lk - lt) <= 5e-6 * max(1.0, abs(lt)), (lk, lt)
    assert float((gk - gt).abs().max()) <= 5e-5 * float(gt.abs().max())
    # the draw itself, from the same generator state
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    u = torch.rand(1000, device=dev)
    o = outputs.to(dev)
    cert = (torch.softmax(o, 1).max(1)[0] > 0.4) | (labels.to(dev) < 10)
    cs = torch.cumsum(cert.to(torch.int64), 0)
    nv = cs[-1]
    r = torch.minimum((u * nv.float()).floor().long(), nv - 1)
    want = torch.searchsorted(cs, r + 1)
    assert torch.equal(slc.to(dev), want)


@pytest.mark.gpu
def test_contrast_slice_duplicates_gradient():
    """few valid points: the slice repeats points (:450-454); their gradients add up onto the same logits row"""
    dev = torch.device("cuda:0")
    outputs, labels = _case(21, 600, 12, 0.02)
    with torch.no_grad():
        valid = int((((torch.softmax(outputs, 1).max(1)[0] > 0.9) | (labels < 10))).sum())
    assert 0 < valid < 50
    draw = torch.randint(0, valid, (1000 - valid,), generator=torch.Generator().manual_seed(3))
    (lk, gk, _, slc), (lt, gt, _, _) = _both_forms(dev, outputs, labels, 90, slice_draw=draw)
    assert len(torch.unique(slc)) == valid
    assert abs(lk - lt) <= 2e-6 * max(1.0, abs(lt))
    assert float((gk - gt).abs().max()) <= 2e-5 * float(gt.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("n,c", [(37, 9), (300, 16), (1000, 4), (5000, 13)])
def test_contrast_small_and_wide_inputs(n, c):
    """fewer points than one workgroup handles, fewer points than slice slots, the widest supported logits (C = 16): the
    kernels against the torch-op form"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n + c)
    outputs = torch.randn(n, c, generator=g) * 1.5
    labels = torch.full((n,), 100, dtype=torch.int64)
    idx = torch.randperm(n, generator=g)[:max(1, n // 5)]
    labels[idx] = torch.randint(0, min(c, 9), (idx.numel(),), generator=g)
    (lk, gk, pk, slc), (lt, gt, pt, _) = _both_forms(dev, outputs, labels, 30, seed=7)
    assert slc.shape == (1000,) and int(slc.max()) < n
    assert abs(lk - lt) <= 5e-6 * max(1.0, abs(lt)), (lk, lt)
    assert float((gk - gt).abs().max()) <= 5e-5 * max(float(gt.abs().max()), 1e-12)


@pytest.mark.gpu
def test_contrast_nothing_valid_gives_zero_loss_and_gradient():
    dev = torch.device("cuda:0")
    outputs, labels = _case(2, 800, 0, 0.01)
    cfg = wcfg.DALESPLConfig()
    cfg.contrast_thd = 99
    o = outputs.to(dev).requires_grad_(True)
    net = KPFCNN.__new__(KPFCNN)
    loss = KPFCNN.contrast_loss(net, o, labels.to(dev), cfg)
    loss.backward()
    assert float(loss) == 0.0 and float(o.grad.abs().max()) == 0.0
