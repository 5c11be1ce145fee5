"""oracle/kpconv_ref.py -- TEST INFRASTRUCTURE ONLY.

Plain-PyTorch (CPU) restatement of the reference's KPConv operator sequence, materialising the
same intermediates the reference does.  Used by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; nothing under weasal_amd/ imports it.

Follows, line by line in meaning (not in text):
  kpconv_gather_ref  models/blocks.py:278-367  (shadow point :278, centring :284, differences :295,
                     sq distances :298, deformable min_d2 / in-range filter :301-325, influences
                     :330-346, closest :349-351, zero shadow feature :357, gather :360, weighted
                     features :363, modulations :366-367)
  kpconv_forward     + the kernel contraction blocks.py:370-374
  max_pool_ref / closest_pool_ref   blocks.py:80-111
Pinned by tests/test_oracle_cpu_kpconv.py against the golden vectors g4/g5/g6 generated from the
reference itself (tests/golden/make_golden.py).

``cpu_reference_mode()`` swaps these functions in for weasal_amd.ops' HIP operators so that the
SAME module classes (weasal_amd.blocks / architectures) can be evaluated on the CPU as the
checker / CPU baseline.  It is a test device: the product never enters it.
"""
import contextlib

import torch


def _influence(sq, extent, influence):
    if influence == "constant":
        return torch.ones_like(sq)
    if influence == "linear":
        return torch.clamp(1 - torch.sqrt(sq) / extent, min=0.0)
    if influence == "gaussian":
        sigma = extent * 0.3
        return torch.exp(-sq / (2 * sigma ** 2 + 1e-9))
    raise ValueError("Unknown influence function type (config.KP_influence)")


def kpconv_gather_ref(x, q_pts, s_pts, inds, kernel_points, extent, influence="linear", aggregation="sum",
                      deformed_kp=None, modulations=None, want_min_d2=False):
    """-> (wf [N,K,Ci], min_d2 [N,K] or None)"""
    inds = inds.long()
    s_pad = torch.cat((s_pts, torch.zeros_like(s_pts[:1, :]) + 1e6), 0)
    neighbors = s_pad[inds, :] - q_pts.unsqueeze(1)                       # [N,H,3]
    kp = deformed_kp.unsqueeze(1) if deformed_kp is not None else kernel_points
    differences = neighbors.unsqueeze(2) - kp                              # [N,H,K,3]
    sq = torch.sum(differences ** 2, dim=3)                                # [N,H,K]
    min_d2 = None
    keep = None
    if deformed_kp is not None:
        min_d2, _ = torch.min(sq, dim=1)
        keep = torch.any(sq < extent ** 2, dim=2)                          # [N,H]
    w = _influence(sq, extent, influence).transpose(1, 2)                  # [N,K,H]
    if aggregation == "closest":
        nn1 = torch.argmin(sq, dim=2)
        w = w * torch.transpose(torch.nn.functional.one_hot(nn1, sq.shape[2]), 1, 2)
    elif aggregation != "sum":
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    x_pad = torch.cat((x, torch.zeros_like(x[:1, :])), 0)
    neighb_x = x_pad[inds]                                                 # [N,H,Ci]
    if keep is not None:
        # neighbours without any kernel point in range are re-pointed to the shadow row (:316-325)
        neighb_x = neighb_x * keep.unsqueeze(2).to(neighb_x.dtype)
    wf = torch.matmul(w, neighb_x)                                         # [N,K,Ci]
    if modulations is not None:
        wf = wf * modulations.unsqueeze(2)
    return wf, (min_d2 if want_min_d2 or deformed_kp is not None else None)


def kpconv_forward(q_pts, s_pts, inds, x, weights, kernel_points, extent, influence="linear", aggregation="sum"):
    """rigid KPConv output [N,Co] (blocks.py:278-374)"""
    wf, _ = kpconv_gather_ref(x, q_pts, s_pts, inds, kernel_points, extent, influence, aggregation)
    return torch.sum(torch.matmul(wf.permute(1, 0, 2), weights), dim=0)


def max_pool_ref(x, inds):
    x_pad = torch.cat((x, torch.zeros_like(x[:1, :])), 0)
    return torch.max(x_pad[inds.long()], 1)[0]


def closest_pool_ref(x, inds):
    x_pad = torch.cat((x, torch.zeros_like(x[:1, :])), 0)
    return x_pad[inds.long()[:, 0]]


def matmul_ref(x, b):
    return torch.matmul(x, b)


def linear_ref(x, weight):
    """nn.Linear without bias (blocks.py:490,497)"""
    return torch.matmul(x, weight.t())


def matmul_epilogue_ref(x, b, bias=None, residual=None, slope=None):
    """x @ b, then the bias of BatchNormBlock (blocks.py:465), the residual sum and LeakyReLU (blocks.py:497-500,709)"""
    y = torch.matmul(x, b)
    if bias is not None:
        y = y + bias
    if residual is not None:
        y = y + residual
    return y if slope is None else torch.nn.functional.leaky_relu(y, slope)


def cross_entropy_ref(logits, labels, lut=None, weight=None):
    """models/architectures.py:362-373: labels -> positions (-1 ignored), CrossEntropyLoss(weight, ignore_index=-1) over
    the transposed, unsqueezed logits [1, C, N]"""
    if lut is not None:
        idx = torch.where((labels >= 0) & (labels < lut.shape[0] - 1), labels, torch.full_like(labels, lut.shape[0] - 1))
        target = lut.to(labels.device)[idx]
    else:
        target = labels
    return torch.nn.functional.cross_entropy(logits.transpose(0, 1).unsqueeze(0), target.unsqueeze(0), weight=weight,
                                             ignore_index=-1)


@contextlib.contextmanager
def cpu_reference_mode():
    """Evaluate weasal_amd.blocks / architectures modules with the restatements above (CPU)."""
    from weasal_amd import ops
    names = ("kpconv_gather", "max_pool", "closest_pool", "matmul", "linear", "matmul_epilogue", "cross_entropy")
    saved = [getattr(ops, n) for n in names]
    for n, f in zip(names, (kpconv_gather_ref, max_pool_ref, closest_pool_ref, matmul_ref, linear_ref,
                            matmul_epilogue_ref, cross_entropy_ref)):
        setattr(ops, n, f)
    try:
        yield
    finally:
        for n, f in zip(names, saved):
            setattr(ops, n, f)
