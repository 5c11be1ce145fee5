"""Forward-only callers of the hot path: nearest-neighbour projection, the voting test loop and the sampler potentials
(second half).

Reference: utils/tester_PseudoLabel.py:149-447 (`ModelTester.cloud_segmentation_test`), utils/metrics.py:35-110,204-230
(`fast_confusion`, `IoU_from_confusions`), datasets/DALES_PseudoLabel.py:888-892 (re-projection indices: `KDTree.query`
of every full-cloud point against the sub-sampled cloud) and :335-350 (Tukey update of the sampling potentials).
The reference does these on the CPU with sklearn KDTrees that it unpickles from `input_{dl}/`; those pickles are code-
bearing files and are not read here.  The searches run on the GPU instead:

  * `nearest_projection`  = the K1 radius search with one column (rows are sorted by distance, so column 0 is the nearest
    sub-cloud point).  A point of a grid-subsampled cloud is never farther than one cell diagonal (dl * sqrt(3)) from its
    cell's barycentre; rows that find nothing inside the radius are searched again with a doubled radius.
  * `VoteAccumulator`     = per-cloud class probabilities resident in HBM; one fused kernel per sphere does softmax + radius
    mask + exponential smoothing at the sphere's input indices (ws_vote_update); re-projection + arg-max + confusion matrix
    in one kernel (ws_project_confusion).
  * `update_potentials`   = ws_potentials_update (float64 Tukey weights around the sphere centre + the new minimum).
`fast_confusion` / `IoU_from_confusions` are host numpy with the reference's semantics (they run once per vote on a
C x C matrix).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops
from ._lib import check, current_stream, ptr


# ---------------------------------------------------------------------------------------------------------------
# metrics (utils/metrics.py)
# ---------------------------------------------------------------------------------------------------------------
def fast_confusion(true, pred, label_values=None):
    """confusion matrix [C, C] (rows = truth), utils/metrics.py:35-110"""
    true = np.squeeze(np.asarray(true))
    pred = np.squeeze(np.asarray(pred))
    if true.ndim != 1:
        raise ValueError('Truth values are stored in a {:d}D array instead of 1D array'.format(true.ndim))
    if pred.ndim != 1:
        raise ValueError('Prediction values are stored in a {:d}D array instead of 1D array'.format(pred.ndim))
    if true.dtype not in [np.int32, np.int64]:
        raise ValueError('Truth values are {:s} instead of int32 or int64'.format(str(true.dtype)))
    if pred.dtype not in [np.int32, np.int64]:
        raise ValueError('Prediction values are {:s} instead of int32 or int64'.format(str(pred.dtype)))
    true = true.astype(np.int32)
    pred = pred.astype(np.int32)
    if label_values is None:
        label_values = np.unique(np.hstack((true, pred)))
    else:
        label_values = np.asarray(label_values)
        if label_values.dtype not in [np.int32, np.int64]:
            raise ValueError('label values are {:s} instead of int32 or int64'.format(str(label_values.dtype)))
        if len(np.unique(label_values)) < len(label_values):
            raise ValueError('Given labels are not unique')
    label_values = np.sort(label_values)
    num_classes = len(label_values)
    if not (label_values[0] == 0 and label_values[-1] == num_classes - 1):
        if label_values[0] < 0:
            raise ValueError('Unsupported negative classes')
        label_map = np.zeros((label_values[-1] + 1,), dtype=np.int32)
        for k, v in enumerate(label_values):
            label_map[v] = k
        pred = label_map[pred]
        true = label_map[true]
    vec_conf = np.bincount(true * num_classes + pred)
    if vec_conf.shape[0] < num_classes ** 2:
        vec_conf = np.pad(vec_conf, (0, num_classes ** 2 - vec_conf.shape[0]), 'constant')
    return vec_conf.reshape((num_classes, num_classes))


def IoU_from_confusions(confusions):
    """per-class IoU from confusion matrices [..., C, C]; absent classes take the mean (utils/metrics.py:204-230)"""
    confusions = np.asarray(confusions)
    TP = np.diagonal(confusions, axis1=-2, axis2=-1)
    TP_plus_FN = np.sum(confusions, axis=-1)
    TP_plus_FP = np.sum(confusions, axis=-2)
    IoU = TP / (TP_plus_FP + TP_plus_FN - TP + 1e-6)
    mask = TP_plus_FN < 1e-3
    counts = np.sum(1 - mask, axis=-1, keepdims=True)
    mIoU = np.sum(IoU, axis=-1, keepdims=True) / (counts + 1e-6)
    IoU += mask * mIoU
    return IoU


# ---------------------------------------------------------------------------------------------------------------
# nearest-neighbour projection (DALES_PseudoLabel.py:888-892)
# ---------------------------------------------------------------------------------------------------------------
def nearest_projection(points, sub_points, radius, chunk=1 << 22):
    """int32 [N]: index of the sub-cloud point nearest to every point (KDTree.query(points, k=1) of the reference), on the
    device.  `radius`: a distance within which every point is expected to have a neighbour (dl * sqrt(3) for a cloud that
    was grid-subsampled with cell dl); points that find none are searched again with the radius doubled."""
    ops._need_cuda(points, sub_points)
    P = ops._f32c(points)
    S = ops._f32c(sub_points)
    n, m = P.shape[0], S.shape[0]
    if m == 0:
        raise RuntimeError("nearest_projection: empty sub-cloud")
    out = torch.empty(n, dtype=torch.int32, device=P.device)
    for a in range(0, n, chunk):
        q = P[a:a + chunk]
        todo = torch.arange(q.shape[0], device=P.device)
        r = float(radius)
        res = torch.full((q.shape[0],), m, dtype=torch.int64, device=P.device)
        for _ in range(24):
            qq = q[todo] if todo.shape[0] != q.shape[0] else q
            try:
                col = ops.radius_neighbors(qq, S, [qq.shape[0]], [m], r, limit=1, dtype=torch.int64)[:, 0]
            except _lib.WeasalHipError as e:
                if "status 4" not in str(e):
                    raise
                col = torch.full((qq.shape[0],), m, dtype=torch.int64, device=P.device)     # nobody found anything
            res[todo] = col
            todo = todo[col >= m]
            if todo.shape[0] == 0:
                break
            r *= 2.0
        else:
            raise RuntimeError("nearest_projection: points without any sub-cloud point within %g" % r)
        out[a:a + chunk] = res.to(torch.int32)
    return out


# ---------------------------------------------------------------------------------------------------------------
# votes (tester_PseudoLabel.py:168-194, 283-320)
# ---------------------------------------------------------------------------------------------------------------
class VoteAccumulator:
    """test_probs of the reference's tester, one [N_cloud, C] float32 tensor per cloud, kept on the device"""

    def __init__(self, cloud_sizes, num_classes, device, test_smooth=0.95):
        self.c = int(num_classes)
        self.smooth = float(test_smooth)
        self.device = device
        self.probs = [torch.zeros((int(n), self.c), dtype=torch.float32, device=device) for n in cloud_sizes]

    def update(self, logits, points0, lengths, input_inds, cloud_inds, radius_mask=0.0):
        """one batch of spheres: logits [sum n, C] (network output), points0 = batch.points[0] (sphere-centred), lengths /
        cloud_inds per sphere (host or device), input_inds [sum n] = index of every point in its cloud.
        radius_mask: test_radius_ratio * in_radius when 0 < test_radius_ratio < 1, else 0."""
        lib = _lib.lib()
        ops._need_cuda(logits)
        lg = logits.detach().float().contiguous()
        pts = ops._f32c(points0)
        inds = input_inds.detach().to(torch.int64).contiguous()
        lens = [int(v) for v in (lengths.tolist() if hasattr(lengths, "tolist") else lengths)]
        clouds = [int(v) for v in (cloud_inds.tolist() if hasattr(cloud_inds, "tolist") else cloud_inds)]
        i0 = 0
        for n, ci in zip(lens, clouds):               # spheres one after the other, like the reference's loop
            p = self.probs[ci]
            check(lib.ws_vote_update(ptr(lg[i0:i0 + n]), n, self.c, ptr(pts[i0:i0 + n]), float(radius_mask),
                                     ptr(inds[i0:i0 + n]), ptr(p), p.shape[0], self.smooth, current_stream()))
            i0 += n

    def predictions(self, cloud, proj=None, labels=None, label_values=None, ignored_labels=()):
        """-> (preds int32 [M] as label VALUES, confusion int64 [C, C] or None): arg-max of the (re-projected) votes and
        the confusion against `labels` (label values; mapped to positions in the sorted label_values like fast_confusion).

        `ignored_labels` (tester_PseudoLabel.py:228-250, 287-307): `label_values` then names ALL labels of the dataset, the
        votes have one column per label that is not ignored.  The reference inserts a zero column per ignored label before
        the arg-max -- which only matters for a point without any vote (all zeros: outside every sphere's 0.7-radius mask):
        it arg-maxes to the FIRST label, ignored or not -- and deletes the ignored rows and columns from the confusion, so a
        point whose target or prediction is an ignored label is not counted.  Both are reproduced: such a point is
        predicted as label_values[0], and leaves the confusion when that label is ignored."""
        lib = _lib.lib()
        p = self.probs[cloud]
        dev = p.device
        m = p.shape[0] if proj is None else proj.shape[0]
        lv_all = np.sort(np.asarray(label_values if label_values is not None else np.arange(self.c))).astype(np.int64)
        ign = np.asarray(sorted(ignored_labels), np.int64)
        lv = np.array([v for v in lv_all if v not in ign], np.int64)                 # the labels the vote columns stand for
        if len(lv) != self.c:
            raise ValueError("label_values minus ignored_labels must name the %d classes of the votes" % self.c)
        pj = None if proj is None else proj.detach().to(torch.int32).contiguous()
        preds = torch.empty(m, dtype=torch.int32, device=dev)
        first_ignored = len(ign) > 0 and lv_all[0] in ign
        voted = None
        if len(ign) > 0:
            rows = p if pj is None else p[pj.long()]
            voted = rows.amax(dim=1) > 0                                            # (votes are probabilities: >= 0)
        conf = lab = None
        if labels is not None:
            lut = torch.full((int(max(lv_all.max(), int(labels.max())) + 2),), -1, dtype=torch.int32, device=dev)
            lut[torch.from_numpy(lv).to(dev)] = torch.arange(self.c, dtype=torch.int32, device=dev)
            lab = lut[labels.detach().to(torch.int64).clamp_min(0)]
            if first_ignored:
                lab = torch.where(voted, lab, torch.full_like(lab, -1))            # predicted as an ignored label: not counted
            lab = lab.contiguous()
            conf = torch.zeros((self.c, self.c), dtype=torch.int64, device=dev)
        check(lib.ws_project_confusion(ptr(p), self.c, ptr(pj), m, ptr(lab), ptr(preds), self.c, ptr(conf), current_stream()))
        values = torch.from_numpy(lv.astype(np.int32)).to(dev)[preds.long()]
        if voted is not None:
            values = torch.where(voted, values, torch.full_like(values, int(lv_all[0])))
        return values, conf


# ---------------------------------------------------------------------------------------------------------------
# potentials (DALES_PseudoLabel.py:335-350)
# ---------------------------------------------------------------------------------------------------------------
def update_potentials(pot_points, potentials, center, radius):
    """potentials += Tukey weights of the coarse points within `radius` of `center` (host float64 [3]); returns
    (min value, arg-min) as device scalars.  pot_points [n,3] f32, potentials [n] f64, both on the device."""
    lib = _lib.lib()
    ops._need_cuda(pot_points, potentials)
    pts = ops._f32c(pot_points)
    if potentials.dtype != torch.float64 or not potentials.is_contiguous():
        raise ValueError("potentials must be a contiguous float64 tensor (updated in place)")
    n = pts.shape[0]
    c = np.ascontiguousarray(np.asarray(center, dtype=np.float64).reshape(3))
    out_min = torch.empty(1, dtype=torch.float64, device=pts.device)
    out_arg = torch.empty(1, dtype=torch.int64, device=pts.device)
    scratch = torch.empty(lib.ws_potentials_scratch_bytes(n), dtype=torch.uint8, device=pts.device)
    check(lib.ws_potentials_update(ptr(pts), n, C.c_void_p(c.ctypes.data), float(radius), ptr(potentials), ptr(out_min),
                                   ptr(out_arg), ptr(scratch), current_stream()))
    return out_min, out_arg


# ---------------------------------------------------------------------------------------------------------------
# the test loop (tester_PseudoLabel.py:149-330), forward only
# ---------------------------------------------------------------------------------------------------------------
def cloud_segmentation_test(net, batches, config, votes, test_radius_ratio=0.7, val_proportions=None, sub_labels=None,
                            label_values=None):
    """One pass over `batches` (an iterable of batch objects carrying .points/.lengths/.input_inds/.cloud_inds like the
    reference's CustomBatch): forward, accumulate the votes.  Returns the sub-cloud confusion-based IoUs when `sub_labels`
    (one label tensor per cloud) is given -- tester_PseudoLabel.py:223-262 incl. the class-proportion rescale -- else None."""
    net.eval()
    rm = test_radius_ratio * config.in_radius if 0 < test_radius_ratio < 1 else 0.0
    with torch.no_grad():
        for batch in batches:
            outputs = net(batch, config)
            votes.update(outputs, batch.points[0], batch.lengths[0], batch.input_inds, batch.cloud_inds, radius_mask=rm)
    if sub_labels is None:
        return None
    confs = []
    for i, lab in enumerate(sub_labels):
        _, conf = votes.predictions(i, labels=lab, label_values=label_values)
        confs.append(conf.cpu().numpy())
    Cm = np.sum(np.stack(confs), axis=0).astype(np.float32)
    if val_proportions is not None:
        Cm *= np.expand_dims(np.asarray(val_proportions, np.float32) / (np.sum(Cm, axis=1) + 1e-6), 1)
    return IoU_from_confusions(Cm)
