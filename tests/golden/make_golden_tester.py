"""Generates tests/golden/g11_tester.npz (run once in the build container; the reference cannot travel):
  * confusion matrices and IoUs from the reference's OWN utils/metrics.py (imported from /root/reference; pure numpy),
  * votes / projection / potentials from oracle/tester_ref.py, whose KDTree calls are the reference's own library calls
    (sklearn.neighbors.KDTree with the reference's arguments).
Usage: python tests/golden/make_golden_tester.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
from oracle import tester_ref  # noqa: E402
from utils.metrics import IoU_from_confusions, fast_confusion  # noqa: E402  (the reference's own file)

rng = np.random.default_rng(11)
C = 9
# one "full" cloud, its grid-subsampled version (barycentres) by the CPU oracle, two spheres of votes
from oracle import geom  # noqa: E402
full = rng.uniform(-6, 6, size=(20000, 3)).astype(np.float32)
full[:, 2] *= 0.25
dl = 0.6
sub = geom.subsample_batch(full, np.array([len(full)], np.int32), sampleDl=dl)[0]
proj = tester_ref.projection_indices(full, sub)
full_labels = rng.integers(0, C, size=len(full)).astype(np.int32)
sub_labels = rng.integers(0, C, size=len(sub)).astype(np.int32)
in_radius = 4.0
test_probs = [np.zeros((len(sub), C), np.float32)]
lengths, points, inds, logits = [], [], [], []
all_state = []
for it in range(3):                                     # three batches of two spheres: overlapping votes, smoothing chain
    lens_b, pts_b, ind_b, log_b = [], [], [], []
    for s in range(2):
        center = sub[rng.integers(0, len(sub))]
        d2 = ((sub - center) ** 2).sum(1)
        sel = np.nonzero(d2 < in_radius ** 2)[0]
        rng.shuffle(sel)
        lens_b.append(len(sel))
        pts_b.append((sub[sel] - center).astype(np.float32))
        ind_b.append(sel.astype(np.int64))
        log_b.append(rng.normal(size=(len(sel), C)).astype(np.float32) * 2)
    lens_b = np.array(lens_b, np.int32)
    pts_b, ind_b, log_b = np.concatenate(pts_b), np.concatenate(ind_b), np.concatenate(log_b)
    test_probs = tester_ref.vote_update(test_probs, log_b, pts_b, lens_b, ind_b, np.array([0, 0]), in_radius, 0.7, 0.95)
    lengths.append(lens_b); points.append(pts_b); inds.append(ind_b); logits.append(log_b)
label_values = np.arange(C).astype(np.int64)
preds_sub = label_values[np.argmax(test_probs[0], axis=1)].astype(np.int32)
conf_sub = fast_confusion(sub_labels, preds_sub, label_values)
proj_probs = test_probs[0][proj, :]
preds_full = label_values[np.argmax(proj_probs, axis=1)].astype(np.int32)
conf_full = fast_confusion(full_labels, preds_full, label_values)
iou_full = IoU_from_confusions(conf_full)
val_prop = np.bincount(full_labels, minlength=C).astype(np.float32)
Cs = conf_sub.astype(np.float32)
Cs *= np.expand_dims(val_prop / (np.sum(Cs, axis=1) + 1e-6), 1)
iou_sub = IoU_from_confusions(Cs)
# non-contiguous label values (the DALES case: an ignored label in the middle) through the reference's label map
lv2 = np.array([0, 1, 2, 3, 5, 6, 7, 8, 10], np.int64)
t2 = lv2[rng.integers(0, C, size=5000)].astype(np.int32)
p2 = lv2[rng.integers(0, C, size=5000)].astype(np.int32)
conf2 = fast_confusion(t2, p2, lv2)
# potentials
pot_points = geom.subsample_batch(sub, np.array([len(sub)], np.int32), sampleDl=in_radius / 10)[0]
pots0 = rng.random(len(pot_points)) * 1e-3
center = pot_points[rng.integers(0, len(pot_points))].astype(np.float64) + rng.normal(scale=in_radius / 10, size=3)
pots1, argmin1 = tester_ref.potentials_update(pot_points, pots0, center, in_radius)
out = dict(full=full, sub=sub, dl=np.float32(dl), proj=proj, full_labels=full_labels, sub_labels=sub_labels,
           in_radius=np.float32(in_radius), test_probs=test_probs[0], preds_sub=preds_sub, conf_sub=conf_sub,
           preds_full=preds_full, conf_full=conf_full, iou_full=iou_full, iou_sub=iou_sub, val_prop=val_prop,
           lv2=lv2, t2=t2, p2=p2, conf2=conf2, pot_points=pot_points, pots0=pots0, center=center, pots1=pots1,
           argmin1=np.int64(argmin1))
for i in range(3):
    out["lengths_%d" % i], out["points_%d" % i] = lengths[i], points[i]
    out["inds_%d" % i], out["logits_%d" % i] = inds[i], logits[i]
np.savez_compressed(os.path.join(HERE, "g11_tester.npz"), **out)
print("wrote g11_tester.npz", {k: getattr(v, "shape", None) for k, v in out.items() if k in ("full", "sub", "proj", "pot_points")})
