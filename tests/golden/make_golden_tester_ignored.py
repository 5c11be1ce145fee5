"""Generates tests/golden/g13_tester_ignored.npz (build container only): the reference tester's handling of IGNORED labels
and of points that never received a vote (utils/tester_PseudoLabel.py:228-258, 287-320): zero columns are inserted into the
vote matrix at the ignored labels' positions before the arg-max, so a point with all-zero votes arg-maxes to column 0, and
the ignored rows / columns are deleted from the confusion.  Votes come from g11 (a sub-cloud with unvisited points); the
confusion / IoU arithmetic is the reference's own utils/metrics.py."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from utils.metrics import IoU_from_confusions, fast_confusion  # noqa: E402  (the reference's own file)

g = np.load(os.path.join(HERE, "g11_tester.npz"))
rng = np.random.default_rng(13)
test_probs, proj = g["test_probs"], g["proj"]
C = test_probs.shape[1]
out = {}
for tag, label_values, ignored in (("first", np.arange(C + 1), [0]),                 # DALES: label 0 = unclassified, ignored
                                   ("middle", np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 11]), [5])):
    label_values = np.asarray(label_values, np.int64)
    targets_sub = label_values[rng.integers(0, len(label_values), size=test_probs.shape[0])].astype(np.int32)
    targets_full = label_values[rng.integers(0, len(label_values), size=proj.shape[0])].astype(np.int32)
    for name, probs, targets in (("sub", np.array(test_probs, copy=True), targets_sub),
                                 ("full", test_probs[proj, :], targets_full)):
        # tester_PseudoLabel.py:231-241 / 287-296
        for l_ind, label_value in enumerate(label_values):
            if label_value in ignored:
                probs = np.insert(probs, l_ind, 0, axis=1)
        preds = label_values[np.argmax(probs, axis=1)].astype(np.int32)
        Cm = fast_confusion(targets, preds, label_values)
        # :246-250 / 303-307
        for l_ind, label_value in reversed(list(enumerate(label_values))):
            if label_value in ignored:
                Cm = np.delete(Cm, l_ind, axis=0)
                Cm = np.delete(Cm, l_ind, axis=1)
        out["%s/%s/targets" % (tag, name)] = targets
        out["%s/%s/preds" % (tag, name)] = preds
        out["%s/%s/conf" % (tag, name)] = Cm
        out["%s/%s/iou" % (tag, name)] = IoU_from_confusions(Cm)
    out["%s/label_values" % tag] = label_values
    out["%s/ignored" % tag] = np.asarray(ignored, np.int64)
    print(tag, "unvisited sub points:", int((test_probs.sum(1) == 0).sum()), "predicted as ignored:",
          int(np.isin(out["%s/sub/preds" % tag], ignored).sum()))
np.savez_compressed(os.path.join(HERE, "g13_tester_ignored.npz"), **out)
