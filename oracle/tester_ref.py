"""oracle/tester_ref.py -- TEST INFRASTRUCTURE ONLY.

numpy / sklearn restatement of the reference's tester arithmetic for the fixtures of tests/golden/g11_tester.npz:
  votes         utils/tester_PseudoLabel.py:168-194  (softmax, radius mask, test_probs smoothing, sphere by sphere)
  projection    datasets/DALES_PseudoLabel.py:888-892 (sklearn.neighbors.KDTree(leaf_size=10).query, the library call of
                the reference itself; sklearn is importable in this container)
  potentials    datasets/DALES_PseudoLabel.py:335-350 (KDTree.query_radius(return_distance=True), Tukey weights)
The confusion / IoU functions are imported from the reference's own utils/metrics.py by tests/golden/make_golden_tester.py
when the fixture is generated; nothing here is used by the product.
"""
import numpy as np


def softmax(x):
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def vote_update(test_probs, logits, points, lengths, in_inds, cloud_inds, in_radius, test_radius_ratio, test_smooth):
    stacked_probs = softmax(logits.astype(np.float32)).astype(np.float32)
    i0 = 0
    for b_i, length in enumerate(lengths):
        pts = points[i0:i0 + length]
        probs = stacked_probs[i0:i0 + length]
        inds = in_inds[i0:i0 + length]
        c_i = cloud_inds[b_i]
        if 0 < test_radius_ratio < 1:
            mask = np.sum(pts ** 2, axis=1) < (test_radius_ratio * in_radius) ** 2
            inds = inds[mask]
            probs = probs[mask]
        test_probs[c_i][inds] = test_smooth * test_probs[c_i][inds] + (1 - test_smooth) * probs
        i0 += length
    return test_probs


def projection_indices(points, sub_points):
    from sklearn.neighbors import KDTree
    tree = KDTree(sub_points, leaf_size=10)
    idxs = tree.query(points, return_distance=False)
    return np.squeeze(idxs).astype(np.int32)


def potentials_update(pot_points, potentials, center, in_radius):
    from sklearn.neighbors import KDTree
    tree = KDTree(pot_points, leaf_size=10)
    pot_inds, dists = tree.query_radius(center.reshape(1, -1), r=in_radius, return_distance=True)
    d2s = np.square(dists[0])
    pot_inds = pot_inds[0]
    tukeys = np.square(1 - d2s / np.square(in_radius))
    tukeys[d2s > np.square(in_radius)] = 0
    potentials = potentials.copy()
    potentials[pot_inds] += tukeys
    return potentials, int(np.argmin(potentials))
