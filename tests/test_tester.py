"""Voting tester / re-projection / potentials against tests/golden/g11_tester.npz: confusion
matrices and IoUs from the reference's own utils/metrics.py, votes / projection / potentials from the reference's tester
arithmetic with the reference's own KDTree library calls (tests/golden/make_golden_tester.py).  CPU part: the numpy
metrics of weasal_amd.tester and the oracle restatement; GPU part: the HIP kernels and the K1-based projection."""
import numpy as np
import pytest
import torch

from conftest import golden


def test_metrics_match_the_reference_functions():
    from weasal_amd.tester import IoU_from_confusions, fast_confusion
    g = golden("g11_tester.npz")
    lv = np.arange(9).astype(np.int64)
    assert np.array_equal(fast_confusion(g["sub_labels"], g["preds_sub"], lv), g["conf_sub"])
    assert np.array_equal(fast_confusion(g["full_labels"], g["preds_full"], lv), g["conf_full"])
    assert np.array_equal(fast_confusion(g["t2"], g["p2"], g["lv2"]), g["conf2"])        # label map through an ignored gap
    assert np.allclose(IoU_from_confusions(g["conf_full"]), g["iou_full"], rtol=0, atol=0)
    Cs = g["conf_sub"].astype(np.float32)
    Cs *= np.expand_dims(g["val_prop"] / (np.sum(Cs, axis=1) + 1e-6), 1)
    assert np.array_equal(IoU_from_confusions(Cs), g["iou_sub"])
    with pytest.raises(ValueError):
        fast_confusion(np.zeros(3, np.float32), np.zeros(3, np.int32))


def test_oracle_restatement_reproduces_the_fixture():
    from oracle import tester_ref
    g = golden("g11_tester.npz")
    probs = [np.zeros_like(g["test_probs"])]
    for i in range(3):
        probs = tester_ref.vote_update(probs, g["logits_%d" % i], g["points_%d" % i], g["lengths_%d" % i], g["inds_%d" % i],
                                       np.array([0, 0]), float(g["in_radius"]), 0.7, 0.95)
    assert np.array_equal(probs[0], g["test_probs"])


@pytest.mark.gpu
def test_projection_votes_confusion_potentials_on_the_gpu(gpu):
    from weasal_amd import tester
    g = golden("g11_tester.npz")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    # ---- nearest-neighbour projection = column 0 of the K1 search
    proj = tester.nearest_projection(dev(g["full"]), dev(g["sub"]), float(g["dl"]) * 3 ** 0.5)
    got = proj.cpu().numpy()
    assert got.dtype == np.int32 and got.shape == g["proj"].shape
    diff = np.nonzero(got != g["proj"])[0]
    if len(diff):        # only where two sub-cloud points are equally near in f32 (the tree decides in f64)
        full, sub = g["full"].astype(np.float64), g["sub"].astype(np.float64)
        d_got = ((full[diff] - sub[got[diff]]) ** 2).sum(1)
        d_ref = ((full[diff] - sub[g["proj"][diff]]) ** 2).sum(1)
        assert len(diff) < 1e-3 * len(got) and np.all(np.abs(d_got - d_ref) <= 1e-6 * d_ref)
    # ---- votes: softmax + radius mask + smoothing, sphere by sphere
    votes = tester.VoteAccumulator([g["sub"].shape[0]], 9, gpu, test_smooth=0.95)
    for i in range(3):
        votes.update(dev(g["logits_%d" % i]), dev(g["points_%d" % i]), g["lengths_%d" % i], dev(g["inds_%d" % i]),
                     np.array([0, 0]), radius_mask=0.7 * float(g["in_radius"]))
    p = votes.probs[0].cpu().numpy()
    assert np.abs(p - g["test_probs"]).max() < 2e-6                      # expf vs numpy's exp: last-ulp differences
    untouched = ~np.isin(np.arange(p.shape[0]), np.concatenate([g["inds_%d" % i] for i in range(3)]))
    assert np.all(p[untouched] == 0)
    # ---- predictions and confusions (sub cloud, then re-projected on the full cloud), from the reference's votes
    votes.probs[0].copy_(dev(g["test_probs"]))
    preds, conf = votes.predictions(0, labels=dev(g["sub_labels"]))
    assert np.array_equal(preds.cpu().numpy(), g["preds_sub"]) and np.array_equal(conf.cpu().numpy(), g["conf_sub"])
    preds, conf = votes.predictions(0, proj=dev(g["proj"]), labels=dev(g["full_labels"]))
    assert np.array_equal(preds.cpu().numpy(), g["preds_full"]) and np.array_equal(conf.cpu().numpy(), g["conf_full"])
    assert np.array_equal(tester.IoU_from_confusions(conf.cpu().numpy()), g["iou_full"])
    # ---- potentials: float64 Tukey update + arg-min
    pots = dev(g["pots0"]).clone()
    mn, am = tester.update_potentials(dev(g["pot_points"]), pots, g["center"], float(g["in_radius"]))
    assert np.abs(pots.cpu().numpy() - g["pots1"]).max() <= 4e-16
    assert int(am.item()) == int(g["argmin1"]) and float(mn.item()) == float(pots.cpu().numpy().min())


@pytest.mark.gpu
def test_ignored_labels_and_unvisited_points(gpu):
    """tests/golden/g13_tester_ignored.npz (the reference tester's zero-column insertion and row / column deletion,
    tester_PseudoLabel.py:228-250, 287-307): an ignored label in front of the valid ones (DALES: 0 = unclassified), or in the
    middle; sub-cloud points no sphere ever voted on"""
    from weasal_amd import tester
    g, s = golden("g11_tester.npz"), golden("g13_tester_ignored.npz")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    votes = tester.VoteAccumulator([g["sub"].shape[0]], 9, gpu, test_smooth=0.95)
    votes.probs[0].copy_(dev(g["test_probs"]))
    assert int((g["test_probs"].sum(1) == 0).sum()) > 0
    for tag in ("first", "middle"):
        lv, ign = s[tag + "/label_values"], s[tag + "/ignored"]
        for name, proj in (("sub", None), ("full", dev(g["proj"]))):
            preds, conf = votes.predictions(0, proj=proj, labels=dev(s["%s/%s/targets" % (tag, name)]), label_values=lv,
                                            ignored_labels=ign)
            assert np.array_equal(preds.cpu().numpy(), s["%s/%s/preds" % (tag, name)]), (tag, name)
            assert np.array_equal(conf.cpu().numpy(), s["%s/%s/conf" % (tag, name)]), (tag, name)
            assert np.array_equal(tester.IoU_from_confusions(conf.cpu().numpy()), s["%s/%s/iou" % (tag, name)])
