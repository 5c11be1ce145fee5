"""CPU: the oracle (own restatement) against the golden vectors generated from the reference,
and against oracle/_ref when that prebuilt library is present."""
import numpy as np
import pytest

from conftest import assert_neighbors_equal, golden, sphere
from oracle import geom

pytestmark = pytest.mark.skipif(not geom.have_port(), reason="oracle/libws_oracle.so not built")


def test_neighbors_vs_golden():
    g = golden("g1_neighbors.npz")
    p, l, sp, sl = g["points"], g["lens"], g["sub_points"], g["sub_lens"]
    for r in (0.6, 1.0):
        got = geom.batch_query(p, p, l, l, r)
        assert_neighbors_equal(p, p, got, g["self_r%.1f" % r], bool(g["tiefree_self_r%.1f" % r]))
        got = geom.batch_query(sp, p, sl, l, r)
        assert_neighbors_equal(sp, p, got, g["pool_r%.1f" % r], bool(g["tiefree_pool_r%.1f" % r]))
        got = geom.batch_query(p, sp, l, sl, 2 * r)
        assert_neighbors_equal(p, sp, got, g["up_r%.1f" % (2 * r)], bool(g["tiefree_up_r%.1f" % (2 * r)]))


def test_subsample_vs_golden():
    g = golden("g2_subsample.npz")
    for dl in (0.3, 0.5, 0.9):
        p, l = geom.subsample_batch(g["points"], g["lens"], sampleDl=dl)
        assert np.array_equal(p, g["p_dl%.1f" % dl]) and np.array_equal(l, g["l_dl%.1f" % dl])
    p, l = geom.subsample_batch(g["points"], g["lens"], sampleDl=0.3, max_p=50)
    assert np.array_equal(p, g["p_dl0.3_maxp50"]) and np.array_equal(l, g["l_dl0.3_maxp50"])
    p, l = geom.subsample_batch(g["points2"], g["lens2"], sampleDl=1.7)
    assert np.array_equal(p, g["p2_dl1.7"]) and np.array_equal(l, g["l2_dl1.7"])


def test_subsample_features_labels_vs_golden():
    g = golden("g3_subsample_fl.npz")
    p, l, f, c = geom.subsample_batch(g["points"], g["lens"], features=g["features"], classes=g["labels"], sampleDl=0.5)
    assert np.array_equal(p, g["b_points"]) and np.array_equal(l, g["b_lens"])
    assert np.array_equal(f, g["b_features"]) and np.array_equal(c, g["b_labels"])
    p1, f1, c1 = geom.subsample(g["points"][:420], features=g["features"][:420], classes=g["labels"][:420], sampleDl=0.5)
    assert np.array_equal(p1, g["s_points"]) and np.array_equal(f1, g["s_features"]) and np.array_equal(c1, g["s_labels"])


def test_empty_result_raises():
    p = np.zeros((0, 3), np.float32)
    with pytest.raises(RuntimeError):
        geom.batch_query(p, p, [0], [0], 1.0)
    with pytest.raises(RuntimeError):
        geom.subsample_batch(p, [0], sampleDl=0.5)


@pytest.mark.skipif(not geom.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_port_vs_reference_build():
    rng = np.random.default_rng(3)
    p = np.concatenate([sphere(rng, 4000, 4.0), sphere(rng, 3000, 4.0, (0.5, 0.2, 0.1))])
    l = np.array([4000, 3000], np.int32)
    for r in (0.6, 1.2):
        a, b = geom.batch_query(p, p, l, l, r, "port"), geom.batch_query(p, p, l, l, r, "ref")
        assert_neighbors_equal(p, p, a, b, False)
    for dl in (0.48, 0.96, 1.92):
        a = geom.subsample_batch(p, l, sampleDl=dl, kind="port")
        b = geom.subsample_batch(p, l, sampleDl=dl, kind="ref")
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_pyramid_vs_golden():
    """the CPU pyramid restatement reproduces the reference's segmentation_inputs (golden g7)"""
    from oracle import pyramid_ref
    from test_oracle_cpu_kpconv import _small_config
    g = golden("g7_pyramid.npz")
    np.random.seed(int(g["np_seed"]))
    li = pyramid_ref.segmentation_inputs(_small_config(), g["points"], g["features"], g["labels"], g["lens"],
                                         list(g["limits"]))
    L = 5
    for l in range(L):
        assert np.array_equal(li[l], g["points_%d" % l])
        assert np.array_equal(li[4 * L + l], g["lengths_%d" % l])
        pts_l = g["points_%d" % l]
        assert_neighbors_equal(pts_l, pts_l, li[L + l], g["neighbors_%d" % l], False)
        if l < L - 1:
            nxt = g["points_%d" % (l + 1)]
            assert_neighbors_equal(nxt, pts_l, li[2 * L + l], g["pools_%d" % l], False)
            assert_neighbors_equal(pts_l, nxt, li[3 * L + l], g["upsamples_%d" % l], False)
