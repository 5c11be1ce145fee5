import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def sphere(rng, n, R, center=(0, 0, 0)):
    """n points i.i.d. uniform in the ball of radius R (rejection from the cube), f32"""
    pts = np.zeros((0, 3), np.float32)
    while len(pts) < n:
        c = rng.uniform(-R, R, size=(2 * n, 3)).astype(np.float32)
        pts = np.concatenate([pts, c[(c.astype(np.float64) ** 2).sum(1) < R * R]])
    return (pts[:n] + np.asarray(center, np.float32)).astype(np.float32)


def d2_rows(q, s, idx):
    """f32 squared distances of index rows, reference recipe, shadow = +inf"""
    sp = np.concatenate([s, np.full((1, 3), np.float32(np.inf))]).astype(np.float32)
    d = q[:, None, :].astype(np.float32) - sp[idx]
    with np.errstate(invalid="ignore"):
        return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def assert_neighbors_equal(q, s, got, want, tie_free):
    """bit-exact when tie_free; otherwise equal up to permutations inside equal-d2 runs
    (the reference's std::sort order on exact distance ties is implementation-defined)"""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    if tie_free:
        assert np.array_equal(got, want)
        return
    dg, dw = d2_rows(q, s, got), d2_rows(q, s, want)
    assert np.array_equal(np.nan_to_num(dg, posinf=3e38), np.nan_to_num(dw, posinf=3e38))
    assert np.array_equal(np.sort(got, axis=1), np.sort(want, axis=1))
    # rows differ only where a tie exists
    bad = (got != want)
    if bad.any():
        r, c = np.nonzero(bad)
        same = np.zeros(len(r), bool)
        for k in (-1, 1):
            cc = np.clip(c + k, 0, got.shape[1] - 1)
            same |= dg[r, cc] == dg[r, c]
        assert same.all()


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
