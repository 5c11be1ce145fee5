"""GPU parity: radius neighbours and grid subsampling (HIP, through the C ABI) against the golden
vectors of the reference and against the CPU oracle on seeded inputs.  Integer outputs and
barycentres are compared bit-exact (ties in d2: see conftest.assert_neighbors_equal)."""
import numpy as np
import pytest
import torch

from conftest import assert_neighbors_equal, golden, sphere
from oracle import geom

pytestmark = pytest.mark.gpu


def dev(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def test_neighbors_vs_golden(gpu):
    from weasal_amd import ops
    g = golden("g1_neighbors.npz")
    p, l, sp, sl = g["points"], g["lens"], g["sub_points"], g["sub_lens"]
    P, SP = dev(p, gpu), dev(sp, gpu)
    for r in (0.6, 1.0):
        got = ops.radius_neighbors(P, P, l, l, r, dtype=torch.int32).cpu().numpy()
        assert_neighbors_equal(p, p, got, g["self_r%.1f" % r], bool(g["tiefree_self_r%.1f" % r]))
        got = ops.radius_neighbors(SP, P, sl, l, r, dtype=torch.int32).cpu().numpy()
        assert_neighbors_equal(sp, p, got, g["pool_r%.1f" % r], bool(g["tiefree_pool_r%.1f" % r]))
        got = ops.radius_neighbors(P, SP, l, sl, 2 * r, dtype=torch.int64).cpu().numpy()
        assert_neighbors_equal(p, sp, got, g["up_r%.1f" % (2 * r)], bool(g["tiefree_up_r%.1f" % (2 * r)]))


def test_neighbors_vs_oracle_random(gpu):
    from weasal_amd import ops
    rng = np.random.default_rng(11)
    p = np.concatenate([sphere(rng, 6000, 4.0), sphere(rng, 1, 4.0), sphere(rng, 5000, 4.0, (3.0, 0.5, 0.1))])
    l = np.array([6000, 1, 5000], np.int32)
    P = dev(p, gpu)
    for r in (0.37, 0.6, 1.2):
        want = geom.batch_query(p, p, l, l, r)
        got, counts = ops.radius_neighbors(P, P, l, l, r, dtype=torch.int32, return_counts=True)
        # ties broken by index on both sides -> bit exact
        assert np.array_equal(got.cpu().numpy(), want)
        assert np.array_equal(counts.cpu().numpy(), (want < p.shape[0]).sum(1))
        # cropped int64 form = leading columns
        lim = max(1, want.shape[1] // 2)
        got64 = ops.radius_neighbors(P, P, l, l, r, limit=lim, dtype=torch.int64).cpu().numpy()
        assert got64.dtype == np.int64 and np.array_equal(got64, want[:, :lim])


def test_neighbors_queries_outside_supports(gpu):
    from weasal_amd import ops
    rng = np.random.default_rng(12)
    s = sphere(rng, 3000, 2.0)
    q = np.concatenate([sphere(rng, 500, 3.5), np.array([[50, 50, 50], [-40, 0, 0]], np.float32)])
    want = geom.batch_query(q, s, [502], [3000], 0.8)
    got = ops.radius_neighbors(dev(q, gpu), dev(s, gpu), [502], [3000], 0.8, dtype=torch.int32).cpu().numpy()
    assert np.array_equal(got, want)


def test_neighbors_empty_raises(gpu):
    from weasal_amd import ops
    p = dev(np.array([[0, 0, 0], [10, 10, 10]], np.float32), gpu)
    q = dev(np.array([[5, 5, 5]], np.float32), gpu)
    with pytest.raises(RuntimeError):
        ops.radius_neighbors(q, p, [1], [2], 0.5)
    with pytest.raises(RuntimeError):
        ops.radius_neighbors(p[:0], p, [0], [2], 0.5)


def test_subsample_vs_golden(gpu):
    from weasal_amd import ops
    g = golden("g2_subsample.npz")
    P = dev(g["points"], gpu)
    for dl in (0.3, 0.5, 0.9):
        p, l = ops.grid_subsample(P, g["lens"], dl)
        assert np.array_equal(l, g["l_dl%.1f" % dl])
        assert np.array_equal(p.cpu().numpy(), g["p_dl%.1f" % dl])
    p, l = ops.grid_subsample(P, g["lens"], 0.3, max_p=50)
    assert np.array_equal(l, g["l_dl0.3_maxp50"]) and np.array_equal(p.cpu().numpy(), g["p_dl0.3_maxp50"])
    p, l = ops.grid_subsample(dev(g["points2"], gpu), g["lens2"], 1.7)
    assert np.array_equal(l, g["l2_dl1.7"]) and np.array_equal(p.cpu().numpy(), g["p2_dl1.7"])


def test_subsample_features_labels_vs_golden(gpu):
    from weasal_amd import ops
    g = golden("g3_subsample_fl.npz")
    p, l, f, c = ops.grid_subsample(dev(g["points"], gpu), g["lens"], 0.5, features=dev(g["features"], gpu),
                                    labels=dev(g["labels"], gpu))
    assert np.array_equal(l, g["b_lens"]) and np.array_equal(p.cpu().numpy(), g["b_points"])
    assert np.array_equal(f.cpu().numpy(), g["b_features"])
    assert np.array_equal(c.cpu().numpy(), g["b_labels"])


@pytest.mark.parametrize("n,R,dl", [(20000, 6.0, 0.48), (50000, 10.0, 0.8), (50000, 10.0, 3.2)])
def test_subsample_vs_oracle_random(gpu, n, R, dl):
    """row order (libstdc++ unordered_map iteration order), keys, counts, barycentres: bit exact"""
    from weasal_amd import ops
    rng = np.random.default_rng(n)
    p = np.concatenate([sphere(rng, n, R), sphere(rng, n // 2, R, (1.0, -2.0, 0.3))])
    l = np.array([n, n // 2], np.int32)
    wp, wl, wk, wc = geom.subsample_batch(p, l, sampleDl=dl, with_keys=True)
    gp, gl, gk, gc = ops.grid_subsample(dev(p, gpu), l, dl, return_keys=True)
    assert np.array_equal(gl, wl)
    assert np.array_equal(gk.cpu().numpy().view(np.uint64), wk)
    assert np.array_equal(gc.cpu().numpy(), wc)
    assert np.array_equal(gp.cpu().numpy(), wp)
    # first-seen order: same rows as a set
    fp, fl = ops.grid_subsample(dev(p, gpu), l, dl, reference_order=False)
    assert np.array_equal(fl, wl)
    a = fp.cpu().numpy()
    i0 = 0
    for n_b in wl:
        ga = a[i0:i0 + n_b]
        wa = wp[i0:i0 + n_b]
        assert np.array_equal(ga[np.lexsort(ga.T)], wa[np.lexsort(wa.T)])
        i0 += n_b


def test_rotate_clouds(gpu):
    from weasal_amd import ops
    rng = np.random.default_rng(2)
    p = rng.normal(size=(1000, 3)).astype(np.float32)
    lens = np.array([300, 700], np.int32)
    R = rng.normal(size=(2, 3, 3)).astype(np.float32)
    want = p.copy()
    want[:300] = np.sum(np.expand_dims(p[:300], 2) * R[0], axis=1)          # datasets/common.py:118
    want[300:] = np.sum(np.expand_dims(p[300:], 2) * R[1], axis=1)
    got = ops.rotate_clouds(dev(p, gpu), dev(lens, gpu), dev(R, gpu)).cpu().numpy()
    assert np.array_equal(got, want)
    want_t = p.copy()
    want_t[:300] = np.sum(np.expand_dims(p[:300], 2) * R[0].T, axis=1)      # datasets/common.py:133
    want_t[300:] = np.sum(np.expand_dims(p[300:], 2) * R[1].T, axis=1)
    got = ops.rotate_clouds(dev(p, gpu), dev(lens, gpu), dev(R, gpu), transpose=True).cpu().numpy()
    assert np.array_equal(got, want_t)
    # host-table form (no H2D copy of lengths / matrices)
    assert np.array_equal(ops.rotate_clouds_host(dev(p, gpu), lens, R).cpu().numpy(), want)
    assert np.array_equal(ops.rotate_clouds_host(dev(p, gpu), lens, R, transpose=True).cpu().numpy(), want_t)


def test_full_size_dales_batch(gpu):
    """BASELINE config 3 sizes (8 spheres x 50k, r = 1 m, dl = 0.8): GPU rows equal the oracle's."""
    from weasal_amd import ops
    rng = np.random.default_rng(77)
    p = np.concatenate([sphere(rng, 50000, 10.0) for _ in range(8)])
    l = np.full(8, 50000, np.int32)
    P = dev(p, gpu)
    want = geom.batch_query(p, p, l, l, 1.0)
    got = ops.radius_neighbors(P, P, l, l, 1.0, dtype=torch.int32).cpu().numpy()
    assert np.array_equal(got, want)
    wp, wl = geom.subsample_batch(p, l, sampleDl=0.8)
    gp, gl = ops.grid_subsample(P, l, 0.8)
    assert np.array_equal(gl, wl) and np.array_equal(gp.cpu().numpy(), wp)


def test_numpy_facades_vs_golden(gpu):
    """the drop-in modules with the reference's call signatures (numpy in / numpy out)"""
    from weasal_amd.cpp_wrappers.cpp_neighbors import radius_neighbors as cpp_neighbors
    from weasal_amd.cpp_wrappers.cpp_subsampling import grid_subsampling as cpp_subsampling
    g = golden("g1_neighbors.npz")
    got = cpp_neighbors.batch_query(g["points"].astype(np.float64), g["points"], list(g["lens"]), g["lens"], radius=0.6)
    assert got.dtype == np.int32
    assert_neighbors_equal(g["points"], g["points"], got, g["self_r0.6"], bool(g["tiefree_self_r0.6"]))
    with pytest.raises(RuntimeError, match="^Error$"):
        cpp_neighbors.batch_query(np.array([[9., 9, 9]]), np.zeros((2, 3)), [1], [2], radius=0.1)
    g3 = golden("g3_subsample_fl.npz")
    p, l, f, c = cpp_subsampling.subsample_batch(g3["points"], g3["lens"], features=g3["features"],
                                                 classes=g3["labels"], sampleDl=0.5)
    assert p.dtype == np.float32 and l.dtype == np.int32 and c.dtype == np.int32 and c.shape == g3["b_labels"].shape
    assert np.array_equal(p, g3["b_points"]) and np.array_equal(l, g3["b_lens"])
    assert np.array_equal(f, g3["b_features"]) and np.array_equal(c, g3["b_labels"])
    p1, f1, c1 = cpp_subsampling.subsample(g3["points"][:420], features=g3["features"][:420],
                                           classes=g3["labels"][:420], sampleDl=0.5)
    assert np.array_equal(p1, g3["s_points"]) and np.array_equal(f1, g3["s_features"]) and np.array_equal(c1, g3["s_labels"])
    pf = cpp_subsampling.subsample(g3["points"][:420], features=g3["features"][:420], sampleDl=0.5)
    assert isinstance(pf, tuple) and np.array_equal(pf[0], g3["sf_points"]) and np.array_equal(pf[1], g3["sf_features"])
    only = cpp_subsampling.subsample(g3["points"][:420], sampleDl=0.5)
    assert isinstance(only, np.ndarray) and np.array_equal(only, g3["s_points"])
    with pytest.raises(RuntimeError, match="^Error$"):
        cpp_subsampling.subsample(np.zeros((0, 3), np.float32), sampleDl=0.5)


def test_dropin_import_paths(gpu):
    """with weasal_amd/dropin first on sys.path the reference's import statements resolve here"""
    import importlib
    import os
    import sys
    from conftest import REPO
    d = os.path.join(REPO, "weasal_amd", "dropin")
    sys.path.insert(0, d)
    try:
        for name in [m for m in list(sys.modules) if m.split(".")[0] in ("models", "cpp_wrappers", "kernels")]:
            del sys.modules[name]
        blocks = importlib.import_module("models.blocks")
        rn = importlib.import_module("cpp_wrappers.cpp_neighbors.radius_neighbors")
        gs = importlib.import_module("cpp_wrappers.cpp_subsampling.grid_subsampling")
        kp = importlib.import_module("kernels.kernel_points")
        import weasal_amd.blocks
        assert blocks.KPConv is weasal_amd.blocks.KPConv and callable(rn.batch_query)
        assert callable(gs.subsample_batch) and callable(kp.load_kernels)
    finally:
        sys.path.remove(d)
        for name in [m for m in list(sys.modules) if m.split(".")[0] in ("models", "cpp_wrappers", "kernels")]:
            del sys.modules[name]


def _spawn_facade_worker(conn):
    """runs in a fresh ("spawn") child: its own HIP runtime, the numpy facades with the reference's signatures"""
    import numpy as np
    try:
        from weasal_amd.cpp_wrappers.cpp_neighbors import radius_neighbors as cpp_neighbors
        from weasal_amd.cpp_wrappers.cpp_subsampling import grid_subsampling as cpp_subsampling
        rng = np.random.default_rng(5)
        pts = rng.uniform(-2, 2, size=(2500, 3)).astype(np.float32)
        lens = np.array([1500, 1000], np.int32)
        inds = cpp_neighbors.batch_query(pts, pts, lens, lens, radius=0.55)
        sub = cpp_subsampling.subsample_batch(pts, lens, sampleDl=0.4, max_p=0, verbose=0)
        conn.send(("ok", pts, lens, inds, sub[0], sub[1]))
    except BaseException as e:             # noqa: BLE001
        conn.send(("error", repr(e)))
    conn.close()


@pytest.mark.timeout(600)
def test_facades_from_a_spawn_worker(gpu):
    """The reference calls the two modules from DataLoader worker processes (datasets/common.py:56-74,126-175,185-196).
    A worker started with the "spawn" method is a fresh child with its own HIP runtime: the facades work there while
    the parent (this process) holds an initialised runtime of its own.  (Fork-started workers cannot: they raise,
    tests/test_abi_cpu.py::test_facades_raise_in_a_forked_child_of_a_gpu_parent.)"""
    import multiprocessing as mp
    from oracle import geom
    torch.zeros(1, device=gpu)                               # the parent's runtime is live
    ctx = mp.get_context("spawn")
    parent, child = ctx.Pipe()
    proc = ctx.Process(target=_spawn_facade_worker, args=(child,))
    proc.start()
    assert parent.poll(500), "spawn worker did not answer"
    msg = parent.recv()
    proc.join(60)
    assert msg[0] == "ok", msg
    _, pts, lens, inds, sub_p, sub_b = msg
    want = geom.batch_query(pts, pts, lens, lens, 0.55)
    assert inds.dtype == np.int32
    assert_neighbors_equal(pts, pts, inds, want, False)
    want_p, want_b = geom.subsample_batch(pts, lens, sampleDl=0.4)[:2]
    assert np.array_equal(sub_p, want_p) and np.array_equal(sub_b, want_b)
