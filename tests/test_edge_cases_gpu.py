"""GPU: shapes off the fast paths -- odd channel counts (scalar row pieces), more than 64 / 128
neighbour columns (several column chunks, the wide sort slab), dense influence modes (entry-pool
overflow -> sub-chunk redo), empty batch elements, many batch elements -- against the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import sphere
from oracle import geom, kpconv_ref

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("ci,h,influence,aggregation", [(6, 130, "linear", "sum"), (5, 40, "constant", "sum"),
                                                        (20, 70, "gaussian", "sum"), (32, 64, "constant", "closest"),
                                                        (1, 17, "linear", "sum"), (48, 200, "linear", "sum"),
                                                        (64, 96, "constant", "sum")])
def test_kpconv_odd_shapes_vs_torch_restatement(gpu, ci, h, influence, aggregation):
    from weasal_amd import ops
    rng = np.random.default_rng(ci * 1000 + h)
    n = 700
    pts = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    inds = rng.integers(0, n + 1, size=(n, h))                 # includes shadow entries
    inds[::9, -3:] = n
    kp = (rng.normal(size=(15, 3)) * 0.4).astype(np.float32)
    x = rng.normal(size=(n, ci)).astype(np.float32)
    dy = rng.normal(size=(n, 15, ci)).astype(np.float32)
    ext = 0.9
    t = lambda a: torch.from_numpy(a)
    xc = t(x).requires_grad_(True)
    want, _ = kpconv_ref.kpconv_gather_ref(xc, t(pts), t(pts), t(inds), t(kp), ext, influence, aggregation)
    (want * t(dy)).sum().backward()
    xg = t(x).to(gpu).requires_grad_(True)
    got, _ = ops.kpconv_gather(xg, t(pts).to(gpu), t(pts).to(gpu), t(inds).to(gpu), t(kp).to(gpu), ext, influence, aggregation)
    (got * t(dy).to(gpu)).sum().backward()
    assert rel(got, want) < 1e-4
    assert rel(xg.grad, xc.grad) < 1e-4


def test_kpconv_different_query_and_support_sets(gpu):
    from weasal_amd import ops
    rng = np.random.default_rng(5)
    nq, ns, h, ci = 333, 901, 33, 12
    q = rng.uniform(-1, 1, size=(nq, 3)).astype(np.float32)
    s = rng.uniform(-1, 1, size=(ns, 3)).astype(np.float32)
    inds = rng.integers(0, ns + 1, size=(nq, h))
    kp = (rng.normal(size=(15, 3)) * 0.4).astype(np.float32)
    x = rng.normal(size=(ns, ci)).astype(np.float32)
    t = lambda a: torch.from_numpy(a)
    xc = t(x).requires_grad_(True)
    want, _ = kpconv_ref.kpconv_gather_ref(xc, t(q), t(s), t(inds), t(kp), 0.8)
    want.square().sum().backward()
    xg = t(x).to(gpu).requires_grad_(True)
    got, _ = ops.kpconv_gather(xg, t(q).to(gpu), t(s).to(gpu), t(inds).to(gpu), t(kp).to(gpu), 0.8)
    got.square().sum().backward()
    assert rel(got, want) < 1e-4 and rel(xg.grad, xc.grad) < 1e-4


def test_neighbors_dense_rows_over_128(gpu):
    """rows of ~300 neighbours: wide sort slab, and the deferred search repeats itself synchronously"""
    from weasal_amd import ops
    rng = np.random.default_rng(8)
    p = sphere(rng, 3000, 1.0)
    lens = np.array([3000], np.int32)
    want = geom.batch_query(p, p, lens, lens, 0.5)
    assert want.shape[1] > 128
    P = torch.from_numpy(p).to(gpu)
    got = ops.radius_neighbors(P, P, lens, lens, 0.5, dtype=torch.int32).cpu().numpy()
    assert np.array_equal(got, want)
    got = ops.radius_neighbors(P, P, lens, lens, 0.5, limit=200, dtype=torch.int64).cpu().numpy()
    assert np.array_equal(got, want[:, :200])
    d = ops.DeferredSearches(gpu)
    d.add(P, P, lens, lens, 0.5, 150)
    d.add(P, P, lens, lens, 0.05, 150)        # few neighbours: trimmed to the true width
    a, b = d.finish()
    assert np.array_equal(a.cpu().numpy(), want[:, :150])
    assert np.array_equal(b.cpu().numpy(), geom.batch_query(p, p, lens, lens, 0.05).astype(np.int64))


def test_empty_and_many_batch_elements(gpu):
    from weasal_amd import ops
    rng = np.random.default_rng(9)
    # an empty element in the middle
    p = np.concatenate([sphere(rng, 900, 2.0), sphere(rng, 700, 2.0, (0.3, 0, 0))])
    lens = np.array([900, 0, 700], np.int32)
    P = torch.from_numpy(p).to(gpu)
    want = geom.batch_query(p, p, lens, lens, 0.5)
    assert np.array_equal(ops.radius_neighbors(P, P, lens, lens, 0.5, dtype=torch.int32).cpu().numpy(), want)
    wp, wl = geom.subsample_batch(p, lens, sampleDl=0.4)
    gp, gl = ops.grid_subsample(P, lens, 0.4)
    assert np.array_equal(gl, wl) and np.array_equal(gp.cpu().numpy(), wp)
    # 70 small elements (beyond the 48 / 64 entry kernel-argument tables)
    sizes = rng.integers(20, 60, size=70).astype(np.int32)
    p = np.concatenate([sphere(rng, int(n), 1.0, (0.01 * i, 0, 0)) for i, n in enumerate(sizes)])
    P = torch.from_numpy(p).to(gpu)
    want = geom.batch_query(p, p, sizes, sizes, 0.4)
    assert np.array_equal(ops.radius_neighbors(P, P, sizes, sizes, 0.4, dtype=torch.int32).cpu().numpy(), want)
    wp, wl = geom.subsample_batch(p, sizes, sampleDl=0.3)
    gp, gl = ops.grid_subsample(P, sizes, 0.3)
    assert np.array_equal(gl, wl) and np.array_equal(gp.cpu().numpy(), wp)
    R = rng.normal(size=(70, 3, 3)).astype(np.float32)
    a = ops.rotate_clouds_host(P, sizes, R).cpu().numpy()
    b = ops.rotate_clouds(P, torch.from_numpy(sizes).to(gpu), torch.from_numpy(R).to(gpu)).cpu().numpy()
    assert np.array_equal(a, b)


@pytest.mark.parametrize("c", [5, 24, 100, 300])
def test_pools_odd_channels(gpu, c):
    from weasal_amd import ops
    rng = np.random.default_rng(c)
    ns, nq, h = 500, 260, 21
    x = rng.normal(size=(ns, c)).astype(np.float32)
    inds = rng.integers(0, ns + 1, size=(nq, h))
    dy = rng.normal(size=(nq, c)).astype(np.float32)
    t = lambda a: torch.from_numpy(a)
    xc = t(x).requires_grad_(True)
    want = kpconv_ref.max_pool_ref(xc, t(inds))
    (want * t(dy)).sum().backward()
    xg = t(x).to(gpu).requires_grad_(True)
    got = ops.max_pool(xg, t(inds).to(gpu))
    (got * t(dy).to(gpu)).sum().backward()
    assert torch.equal(got.detach().cpu(), want.detach()) and rel(xg.grad, xc.grad) < 1e-6
    xc2 = t(x).requires_grad_(True)
    w2 = kpconv_ref.closest_pool_ref(xc2, t(inds))
    (w2 * t(dy)).sum().backward()
    xg2 = t(x).to(gpu).requires_grad_(True)
    g2 = ops.closest_pool(xg2, t(inds).to(gpu))
    (g2 * t(dy).to(gpu)).sum().backward()
    assert torch.equal(g2.detach().cpu(), w2.detach()) and rel(xg2.grad, xc2.grad) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("m,n", [(0, 16), (1, 4), (130, 9), (5000, 33), (4099, 128)])
def test_act_bwd_colsum_edge_shapes(m, n):
    """ws_act_bwd_colsum: empty input, single row, odd widths (scalar path), rows that do not fill a chunk"""
    from weasal_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(m + n)
    dy, y = torch.randn(m, n, device=dev), torch.randn(m, n, device=dev)
    for act in (True, False):
        dz, cs = ops._act_bwd_colsum(dy, y if act else None, 0.1 if act else None, True)
        ref = torch.where(y > 0, dy, dy * 0.1) if act else dy
        assert torch.equal(dz, ref)
        want = ref.double().sum(0)
        assert cs.shape == (n,)
        assert float((cs.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())) if m else float(cs.abs().max()) == 0.0


@pytest.mark.gpu
def test_contrast_rows_small_slice_and_wide_logits():
    """ws_contrast_rows with fewer slice rows than 1000, 16 logit channels, and a point that is its own slice row"""
    from weasal_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    n, c, s = 777, 16, 130
    on = torch.nn.functional.normalize(torch.randn(n, c, device=dev), dim=1).requires_grad_(True)
    slc = torch.randint(0, n, (s,), device=dev); slc[0] = 5
    cert = torch.rand(n, device=dev) > 0.3
    lbl = torch.randint(0, 7, (n,), device=dev)
    loss = ops.contrast_rows(on, on[slc], slc, cert, lbl, 0.1, 1e-8)
    # dense restatement of architectures.py:455-497 in float64
    o = on.detach().double()
    mul = o @ o[slc].T / 0.1
    use = (torch.arange(n, device=dev)[:, None] != slc[None, :]) & (cert[slc][None, :] == cert[:, None])
    pos = use & (lbl[slc][None, :] == lbl[:, None])
    lg = mul - mul.max(1, keepdim=True)[0]
    e = (lg.exp() * use).sum(1, keepdim=True)
    lp = (lg - torch.log(e + 1e-8)) * use
    ref = -0.1 * (pos * lp).sum(1) / (pos.sum(1) + 1e-12)
    assert float((loss.detach().double() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    loss.sum().backward()
    assert bool(torch.isfinite(on.grad).all())


@pytest.mark.gpu
@pytest.mark.parametrize("nc,nf,c", [(7000, 40000, 128), (300, 5000, 256), (1, 77, 64), (5000, 5000, 32), (900, 3000, 20), (280, 1500, 1024), (1500, 10000, 512)])
def test_closest_pool_backward_vector_form(gpu, nc, nf, c):
    """nearest-upsampling backward with float4 lanes and several incoming rows side by side (closest_pool_bwd_vec_kernel) against
    index_add in float64 and against the scalar kernel: shadow indices, supports nobody points at, one support for everything"""
    import ctypes as C
    from weasal_amd import _lib, ops
    lib = _lib.lib()
    torch.manual_seed(nc + c)
    x = torch.randn(nc, c, device=gpu, requires_grad=True)
    ups = torch.randint(0, nc, (nf, 4), device=gpu)
    if nc > 10:
        ups[:, 0] = torch.where(ups[:, 0] % 5 == 0, torch.full_like(ups[:, 0], 3), ups[:, 0])     # a crowded support, empty ones
    ups[::13, 0] = nc                                                                              # shadow rows
    g = torch.randn(nf, c, device=gpu)
    flag = C.c_int.in_dll(lib, "ws_closest_bwd_vec")
    grads = []
    try:
        for v in (1, 0):
            flag.value = v
            ops.clear_table_cache()
            y = ops.closest_pool(x, ups)
            (dx,) = torch.autograd.grad(y, x, g)
            torch.cuda.synchronize()
            grads.append(dx)
    finally:
        flag.value = 1
    real = ups[:, 0] < nc
    want = torch.zeros(nc, c, device=gpu, dtype=torch.float64).index_add_(0, ups[real, 0], g[real].double())
    scale = float(want.abs().max()) + 1e-30
    assert float((grads[0].double() - want).abs().max()) <= 1e-5 * scale
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * scale
