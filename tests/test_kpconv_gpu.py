"""GPU parity: fused KPConv kernels (through the C ABI) against the golden vectors generated from
the reference's models/blocks.py.  Tolerance (BASELINE.json north_star): 1e-4 relative on fp32
activations, taken as max|a-b| <= 1e-4 * max|ref| per tensor (summation order differs from
torch's bmm); gradients use the same bound."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def build_conv(g, device, ci, co, **kw):
    from weasal_amd.blocks import KPConv
    np.random.seed(0)
    conv = KPConv(15, 3, ci, co, float(g["KP_extent"]), float(g["radius"]), **kw)
    with torch.no_grad():
        conv.weights.copy_(torch.from_numpy(g["weights"]))
        conv.kernel_points.copy_(torch.from_numpy(g["kernel_points"]))
        if kw.get("deformable"):
            conv.offset_conv.weights.copy_(torch.from_numpy(g["offset_weights"]))
            conv.offset_conv.kernel_points.copy_(torch.from_numpy(g["offset_kernel_points"]))
            conv.offset_bias.copy_(torch.from_numpy(g["offset_bias"]))
    return conv.to(device)


RIGID = [("g4_kpconv_3_64.npz", {}), ("g4_kpconv_32_32.npz", {}), ("g4_kpconv_32_32_strided.npz", {}),
         ("g4_kpconv_64_64.npz", {}), ("g4_kpconv_16_16_gaussian.npz", {"KP_influence": "gaussian"}),
         ("g4_kpconv_16_16_constant.npz", {"KP_influence": "constant"}),
         ("g4_kpconv_16_16_closest.npz", {"aggregation_mode": "closest"})]


@pytest.mark.parametrize("name,kw", RIGID)
def test_rigid_kpconv_forward_backward(gpu, name, kw):
    g = golden(name)
    ci, co = g["x"].shape[1], g["out"].shape[1]
    conv = build_conv(g, gpu, ci, co, **kw)
    x = torch.from_numpy(g["x"]).to(gpu).requires_grad_(True)
    out = conv(torch.from_numpy(g["q_pts"]).to(gpu), torch.from_numpy(g["s_pts"]).to(gpu),
               torch.from_numpy(g["inds"]).to(gpu), x)
    assert rel(out.detach().cpu().numpy(), g["out"]) < TOL
    (out * torch.from_numpy(g["dy"]).to(gpu)).sum().backward()
    assert rel(x.grad.cpu().numpy(), g["grad_x"]) < TOL
    assert rel(conv.weights.grad.cpu().numpy(), g["grad_weights"]) < TOL
    assert conv.kernel_points.grad is None


DEFORM = [("g5_kpconv_deform_16_16.npz", False), ("g5_kpconv_deform_mod_16_32.npz", True),
          ("g5_kpconv_deform_strided_16_16.npz", False)]


@pytest.mark.parametrize("name,modulated", DEFORM)
def test_deformable_kpconv(gpu, name, modulated):
    import types
    from weasal_amd.architectures import p2p_fitting_regularizer
    g = golden(name)
    ci, co = g["x"].shape[1], g["out"].shape[1]
    conv = build_conv(g, gpu, ci, co, deformable=True, modulated=modulated)
    x = torch.from_numpy(g["x"]).to(gpu).requires_grad_(True)
    out = conv(torch.from_numpy(g["q_pts"]).to(gpu), torch.from_numpy(g["s_pts"]).to(gpu),
               torch.from_numpy(g["inds"]).to(gpu), x)
    assert rel(conv.offset_features.detach().cpu().numpy(), g["offset_features"]) < TOL
    assert rel(conv.deformed_KP.detach().cpu().numpy(), g["deformed_KP"]) < TOL
    assert rel(conv.min_d2.detach().cpu().numpy(), g["min_d2"]) < TOL
    assert rel(out.detach().cpu().numpy(), g["out"]) < TOL
    net = types.SimpleNamespace(modules=lambda: [conv], l1=torch.nn.L1Loss(), K=15, repulse_extent=1.2,
                                deform_fitting_power=1.0)
    reg = p2p_fitting_regularizer(net)
    assert abs(float(reg) - float(g["reg_loss"])) <= TOL * abs(float(g["reg_loss"]))
    ((out * torch.from_numpy(g["dy"]).to(gpu)).sum() + reg).backward()
    # offset gradients chain two KPConvs and a sqrt: 5e-4 of the tensor's max
    assert rel(x.grad.cpu().numpy(), g["grad_x"]) < 5 * TOL
    assert rel(conv.weights.grad.cpu().numpy(), g["grad_weights"]) < TOL
    assert rel(conv.offset_conv.weights.grad.cpu().numpy(), g["grad_offset_weights"]) < 5 * TOL
    assert rel(conv.offset_bias.grad.cpu().numpy(), g["grad_offset_bias"]) < 5 * TOL


def test_pools(gpu):
    from weasal_amd import blocks
    g = golden("g6_pools.npz")
    x = torch.from_numpy(g["x"]).to(gpu).requires_grad_(True)
    mp = blocks.max_pool(x, torch.from_numpy(g["inds"]).to(gpu))
    assert np.array_equal(mp.detach().cpu().numpy(), g["max_pool"])
    (mp * torch.from_numpy(g["dy"]).to(gpu)).sum().backward()
    assert rel(x.grad.cpu().numpy(), g["grad_max_pool"]) < 1e-6
    xc = torch.from_numpy(g["xc"]).to(gpu).requires_grad_(True)
    cp = blocks.closest_pool(xc, torch.from_numpy(g["up"]).to(gpu))
    assert np.array_equal(cp.detach().cpu().numpy(), g["closest_pool"])
    (cp * torch.from_numpy(g["dyc"]).to(gpu)).sum().backward()
    assert rel(xc.grad.cpu().numpy(), g["grad_closest_pool"]) < 1e-6
    ga = blocks.global_average(x.detach(), g["lengths"])
    assert rel(ga.cpu().numpy(), g["global_average"]) < 1e-6


def test_cpu_tensors_are_rejected():
    """no CPU fallback: the operators refuse host tensors loudly"""
    from weasal_amd import _lib, ops
    x = torch.zeros(4, 8)
    with pytest.raises(_lib.WeasalHipError):
        ops.max_pool(x, torch.zeros(4, 2, dtype=torch.int64))


def test_transposed_table_matches_scatter(gpu):
    from weasal_amd import ops
    rng = np.random.default_rng(0)
    ns, nq, h = 1000, 700, 37
    inds = torch.from_numpy(rng.integers(0, ns + 1, size=(nq, h))).to(gpu)
    t = ops.TransposedTable(inds, ns)
    off = t.offsets.cpu().numpy()
    pairs = t.pairs.cpu().numpy()
    flat = inds.cpu().numpy().reshape(-1)
    assert off[0] == 0 and off[ns + 1] == int((flat < ns).sum())
    for s in list(range(0, ns, 97)) + [ns - 1]:
        want = np.nonzero(flat == s)[0]
        assert np.array_equal(pairs[off[s]:off[s + 1]], want)


def test_large_kpconv_properties(gpu):
    """DALES-sized layer (BASELINE config 3, enc1: N=400k would need the pyramid; here 60k x 59,
    32->32): linearity in x and agreement of the backward with a finite-difference direction."""
    from weasal_amd import ops
    rng = np.random.default_rng(5)
    n, h, ci = 60000, 59, 32
    pts = torch.from_numpy(rng.uniform(-10, 10, size=(n, 3)).astype(np.float32)).to(gpu)
    inds = torch.from_numpy(rng.integers(0, n + 1, size=(n, h))).to(gpu)
    # make neighbours geometrically close: q + small offset points
    s_pts = pts
    kp = torch.from_numpy(rng.normal(scale=0.4, size=(15, 3)).astype(np.float32)).to(gpu)
    x1 = torch.randn(n, ci, device=gpu)
    x2 = torch.randn(n, ci, device=gpu)
    f = lambda x: ops.kpconv_gather(x, pts, s_pts, inds, kp, 20.0)[0]
    a, b, c = f(x1), f(x2), f(x1 + 2 * x2)
    assert torch.allclose(c, a + 2 * b, rtol=1e-4, atol=1e-3)
    xg = x1.clone().requires_grad_(True)
    wf = ops.kpconv_gather(xg, pts, s_pts, inds, kp, 20.0)[0]
    dy = torch.randn_like(wf)
    (wf * dy).sum().backward()
    # <dy, f(x2)> == <grad, x2> by linearity
    lhs = (dy * b).sum().item()
    rhs = (xg.grad * x2).sum().item()
    assert abs(lhs - rhs) <= 1e-3 * max(abs(lhs), 1.0)


@pytest.mark.parametrize("m,k,n", [(5000, 64, 32), (70001, 480, 32), (4097, 45, 64), (20000, 384, 128),
                                   (9000, 128, 9), (12345, 960, 64), (8192, 100, 300)])
def test_skinny_gemm_vs_float64(gpu, m, k, n):
    """MFMA f32 GEMMs (forward, dx, dW) against a float64 reference"""
    from weasal_amd import ops
    torch.manual_seed(m)
    x = torch.randn(m, k, device=gpu, requires_grad=True)
    b = (torch.randn(k, n, device=gpu) / k ** 0.5).requires_grad_(True)
    dy = torch.randn(m, n, device=gpu)
    y = ops.matmul(x, b)
    y.backward(dy)
    xd, bd, dyd = x.detach().double(), b.detach().double(), dy.double()

    def rel(a, ref):
        return ((a.double() - ref).abs().max() / ref.abs().max()).item()

    assert rel(y.detach(), xd @ bd) < 2e-6
    assert rel(x.grad, dyd @ bd.t()) < 2e-6
    assert rel(b.grad, xd.t() @ dyd) < 2e-5      # m-long fp32 sums
    # strided x (a column slice) and nn.Linear form
    w = torch.randn(n, k, device=gpu) / k ** 0.5
    xs = torch.randn(m, k + 8, device=gpu)[:, 4:4 + k]
    assert rel(ops.linear(xs, w), xs.double() @ w.double().t()) < 2e-6


@pytest.mark.parametrize("m,k,n,with_bias,with_res,slope", [(5000, 64, 32, True, False, 0.1), (20000, 32, 128, False, True, 0.1),
                                                             (9000, 128, 9, True, False, 0.1), (6000, 480, 64, True, True, None)])
def test_gemm_epilogue_vs_torch(gpu, m, k, n, with_bias, with_res, slope):
    """bias + residual + LeakyReLU in the GEMM epilogue == the separate torch ops (blocks.py:497-500,709)"""
    from weasal_amd import ops
    torch.manual_seed(n)
    mk = lambda *s: torch.randn(*s, device=gpu, dtype=torch.float64)
    x, b = mk(m, k), mk(k, n) / k ** 0.5
    bias = mk(n) if with_bias else None
    res = mk(m, n) if with_res else None
    dy = mk(m, n)
    leaves64 = [t.clone().requires_grad_(True) for t in (x, b) + ((bias,) if with_bias else ()) + ((res,) if with_res else ())]
    leaves32 = [t.detach().float().clone().requires_grad_(True) for t in leaves64]

    def run(leaves, fused):
        it = iter(leaves)
        xx, bb = next(it), next(it)
        bi = next(it) if with_bias else None
        rr = next(it) if with_res else None
        if fused:
            return ops.matmul_epilogue(xx, bb, bias=bi, residual=rr, slope=slope)
        y = xx @ bb
        if bi is not None:
            y = y + bi
        if rr is not None:
            y = y + rr
        return y if slope is None else torch.nn.functional.leaky_relu(y, slope)

    y64 = run(leaves64, False)
    y64.backward(dy)
    y32 = run(leaves32, True)
    y32.backward(dy.float())
    rel = lambda a, r: ((a.double() - r).abs().max() / r.abs().max()).item()
    assert rel(y32.detach(), y64.detach()) < 5e-6
    for a, r in zip(leaves32, leaves64):
        assert rel(a.grad, r.grad) < 5e-5


@pytest.mark.parametrize("m,k,n,ep", [(1500, 3840, 256, 0), (270, 7680, 512, 5), (10257, 1920, 128, 7), (4096, 1024, 96, 4),
                                      (33, 512, 64, 3), (400000, 128, 128, 6)])
def test_gemm_splitk_and_wave_grids_through_the_c_abi(gpu, m, k, n, ep):
    """ws_gemm_xb_epilogue_splitk called directly (short deep products split K over up to 16 workgroup layers; the
    last shape takes the un-split 4x1 wave grid) against float64: bias (1), residual (2), LeakyReLU (4)"""
    from weasal_amd import _lib
    from weasal_amd._lib import check, current_stream, ptr
    lib = _lib.lib()
    torch.manual_seed(k + n)
    x = torch.randn(m, k, device=gpu)
    b = torch.randn(k, n, device=gpu) / k ** 0.5
    bias = torch.randn(n, device=gpu) if ep & 1 else None
    res = torch.randn(m, n, device=gpu) if ep & 2 else None
    y = torch.empty(m, n, device=gpu)
    sb = lib.ws_gemm_xb_scratch_bytes(m, k, n)
    scratch = torch.empty(max(sb, 16), dtype=torch.uint8, device=gpu)
    check(lib.ws_gemm_xb_epilogue_splitk(ptr(x), m, k, k, ptr(b), n, ptr(bias), ptr(res), n, 1 if ep & 4 else 0, 0.1,
                                         ptr(y), n, ptr(scratch) if sb else None, sb, current_stream()))
    ref = x.double() @ b.double()
    if bias is not None:
        ref = ref + bias.double()
    if res is not None:
        ref = ref + res.double()
    if ep & 4:
        ref = torch.nn.functional.leaky_relu(ref, 0.1)
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() < 5e-6
    if m < 32768 and k >= 512:
        assert sb > 0          # these shapes are the ones that split


@pytest.mark.parametrize("m,k,n,gated,masked", [(5000, 128, 32, True, False), (5000, 128, 128, True, True), (300, 2048, 512, True, True),
                                                 (70001, 9, 128, False, True), (1000, 64, 64, False, False)])
def test_gemm_epilogue_gates_through_the_c_abi(gpu, m, k, n, gated, masked):
    """ws_gemm_xb_gated_strided: LeakyReLU'(gate_y) and the dropout mask applied after bias / residual / activation --
    rows-on-lanes kernel, its split-K form (300 x 2048) and the LDS-staged kernel (k = 9) -- against float64"""
    from weasal_amd import _lib
    from weasal_amd._lib import check, current_stream, ptr
    lib = _lib.lib()
    torch.manual_seed(m + n)
    x = torch.randn(m, k, device=gpu)
    b = torch.randn(k, n, device=gpu) / k ** 0.5
    bias = torch.randn(n, device=gpu)
    res = torch.randn(m, n, device=gpu)
    gy = torch.randn(m, n, device=gpu) if gated else None
    mask = (torch.rand(m, n, device=gpu) < 0.5).to(torch.uint8) if masked else None
    y = torch.empty(m, n, device=gpu)
    sb = lib.ws_gemm_xb_scratch_bytes(m, k, n)
    scratch = torch.empty(max(sb, 16), dtype=torch.uint8, device=gpu)
    check(lib.ws_gemm_xb_gated_strided(ptr(x), m, k, k, ptr(b), n, 1, n, ptr(bias), ptr(res), n, 1, 0.1, ptr(gy), n, 0.1, ptr(mask), n, 2.0,
                                       ptr(y), n, ptr(scratch) if sb else None, sb, current_stream()))
    ref = torch.nn.functional.leaky_relu(x.double() @ b.double() + bias.double() + res.double(), 0.1)
    if gated:
        ref = ref * torch.where(gy > 0, 1.0, 0.1).double()
    if masked:
        ref = ref * mask.double() * 2.0
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() < 5e-6


def test_gated_gather_backward_is_the_plain_one_times_the_activation_derivative(gpu):
    """ws_kpconv_gather_bwd_x_gated / _grid_gated == ws_kpconv_gather_bwd_x / _grid followed by * LeakyReLU'(gate_y), bit for bit"""
    from weasal_amd import _lib, ops
    from weasal_amd._lib import check, current_stream, ptr
    from conftest import sphere
    lib = _lib.lib()
    rng = np.random.RandomState(11)
    pts = torch.from_numpy(sphere(rng, 6000, 2.0)).to(gpu)
    lens = [6000]
    radius, extent, ci = 0.25, 0.12, 32
    searches = ops.DeferredSearches(gpu)
    _, _, grid = searches.add(pts, pts, lens, lens, radius, 10, want_grid=True)      # ~12 neighbours on average: some rows truncated
    inds, = searches.finish()
    assert 1 < inds.shape[1] <= 10 and searches.last_counts[0] <= 128
    kp = torch.from_numpy((rng.randn(15, 3) * 0.08).astype(np.float32)).to(gpu)
    dwf = torch.randn(6000, 15 * ci, device=gpu)
    gate = torch.randn(6000, ci, device=gpu)
    table = ops.transposed_table(inds, 6000)
    plain, gated = torch.empty(6000, ci, device=gpu), torch.empty(6000, ci, device=gpu)
    check(lib.ws_kpconv_gather_bwd_x(ptr(pts), 6000, ptr(pts), 6000, ptr(inds), inds.shape[1], ptr(table.offsets), ptr(table.pairs), ptr(dwf),
                                     ci, ptr(kp), 15, None, None, extent, 0, 0, None, ptr(plain), current_stream()))
    check(lib.ws_kpconv_gather_bwd_x_gated(ptr(pts), 6000, ptr(pts), 6000, ptr(inds), inds.shape[1], ptr(table.offsets), ptr(table.pairs),
                                           ptr(dwf), ci, ptr(kp), 15, None, None, extent, 0, 0, None, ptr(gate), 0.1, ptr(gated),
                                           current_stream()))
    assert torch.equal(gated, plain * torch.where(gate > 0, 1.0, 0.1).to(plain.dtype))
    plain_g, gated_g = torch.empty(6000, ci, device=gpu), torch.empty(6000, ci, device=gpu)
    check(lib.ws_kpconv_gather_bwd_x_grid(ptr(pts), 6000, ptr(grid.blob), grid.nb, grid.cells, ptr(grid.key_last), radius, ptr(dwf), ci,
                                          ptr(kp), 15, None, None, extent, 0, 0, None, ptr(plain_g), ptr(grid.overflow), current_stream()))
    check(lib.ws_kpconv_gather_bwd_x_grid_gated(ptr(pts), 6000, ptr(grid.blob), grid.nb, grid.cells, ptr(grid.key_last), radius, ptr(dwf),
                                                ci, ptr(kp), 15, None, None, extent, 0, 0, None, ptr(gate), 0.1, None, 0, ptr(gated_g),
                                                ptr(grid.overflow), current_stream()))
    assert torch.equal(gated_g, plain_g * torch.where(gate > 0, 1.0, 0.1).to(plain_g.dtype))
    # candidates from the supports' own untruncated rows (limit 10 truncates about half of the rows: both paths are live): the same pairs,
    # summed in another order
    rows_g = torch.empty(6000, ci, device=gpu)
    check(lib.ws_kpconv_gather_bwd_x_grid_gated(ptr(pts), 6000, ptr(grid.blob), grid.nb, grid.cells, ptr(grid.key_last), radius, ptr(dwf),
                                                ci, ptr(kp), 15, None, None, extent, 0, 0, None, None, 0.0, ptr(inds), inds.shape[1],
                                                ptr(rows_g), ptr(grid.overflow), current_stream()))
    untruncated = int((grid.key_last == -1).sum())
    assert 0 < untruncated < 6000
    assert float((rows_g - plain_g).abs().max()) <= 2e-6 * float(plain_g.abs().max())
    assert int(grid.overflow.item()) == 0


@pytest.mark.parametrize("m,k,n", [(100000, 128, 128), (10257, 1920, 128), (380, 7680, 512), (5000, 64, 32)])
def test_split_bf16_products_are_fp32_accurate(gpu, m, k, n):
    """the opt-in gemm_xb3 path (ws_gemm_split = 1: fp32 products as six bf16 MFMA partial products of exact three-way
    splits) against float64, next to the default f32-input MFMA kernels on the same operands: its error is not larger"""
    import ctypes
    from weasal_amd import _lib, ops
    try:
        sw = ctypes.c_int.in_dll(_lib.lib(), "ws_gemm_split")
    except ValueError:
        pytest.skip("lab-only kernels (make CXXFLAGS+=-DWS_LAB_SPLIT_GEMM): not compiled into the product library")
    torch.manual_seed(m + k)
    x = torch.randn(m, k, device=gpu) * torch.exp(torch.randn(m, 1, device=gpu))       # rows of very different scale
    b = torch.randn(k, n, device=gpu) / k ** 0.5
    ref = x.double() @ b.double()
    err = {}
    try:
        for mode in (0, 1):
            sw.value = mode
            y = ops._gemm_xb(x, b)
            err[mode] = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    finally:
        sw.value = 0
    assert err[0] < 5e-6 and err[1] < 5e-6
    assert err[1] <= 1.5 * err[0] + 1e-7
    # dW = x^T dy on the same path
    dy = torch.randn(m, n, device=gpu) * torch.exp(torch.randn(m, 1, device=gpu))
    xs = torch.randn(m, min(k, 512), device=gpu)
    ref_w = xs.double().t() @ dy.double()
    errw = {}
    try:
        for mode in (0, 1):
            sw.value = mode
            o = ops._gemm_xty(_lib.lib(), xs, dy)
            errw[mode] = ((o.double() - ref_w).abs().max() / ref_w.abs().max()).item()
    finally:
        sw.value = 0
    assert errw[0] < 5e-6 and errw[1] < 5e-6 and errw[1] <= 1.5 * errw[0] + 2e-7


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,n", [(1, 32, 4), (31, 64, 36), (33, 32, 100), (4099, 96, 132), (70001, 32, 480), (12000, 128, 64)])
def test_gemm_staged_epilogue_is_bit_identical(gpu, m, k, n):
    """the LDS-turned epilogue of gemm_xb2 (whole row segments per store) against the per-lane one: same bits for ragged
    row / column counts, with bias + residual + LeakyReLU"""
    import ctypes as C
    from weasal_amd import _lib
    from weasal_amd._lib import check, current_stream, ptr
    lib = _lib.lib()
    torch.manual_seed(m + n)
    x = torch.randn(m, k, device=gpu)
    b = torch.randn(k, n, device=gpu)
    bias = torch.randn(n, device=gpu)
    res = torch.randn(m, n, device=gpu)
    outs = []
    flag = C.c_int.in_dll(lib, "ws_gemm_staged")
    try:
        for staged in (0, 1):
            flag.value = staged
            y = torch.full((m, n), float("nan"), device=gpu)
            check(lib.ws_gemm_xb_epilogue(ptr(x), m, k, k, ptr(b), n, ptr(bias), ptr(res), n, 1, 0.1, ptr(y), n, current_stream()))
            torch.cuda.synchronize()
            outs.append(y)
    finally:
        flag.value = 1
    assert bool(torch.isfinite(outs[1]).all())
    assert torch.equal(outs[0], outs[1])
    want = torch.nn.functional.leaky_relu((x.double() @ b.double()) + bias.double() + res.double(), 0.1)
    assert float((outs[1].double() - want).abs().max()) <= 1e-4 * float(want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,n", [(40003, 9, 128), (400000, 45, 64), (5000, 7, 4), (4096, 64 - 1, 192), (33333, 3, 1024)])
def test_gemm_shallow_contractions_stream(gpu, m, k, n):
    """gemm_xb_shallow_kernel (k <= 64 not a multiple of 32: the logits' dX, the input layer's contraction) against float64, with
    bias + residual + LeakyReLU and with the LeakyReLU' gate + dropout of ws_gemm_xb_gate_dropout; and against gemm_xb_kernel
    (switch off) on the same inputs"""
    import ctypes as C
    from weasal_amd import _lib
    from weasal_amd._lib import check, current_stream, ptr
    lib = _lib.lib()
    torch.manual_seed(m + k)
    x = torch.randn(m, k, device=gpu)
    b = torch.randn(k, n, device=gpu)
    bias = torch.randn(n, device=gpu)
    res = torch.randn(m, n, device=gpu)
    flag = C.c_int.in_dll(lib, "ws_gemm_shallow")
    outs = []
    try:
        for v in (1, 0):
            flag.value = v
            y = torch.full((m, n), float("nan"), device=gpu)
            check(lib.ws_gemm_xb_epilogue(ptr(x), m, k, k, ptr(b), n, ptr(bias), ptr(res), n, 1, 0.1, ptr(y), n, current_stream()))
            torch.cuda.synchronize()
            outs.append(y)
    finally:
        flag.value = 1
    want = torch.nn.functional.leaky_relu((x.double() @ b.double()) + bias.double() + res.double(), 0.1)
    scale = float(want.abs().max())
    assert bool(torch.isfinite(outs[0]).all())
    assert float((outs[0].double() - want).abs().max()) <= 1e-5 * scale
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-5 * scale
    # gate + dropout form: dropout_bwd first, then the gate (the order of the separate passes)
    gy = torch.randn(m, n, device=gpu)
    p, seed = 0.5, 13579
    plain = torch.empty((m, n), device=gpu)
    check(lib.ws_gemm_xb_epilogue(ptr(x), m, k, k, ptr(b), n, None, None, 0, 0, 0.0, ptr(plain), n, current_stream()))
    dropped = torch.empty_like(plain)
    check(lib.ws_dropout_apply(ptr(plain), plain.numel(), p, seed, ptr(dropped), current_stream()))
    ref = dropped * torch.where(gy > 0, torch.ones_like(gy), torch.full_like(gy, 0.1))
    got = torch.full((m, n), float("nan"), device=gpu)
    scratch = torch.empty(256, dtype=torch.uint8, device=gpu)
    check(lib.ws_gemm_xb_gate_dropout(ptr(x), m, k, k, ptr(b), n, ptr(gy), n, 0.1, p, seed, ptr(got), n, ptr(scratch), 0, current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
