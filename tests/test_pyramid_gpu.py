"""GPU: the device pyramid against the reference's segmentation_inputs (golden g7) and the whole
network forward/backward/SGD step against golden g8 (the reference's KPFCNN on that pyramid)."""
import types

import numpy as np
import pytest
import torch

from conftest import assert_neighbors_equal, golden

pytestmark = pytest.mark.gpu


def _cfg():
    from test_oracle_cpu_kpconv import _small_config
    return _small_config()


def test_pyramid_vs_golden(gpu):
    from weasal_amd import pyramid
    g = golden("g7_pyramid.npz")
    cfg = _cfg()
    np.random.seed(int(g["np_seed"]))
    li = pyramid.segmentation_inputs(cfg, torch.from_numpy(g["points"]).to(gpu), torch.from_numpy(g["features"]).to(gpu),
                                     torch.from_numpy(g["labels"]).to(gpu), g["lens"], list(g["limits"]))
    L = 5
    assert len(li) == 5 * L + 2
    for l in range(L):
        assert np.array_equal(li[l].cpu().numpy(), g["points_%d" % l]), l
        assert np.array_equal(li[4 * L + l].cpu().numpy(), g["lengths_%d" % l])
    for l in range(L):
        pts_l = g["points_%d" % l]
        got = li[L + l].cpu().numpy()
        assert got.dtype == np.int64
        assert_neighbors_equal(pts_l, pts_l, got, g["neighbors_%d" % l], False)
        if l < L - 1:
            nxt = g["points_%d" % (l + 1)]
            assert_neighbors_equal(nxt, pts_l, li[2 * L + l].cpu().numpy(), g["pools_%d" % l], False)
            assert_neighbors_equal(pts_l, nxt, li[3 * L + l].cpu().numpy(), g["upsamples_%d" % l], False)
        else:
            assert li[2 * L + l].shape == (0, 1) and li[3 * L + l].shape == (0, 1)


def test_kpfcnn_step_vs_golden(gpu):
    from weasal_amd import pyramid
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.trainer import make_optimizer, train_step
    g8, g7 = golden("g8_kpfcnn.npz"), golden("g7_pyramid.npz")
    cfg = _cfg()
    np.random.seed(0)
    net = KPFCNN(cfg, np.arange(9), [])
    sd = {k[4:]: torch.from_numpy(g8[k]) for k in g8.files if k.startswith("sd0/")}
    net.load_state_dict(sd, strict=False)
    net.to(gpu).train()
    L = 5
    flat = ([g7["points_%d" % l] for l in range(L)] + [g7["neighbors_%d" % l] for l in range(L)]
            + [g7["pools_%d" % l] for l in range(L)] + [g7["upsamples_%d" % l] for l in range(L)]
            + [g7["lengths_%d" % l] for l in range(L)] + [g7["features"], g7["labels"]])
    batch = pyramid.PyramidBatch([torch.from_numpy(np.ascontiguousarray(a)) for a in flat]).to(gpu)
    opt = make_optimizer(net, cfg)
    loss, out = train_step(net, opt, batch, cfg)

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)

    assert rel(out.detach().cpu().numpy(), g8["logits"]) < 1e-4
    assert abs(loss.item() - float(g8["loss"])) < 1e-5
    params = dict(net.named_parameters())
    for k in g8.files:
        if k.startswith("grad/"):
            assert rel(params[k[5:]].grad.cpu().numpy(), g8[k]) < 1e-3, k
    sd1 = net.state_dict()
    for k in g8.files:
        if k.startswith("sd1/"):
            assert rel(sd1[k[4:]].cpu().numpy(), g8[k]) < 1e-4, k


def test_calibrator_on_unlimited_pyramid(gpu):
    """device histograms == numpy histograms of the same un-limited pyramid (CPU oracle)"""
    from oracle import pyramid_ref
    from weasal_amd import calibration, pyramid, synthetic
    cfg = _cfg()
    pts, feats, labels, lens = synthetic.make_inputs(3, 2, 1500, 2.5, cfg.in_features_dim)
    np.random.seed(11)
    batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu),
                                torch.from_numpy(labels).to(gpu), lens, ())
    cal = calibration.NeighborhoodCalibrator(cfg).update(batch)
    np.random.seed(11)
    li = pyramid_ref.segmentation_inputs(cfg, pts, feats, labels, lens, ())
    L = cfg.num_layers
    want = np.vstack([np.bincount((m < m.shape[0]).sum(1), minlength=cal.hist_n)[:cal.hist_n] for m in li[L:2 * L]])
    assert np.array_equal(cal.hists.cpu().numpy(), want)
    assert np.array_equal(cal.limits(), calibration.limits_from_histograms(want, 0.9))


def test_prefetcher_matches_direct_build(gpu):
    """batches built ahead on the side stream == batches built in line (same np.random stream), and a
    training step on a prefetched batch gives the same loss"""
    from weasal_amd import pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.prefetch import PyramidPrefetcher
    from weasal_amd.trainer import make_optimizer, train_step
    cfg = _cfg()
    raw = []
    for i in range(3):
        p, f, l, le = synthetic.make_inputs(50 + i, 2, 1200, 2.5, cfg.in_features_dim)
        raw.append((torch.from_numpy(p).to(gpu), torch.from_numpy(f).to(gpu), torch.from_numpy(l).to(gpu), le))
    limits = [20, 24, 26, 26, 20]
    np.random.seed(77)
    direct = [pyramid.build_batch(cfg, *r, limits) for r in raw]
    torch.cuda.synchronize()
    np.random.seed(77)
    pf = PyramidPrefetcher(cfg, iter(raw), limits, depth=2, device=gpu)
    got = list(pf)
    pf.close()
    assert len(got) == 3
    for a, b in zip(got, direct):
        a.activate()
        for name in ("points", "neighbors", "pools", "upsamples", "lengths"):
            for x, y in zip(getattr(a, name), getattr(b, name)):
                assert torch.equal(x, y), name
        assert len(a.point_orders) == len(b.point_orders) == 5
    losses = []
    for batches in (direct, got):
        np.random.seed(3); torch.manual_seed(3)
        net = KPFCNN(cfg, np.arange(9), []).to(gpu).train()
        opt = make_optimizer(net, cfg)
        out = [train_step(net, opt, bt, cfg)[0].item() for bt in batches]
        losses.append(out)
    assert np.allclose(losses[0], losses[1], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("index_order", [True, False])
def test_grid_backward_matches_transposed_table(index_order):
    """table-free KPConv backward (ws_kpconv_gather_bwd_x_grid) vs the transposed-table K4 on a pyramid whose rows are
    truncated by the neighbourhood limits: the same pairs in both; summed in index order (ws_kpconv_grid_sorted = 1, the
    pair order of the table) the results are bit-identical, in the order of the grid walk (the default: a third fewer
    instructions, just as deterministic) they agree to fp32 re-association"""
    import ctypes as C
    import numpy as np
    from weasal_amd import _lib
    sorted_switch = C.c_int.in_dll(_lib.lib(), "ws_kpconv_grid_sorted")
    from weasal_amd import config as wcfg, ops, pyramid, synthetic
    from weasal_amd.kernel_points import load_kernels
    dev = torch.device("cuda:0")
    cfg = wcfg.DALESPLConfig()
    pts, feats, labels, lens = synthetic.make_inputs(7, 3, 20000, 10.0, cfg.in_features_dim)
    pts, feats, labels = (torch.from_numpy(a).to(dev) for a in (pts, feats, labels))
    limits = [20, 30, 40, 40, 30]                 # well below the true counts: truncated rows everywhere
    batch = pyramid.build_batch(cfg, pts, feats, labels, lens, limits)
    assert len(batch.search_grids) >= 3
    batch.activate()
    torch.manual_seed(0)
    for l, (inds, grid) in enumerate(batch.search_grids[:3]):
        p = batch.points[l]
        assert inds.data_ptr() == batch.neighbors[l].data_ptr()
        r = cfg.first_subsampling_dl * cfg.conv_radius * 2 ** l
        kp = torch.from_numpy(load_kernels(r * cfg.KP_extent / cfg.conv_radius * 1.0, 15, dimension=3, fixed="center")
                              .astype(np.float32)).to(dev) if False else torch.randn(15, 3, device=dev) * (0.6 * r)
        extent = r * cfg.KP_extent / cfg.conv_radius
        for ci, variant in ((32, "rigid"), (3, "rigid"), (16, "deformable"), (64, "gaussian-closest")):
            x = torch.randn(p.shape[0], ci, device=dev)
            kw = {}
            if variant == "deformable":       # per-query kernel points + modulations (blocks.py:262-276, 366-367)
                kw = dict(deformed_kp=kp[None] + 0.1 * r * torch.randn(p.shape[0], 15, 3, device=dev),
                          modulations=torch.rand(p.shape[0], 15, device=dev), want_min_d2=True)
            elif variant == "gaussian-closest":
                kw = dict(influence="gaussian", aggregation="closest")
            outs = []
            for use_grid in (True, False):
                sorted_switch.value = 1 if index_order else 0
                ops.GRID_BACKWARD = use_grid
                ops.clear_table_cache()
                xx = x.clone().requires_grad_(True)
                wf, _ = ops.kpconv_gather(xx, p, p, inds, kp, extent, **kw)
                wf.backward(torch.ones_like(wf) * 0.5 + wf.detach() * 0.1)
                outs.append(xx.grad.clone())
            ops.GRID_BACKWARD = True
            sorted_switch.value = 0
            assert int(grid.overflow.item()) == 0
            if index_order:
                assert torch.equal(outs[0], outs[1]), (l, ci, variant, float((outs[0] - outs[1]).abs().max()))
            else:
                err = float((outs[0] - outs[1]).abs().max() / outs[1].abs().max())
                assert err < 2e-6, (l, ci, variant, err)
                again = []                                  # the grid-walk order is deterministic: same bits run to run
                for _ in range(2):
                    xx = x.clone().requires_grad_(True)
                    wf, _ = ops.kpconv_gather(xx, p, p, inds, kp, extent, **kw)
                    wf.backward(torch.ones_like(wf) * 0.5 + wf.detach() * 0.1)
                    again.append(xx.grad.clone())
                assert torch.equal(again[0], again[1]) and torch.equal(again[0], outs[0])


@pytest.mark.gpu
def test_training_converges_on_synthetic_spheres():
    """end to end: prefetched pyramids -> KPFCNN step (grid backward, fused GEMM epilogues, SGD) for 40 steps on
    a separable synthetic task (the label is the octant-parity of the point) must reduce the loss clearly"""
    import numpy as np
    from weasal_amd import config as wcfg, synthetic
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.prefetch import PyramidPrefetcher
    from weasal_amd.trainer import make_optimizer, train_step, freeze_gc, InFlightLimiter
    dev = torch.device("cuda:0")
    cfg = wcfg.DALESPLConfig()
    cfg.learning_rate = 1e-2
    np.random.seed(3); torch.manual_seed(3)
    net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
    opt = make_optimizer(net, cfg)
    batches = []
    for i in range(4):
        pts, feats, labels, lens = synthetic.make_inputs(100 + i, 2, 12000, 10.0, cfg.in_features_dim)
        off = np.concatenate([[0], np.cumsum(lens)])
        for b in range(len(lens)):                      # label = which side of the sphere's centre plane (classes 0 / 1)
            seg = pts[off[b]:off[b + 1]]
            labels[off[b]:off[b + 1]] = (seg[:, 2] > seg[:, 2].mean()).astype(labels.dtype)
        feats[:, 0] = 1.0
        feats[:, 1:] = pts[:, 2:3] - pts[:, 2:3].mean()   # height as input feature: learnable
        batches.append(tuple(torch.from_numpy(a).to(dev) for a in (pts, feats, labels)) + (lens,))

    def source():
        i = 0
        while True:
            yield batches[i % len(batches)]
            i += 1
    pf = PyramidPrefetcher(cfg, source(), [25, 35, 40, 40, 35], depth=2)
    lim = InFlightLimiter(3)
    losses = []
    try:
        for step in range(40):
            loss, _ = train_step(net, opt, next(pf), cfg)
            lim.tick()
            losses.append(loss.detach())
            if step == 5:
                freeze_gc()
    finally:
        pf.close()
    losses = torch.stack(losses).cpu().numpy()
    assert np.isfinite(losses).all()
    assert losses[-5:].mean() < 0.5 * losses[:3].mean(), losses


@pytest.mark.gpu
def test_kpfcnn_mprm_vs_golden():
    """weak-label network (attention blocks, class logits, CAMs, both losses, gradients) against the reference's own
    KPFCNN_mprm run on CPU (tests/golden/make_golden_mprm.py): 1e-4 relative per tensor, 1e-3 on parameter gradients"""
    import numpy as np
    from weasal_amd import config as wcfg
    from weasal_amd.architectures import KPFCNN_mprm
    from weasal_amd.pyramid import PyramidBatch
    g = golden("g10_mprm.npz")
    dev = torch.device("cuda:0")

    class Cfg(wcfg.Config):
        dataset = "GoldenWL"
        num_classes = 6
        architecture = ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
                        'nearest_upsample', 'nearest_upsample']
        num_kernel_points = 15
        first_subsampling_dl = 0.3
        conv_radius = 2.5
        deform_radius = 1.0
        KP_extent = 1.0
        KP_influence = 'linear'
        aggregation_mode = 'sum'
        first_features_dim = 16
        in_features_dim = 4
        modulated = False
        use_batch_norm = True
        batch_norm_momentum = 0.02
        deform_fitting_mode = 'point2point'
        deform_fitting_power = 1.0
        deform_lr_factor = 0.1
        repulse_extent = 1.2
        class_w = []
    cfg = Cfg()
    L = 3
    li = [torch.from_numpy(g["points_%d" % l]).to(dev) for l in range(L)]
    li += [torch.from_numpy(g["neighbors_%d" % l].astype(np.int64)).to(dev) for l in range(L)]
    li += [torch.from_numpy(g["pools_%d" % l].astype(np.int64)).to(dev) for l in range(L)]
    li += [torch.from_numpy(g["upsamples_%d" % l].astype(np.int64)).to(dev) for l in range(L)]
    li += [torch.from_numpy(g["lengths_%d" % l].astype(np.int32)).to(dev) for l in range(L)]
    li += [torch.from_numpy(g["features"]).to(dev), torch.from_numpy(g["labels"]).to(dev)]
    batch = PyramidBatch(li)
    batch.center_pts = torch.from_numpy(g["center_pts"]).to(dev)
    np.random.seed(0); torch.manual_seed(0)
    net = KPFCNN_mprm(cfg, np.arange(6), []).to(dev).train()
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd0/")}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all("num_batches_tracked" in m for m in missing), (missing, unexpected)

    x, cla, cam = net(batch, cfg)

    def close(a, ref, tol, what):
        ref = torch.from_numpy(np.asarray(ref)).to(a.device)
        err = float((a.detach() - ref).abs().max()); scale = float(ref.abs().max())
        assert err <= tol * max(scale, 1e-6), (what, err, scale)
    close(x, g["x"], 1e-4, "x")
    for i in range(4):
        close(cla[i], g["cla_logits_%d" % i], 1e-4, "cla %d" % i)
        close(cam[i], g["cam_%d" % i], 1e-4, "cam %d" % i)
    loss_cls = net.class_logits_loss(cla, torch.from_numpy(g["cloud_lb"]).to(dev))
    sizes = g["region_sizes"]; flat = g["regions_flat"]
    regions = [[flat[:sizes[0]], flat[sizes[0]:sizes[0] + sizes[1]]], []]
    regions_lb = [[g["regions_lb"][0], g["regions_lb"][1]], []]
    loss_reg = net.region_mprm_loss(cam, regions, regions_lb, batch.lengths[0].cpu())
    assert abs(float(loss_cls.detach()) - float(g["loss_cls"])) <= 1e-4 * abs(float(g["loss_cls"]))
    assert abs(float(loss_reg.detach()) - float(g["loss_reg"])) <= 1e-4 * abs(float(g["loss_reg"]))
    assert abs(net.accuracy(x, batch.labels) - float(g["acc"])) < 1e-6
    (loss_cls + loss_reg).backward()
    grads = {k: v.grad for k, v in net.named_parameters() if v.grad is not None}
    names = [str(n) for n in g["grad_names"]]
    assert sorted(grads.keys()) == names
    for n, ref_norm in zip(names, g["grad_norms"]):
        got = float(grads[n].double().norm())
        assert abs(got - float(ref_norm)) <= 1e-3 * max(float(ref_norm), 1e-8), (n, got, float(ref_norm))
        if ("grad/" + n) in g.files:
            close(grads[n], g["grad/" + n], 1e-3, "grad " + n)


@pytest.mark.gpu
def test_grid_backward_pair_conservation_full_size():
    """BASELINE config 3 size (8 x 50 000 points): with constant influence and unit inputs the table-free backward
    must return, for every support, 15 x (number of rows that contain it) -- a size-independent property that
    checks the membership test against the index matrix itself, on every level that has a grid"""
    import numpy as np
    from weasal_amd import config as wcfg, ops, pyramid, synthetic
    dev = torch.device("cuda:0")
    cfg = wcfg.DALESPLConfig()
    wl = synthetic.WORKLOADS["dales"]
    pts, feats, labels, lens = synthetic.make_inputs(11, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
    pts, feats, labels = (torch.from_numpy(a).to(dev) for a in (pts, feats, labels))
    batch = pyramid.build_batch(cfg, pts, feats, labels, lens, wl["limits"])
    batch.activate()
    assert len(batch.search_grids) >= 4
    for l, (inds, grid) in enumerate(batch.search_grids):
        p = batch.points[l]
        ns = p.shape[0]
        indeg = torch.bincount(inds[inds < ns].flatten(), minlength=ns)
        x = torch.ones(ns, 1, device=dev, requires_grad=True)
        wf, _ = ops.kpconv_gather(x, p, p, inds, torch.zeros(15, 3, device=dev), 1.0, influence="constant")
        wf.backward(torch.ones_like(wf))
        got = torch.round(x.grad[:, 0].double() / 15).long()
        assert int(grid.overflow.item()) == 0
        assert torch.equal(got, indeg), (l, int((got != indeg).sum()))


@pytest.mark.gpu
def test_deformable_modulated_network_steps():
    """BASELINE config 5's shape in fp32 (deformable + modulated KPConv in the two deepest levels, searched with the
    deformable radius): three training steps on small spheres stay finite and move the offset / modulation weights"""
    import numpy as np
    from weasal_amd import config as wcfg, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.trainer import make_optimizer, train_step
    dev = torch.device("cuda:0")
    cfg = wcfg.DALESDeformConfig()
    np.random.seed(2); torch.manual_seed(2)
    net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
    opt = make_optimizer(net, cfg)
    names = [n for n, _ in net.named_parameters() if "offset" in n]
    assert names, "no deformable offset parameters in the network"
    before = {n: p.detach().clone() for n, p in net.named_parameters() if n in names[:4]}
    pts, feats, labels, lens = synthetic.make_inputs(5, 2, 20000, 10.0, cfg.in_features_dim)
    pts, feats, labels = (torch.from_numpy(a).to(dev) for a in (pts, feats, labels))
    losses = []
    for _ in range(3):
        batch = pyramid.build_batch(cfg, pts, feats, labels, lens, [40, 50, 60, 200, 150])
        loss, out = train_step(net, opt, batch, cfg)
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and torch.isfinite(out).all()
    after = dict(net.named_parameters())
    assert any(not torch.equal(before[n], after[n].detach()) for n in before)


@pytest.mark.gpu
def test_one_call_pyramid_equals_the_per_call_loop():
    """ws_pyramid_build (one library call per batch) against the per-call loop of pyramid.segmentation_inputs on the same
    inputs and the same np.random stream: points, index matrices, lengths, cell orders, search grids and transposed tables
    bit-identical; also after a deliberately tiny arena (the WS_ERR_CAPACITY round trip)"""
    from weasal_amd import config as wcfg, pyramid, synthetic
    dev = torch.device("cuda:0")
    for name, cfgn in (("vaihingen", "Vaihingen3DPLConfig"), ("dales_deform", "DALESDeformConfig")):
        wl = synthetic.WORKLOADS[name]
        cfg = getattr(wcfg, cfgn)()
        spheres, points = (3, 2500) if name == "vaihingen" else (2, 6000)
        pts, feats, labels, lens = synthetic.make_inputs(5, spheres, points, wl["radius"] * (0.45 if name == "dales_deform" else 1.0),
                                                         cfg.in_features_dim)
        args = (cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev), lens, wl["limits"])
        batches = []
        for native, tiny in ((False, False), (True, False), (True, True)):
            pyramid.NATIVE_PYRAMID = native
            if tiny:
                pyramid._arena_hint.clear()
                real_schedule = pyramid._schedule
                pyramid._arena_hint.update({k: 1 for k in []})
                # make the first guess hopeless: one byte per key is not enough for anything
                import threading
                key = (dev.index or 0, threading.get_ident(), int(args[1].shape[0]), tuple(int(v) for v in wl["limits"]), len(real_schedule(cfg, wl["limits"])))
                pyramid._arena_hint[key] = 1
            try:
                np.random.seed(9)
                batches.append(pyramid.build_batch(*args))
            finally:
                pyramid.NATIVE_PYRAMID = True
        torch.cuda.synchronize()
        ref = batches[0]
        for got in batches[1:]:
            for a, b in zip(ref.points + ref.neighbors + ref.pools + ref.upsamples + ref.lengths,
                            got.points + got.neighbors + got.pools + got.upsamples + got.lengths):
                assert a.shape == b.shape and torch.equal(a, b)
            assert len(ref.point_orders) == len(got.point_orders)
            for (_, oa), (_, ob) in zip(ref.point_orders, got.point_orders):
                assert torch.equal(oa, ob)
            assert len(ref.search_grids) == len(got.search_grids)
            for (ma, ga), (mb, gb) in zip(ref.search_grids, got.search_grids):
                assert torch.equal(ma, mb) and torch.equal(ga.key_last, gb.key_last)
                # the exported grid: [CloudGrid nb | cell_start cells + 2 | sorted ns] at 256-byte aligned offsets (csrc/ws_grid.h);
                # the gaps between the sections are never written
                al = lambda v: (v + 255) // 256 * 256
                c_off = al(ga.nb * 52)
                s_off = c_off + al((ga.cells + 2) * 4)
                for lo, hi in ((0, ga.nb * 52), (c_off, c_off + (ga.cells + 1) * 4), (s_off, s_off + ga.ns * 16)):
                    assert torch.equal(ga.blob[lo:hi], gb.blob[lo:hi]), (lo, hi)
                assert (ga.nb, ga.cells, ga.ns, ga.max_count, ga.cap) == (gb.nb, gb.cells, gb.ns, gb.max_count, gb.cap)
                assert abs(ga.radius - gb.radius) == 0.0
            for ta, tb in ((ref.tables, got.tables), (ref.col0_tables, got.col0_tables)):
                # (a convolution level whose search overflowed the asynchronous slab has no grid: the per-call loop pre-builds
                # its table, the one-call path leaves it to the first backward -- every table the latter has must match)
                assert len(tb) <= len(ta) and len(tb) >= len(ta) - len(ref.points)
                for mb, nsb, xb in tb:
                    hits = [(ma, nsa, xa) for ma, nsa, xa in ta if nsa == nsb and ma.shape == mb.shape and torch.equal(ma, mb)]
                    assert len(hits) >= 1
                    ma, nsa, xa = hits[0]
                    assert torch.equal(xa.offsets[:nsa + 1], xb.offsets[:nsb + 1])
                    n_pairs = int(xa.offsets[nsa])
                    assert torch.equal(xa.pairs[:n_pairs], xb.pairs[:n_pairs])
            assert [float(r) for _, r in ref.search_radii] == [float(r) for _, r in got.search_radii]


@pytest.mark.gpu
def test_one_call_pyramid_errors_and_fallbacks():
    """an empty sphere raises the reference's empty-result error from either path; more than 64 spheres, or no limits, take
    the per-call loop (ws_pyramid_build's bounds) and still build"""
    from weasal_amd import config as wcfg, pyramid, synthetic
    dev = torch.device("cuda:0")
    cfg = wcfg.Vaihingen3DPLConfig()
    wl = synthetic.WORKLOADS["vaihingen"]
    pts, feats, labels, lens = synthetic.make_inputs(3, 2, 1500, wl["radius"], cfg.in_features_dim)
    P, F, L = torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev)
    assert pyramid.native_eligible(cfg, P, lens, wl["limits"])
    assert not pyramid.native_eligible(cfg, P, lens, [])
    assert not pyramid.native_eligible(cfg, P, np.ones(65, np.int32), wl["limits"])
    np.random.seed(2)
    b = pyramid.build_batch(cfg, P, F, L, lens, [])                 # no limits: the two-call protocol, full widths
    assert b.neighbors[0].shape[0] == 3000 and b.neighbors[0].shape[1] >= 5
    # a far-away lonely point as its own sphere: every search of it finds itself only; an EMPTY sphere is the reference's error
    lens_bad = np.array([int(lens[0]), 0, int(lens[1])], dtype=np.int32)
    for native in (True, False):
        pyramid.NATIVE_PYRAMID = native
        try:
            with pytest.raises(RuntimeError):
                np.random.seed(2)
                pyramid.build_batch(cfg, P[:0], F[:0], L[:0], np.zeros(1, np.int32), wl["limits"])
            np.random.seed(2)
            pyramid.build_batch(cfg, P, F, L, lens_bad, wl["limits"])       # an empty element next to full ones is legal
        finally:
            pyramid.NATIVE_PYRAMID = True


@pytest.mark.gpu
def test_nearest_only_upsampling_is_column_zero_and_trains_identically():
    """opt-in nearest-only upsampling search (ws_radius_neighbors_nearest_async; config.nearest_upsample_only): the [N, 1]
    matrices equal column 0 of the full rows bit for bit (same (distance, index) order, ns where the radius is empty), every
    other tensor of the batch is unchanged, and a training step gives bit-identical logits, loss and weights -- the network
    reads nothing else of an upsampling matrix"""
    import copy
    from weasal_amd import config as wcfg, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.trainer import make_optimizer, train_step
    dev = torch.device("cuda:0")
    for name, cfgn, spheres, points, scale in (("vaihingen", "Vaihingen3DPLConfig", 3, 2500, 1.0), ("dales_deform", "DALESDeformConfig", 2, 6000, 0.45)):
        wl = synthetic.WORKLOADS[name]
        cfg = getattr(wcfg, cfgn)()
        cfg.dropout = 0.0
        pts, feats, labels, lens = synthetic.make_inputs(8, spheres, points, wl["radius"] * scale, cfg.in_features_dim)
        args = (torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev), lens, wl["limits"])
        np.random.seed(4)
        full = pyramid.build_batch(cfg, *args)
        cfg_n = copy.copy(cfg)
        cfg_n.nearest_upsample_only = True
        np.random.seed(4)
        near = pyramid.build_batch(cfg_n, *args)
        L = len(full.points)
        for l in range(L):
            assert torch.equal(full.points[l], near.points[l]) and torch.equal(full.neighbors[l], near.neighbors[l])
            assert torch.equal(full.pools[l], near.pools[l])
            if l + 1 < L:
                assert near.upsamples[l].shape == (full.points[l].shape[0], 1)
                assert torch.equal(near.upsamples[l][:, 0], full.upsamples[l][:, 0])
        if cfgn == "DALESDeformConfig":
            continue                                   # (the bf16 step is compared on the f32 network below only)
        outs = []
        for batch in (full, near):
            np.random.seed(5)
            torch.manual_seed(5)
            net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
            opt = make_optimizer(net, cfg)
            loss, out = train_step(net, opt, batch, cfg)
            torch.cuda.synchronize()
            outs.append((float(loss.detach()), out.detach().clone(), {k: v.detach().clone() for k, v in net.state_dict().items()}))
        assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
        for k in outs[0][2]:
            assert torch.equal(outs[0][2][k], outs[1][2][k]), k
