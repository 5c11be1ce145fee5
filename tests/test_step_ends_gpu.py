"""GPU parity of the two ends of the training step: the fused loss (ws_softmax_ce_fwd / _bwd, models/architectures.py:362-373)
against the oracle's statement of the reference criterion on the CPU, and the fused parameter update (ws_sgd_step,
utils/trainer_PseudoLabel.py:72-82,216-218) against torch.nn.utils.clip_grad_value_ + torch.optim.SGD on CPU copies.
Tolerances: loss 1e-6 of max(|loss|, 1) (f32 sums in another order), gradients 1e-6 of the largest entry, parameters after three
steps 1e-6 relative (the stock kernels may contract a * b + c into one rounding)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lut(valid, device):
    vmax = int(max(valid))
    lut = -torch.ones(vmax + 2, dtype=torch.int64)
    for i, c in enumerate(sorted(valid)):
        lut[c] = i
    return lut.to(device)


@pytest.mark.parametrize("n,c,weighted", [(5000, 9, False), (5000, 9, True), (257, 4, True), (1, 9, False), (70001, 64, False)])
def test_cross_entropy_matches_the_reference_criterion(gpu, n, c, weighted):
    from weasal_amd import ops
    from oracle import kpconv_ref
    rng = np.random.RandomState(3)
    valid = list(range(1, c + 1))                                  # label values 1..c are classes, 0 and 99 are ignored
    logits = torch.from_numpy((rng.randn(n, c) * 3).astype(np.float32))
    labels = torch.from_numpy(rng.choice([0, 99] + valid, size=n).astype(np.int64))
    labels[0] = valid[0]                                           # at least one valid row
    w = torch.from_numpy(rng.uniform(0.5, 2.0, c).astype(np.float32)) if weighted else None
    x_ref = logits.clone().requires_grad_(True)
    loss_ref = kpconv_ref.cross_entropy_ref(x_ref, labels, _lut(valid, "cpu"), w)
    (loss_ref * 1.7).backward()
    x = logits.to(gpu).requires_grad_(True)
    loss = ops.cross_entropy(x, labels.to(gpu), _lut(valid, gpu), None if w is None else w.to(gpu))
    (loss * 1.7).backward()
    # a row's term is logsumexp - x[t], a difference of O(|logit|) numbers: an ulp of those, so absolute for small losses
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-6 * max(abs(float(loss_ref.detach())), 1.0)
    g, g_ref = x.grad.cpu(), x_ref.grad
    assert float((g - g_ref).abs().max()) <= 1e-6 * float(g_ref.abs().max())
    ignored = ~torch.isin(labels, torch.tensor(valid))
    assert float(g[ignored].abs().sum()) == 0.0                    # ignored rows: exact zeros


def test_cross_entropy_edge_cases(gpu):
    from weasal_amd import ops, _lib
    # labels already positions (no table), a strided logits view, every row ignored -> nan like the stock loss
    rng = np.random.RandomState(4)
    wide = torch.from_numpy(rng.randn(300, 16).astype(np.float32)).to(gpu)
    x = wide[:, :9]
    t = torch.from_numpy(rng.randint(-1, 9, 300).astype(np.int64))
    ref = torch.nn.functional.cross_entropy(x.cpu(), t, ignore_index=-1)
    got = ops.cross_entropy(x, t.to(gpu))
    assert abs(float(got) - float(ref)) <= 1e-6 * abs(float(ref))
    none = ops.cross_entropy(x, torch.full((300,), -1, dtype=torch.int64, device=gpu))
    assert torch.isnan(none).item()
    with pytest.raises(_lib.WeasalHipError):
        ops.cross_entropy(x.cpu(), t)
    with pytest.raises(_lib.WeasalHipError):
        ops.cross_entropy(torch.zeros(4, 65, device=gpu), torch.zeros(4, dtype=torch.int64, device=gpu))


def _params(rng, device):
    shapes = [(1,), (3,), (5, 7), (4097,), (64, 15, 33), (100003,), (0,), (1024, 256)]
    return [torch.nn.Parameter(torch.from_numpy(rng.randn(*s).astype(np.float32)).to(device)) for s in shapes]


@pytest.mark.parametrize("momentum,wd,clip", [(0.98, 1e-3, 100.0), (0.9, 0.0, 0.05), (0.0, 1e-3, 0.0)])
def test_fused_sgd_matches_clip_and_torch_sgd(gpu, momentum, wd, clip):
    from weasal_amd.trainer import FusedSGD
    rng = np.random.RandomState(5)
    ps_ref = _params(rng, "cpu")
    ps = [torch.nn.Parameter(p.detach().clone().to(gpu)) for p in ps_ref]
    groups = lambda q: [{'params': q[:5]}, {'params': q[5:], 'lr': 0.003}]
    ref = torch.optim.SGD(groups(ps_ref), lr=0.01, momentum=momentum, weight_decay=wd)
    opt = FusedSGD(groups(ps), lr=0.01, momentum=momentum, weight_decay=wd)
    for step in range(3):
        for p_ref, p in zip(ps_ref, ps):
            g = torch.from_numpy((rng.randn(*p_ref.shape) * (0.1 if step else 1.0)).astype(np.float32))
            p_ref.grad = g.clone()
            p.grad = g.to(gpu)
        if step == 1:                                              # a parameter the graph skipped this step
            ps_ref[2].grad = None
            ps[2].grad = None
        if clip > 0:
            torch.nn.utils.clip_grad_value_([p for p in ps_ref if p.grad is not None], clip)
        ref.step()
        opt.step(clip_value=clip)
        if step == 1:
            opt.param_groups[0]['lr'] = ref.param_groups[0]['lr'] = 0.02      # what the scheduler does between epochs
    for i, (p_ref, p) in enumerate(zip(ps_ref, ps)):
        if p_ref.numel() == 0:
            continue
        err = float((p.detach().cpu() - p_ref.detach()).abs().max()) / float(p_ref.detach().abs().max())
        assert err <= 1e-6, (i, err)
        if momentum:
            b, b_ref = opt.state[p]['momentum_buffer'].cpu(), ref.state[p_ref]['momentum_buffer']
            assert float((b - b_ref).abs().max()) <= 1e-6 * float(b_ref.abs().max()), i
    # the state is torch.optim.SGD's: it loads into the stock optimizer
    stock = torch.optim.SGD(groups([torch.nn.Parameter(p.detach().clone()) for p in ps]), lr=0.01, momentum=momentum, weight_decay=wd)
    stock.load_state_dict(opt.state_dict())


def test_fused_sgd_hands_unsupported_cases_to_torch(gpu):
    from weasal_amd.trainer import FusedSGD
    p = torch.nn.Parameter(torch.ones(10, device=gpu))
    q = torch.nn.Parameter(torch.ones(10, device=gpu))
    a = FusedSGD([p], lr=0.1, momentum=0.9, nesterov=True)
    b = torch.optim.SGD([q], lr=0.1, momentum=0.9, nesterov=True)
    for _ in range(2):
        p.grad = torch.full((10,), 3.0, device=gpu)
        q.grad = torch.full((10,), 3.0, device=gpu)
        torch.nn.utils.clip_grad_value_([q], 1.0)
        a.step(clip_value=1.0)
        b.step()
    assert torch.equal(p.detach(), q.detach())


def test_run_ahead_limiter_reports_a_set_capacity_flag(gpu):
    """InFlightLimiter folds the K4G capacity flags of the batches on the device and looks at them off the step: a set flag
    raises at the periodic check (after check_every + depth ticks at the latest) or in finish(), a clean run stays silent"""
    import types
    from weasal_amd.trainer import InFlightLimiter

    def batch(value, grids=3):
        return types.SimpleNamespace(search_grids=[(None, types.SimpleNamespace(overflow=torch.full((1,), value, dtype=torch.int32, device=gpu)))
                                                   for _ in range(grids)])

    lim = InFlightLimiter(depth=2, check_every=4)
    for _ in range(12):
        lim.tick(batch(0))
    lim.finish()
    lim = InFlightLimiter(depth=2, check_every=4)
    lim.tick(batch(0))
    lim.tick(batch(300))                       # folded into the running maximum, not yet on the host
    with pytest.raises(RuntimeError, match="overflowed"):
        for _ in range(8):
            lim.tick(batch(0))
    lim = InFlightLimiter(depth=2, check_every=100)
    lim.tick(batch(0))
    lim.tick(batch(7, grids=2))                # another number of grids: the folded flags go out at once
    lim.tick(batch(300, grids=2))
    with pytest.raises(RuntimeError, match="300"):
        lim.finish()


@pytest.mark.gpu
def test_dropout_kernel_mask_statistics_and_backward():
    """ops.dropout (ws_dropout_apply): Bernoulli(1 - p) keep mask from (seed, index), kept values scaled by 1 / (1 - p), the
    backward applies the SAME mask (recomputed, not stored); a seed reproduces the mask, another seed gives another"""
    from weasal_amd import ops
    dev = torch.device("cuda:0")
    x = (torch.rand(400_003, 7, device=dev) + 0.5).requires_grad_(True)     # no zeros; size not a multiple of 4
    for p in (0.5, 0.1):
        y = ops.dropout(x, p, seed=1234)
        kept = y != 0
        frac = float(kept.float().mean())
        assert abs(frac - (1 - p)) < 4 * (p * (1 - p) / x.numel()) ** 0.5 + 1e-4, (p, frac)
        assert torch.equal(y[kept], (x.detach() * (1.0 / (1.0 - p)))[kept])
        g = torch.randn_like(y)
        (dx,) = torch.autograd.grad(y, x, g)
        assert torch.equal(dx != 0, kept & (g != 0))
        assert torch.equal(dx[kept], (g * (1.0 / (1.0 - p)))[kept])
        assert torch.equal(ops.dropout(x, p, seed=1234), y)
        other = ops.dropout(x, p, seed=1235) != 0
        assert 0.2 < float((other != kept).float().mean()) / (2 * p * (1 - p)) < 5.0
        # no structure along rows or columns (a counter-based generator must not alias with the row length)
        assert float(kept.float().mean(0).std()) < 0.01 and float(kept.float().mean(1).std()) < 0.5
    assert torch.equal(ops.dropout(x, 0.0, seed=5), x.detach())
    torch.manual_seed(3)
    a = ops.dropout(x, 0.5)
    torch.manual_seed(3)
    assert torch.equal(ops.dropout(x, 0.5), a)


@pytest.mark.gpu
def test_training_steps_are_bit_reproducible():
    """three Vaihingen training steps (one-call pyramid with limits, grid-walk backward, split-K products, fused update), run
    twice from the same seeds in one process: identical index matrices, logits, losses and final weights -- no kernel of the
    step sums in an order that depends on scheduling (the search grid's cells are filled through an atomic cursor and then
    put in index order: csrc/neighbors.hip, nb_cell_rank_kernel)"""
    from weasal_amd import config as wcfg, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.trainer import make_optimizer, train_step
    dev = torch.device("cuda:0")

    def run():
        cfg = wcfg.Vaihingen3DPLConfig()
        np.random.seed(1)
        torch.manual_seed(1)
        net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
        opt = make_optimizer(net, cfg)
        wl = synthetic.WORKLOADS["vaihingen"]
        seen = []
        for step in range(3):
            pts, feats, labels, lens = synthetic.make_inputs(40 + step, 2, wl["points"], wl["radius"], cfg.in_features_dim)
            np.random.seed(step)
            batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev),
                                        torch.from_numpy(labels).to(dev), lens, wl["limits"])
            loss, out = train_step(net, opt, batch, cfg, epoch=0)          # dropout and the contrastive term included
            seen.append((float(loss.detach()), out.detach().clone(), [m.clone() for m in batch.neighbors + batch.pools]))
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in net.state_dict().items()}, seen

    a, sa = run()
    b, sb = run()
    for (la, oa, ma), (lb, ob, mb) in zip(sa, sb):
        assert la == lb and torch.equal(oa, ob)
        assert all(torch.equal(x, y) for x, y in zip(ma, mb))
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,n", [(40003, 64, 128), (300, 2048, 256), (5001, 32, 36), (777, 9, 128)])
def test_dropout_in_the_product_epilogue_is_the_dropout_kernel(gpu, m, k, n):
    """ws_gemm_xb_dropout_strided (dropout of the activated output recomputed from (seed, row * n + col) in the epilogue) against
    the product followed by ws_dropout_apply: same bits -- on the rows-on-lanes kernel, its split-K form (300 x 2048) and the
    LDS-staged kernel (k = 9); and ws_act_bwd_colsum_dropout against dropout backward + ws_act_bwd_colsum (dz and the sums)"""
    from weasal_amd import _lib
    from weasal_amd._lib import check, current_stream, ptr
    lib = _lib.lib()
    torch.manual_seed(m + n)
    x = torch.randn(m, k, device=gpu)
    b = torch.randn(k, n, device=gpu)
    bias = torch.randn(n, device=gpu)
    res = torch.randn(m, n, device=gpu)
    p, seed = 0.5, 987654321012345
    scratch = torch.empty(max(int(lib.ws_gemm_xb_scratch_bytes(m, k, n)), 256), dtype=torch.uint8, device=gpu)
    plain = torch.full((m, n), float("nan"), device=gpu)
    fused = torch.full((m, n), float("nan"), device=gpu)
    check(lib.ws_gemm_xb_epilogue_strided(ptr(x), m, k, k, ptr(b), n, 1, n, ptr(bias), ptr(res), n, 1, 0.1, ptr(plain), n, ptr(scratch),
                                          scratch.numel(), current_stream()))
    want = torch.empty_like(plain)
    check(lib.ws_dropout_apply(ptr(plain), plain.numel(), p, seed, ptr(want), current_stream()))
    check(lib.ws_gemm_xb_dropout_strided(ptr(x), m, k, k, ptr(b), n, 1, n, ptr(bias), ptr(res), n, 1, 0.1, p, seed, ptr(fused), n,
                                         ptr(scratch), scratch.numel(), current_stream()))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(fused).all())
    assert torch.equal(fused, want)
    assert 0.45 < float((fused == 0).float().mean()) < 0.55
    # backward: dy = gradient of the dropped tensor `want`
    dy = torch.randn(m, n, device=gpu)
    ddrop = torch.empty_like(dy)
    check(lib.ws_dropout_apply(ptr(dy), dy.numel(), p, seed, ptr(ddrop), current_stream()))
    cs = torch.empty(max(int(lib.ws_act_bwd_colsum_scratch_bytes(m, n)), 256), dtype=torch.uint8, device=gpu)
    dz0, dz1 = torch.empty_like(dy), torch.empty_like(dy)
    s0, s1 = torch.empty(n, device=gpu), torch.empty(n, device=gpu)
    check(lib.ws_act_bwd_colsum(ptr(ddrop), m, n, n, ptr(plain), n, 0.1, ptr(dz0), n, ptr(s0), ptr(cs), current_stream()))
    check(lib.ws_act_bwd_colsum_dropout(ptr(dy), m, n, n, ptr(want), n, 0.1, p, seed, ptr(dz1), n, ptr(s1), ptr(cs), current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(dz0, dz1) and torch.equal(s0, s1)


@pytest.mark.gpu
def test_fused_dropout_step_is_the_unfused_step(monkeypatch):
    """one Vaihingen training step with the droplayer fused into the last decoder step (forward epilogue + backward's first
    pass) and with the dropout kernel as a pass of its own: the same seed draw, identical logits, loss and every gradient"""
    from weasal_amd import architectures, config as wcfg, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    dev = torch.device("cuda:0")
    cfg = wcfg.Vaihingen3DPLConfig()
    wl = synthetic.WORKLOADS["vaihingen"]
    pts, feats, labels, lens = synthetic.make_inputs(7, 2, wl["points"], wl["radius"], cfg.in_features_dim)

    def run(fused_on):
        monkeypatch.setattr(architectures, "DROPOUT_FUSED", fused_on)
        np.random.seed(1)
        torch.manual_seed(1)
        net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
        np.random.seed(2)
        batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev),
                                    torch.from_numpy(labels).to(dev), lens, wl["limits"])
        torch.manual_seed(5)
        out = net(batch, cfg)
        loss = net.loss(out, batch.labels)
        loss.backward()
        torch.cuda.synchronize()
        return out.detach().clone(), float(loss.detach()), {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None}

    oa, la, ga = run(True)
    ob, lb, gb = run(False)
    assert float((oa == 0).float().mean()) < 0.9
    assert torch.equal(oa, ob) and la == lb
    assert ga.keys() == gb.keys() and len(ga) > 20
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


@pytest.mark.gpu
def test_training_steps_hold_no_reference_cycles():
    """40 training steps with the cyclic collector OFF: device memory does not grow -- nothing a step creates (autograd nodes,
    skip slots, gate links: fused.GateLink keeps only a weak reference to the tensor whose node it hangs off) needs the
    collector to be freed.  (A strong reference there leaked 0.7 GB per DALES step: out of memory after ~400 steps.)"""
    import gc
    from weasal_amd import config as wcfg, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.trainer import make_optimizer, train_step
    dev = torch.device("cuda:0")
    cfg = wcfg.Vaihingen3DPLConfig()
    wl = synthetic.WORKLOADS["vaihingen"]
    torch.manual_seed(0)
    np.random.seed(0)
    net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
    opt = make_optimizer(net, cfg)
    p, f, l, le = synthetic.make_inputs(0, 2, wl["points"], wl["radius"], cfg.in_features_dim)
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        mem = []
        for step in range(40):
            b = pyramid.build_batch(cfg, torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev), le,
                                    wl["limits"])
            train_step(net, opt, b, cfg, epoch=0)
            del b
            if step % 10 == 9:
                torch.cuda.synchronize()
                mem.append(torch.cuda.memory_allocated() >> 20)
    finally:
        if was:
            gc.enable()
    assert mem[-1] <= mem[1] + 8, mem
