"""GPU: whole blocks behind one C call each way (ws_kpblock_fwd/_bwd, ws_upunary_fwd/_bwd; weasal_amd/fused.py) against
the operator-by-operator path of weasal_amd/blocks.py (the form round 1 pinned against goldens g4/g8) on the same
inputs.  Same kernels and the same order inside every sum; the only re-association is where a gradient accumulation
became the residual operand of a GEMM epilogue (a + b in one order instead of the other), and that layers of fewer than
4 096 rows stay on the MFMA kernels (split-K) inside the block calls where the operator path hands them to rocBLAS:
outputs and parameters after the step must agree to 2e-5 of max|ref|, the gradient as one vector to 5e-4 in relative L2
(fp32 re-association only; single cancellation-heavy gradient tensors move by up to 6e-3 of their maximum (measured, |grad| ~ 1e-10): held at 2e-2).  The golden network test (tests/test_pyramid_gpu.py::
test_kpfcnn_step_vs_golden) and the full-width oracle tests run through the block calls as well."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _run(gpu, cfg, fused_on, seed=3, steps=1):
    from weasal_amd import fused, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.trainer import make_optimizer, train_step
    fused.FUSED_BLOCKS = fused_on
    min_rows, fused.MIN_ROWS = fused.MIN_ROWS, 0          # every block through the calls (the product's default; the switch is a diagnostic)
    try:
        np.random.seed(seed)
        torch.manual_seed(seed)
        net = KPFCNN(cfg, np.arange(9), []).to(gpu).train()
        if not cfg.use_batch_norm:                         # learned biases (BatchNormBlock with use_bn False): make them count
            gen = torch.Generator(device="cpu").manual_seed(17)
            with torch.no_grad():
                for name, p in net.named_parameters():
                    if name.endswith(".bias"):
                        p.copy_(0.1 * torch.randn(p.shape, generator=gen).to(gpu))
        opt = make_optimizer(net, cfg)
        pts, feats, labels, lens = synthetic.make_inputs(21, 3, 4000, 4.0, cfg.in_features_dim)
        np.random.seed(8)
        batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu),
                                    torch.from_numpy(labels).to(gpu), lens, [40, 45, 50, 50, 40])
        for _ in range(steps):
            loss, out = train_step(net, opt, batch, cfg)
        torch.cuda.synchronize()
        grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        params = {k: p.detach().clone() for k, p in net.named_parameters()}
        return out.detach().clone(), loss.item(), grads, params
    finally:
        fused.FUSED_BLOCKS = True
        fused.MIN_ROWS = min_rows


@pytest.mark.parametrize("cfg_name,use_bn", [("DALESPLConfig", True), ("Vaihingen3DPLConfig", True), ("DALESPLConfig", False)])
def test_block_calls_match_operator_path(gpu, cfg_name, use_bn):
    """use_bn False: BatchNormBlock is a learned bias (blocks.py:465): b1 / bk / b2 / bs and their gradients are live"""
    from weasal_amd import config as wcfg
    cfg = getattr(wcfg, cfg_name)()
    cfg.dropout = 0.0
    cfg.use_batch_norm = use_bn
    out_f, loss_f, g_f, p_f = _run(gpu, cfg, True)
    out_o, loss_o, g_o, p_o = _run(gpu, cfg, False)
    assert rel(out_f, out_o) < 2e-5
    assert abs(loss_f - loss_o) < 1e-5 * abs(loss_o)
    assert set(g_f) == set(g_o) and len(g_f) > 30
    num = sum(float(((g_f[k].double() - g_o[k].double()) ** 2).sum()) for k in g_o)
    den = sum(float((g_o[k].double() ** 2).sum()) for k in g_o)
    assert (num / den) ** 0.5 < 5e-4                      # the gradient as one vector (measured 1.5e-4)
    for k in g_o:                                         # single cancellation-heavy tensors (gradients of ~1e-10 in the deep
        assert rel(g_f[k], g_o[k]) < 5e-2, k              # blocks) move by ~1e-2 under any re-association of the dW row sums
    for k in p_o:
        assert rel(p_f[k], p_o[k]) < 2e-5, k


def test_block_calls_are_taken_and_launch_fewer_kernels(gpu):
    """the rigid f32 network really runs through the block calls (one autograd node per block)"""
    from weasal_amd import config as wcfg, fused
    cfg = wcfg.DALESPLConfig()
    cfg.dropout = 0.0
    calls = {"kp": 0, "up": 0}
    kp_apply, up_apply = fused._KPBlockFn.apply, fused._UpUnaryFn.apply

    def kp(*a):
        calls["kp"] += 1
        return kp_apply(*a)

    def up(*a):
        calls["up"] += 1
        return up_apply(*a)
    fused._KPBlockFn.apply, fused._UpUnaryFn.apply = kp, up
    try:
        _run(gpu, cfg, True)
    finally:
        fused._KPBlockFn.apply, fused._UpUnaryFn.apply = kp_apply, up_apply
    assert calls == {"kp": 10, "up": 4}, calls


def test_kpblock_rejects_unsupported_shapes_loudly(gpu):
    """descriptor validation: K != 15 and widths that are not multiples of 4 are WS_ERR_UNSUPPORTED, not silence"""
    import ctypes as C
    from weasal_amd import _lib, fused
    lib = _lib.lib()
    d = fused.KPBlockDesc()
    x = torch.zeros(64, device=gpu)
    for f in ("q_pts", "s_pts", "inds", "kernel_points", "feat", "wk", "wf", "out"):
        setattr(d, f, x.data_ptr())
    d.nq = d.ns = 4
    d.h, d.k, d.extent = 3, 7, 1.0
    d.in_dim = d.conv_in = 8
    d.conv_out = d.out_dim = 8
    assert lib.ws_kpblock_fwd_scratch_bytes(C.byref(d)) == -1 and b"K=15" in lib.ws_last_error()
    d.k = 15
    d.conv_out = d.out_dim = 6
    assert lib.ws_kpblock_fwd_scratch_bytes(C.byref(d)) == -1 and b"multiples of 4" in lib.ws_last_error()
    d.conv_out = d.out_dim = 8
    assert lib.ws_kpblock_fwd_scratch_bytes(C.byref(d)) > 0


@pytest.mark.gpu
def test_skip_gradients_summed_inside_the_strided_block(monkeypatch):
    """an encoder tensor read by the next strided block and by the decoder's skip connection: with the slots (fused.SkipSlot)
    the decoder's share is summed into the block's input gradient by the pool backward's store; without them autograd adds
    the two.  Same gradients (the association of the three-term sum differs: 1e-6), every level's slot used, and a backward
    whose nodes run in the other order (the block first: torch.autograd.grad on the encoder alone, then the tap) stays exact"""
    from weasal_amd import config as wcfg, fused, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    dev = torch.device("cuda:0")
    cfg = wcfg.Vaihingen3DPLConfig()
    wl = synthetic.WORKLOADS["vaihingen"]
    pts, feats, labels, lens = synthetic.make_inputs(11, 2, wl["points"], wl["radius"], cfg.in_features_dim)

    def run(on):
        monkeypatch.setattr(fused, "SKIP_SLOTS", on)
        np.random.seed(1)
        torch.manual_seed(1)
        net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
        np.random.seed(2)
        batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev),
                                    torch.from_numpy(labels).to(dev), lens, wl["limits"])
        torch.manual_seed(5)
        hits = fused.skip_slot_hits
        out = net(batch, cfg)
        loss = net.loss(out, batch.labels)
        loss.backward()
        torch.cuda.synchronize()
        return out.detach().clone(), {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None}, \
            fused.skip_slot_hits - hits

    oa, ga, ha = run(True)
    ob, gb, hb = run(False)
    assert ha == cfg.num_layers - 1 and hb == 0
    assert torch.equal(oa, ob)
    for k in ga:
        scale = float(gb[k].abs().max()) + 1e-30
        assert float((ga[k] - gb[k]).abs().max()) <= 2e-5 * scale, k

    # the other order: the block's backward before the tap's
    slot = fused.SkipSlot()
    slot.armed = True
    x = torch.randn(1000, 32, device=dev, requires_grad=True)
    tapped = fused.skip_tap(x, slot)
    slot.taken = True                          # what a block backward that found the slot empty leaves behind
    (g,) = torch.autograd.grad(tapped.sum(), x)
    assert torch.equal(g, torch.ones_like(x)) and slot.grad is None


@pytest.mark.gpu
@pytest.mark.parametrize("nc,nf,c_up,c_skip,out_dim,drop", [(900, 5003, 64, 32, 32, 0.0), (37, 300, 256, 128, 128, 0.5), (4000, 40001, 128, 64, 64, 0.5)])
def test_decoder_step_gathers_the_upsampled_rows_in_its_epilogue(gpu, nc, nf, c_up, c_skip, out_dim, drop):
    """ws_upunary_fwd with the nearest-upsampled rows read by the last product's epilogue (yc[ups[r, 0]] as a gathered
    residual; shadow indices add nothing) against the form that writes them out first (ws_closest_pool_fwd): same bits, with
    and without the fused dropout, rows-on-lanes and split-K products; and against upsample -> concat -> unary in float64"""
    import ctypes as C
    from weasal_amd import _lib, fused
    from weasal_amd.blocks import UnaryBlock
    from weasal_amd import config as wcfg
    lib = fused._bind()
    torch.manual_seed(nc)
    cfg = wcfg.Vaihingen3DPLConfig()
    unary = UnaryBlock(c_up + c_skip, out_dim, False, 0).to(gpu)
    with torch.no_grad():
        unary.batch_norm.bias.normal_()
    x = torch.randn(nc, c_up, device=gpu)
    skip = torch.randn(nf, c_skip, device=gpu)
    ups = torch.randint(0, nc, (nf, 3), device=gpu)
    ups[::17, 0] = nc                                           # shadow rows: the zero feature (blocks.py:80-92)
    flag = C.c_int.in_dll(lib, "ws_block_gather_residual")
    outs = []
    try:
        for v in (0, 1):
            flag.value = v
            with torch.no_grad():
                outs.append(fused.upunary(x, skip, unary, ups, (drop, 424242) if drop else None))
            torch.cuda.synchronize()
    finally:
        flag.value = 1
    assert torch.equal(outs[0], outs[1])
    xp = torch.cat([x, torch.zeros(1, c_up, device=gpu)]).double()
    want = torch.nn.functional.leaky_relu(torch.cat([xp[ups[:, 0]], skip.double()], 1) @ unary.mlp.weight.double().t()
                                          + unary.batch_norm.bias.double(), 0.1)
    got = outs[1].double()
    if drop:
        keep = got != 0
        assert 0.4 < float(keep.float().mean()) < 0.6
        want = torch.where(keep, want / (1 - drop), torch.zeros_like(want))
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())


@pytest.mark.gpu
def test_gate_links_hand_the_activation_backward_to_the_consumer(monkeypatch):
    """consecutive block calls (fused.GateLink): the consumer writes its input gradient times LeakyReLU'(input) and the producer
    skips its first backward pass -- the same products in the same order, so every gradient keeps its bits; the links are
    taken along the whole encoder and decoder chain, with the skip connections' shares summed before the gate"""
    from weasal_amd import config as wcfg, fused, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    dev = torch.device("cuda:0")
    cfg = wcfg.Vaihingen3DPLConfig()
    wl = synthetic.WORKLOADS["vaihingen"]
    pts, feats, labels, lens = synthetic.make_inputs(13, 2, wl["points"], wl["radius"], cfg.in_features_dim)

    def run(on):
        monkeypatch.setattr(fused, "GATE_LINKS", on)
        np.random.seed(1)
        torch.manual_seed(1)
        net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
        np.random.seed(2)
        batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev),
                                    torch.from_numpy(labels).to(dev), lens, wl["limits"])
        torch.manual_seed(5)
        hits = fused.gate_link_hits
        out = net(batch, cfg)
        loss = net.loss(out, batch.labels)
        loss.backward()
        torch.cuda.synchronize()
        return out.detach().clone(), {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None}, \
            fused.gate_link_hits - hits

    oa, ga, ha = run(True)
    ob, gb, hb = run(False)
    n_enc = len([b for b in cfg.architecture if 'upsample' not in b and 'unary' not in b])
    assert hb == 0 and ha >= n_enc - 1 + 3, (ha, n_enc)
    assert torch.equal(oa, ob)
    assert ga.keys() == gb.keys()
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k
