"""CPU, 2 processes over gloo: the data-parallel step (flat gradient all-reduce between backward
and clip) gives every rank the mean gradient and identical parameters after the SGD step."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


STEPS = 3


def _make(rank_seed):
    """tiny KPFCNN + its own batch, evaluated on the CPU through the oracle restatement"""
    from oracle import pyramid_ref
    from test_oracle_cpu_kpconv import _small_config
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.pyramid import PyramidBatch
    from weasal_amd.synthetic import make_inputs
    cfg = _small_config()
    np.random.seed(7)
    torch.manual_seed(7)
    net = KPFCNN(cfg, np.arange(9), [])
    net.train()
    pts, feats, labels, lens = make_inputs(100 + rank_seed, 2, 400, 1.5, cfg.in_features_dim)
    np.random.seed(5)
    li = pyramid_ref.segmentation_inputs(cfg, pts, feats, labels, lens, [])
    batch = PyramidBatch([torch.from_numpy(np.ascontiguousarray(a)) for a in li])
    return cfg, net, batch


def _worker(rank, world, port, out, buckets=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import kpconv_ref
    from weasal_amd import dp
    from weasal_amd.trainer import make_optimizer, train_step
    r, lr, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cfg, net, batch = _make(rank)
    dp.broadcast_parameters(net)
    opt = make_optimizer(net, cfg)
    sync = dp.GradSync(buckets=buckets)
    with kpconv_ref.cpu_reference_mode():
        for _ in range(STEPS):   # later steps exercise the re-homed gradient views (and the hooks of the bucketed form)
            loss, _o = train_step(net, opt, batch, cfg, grad_sync=sync)
    sync.verify()
    res = {k: v.detach().clone() for k, v in net.state_dict().items()}
    res["__grad__"] = sync.flat.clone()
    res["__bytes__"] = sync.nbytes()
    res["__points__"] = batch.points[0].clone()
    out[rank] = res
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("buckets", [1, 3])
def test_two_rank_step_matches_mean_gradient(buckets):
    """buckets = 3: the opt-in overlapped form (asynchronous all-reduce per range, launched from gradient hooks)"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, buckets), nprocs=world, join=True)
    a, b = out[0], out[1]
    for k in a:
        if k.startswith("__"):
            continue
        assert torch.equal(a[k], b[k]), k                 # replicas stay identical
    assert torch.equal(a["__grad__"], b["__grad__"])
    assert a["__bytes__"] > 0
    assert a["__points__"].shape != b["__points__"].shape or not torch.equal(a["__points__"], b["__points__"])   # rank-seeded inputs
    # reference: one process, both batches, mean of the two gradients, same two steps
    from oracle import kpconv_ref
    from weasal_amd.trainer import make_optimizer
    cfg, net, b0 = _make(0)
    _, _, b1 = _make(1)
    opt = make_optimizer(net, cfg)
    with kpconv_ref.cpu_reference_mode():
        for _ in range(STEPS):
            opt.zero_grad()
            grads = []
            for batch in (b0, b1):
                net.zero_grad()
                net.loss(net(batch, cfg), batch.labels).backward()
                grads.append([None if p.grad is None else p.grad.clone() for p in net.parameters()])
            for p, g0, g1 in zip(net.parameters(), *grads):
                p.grad = None if g0 is None else (g0 + g1) / 2
            torch.nn.utils.clip_grad_value_(net.parameters(), cfg.grad_clip_norm)
            opt.step()
    for k, v in net.state_dict().items():
        assert torch.allclose(v, a[k], rtol=1e-5, atol=1e-6), k


# ---------------------------------------------------------------------------------------------------------------
# gradient-set changes (VERDICT r1 item 6 / ADVICE r1): a parameter whose gradient first appears at a later step is
# reduced from that step on; ranks that disagree on the set raise; a second backward under the hooks raises
# ---------------------------------------------------------------------------------------------------------------
class _TwoBranch(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.a = torch.nn.Linear(6, 5)
        self.late = torch.nn.Linear(6, 5)          # joins the loss from `late_from` on
        self.never = torch.nn.Linear(6, 5)         # never used: its gradient stays None

    def loss(self, x, use_late):
        y = self.a(x).square().mean()
        if use_late:
            y = y + self.late(x).sin().mean()
        return y


def _branch_worker(rank, world, port, out, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from weasal_amd import dp
    dp.init_from_env(backend="gloo")
    net = _TwoBranch()
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    sync = dp.GradSync(buckets=3 if mode == "second_backward" else 1)
    g = torch.Generator().manual_seed(50 + rank)
    msg = ""
    try:
        for step in range(4):
            x = torch.randn(7, 6, generator=g)
            if mode == "asymmetric":
                use_late = step >= 1 and rank == 1
            elif mode == "second_backward":
                use_late = True
            else:
                use_late = step >= 1
            opt.zero_grad(set_to_none=False)
            loss = net.loss(x, use_late)
            sync.arm()
            loss.backward()
            if mode == "second_backward" and step == 2:
                net.loss(x, use_late).backward()          # not armed again: the hooks must refuse
            sync(net)
            opt.step()
        sync.verify()
    except RuntimeError as e:
        msg = str(e)
    out[rank] = {"state": {k: v.clone() for k, v in net.state_dict().items()}, "error": msg,
                 "never_none": net.never.weight.grad is None}
    if mode == "late":
        dist.barrier()
    dist.destroy_process_group()


def _run_branch(mode):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_branch_worker, args=(2, _free_port(), out, mode), nprocs=2, join=True)
    return out[0], out[1]


@pytest.mark.timeout(300)
def test_gradient_that_first_appears_at_step_2_is_reduced():
    a, b = _run_branch("late")
    assert a["error"] == "" and b["error"] == ""
    assert a["never_none"] and b["never_none"]          # untouched parameters keep grad None, as single-process
    for k in a["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), k
    # single process, mean of the two ranks' gradients
    net = _TwoBranch()
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    gens = [torch.Generator().manual_seed(50 + r) for r in range(2)]
    for step in range(4):
        xs = [torch.randn(7, 6, generator=g) for g in gens]
        opt.zero_grad()
        (sum(net.loss(x, step >= 1) for x in xs) / 2).backward()
        opt.step()
    for k, v in net.state_dict().items():
        assert torch.allclose(v, a["state"][k], rtol=1e-5, atol=1e-7), k
    assert not torch.equal(net.late.weight, _TwoBranch().late.weight)     # the late branch really trained


@pytest.mark.timeout(300)
def test_ranks_that_disagree_on_the_gradient_set_raise():
    a, b = _run_branch("asymmetric")
    assert "disagree" in a["error"] and "disagree" in b["error"]


@pytest.mark.timeout(300)
def test_second_backward_under_the_bucket_hooks_raises():
    a, b = _run_branch("second_backward")
    assert "backward ran without arm()" in a["error"] and "backward ran without arm()" in b["error"]
