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
        for _ in range(2):       # second step exercises the re-homed gradient views
            loss, _o = train_step(net, opt, batch, cfg, grad_sync=sync)
    res = {k: v.detach().clone() for k, v in net.state_dict().items()}
    res["__grad__"] = sync.flat.clone()
    res["__bytes__"] = sync.nbytes()
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
    # reference: one process, both batches, mean of the two gradients, same two steps
    from oracle import kpconv_ref
    from weasal_amd.trainer import make_optimizer
    cfg, net, b0 = _make(0)
    _, _, b1 = _make(1)
    opt = make_optimizer(net, cfg)
    with kpconv_ref.cpu_reference_mode():
        for _ in range(2):
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
