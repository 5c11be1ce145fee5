"""GPU, ONE rank, backend "nccl" (= RCCL): the data-parallel exchange executed on the device -- process group with
device_id, flat fp32 gradient buffer in HBM, the blocking all-reduce and the bucketed asynchronous one launched from the
gradient hooks (autograd thread), the pinned signature copy and verify().  A self all-reduce moves no data between
GPUs (no multi-GPU box is available to the suite); what this pins is that the RCCL branch of weasal_amd.dp runs and
leaves the step's results unchanged: parameters after 3 steps equal those of the single-process step bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(use_dp, buckets, port, out):
    import torch.distributed as dist
    from weasal_amd import config as wcfg, dp, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.trainer import make_optimizer, train_step
    dev = torch.device("cuda:0")
    sync = None
    if use_dp:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        r, lr, w = dp.init_from_env(backend="nccl", force=True)
        assert (r, lr, w) == (0, 0, 1) and dist.get_backend() == "nccl"
        sync = dp.GradSync(buckets=buckets, single_rank_exchange=True)
    cfg = wcfg.Vaihingen3DPLConfig()
    cfg.dropout = 0.0
    np.random.seed(1)
    torch.manual_seed(1)
    net = KPFCNN(cfg, np.arange(9), []).to(dev).train()
    opt = make_optimizer(net, cfg)
    wl = synthetic.WORKLOADS["vaihingen"]
    for step in range(3):
        pts, feats, labels, lens = synthetic.make_inputs(40 + step, 2, wl["points"], wl["radius"], cfg.in_features_dim)
        np.random.seed(step)
        batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev),
                                    lens, wl["limits"])
        train_step(net, opt, batch, cfg, grad_sync=sync)
    if sync is not None:
        sync.verify()
        assert sync.flat.is_cuda and sync.nbytes() > 1e6
        dist.destroy_process_group()
    torch.cuda.synchronize()
    out[(use_dp, buckets)] = {k: v.detach().cpu() for k, v in net.state_dict().items()}


def _worker(_rank, port, out):
    _run(False, 1, port, out)
    _run(True, 1, port, out)


def _worker_buckets(_rank, port, out):
    _run(True, 4, port, out)


@pytest.mark.timeout(600)
def test_rccl_branch_runs_on_one_rank_and_leaves_the_step_unchanged(gpu):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), out), nprocs=1, join=True)           # a fresh child: never a re-exec of this process
    mp.spawn(_worker_buckets, args=(_free_port(), out), nprocs=1, join=True)
    ref, one, four = out[(False, 1)], out[(True, 1)], out[(True, 4)]
    assert len(ref) > 100
    for k in ref:
        assert torch.equal(ref[k], one[k]), k          # sum over one rank, / 1: the same gradients, the same update
        assert torch.equal(ref[k], four[k]), k


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the parent touches no GPU, starts two fresh rank
    processes, relays rank 0's line.  On this one-GPU box the two ranks share the card over gloo (the rehearsal backend);
    what is pinned is the launch path and that the line describes the group that really ran (n_gpus, dist_world_size)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WEASAL_DIST_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--workload", "vaihingen"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["dist_world_size"] == 2
    assert out["config"]["dist_backend"] == "gloo" and out["config"]["parallelism"] == "dp2"
    assert out["config"]["allreduce_bytes"] == 4 * 4097993          # the Vaihingen KPFCNN's flat fp32 gradient
    assert out["scaling"] == "weak" and out["value"] > 0
