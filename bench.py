#!/usr/bin/env python3
"""bench.py -- points/sec of one KP-FCNN training step on synthetic DALES-shaped spheres.

One "step" = one pass of the whole hot path over one batch that is already resident in HBM:
  input pyramid on the GPU (13 radius-neighbour searches + 4 grid subsamplings, HIP)
  -> KPFCNN forward -> loss -> backward -> [flat RCCL all-reduce when N > 1]
  -> clip_grad_value_ -> SGD step.
`value` = level-0 points of all ranks per second (weak scaling: every rank owns its own batch of
8 x 50k-point spheres, BASELINE.json configs[2]; configs[3] is the same per rank on 8 GPUs).

Besides the contract line this prints, in the same JSON object:
  roofline      the fused KPConv gather kernel (K3) of the largest layer (enc1: N=400k, H=59, 32->32):
                algorithmic bytes B_fwd / average launch duration measured with
                HIP events on the launch stream inside the timed region, against the 8 TB/s HBM peak.
  cpu_baseline  the CPU path (reference geometry core from oracle/_ref when present, else the port;
                plain-torch KPConv restatement) timed on this box's host cores on ONE sphere.

Launch: `python bench.py` (1 GPU), or for N ranks either
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
or plainly `python bench.py --gpus N ...`: without WORLD_SIZE in the environment this process touches no GPU, starts the
N ranks itself as fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relays rank 0's JSON line and
exits non-zero if any rank fails.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def b_fwd(n, h, ci, co, es=4):
    """algorithmic bytes of one fused KPConv forward (the "logical gather" model, every neighbour
    reference counted once); es = bytes per feature element (4 = f32 rows, 2 = bf16 rows of config 5)"""
    return n * h * (8 + 12 + es * ci) + n * (12 + es * co) + 4 * 15 * ci * co + 180


def u_fwd(n, ns, h, ci, co, es=4):
    """compulsory-unique bytes of the same layer: every index once, every support row once"""
    return 8 * n * h + ns * (12 + es * ci) + n * (12 + es * co) + 4 * 15 * ci * co


def rows_inside_reach(net, cfg, inp, limits, h, ci):
    """mean number of neighbours per query that lie within the reach (largest kernel-point norm + influence extent) of the
    first KPConv with `ci` input channels whose index matrix has `h` columns -- what a sorted-row cutoff launch gathers"""
    import torch
    from weasal_amd import pyramid
    from weasal_amd.blocks import KPConv
    pts, feats, labels, lens = inp
    b = pyramid.build_batch(cfg, pts, feats, labels, lens, limits)
    lvl = [l for l, m in enumerate(b.neighbors) if m.shape[1] == h]
    mods = [m for m in net.modules() if isinstance(m, KPConv) and m.in_channels == ci and not m.deformable]
    if not lvl or not mods:
        return None
    l, m = lvl[0], mods[0]
    reach = float(m.kernel_points.norm(dim=1).max()) + float(m.KP_extent)
    q = b.points[l]
    sp = torch.cat([q, torch.full((1, 3), 1e6, device=q.device)])
    tot = 0
    for a in range(0, q.shape[0], 32768):
        d2 = ((sp[b.neighbors[l][a:a + 32768]] - q[a:a + 32768, None, :]) ** 2).sum(-1)
        tot += int((d2 <= reach * reach).sum())
    return tot / q.shape[0]


class KernelTimer:
    """HIP-event timing of selected launches on torch's current stream (= the launch stream)"""

    def __init__(self):
        self.enabled = False
        self.records = {}

    def begin(self, key):
        if not self.enabled:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (key, e0, e1)

    def end(self, tok):
        if tok is None:
            return
        key, e0, e1 = tok
        e1.record()
        self.records.setdefault(key, []).append((e0, e1))

    def summary(self):
        out = {}
        for key, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[key] = (float(np.mean(ms)), len(ms))
        return out


def _pyramid_worker(args):
    """one DataLoader-worker's share of the CPU pyramid (its own process, one thread)"""
    import numpy as _np
    from oracle import geom as _geom, pyramid_ref as _pr
    from weasal_amd import config as _wcfg
    from weasal_amd.synthetic import make_inputs as _mk
    cfg_name, wl, seed = args
    cfg = getattr(_wcfg, cfg_name)()
    cfg.feature_dtype = 'f32'
    pts, feats, labels, lens = _mk(seed, 1, wl["points"], wl["radius"], cfg.in_features_dim)
    _np.random.seed(seed)
    t0 = time.perf_counter()
    _pr.segmentation_inputs(cfg, pts, feats, labels, lens, wl["limits"], kind="ref" if _geom.have_ref() else "port")
    return time.perf_counter() - t0, int(lens.sum())


def cpu_pyramid_rate(cfg_name, wl, procs):
    """points/s of the CPU input pyramid with `procs` worker processes (the reference's input_threads = 10
    DataLoader workers, train_DALES_PseudoLabel.py:291-296), one sphere each"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")              # fresh children: never fork a process that holds the GPU
    with ctx.Pool(procs) as pool:
        pool.map(_pyramid_worker, [(cfg_name, wl, 900 + i) for i in range(procs)])       # warm-up: imports, page-in
        t0 = time.perf_counter()
        res = pool.map(_pyramid_worker, [(cfg_name, wl, 100 + i) for i in range(procs)])
        wall = time.perf_counter() - t0
    return sum(n for _, n in res) / wall, wall


def cpu_baseline(cfg_cls, wl, threads):
    """Reference CPU path on a bounded sample: up to FOUR spheres of the workload (half a DALES batch) through
    (a) the pyramid on the CPU geometry core, single thread like one DataLoader worker,
    (b) KPFCNN forward + loss + backward in plain torch on all host threads;
    plus (c) the pyramid rate of 10 worker processes (the reference's input_threads), one sphere each."""
    from oracle import geom, kpconv_ref, pyramid_ref
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.pyramid import PyramidBatch
    from weasal_amd.synthetic import make_inputs
    torch.set_num_threads(threads)
    cfg = cfg_cls()
    cfg.feature_dtype = 'f32'          # the CPU restatement is the reference's fp32 arithmetic
    kind = "ref" if geom.have_ref() else "port"
    n_sph = min(4, int(wl["spheres"]))      # a bounded sample: ~10 s of CPU work on the DALES workload
    pts, feats, labels, lens = make_inputs(12345, n_sph, wl["points"], wl["radius"], cfg.in_features_dim)
    np.random.seed(0)
    t0 = time.perf_counter()
    li = pyramid_ref.segmentation_inputs(cfg, pts, feats, labels, lens, wl["limits"], kind=kind)
    t_pyr = time.perf_counter() - t0
    batch = PyramidBatch([torch.from_numpy(np.ascontiguousarray(a)) for a in li])
    net = KPFCNN(cfg, np.arange(9), [])
    net.train()
    with kpconv_ref.cpu_reference_mode():
        t0 = time.perf_counter()
        out = net(batch, cfg)
        loss = net.loss(out, batch.labels)
        loss.backward()
        t_model = time.perf_counter() - t0
    n = int(lens.sum())
    procs = 10
    try:
        pyr10, wall10 = cpu_pyramid_rate(cfg_cls.__name__, wl, procs)
    except Exception as e:          # reported, never fatal
        pyr10, wall10 = None, repr(e)
    return {"value": n / (t_pyr + t_model), "unit": "points/s", "cores": threads,
            "pyramid_points_per_s_1_thread": n / t_pyr,
            "pyramid_points_per_s_10_processes": pyr10, "pyramid_10_processes_wall_s": wall10,
            "model_points_per_s": n / t_model,
            "kind": "reference" if kind == "ref" else "port",
            "sample": "%d sphere(s), %d pts: pyramid %.2fs on 1 thread (%s geometry core) + KPFCNN fwd+bwd %.2fs "
                      "in plain torch (oracle/kpconv_ref.py restatement of models/blocks.py) on %d threads"
                      % (n_sph, n, t_pyr, "oracle/_ref = the reference's own C++" if kind == "ref" else "oracle port", t_model, threads)}


def self_launch(n):
    """`--gpus N` without a launcher: start the N ranks as FRESH child processes of this one, which has made no GPU call
    (importing torch and counting devices does not initialise HIP; a process that did must never be re-executed).
    Rank 0's stdout (the JSON line) and every rank's stderr are relayed; the exit code is the first non-zero one."""
    import socket
    import subprocess
    backend = os.environ.get("WEASAL_DIST_BACKEND", "")
    ndev = torch.cuda.device_count()
    if backend != "gloo" and ndev < n:
        print("bench.py --gpus %d: only %d GPU(s) visible; RCCL needs one GPU per rank (WEASAL_DIST_BACKEND=gloo lets "
              "ranks share a GPU for a rehearsal)" % (n, ndev), file=sys.stderr)
        return 2
    with socket.socket() as s:                      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for other in procs:             # a dead rank leaves the others waiting in a collective
                        other.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nearest-upsample", type=int, default=0,
                    help="1: opt-in nearest-only upsampling searches (the [N, 1] matrices KP-FCNN actually reads; the reference's "
                         "batch carries the full rows, which stays the default and the contract line)")
    ap.add_argument("--workload", default="dales", choices=["dales", "vaihingen", "dales_deform", "dales_deform_f32", "vaihingen_wl"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--distinct-batches", type=int, default=4)
    ap.add_argument("--dp-buckets", type=int, default=int(os.environ.get("WEASAL_DP_BUCKETS", "1")),
                    help="N > 1 only: >1 cuts the gradient buffer into that many ranges whose all-reduce is launched "
                         "from gradient hooks during backward (opt-in; 1 = one all-reduce after backward)")
    ap.add_argument("--contrast", type=int, default=1,
                    help="1 (default): the step includes KPFCNN.contrast_loss, as the reference's pseudo-label step does from "
                         "epoch 0 (trainer_PseudoLabel.py:204-208, contrast_start = 0); 0: cross entropy (+ regulariser) only")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="infer: a step = GPU pyramid + forward only under no_grad (the testers' voting passes, "
                         "utils/tester_PseudoLabel.py:164); the level-0 32 -> 32 KPConv layers then run as one launch each")
    ap.add_argument("--blas", default="", help="torch.backends.cuda.preferred_blas_library (A/B only)")
    ap.add_argument("--prefetch", type=int, default=1,
                    help="1: build the pyramid of the next batches on a second HIP stream / host thread while the "
                         "current step trains (weasal_amd.prefetch, the GPU counterpart of the reference's DataLoader "
                         "workers); 0: pyramid and training strictly one after the other on one stream")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch.distributed as dist
    from weasal_amd import config as wcfg, dp, ops, pyramid, synthetic
    from weasal_amd.architectures import KPFCNN
    from weasal_amd.trainer import make_optimizer, train_step

    if args.blas:
        torch.backends.cuda.preferred_blas_library(args.blas)
    rank, local_rank, world = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())   # (% only matters for gloo rehearsals)
    torch.cuda.set_device(dev)

    wl = synthetic.WORKLOADS[args.workload]
    cfg_cls = getattr(wcfg, wl["config"])
    cfg = cfg_cls()
    if args.nearest_upsample:
        cfg.nearest_upsample_only = True
    np.random.seed(1234 + rank)
    torch.manual_seed(1234)            # same initial replica everywhere
    weak = getattr(cfg, "model_name", "") == "KPFCNN_mprm"          # BASELINE config 1: the weak-label step
    if weak:
        from weasal_amd.architectures import KPFCNN_mprm
        from weasal_amd.trainer import train_step_weak
        net = KPFCNN_mprm(cfg, np.arange(cfg.num_classes), []).to(dev)
    else:
        net = KPFCNN(cfg, np.arange(9), []).to(dev)
    net.train() if args.mode == "train" else net.eval()
    dp.broadcast_parameters(net)
    opt = make_optimizer(net, cfg)
    sync = dp.GradSync(buckets=args.dp_buckets) if world > 1 else None
    if sync is not None:
        sync.timed = True          # HIP events around the exchange (reported as config.allreduce_ms)

    # inputs resident in HBM before the timed region (seed = 1000*rank + step)
    nd = max(1, min(args.distinct_batches, args.steps + args.warmup))
    inputs, weak_labels = [], []
    for i in range(nd):
        pts, feats, labels, lens = synthetic.make_inputs(1000 * rank + i, wl["spheres"], wl["points"], wl["radius"],
                                                         cfg.in_features_dim)
        inputs.append((torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev),
                       torch.from_numpy(labels).to(dev), lens))
        if weak:
            region, region_lb, cloud_lb, centers = synthetic.make_weak_labels(1000 * rank + i, pts, labels, lens,
                                                                              num_classes=cfg.num_classes)
            weak_labels.append((region, region_lb, torch.from_numpy(cloud_lb).to(dev), torch.from_numpy(centers).to(dev)))
    n_points = int(inputs[0][3].sum())

    timer = KernelTimer()
    ops.set_kernel_timer(timer)
    from weasal_amd import fused

    prefetcher = None
    if args.prefetch:
        from weasal_amd.prefetch import PyramidPrefetcher

        def endless():
            i = 0
            while True:
                yield inputs[i % nd]
                i += 1
        prefetcher = PyramidPrefetcher(cfg, endless(), wl["limits"], depth=2, device=dev, for_training=args.mode == "train")

    from weasal_amd.trainer import InFlightLimiter
    limiter = InFlightLimiter(depth=4)

    waits = {"prefetch": 0.0, "limiter": 0.0}

    served = [0]

    def step(i):
        if prefetcher is not None:
            tw0 = time.perf_counter()
            batch = next(prefetcher)              # blocks while the side stream is still building this batch
            waits["prefetch"] += time.perf_counter() - tw0
            i = served[0]                         # the prefetcher hands the inputs out in source order
            served[0] += 1
        else:
            pts, feats, labels, lens = inputs[i % nd]
            batch = pyramid.build_batch(cfg, pts, feats, labels, lens, wl["limits"], for_training=args.mode == "train")
        if args.mode == "infer":
            with torch.no_grad():
                out = net(batch, cfg)
            loss = out[0].sum() if isinstance(out, tuple) else out.sum()      # (something to read back at the end)
        elif weak:
            batch.region, batch.region_lb, batch.cloud_lb, batch.center_pts = weak_labels[i % nd]
            loss, _ = train_step_weak(net, opt, batch, cfg, grad_sync=sync)
        else:
            loss, _ = train_step(net, opt, batch, cfg, grad_sync=sync, epoch=0 if args.contrast else None)
        tw0 = time.perf_counter()
        limiter.tick(batch)          # bounds the host's run-ahead (4 steps); checks the K4G capacity flags off the hot path
        waits["limiter"] += time.perf_counter() - tw0
        return loss

    # Untimed pre-warm before the W warm-up steps (allocator caches, lazy module loads), then the training
    # loop's usual garbage-collector hygiene: without gc.freeze() CPython's full collection walks the module
    # tree and the autograd graphs for 80-90 ms at about the 17th step of a fresh process
    # (tools/stall_diag.py) -- a host stall, unrelated to the kernels, that drains the launch queue.
    from weasal_amd.trainer import freeze_gc
    if os.environ.get("WEASAL_TRAIN_STREAM_PRIORITY", "0") != "0":
        # experiment: the training work on a high-priority stream (the pyramid builders keep normal priority)
        hp = torch.cuda.Stream(device=dev, priority=-1)
        hp.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(hp)
    for i in range(max(0, 12 - args.warmup)):
        step(i)
    torch.cuda.synchronize()
    freeze_gc()
    for i in range(args.warmup):
        tw = time.perf_counter()
        step(i)
        torch.cuda.synchronize()
        if rank == 0:   # progress on stderr (untimed region) so that a long run is visibly alive
            print("[bench] warmup step %d: %.1f ms" % (i, 1e3 * (time.perf_counter() - tw)), file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    if os.environ.get("WEASAL_STALL_DIAG"):
        pyramid.ACTIVATE_STALLS = []
    waits["prefetch"] = waits["limiter"] = 0.0
    fused.timer_reset()
    # HIP events around the K3 launches inside the block calls (launch stream); WEASAL_TIMED_MIN_ROWS: only layers with at
    # least that many query rows (A/B of the events' own cost)
    fused.set_timed(True, int(os.environ.get("WEASAL_TIMED_MIN_ROWS", "0")))
    from weasal_amd import _lib as _wlib
    launches0 = int(_wlib.lib().ws_launch_count())
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    t_enqueued = time.perf_counter() - t0      # host time to issue the K steps (GPU still running)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    launches = int(_wlib.lib().ws_launch_count()) - launches0      # (the prefetch threads run ahead by up to `depth` batches: +- a pyramid)
    limiter.finish()                 # the K4G capacity flags of the last steps (outside the timed region: everything has completed)
    timer.enabled = False
    fused.set_timed(False)
    if prefetcher is not None:
        prefetcher.close()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0 and pyramid.ACTIVATE_STALLS:
        st = [a.elapsed_time(b) for a, b in pyramid.ACTIVATE_STALLS]
        print("[bench] training stream waiting for the batch's pyramid at the hand-over: mean %.3f ms, max %.3f ms over %d steps"
              % (float(np.mean(st)), float(np.max(st)), len(st)), file=sys.stderr)
    if rank == 0:
        ms = 1000.0 * dt / args.steps
        bf16 = getattr(cfg, 'feature_dtype', 'f32') == 'bf16'
        es = 2 if bf16 else 4
        backend = dist.get_backend() if world > 1 else None
        exch = "" if world == 1 else (" + RCCL grad all-reduce" if backend == "nccl" else " + %s grad all-reduce (NOT RCCL: rehearsal backend)" % backend)
        res = {"metric": "points/sec fwd+bwd KPFCNN on DALES spheres; achieved HBM GB/s on KPConv gather" if args.mode == "train" else
                         "points/sec forward-only (inference pass) KPFCNN on DALES spheres; achieved HBM GB/s on the KPConv layer",
               "value": world * n_points * args.steps / dt, "unit": "points/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
               # host side of the training thread: wall time to issue the K steps, and the same without the time it spent
               # blocked (waiting for the side stream's next batch / for the GPU through the run-ahead limiter)
               "host_issue_ms_per_step": 1000.0 * t_enqueued / args.steps,
               "host_busy_ms_per_step": 1000.0 * (t_enqueued - waits["prefetch"] - waits["limiter"]) / args.steps,
               "host_wait_prefetch_ms_per_step": 1000.0 * waits["prefetch"] / args.steps,
               "host_wait_gpu_ms_per_step": 1000.0 * waits["limiter"] / args.steps,
               "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
               "config": {"workload": wl["name"] + (", bf16 feature rows / fp32 accumulate / fp32 geometry" if bf16 else ", fp32")
                          + (", step = GPU pyramid + fwd + loss + bwd" + exch + " + SGD" if args.mode == "train" else
                             ", step = GPU pyramid + forward under no_grad (the voting test's pass)")
                          + ("; contrast_loss term included (trainer_PseudoLabel.py:204-208)" if args.contrast else "; contrast_loss term (trainer_PseudoLabel.py:204-208) left out (--contrast 0)")
                          + ("; pyramid of the next batch overlapped on a second stream" if args.prefetch else "")
                          + ("; OPT-IN nearest-only upsampling searches ([N, 1] matrices instead of the reference's full rows)"
                             if pyramid.nearest_upsample_only(cfg) else ""),
                          "points_per_step_per_gpu": n_points, "parallelism": "dp%d" % world,
                          "dist_backend": backend,
                          # what the process group itself says (not the flag): ranks that took part in the exchange
                          "dist_world_size": dist.get_world_size() if world > 1 else 1,
                          "allreduce_bytes": sync.nbytes() if sync is not None else 0,
                          "allreduce_ms": sync.mean_ms() if sync is not None else None,
                          # kernels this library launched per step, both streams (framework kernels -- dropout, the random draw of
                          # the contrastive slice, autograd's gradient sums -- are not counted: ~15 per step)
                          "library_launches_per_step": round(launches / args.steps, 1),
                          "host_threads_per_rank": "1 training + %d pyramid prefetch" % (prefetcher.workers if prefetcher is not None else 0),
                          "final_loss": float(loss.item())}}
        # ---- roofline of the fused KPConv gather kernel (K3) on the largest layer: HIP events on the launch stream,
        #      recorded inside the block calls (ws_timer_*) or around the operator launch (ops timer)
        summ = timer.summary()
        recs, layer_ms = {}, {}
        for nq_, h_, ci_, ms_, ml_ in fused.timer_records(layer=True):
            recs.setdefault(("kpconv_gather_fwd", nq_, h_, ci_), []).append(ms_)
            layer_ms.setdefault(("kpconv_gather_fwd", nq_, h_, ci_), []).append(ml_)
        for k_, v_ in recs.items():
            if k_ in summ:
                tot = summ[k_][0] * summ[k_][1] + sum(v_)
                summ[k_] = (tot / (summ[k_][1] + len(v_)), summ[k_][1] + len(v_))
            else:
                summ[k_] = (float(np.mean(v_)), len(v_))
        fwd = {k: v for k, v in summ.items() if k[0] == "kpconv_gather_fwd" and k[3] >= 8}
        if fwd:
            key = max(fwd, key=lambda k: b_fwd(k[1], k[2], k[3], k[3], es))     # the launch that moves the most bytes
            ms_k, count = fwd[key]
            _, nq, h, ci = key
            bytes_alg = b_fwd(nq, h, ci, ci, es)
            deform = "deform" in args.workload
            h_reach = None
            if deform:
                # rows searched with the deformable radius, walked only up to the kernel's reach (sorted-row cutoff): price the
                # launch on the neighbours INSIDE the reach (the ones with a non-zero influence, which any implementation has
                # to gather), not on all H columns -- otherwise skipped columns would count as bytes moved
                h_reach = rows_inside_reach(net, cfg, inputs[0], wl["limits"], h, ci)
                if h_reach is not None:
                    bytes_alg = int(nq * h_reach * (8 + 12 + es * ci) + nq * (12 + es * ci) + 4 * 15 * ci * ci + 180)
            achieved = bytes_alg / (ms_k * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(REPO, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    tkey = ("%s:%d:%d" if args.mode == "train" else "infer:%s:%d:%d") % (args.workload, h, ci)
                    traffic = tj.get("kpconv_gather_fwd_bytes_per_launch", {}).get(tkey)
                except Exception:
                    traffic = None
            # the kernel the library's dispatcher launches for this layer (ws_kpconv_gather_fwd_variant: the same selection
            # rules as the launch).  In the deformable workloads the timed launch is the rigid OFFSET convolution of the
            # first deformable block; rows searched with the deformable radius take the sorted-row cutoff.
            import ctypes as _C
            from weasal_amd import _lib as _wl
            name = _C.create_string_buffer(256)
            _wl.check(_wl.lib().ws_kpconv_gather_fwd_variant(ci, 0, 0, 0, 1 if bf16 else 0, 1 if deform else 0, name, 256))
            one_launch = (args.mode == "infer" and ci == 32 and not bf16 and not deform and fused.FUSED_INFER)
            if one_launch:
                kname = ("kpconv_gather_fwd_mfma_kernel<NT=2, MODE=0, DEF=false, VECROW=true, float, GS=2, CUT=false, FUSE=true> "
                         "(gather AND the 15 Ci x Co contraction in one launch: ws_kpconv_layer_fwd_fused)")
            else:
                kname = name.value.decode()
            res["roofline"] = {"bound": "hbm",
                               "kernel": kname + " on N=%d queries, H=%d, Ci=%d" % (nq, h, ci)
                                         + ("; rows from the deformable search radius: the kernel walks each (distance-sorted) row only "
                                            "up to the reach of the kernel points (%.1f of the %d columns on average): the algorithmic bytes "
                                            "count those neighbours only" % (h_reach, h) if deform and h_reach is not None else ""),
                               "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "achieved_is": "SURVEY 8d logical-gather bytes B_fwd / launch time (every neighbour reference "
                                              "counted once; L2 reuse lets it exceed what HBM alone delivers)",
                               "achieved_hbm_measured": (traffic / (ms_k * 1e-3) / 1e9) if traffic else None,
                               "u_fwd_bytes": u_fwd(nq, nq, h, ci, ci, es),
                               "algorithmic_bytes_per_launch": bytes_alg, "avg_launch_ms": ms_k, "launches_timed": count}
            if key in layer_ms:
                # the whole layer B_fwd describes (gather + the contraction wf x W that follows it: two launches, `wf` makes
                # a round trip through HBM between them), by HIP events from the start of the first to the end of the second
                ml = float(np.mean(layer_ms[key]))
                res["roofline"]["layer"] = {"what": "K3 gather + contraction [N,15Ci]x[15Ci,Co] of the same layer (two launches)"
                                                    if not one_launch else
                                                    "the whole KPConv layer in one launch (gather + contraction + bias + LeakyReLU)",
                                            "avg_ms": ml, "achieved": bytes_alg / (ml * 1e-3) / 1e9,
                                            "frac": bytes_alg / (ml * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # per-launch means of the KPConv gather kernels, grouped by layer (N varies by a few points from
            # batch to batch with the random grid orientation: rounded to two significant digits)
            agg = {}
            for k, v in summ.items():
                if k[1] < 4096:       # the deep levels (a few hundred rows, H varying from batch to batch) would only clutter the line
                    continue
                nr = int(float("%.2g" % k[1]))
                key = "%s N~%d H=%d C=%d" % (k[0], nr, k[2], k[3])
                a = agg.setdefault(key, [0.0, 0])
                a[0] += v[0] * v[1]
                a[1] += v[1]
            res["kernels_ms"] = {k: round(a[0] / a[1], 4) for k, a in sorted(agg.items())}
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(cfg_cls, wl, min(os.cpu_count() or 1, 16))
            except Exception as e:   # the baseline is a report, never a reason to lose the GPU line
                res["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
