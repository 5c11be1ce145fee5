"""Data parallelism over the 8 GPUs of one node: one process per GPU, independent sphere batches
per rank, ONE flat gradient all-reduce per step (RCCL over xGMI via torch.distributed, backend
"nccl" == RCCL on ROCm; "gloo" on CPU for tests).

The reference is single-process (utils/trainer_PseudoLabel.py:90-94); the exchange sits between
``loss.backward()`` and ``clip_grad_value_`` (:214-218).  BatchNorm is an identity on this path
(models/blocks.py:453-463), so gradients are the only cross-rank state.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """-> (rank, local_rank, world).  Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).
    force: create the process group also for WORLD_SIZE = 1 (tests that run the RCCL branch on one GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # WEASAL_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box)
            backend = os.environ.get("WEASAL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def broadcast_parameters(net, src=0):
    """same initial replica on every rank"""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(net.parameters()) + list(net.buffers()):
            dist.broadcast(t.data, src)


class GradSync:
    """Flat-bucket gradient averaging between ``loss.backward()`` and ``clip_grad_value_``.

    Layout: ONE contiguous fp32 buffer over every ``requires_grad`` parameter in parameter order -- a function of
    the module alone, so it is the same on every rank whatever each rank's graph touched.  A parameter's gradient
    is re-homed as a view of its range the first time it exists (also when it first appears at a later step: a loss
    term switched on by epoch, an unfrozen layer); a parameter the graph never touches keeps ``grad is None`` --
    its range stays zero and the optimizer skips it exactly as in the single-process reference (the BatchNorm1d
    weights that never run, models/blocks.py:453-463).  Every call is one all-reduce(sum) + one scale.

    Replica agreement: the buffer carries two trailing floats, (number of gradients present on this rank, checksum
    of which ones).  Their sums come back with the same all-reduce and are read on the host one step later (an
    asynchronous copy, no synchronisation on the step): if the ranks did not agree on the SET of parameters that have
    gradients -- the one way this scheme could let replicas drift -- the next call raises.  ``fill_missing=True``
    makes the set static instead: every requires_grad parameter always has a (possibly zero) gradient; note that SGD's
    weight decay then also reaches parameters that never ran.

    ``buckets > 1`` (opt-in) overlaps the exchange with backward: the buffer is cut into that many contiguous ranges
    in parameter order; backward fills it from the end (the head's gradients come first) and a post-accumulate hook
    per parameter launches the asynchronous all-reduce of a range as soon as the gradients it had in the previous step
    have landed.  The hooks only run between ``arm()`` (called by ``trainer.train_step`` right before backward) and
    ``__call__``: a backward that is not armed, or a gradient that arrives for a range already reduced (a second
    backward before the exchange, a gradient that first appears in this backward), raises instead of adding into
    reduced data.  Element-wise the result is the same sum (bit-identical on 2 ranks: tests/test_dp_cpu.py).

    The RCCL path (backend "nccl") has not been executed on a multi-GPU node yet (no such node was available to
    rounds 1-2): the multi-rank semantics are tested over gloo, the RCCL branch itself on one rank
    (``single_rank_exchange``, tests/test_dp_gpu.py)."""

    TAIL = 2

    def __init__(self, group=None, buckets=1, fill_missing=False, single_rank_exchange=False):
        self.group = group
        # single_rank_exchange: run the exchange also in a group of ONE rank (a self all-reduce) -- the only way to execute the
        # RCCL branch (device buffer, asynchronous handles from the autograd thread, pinned signature copy) on a 1-GPU box
        self._min_world = 1 if single_rank_exchange else 2
        self.flat = None
        self.params = None
        self.buckets = max(1, int(buckets))
        self.fill_missing = bool(fill_missing)
        self.total = 0
        self._views = []
        self._present = []
        self._index = {}
        self._ranges = []        # (start, end) element ranges of the buckets
        self._members = []       # parameter indices per bucket
        self._pending = []
        self._handles = []
        self._launched = []
        self._bucket_of = []
        self._armed = False
        self._ran = False        # the armed backward has finished (overlapped form)
        self._cb_queued = False
        self._sig_dev = None     # this rank's (count, checksum), on the device
        self._sig_local = None
        self._sig_check = None   # (pinned host sums, event, expected) of the previous call
        self._host = None
        self.timed = False       # bench.py: HIP events around the exchange (GPU tensors only)
        self._times = []

    def mean_ms(self):
        """mean duration of the exchanges timed so far (HIP events on the stream the all-reduce was issued from:
        buckets == 1 only -- the overlapped form's ranges run under backward); None without records"""
        if not self._times:
            return None
        ms = [a.elapsed_time(b) for a, b in self._times]
        self._times = []
        return float(sum(ms) / len(ms))

    # ---- layout ---------------------------------------------------------------------------------------------
    def _layout(self, net):
        self.params = [p for p in net.parameters() if p.requires_grad]
        if not self.params:
            raise RuntimeError("GradSync: the module has no trainable parameter")
        dev = self.params[0].device
        spans, off = [], 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise RuntimeError("GradSync: fp32 master parameters expected")
            spans.append((off, off + p.numel()))
            off += p.numel()
        self.total = off
        self.flat = torch.zeros(off + self.TAIL, dtype=torch.float32, device=dev)
        self._views = [self.flat[a:b].view_as(p) for (a, b), p in zip(spans, self.params)]
        self._present = [False] * len(self.params)
        self._index = {p: i for i, p in enumerate(self.params)}
        self._sig_dev = torch.zeros(self.TAIL, dtype=torch.float32, device=dev)
        if self.buckets > 1:
            target = off / float(self.buckets)
            self._ranges, self._members, self._bucket_of = [], [], [0] * len(self.params)
            start_i = 0
            for b in range(self.buckets):
                end_i = start_i
                limit = off if b == self.buckets - 1 else (b + 1) * target
                while end_i < len(self.params) and (spans[end_i][1] <= limit or end_i == start_i):
                    end_i += 1
                if b == self.buckets - 1:
                    end_i = len(self.params)
                if end_i > start_i:
                    self._ranges.append((spans[start_i][0], spans[end_i - 1][1]))
                    self._members.append(list(range(start_i, end_i)))
                    for i in range(start_i, end_i):
                        self._bucket_of[i] = len(self._ranges) - 1
                start_i = end_i
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _rehome(self, i):
        """make parameter i's gradient the view of its range; -> True if it has a gradient"""
        p, v = self.params[i], self._views[i]
        g = p.grad
        if g is None:
            if self.fill_missing:
                p.grad = v
                return True
            if self._present[i]:
                v.zero_()                      # had a gradient before (zero_grad(set_to_none=True)): range back to zero
            return False
        if g.data_ptr() != v.data_ptr() or g.shape != v.shape:
            v.copy_(g)
            p.grad = v
        return True

    # ---- overlapped form --------------------------------------------------------------------------------------
    def arm(self):
        """call right before ``backward`` (trainer.train_step does): lets the gradient hooks of the overlapped form
        run for exactly one backward"""
        if self.buckets > 1 and self.flat is not None:
            if (self._armed or self._ran) and any(self._launched):
                raise RuntimeError("GradSync.arm(): the previous backward's exchange was never finished (call the "
                                   "GradSync object between backward and the optimizer step)")
            self._pending = [sum(1 for i in m if self._present[i]) for m in self._members]
            self._launched = [False] * len(self._ranges)
            self._handles = []
        self._armed = True
        self._ran = False

    def _backward_done(self):
        self._armed = False
        self._ran = True
        self._cb_queued = False

    def _launch(self, b):
        s, e = self._ranges[b]
        self._handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._launched[b] = True

    def _on_grad(self, p):
        if not (dist.is_initialized() and dist.get_world_size(self.group) >= self._min_world):
            return
        if not self._armed:
            raise RuntimeError("GradSync(buckets>1): backward ran without arm() -- a second backward before the exchange "
                               "(gradient accumulation, an auxiliary loss) would add into ranges that are already "
                               "reduced; sum the losses into one backward or use buckets=1")
        if not self._cb_queued:      # the first gradient of this backward: disarm when this backward ends
            self._cb_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._backward_done)
        i = self._index[p]
        b = self._bucket_of[i]
        if self._launched[b]:
            raise RuntimeError("GradSync(buckets>1): a gradient arrived for a range whose all-reduce is already in flight "
                               "(parameter %d first got a gradient in this backward, or backward ran twice); use "
                               "buckets=1 for a step whose gradient set changes" % i)
        was = self._present[i]
        self._rehome(i)
        if was:
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)

    # ---- the exchange -------------------------------------------------------------------------------------------
    def nbytes(self):
        return 0 if self.flat is None else self.total * 4

    def verify(self):
        """raise if the ranks disagreed on the set of parameters with gradients in the previous exchange"""
        chk, self._sig_check = self._sig_check, None
        if chk is None:
            return
        host, event, expect = chk
        if event is not None:
            event.synchronize()
        got = (float(host[0]), float(host[1]))
        if got != expect:
            raise RuntimeError("GradSync: ranks disagree on which parameters have gradients (sum of counts/checksums %r, "
                               "expected %r): the replicas have diverged -- make the loss terms the same on every rank "
                               "or use GradSync(fill_missing=True)" % (got, expect))

    def __call__(self, net):
        if not (dist.is_initialized() and dist.get_world_size(self.group) >= self._min_world):
            self._armed = False
            return
        world = dist.get_world_size(self.group)
        self.verify()
        first = self.flat is None
        if first:
            self._layout(net)
        count, chk = 0, 0
        for i in range(len(self.params)):
            here = self._rehome(i)
            if here and not self._present[i] and not first and self.buckets > 1 and self._launched and \
                    self._launched[self._bucket_of[i]]:
                raise RuntimeError("GradSync(buckets>1): parameter %d got its first gradient after its range was reduced" % i)
            self._present[i] = here
            if here:
                count += 1
                chk = (chk * 31 + i + 1) % 8191
        sig = (float(count), float(chk))
        if sig != self._sig_local:
            self._sig_local = sig
            self._sig_dev.copy_(torch.tensor(sig, dtype=torch.float32))
        tail = self.flat[self.total:]
        tail.copy_(self._sig_dev)
        if self.buckets > 1 and not first and (self._armed or self._ran):
            for b in range(len(self._ranges)):      # ranges whose hooks did not all fire
                if not self._launched[b]:
                    self._launch(b)
            self._handles.append(dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            for h in self._handles:
                h.wait()
            self._handles = []
            self._launched = [False] * len(self._ranges)
        else:
            ev = None
            if self.timed and self.flat.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if ev is not None:
                ev[1].record()
                self._times.append(ev)
                del self._times[:-64]
        self._armed = self._ran = False
        self.flat[:self.total].div_(world)
        # the summed signature travels to the host asynchronously and is checked at the next call
        if self.flat.is_cuda:
            if self._host is None:          # two pinned landing buffers, used alternately
                self._host = [torch.empty(self.TAIL, dtype=torch.float32, pin_memory=True) for _ in range(2)]
            self._host.reverse()
            host = self._host[0]
            host.copy_(tail, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
        else:
            host, event = tail.clone(), None
        self._sig_check = (host, event, (sig[0] * world, sig[1] * world))
