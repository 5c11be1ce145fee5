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


def init_from_env(backend=None):
    """-> (rank, local_rank, world).  Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
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
    """Flat-bucket gradient averaging.  On the first call the gradients that exist after backward
    are re-homed as views of one contiguous fp32 buffer (parameters the graph never touches -- e.g.
    the BatchNorm1d weights that never run -- keep grad None, so the optimizer skips them exactly as
    in the single-process reference); every call is then one all-reduce(sum) + one scale.

    ``buckets > 1`` (opt-in) overlaps the exchange with backward: the flat buffer is cut into that many
    contiguous ranges in parameter order; backward fills it from the end (the head's gradients come first),
    and a post-accumulate hook per parameter launches the asynchronous all-reduce of a range as soon as its
    last gradient has landed.  ``__call__`` then only waits and scales.  Element-wise the result is the same
    sum (bit-identical on 2 ranks: tests/test_dp_cpu.py)."""

    def __init__(self, group=None, buckets=1):
        self.group = group
        self.flat = None
        self.params = None
        self.buckets = max(1, int(buckets))
        self._ranges = []        # (start, end) element ranges of the buckets
        self._pending = []       # gradients still missing per bucket in the current backward
        self._sizes = []
        self._handles = []
        self._launched = []
        self._bucket_of = {}

    def _build(self, net):
        self.params = [p for p in net.parameters() if p.grad is not None]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        spans = []
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            view.copy_(p.grad)
            p.grad = view
            spans.append((off, off + n))
            off += n
        if self.buckets > 1:
            target = total / float(self.buckets)
            self._ranges, self._sizes = [], []
            start_i = 0
            for b in range(self.buckets):
                end_i = start_i
                limit = total if b == self.buckets - 1 else (b + 1) * target
                while end_i < len(self.params) and (spans[end_i][1] <= limit or end_i == start_i):
                    end_i += 1
                if b == self.buckets - 1:
                    end_i = len(self.params)
                if end_i > start_i:
                    self._ranges.append((spans[start_i][0], spans[end_i - 1][1]))
                    self._sizes.append(end_i - start_i)
                    for p in self.params[start_i:end_i]:
                        self._bucket_of[p] = len(self._ranges) - 1
                start_i = end_i
            self._reset()
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _reset(self):
        self._pending = list(self._sizes)
        self._launched = [False] * len(self._ranges)
        self._handles = []

    def _launch(self, b):
        s, e = self._ranges[b]
        self._handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._launched[b] = True

    def _on_grad(self, p):
        b = self._bucket_of.get(p)
        if b is None or self._launched[b]:
            return
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def nbytes(self):
        return 0 if self.flat is None else self.flat.numel() * 4

    def __call__(self, net):
        if not (dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return
        first = self.flat is None
        if first:
            self._build(net)
        else:
            for p in self.params:       # a grad replaced by autograd (set_to_none) would break the views
                if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                        p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * 4:
                    raise RuntimeError("GradSync: gradient left the flat bucket; use zero_grad(set_to_none=False)")
        if self.buckets > 1 and not first:
            for b in range(len(self._ranges)):      # a range whose hooks did not all fire (never on a static graph)
                if not self._launched[b]:
                    self._launch(b)
            for h in self._handles:
                h.wait()
            self._reset()
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(dist.get_world_size(self.group))
