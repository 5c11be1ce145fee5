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
    in the single-process reference); every call is then one all-reduce(sum) + one scale."""

    def __init__(self, group=None):
        self.group = group
        self.flat = None
        self.params = None

    def _build(self, net):
        self.params = [p for p in net.parameters() if p.grad is not None]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            view.copy_(p.grad)
            p.grad = view
            off += n

    def nbytes(self):
        return 0 if self.flat is None else self.flat.numel() * 4

    def __call__(self, net):
        if not (dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return
        if self.flat is None:
            self._build(net)
        else:
            for p in self.params:       # a grad replaced by autograd (set_to_none) would break the views
                if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                        p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * 4:
                    raise RuntimeError("GradSync: gradient left the flat bucket; use zero_grad(set_to_none=False)")
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(dist.get_world_size(self.group))
