"""One training step with the reference's recipe (utils/trainer_PseudoLabel.py:80-87,199-219),
plus the data-parallel gradient exchange the reference does not have.

  zero_grad -> net(batch, config) -> net.loss -> backward -> [all-reduce grads / world] ->
  clip_grad_value_(grad_clip_norm) -> SGD(momentum, weight_decay; 'offset' params at
  lr * deform_lr_factor).step()

The reference's per-step torch.cuda.empty_cache() + synchronize() (:221-222) serialise the GPU
and are deliberately not reproduced.
"""
import torch


class FusedSGD(torch.optim.SGD):
    """torch.optim.SGD (same constructor, param_groups, state / state_dict: `momentum_buffer` per parameter) whose step on
    GPU parameters is ONE pass of the HIP kernel ws_sgd_step per parameter group -- value clipping
    (torch.nn.utils.clip_grad_value_, trainer_PseudoLabel.py:216), weight decay, momentum and the update read and write
    every element once, instead of a clamp and three `foreach` passes.  Anything the kernel does not cover (CPU
    parameters as in the gloo / oracle tests, Nesterov, dampening, maximize, non-f32 or sparse gradients) takes
    torch's own step after the same clipping."""

    def _fusable(self, group, params):
        if group.get('nesterov') or group.get('dampening', 0) != 0 or group.get('maximize'):
            return False
        for p in params:
            g = p.grad
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.dtype == torch.float32
                    and not g.is_sparse and g.is_contiguous() and g.device == p.device):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None, clip_value=0.0):
        import ctypes
        from . import _lib
        groups = [(g, [p for p in g['params'] if p.grad is not None]) for g in self.param_groups]
        if closure is not None or not all(self._fusable(g, ps) for g, ps in groups if ps):
            if clip_value and clip_value > 0:
                torch.nn.utils.clip_grad_value_([p for _, ps in groups for p in ps], clip_value)
            return super().step(closure)
        lib = _lib.lib()
        for group, params in groups:
            momentum = float(group['momentum'])
            # parameters without a momentum buffer yet take torch's first step (buf = g): normally all of them or none
            fresh = [p for p in params if self.state[p].get('momentum_buffer') is None] if momentum != 0 else []
            seasoned = [p for p in params if self.state[p].get('momentum_buffer') is not None] if momentum != 0 else params
            for first, sel in ((True, fresh), (False, seasoned)):
                if not sel:
                    continue
                with torch.cuda.device(sel[0].device):
                    n = len(sel)
                    arr = ctypes.c_void_p * n
                    if momentum != 0:
                        if first:
                            for p in sel:
                                self.state[p]['momentum_buffer'] = torch.empty_like(p, memory_format=torch.contiguous_format)
                        bufs = arr(*[self.state[p]['momentum_buffer'].data_ptr() for p in sel])
                    else:
                        bufs = None
                    _lib.check(lib.ws_sgd_step(arr(*[p.data_ptr() for p in sel]), arr(*[p.grad.data_ptr() for p in sel]), bufs,
                                               (ctypes.c_int64 * n)(*[p.numel() for p in sel]), n, float(group['lr']), momentum,
                                               float(group['weight_decay']), float(clip_value or 0.0), int(first),
                                               _lib.current_stream()))
        return None


def make_optimizer(net, config):
    """SGD with a second parameter group for the deformable offsets (trainer_PseudoLabel.py:80-87); on the GPU the
    step is the fused kernel of FusedSGD"""
    deform_params = [v for k, v in net.named_parameters() if 'offset' in k]
    other_params = [v for k, v in net.named_parameters() if 'offset' not in k]
    deform_lr = config.learning_rate * config.deform_lr_factor
    # foreach: for the cases FusedSGD hands to torch (same arithmetic, multi-tensor launches)
    foreach = bool(other_params) and other_params[0].is_cuda
    return FusedSGD([{'params': other_params}, {'params': deform_params, 'lr': deform_lr}],
                    lr=config.learning_rate, momentum=config.momentum,
                    weight_decay=config.weight_decay, foreach=foreach or None)


def train_step(net, optimizer, batch, config, grad_sync=None, epoch=None):
    """-> (loss tensor, logits).  grad_sync: optional callable(net) run between backward and clip
    (weasal_amd.dp.GradSync: one flat RCCL all-reduce).  epoch: when given, the supervised contrastive
    loss joins from `config.contrast_start` on (trainer_PseudoLabel.py:204-208)."""
    # single process: gradients are dropped and re-assigned (autograd adopts the tensor the backward produced: no zero-fill
    # and no accumulate-add kernel per parameter); data parallel: they stay the views of GradSync's flat buffer
    optimizer.zero_grad(set_to_none=grad_sync is None)
    outputs = net(batch, config)
    loss = net.loss(outputs, batch.labels)
    if epoch is not None and epoch >= getattr(config, 'contrast_start', 1 << 30):
        loss = loss + net.contrast_loss(outputs, batch.labels, config)
    if grad_sync is not None and hasattr(grad_sync, "arm"):
        grad_sync.arm()          # the overlapped exchange's gradient hooks run for this backward only
    loss.backward()
    if grad_sync is not None:
        grad_sync(net)
    if isinstance(optimizer, FusedSGD):
        optimizer.step(clip_value=config.grad_clip_norm)     # clip + weight decay + momentum + update in one pass
        return loss, outputs
    if config.grad_clip_norm > 0:
        params = getattr(net, "_param_list", None)
        if params is None:                   # the parameter set is static: no module-tree walk per step
            params = [p for p in net.parameters()]
            net._param_list = params
        torch.nn.utils.clip_grad_value_(params, config.grad_clip_norm)
    optimizer.step()
    return loss, outputs


def train_step_weak(net, optimizer, batch, config, grad_sync=None):
    """One step of the WEAK-LABEL trainer (utils/trainer_WeakLabel.py:181-216) for KPFCNN_mprm:
      skip batches without sub-region labels (:181-184) -> zero_grad -> logits, class_logits, cam = net(batch, config) ->
      region_mprm_loss(cam, batch.region, batch.region_lb, batch.lengths[0]) or class_logits_loss(class_logits,
      batch.cloud_lb) by config.loss_type (:203-207) -> backward -> [all-reduce] -> clip_grad_NORM_(grad_clip_norm) (:216;
      the pseudo-label trainer clips by value) -> SGD step.
    -> (loss, (logits, class_logits, cam)), or (None, None) for a skipped batch."""
    if not any(len(r) > 0 for r in batch.region):
        return None, None
    optimizer.zero_grad(set_to_none=grad_sync is None)
    logits, class_logits, cam = net(batch, config)
    if config.loss_type == 'region_mprm_loss':
        lens0 = batch.lengths_host[0] if getattr(batch, 'lengths_host', None) is not None else batch.lengths[0]
        loss = net.region_mprm_loss(cam, batch.region, batch.region_lb, lens0)       # (host lengths: no device read-back)
    else:
        loss = net.class_logits_loss(class_logits, batch.cloud_lb)
    if grad_sync is not None and hasattr(grad_sync, "arm"):
        grad_sync.arm()
    loss.backward()
    if grad_sync is not None:
        grad_sync(net)
    if config.grad_clip_norm > 0:
        params = getattr(net, "_param_list", None)
        if params is None:
            params = [p for p in net.parameters()]
            net._param_list = params
        # total norm, coefficient and scaling stay on the device (no synchronisation); the value is kept for the log line
        net.grad_norm = torch.nn.utils.clip_grad_norm_(params, config.grad_clip_norm)
    if isinstance(optimizer, FusedSGD):
        optimizer.step(clip_value=0.0)
    else:
        optimizer.step()
    return loss, (logits, class_logits, cam)


def freeze_gc():
    """Call once after the model, optimizer and the first batches exist: moves everything alive into the
    garbage collector's permanent generation.  Without it CPython's full (generation-2) collection walks the
    module tree, the autograd graphs and every cached tensor wrapper -- an 80-90 ms host stall every few
    hundred thousand allocations (step ~17 of a fresh process, tools/stall_diag.py), long enough to drain the
    GPU's launch queue.  The collector stays enabled; young generations are cheap."""
    import gc
    gc.collect()
    gc.freeze()


class InFlightLimiter:
    """Bounds how far the host runs ahead of the GPU: call once per training step; it records an event and
    waits for the event of `depth` steps ago.  With the host issuing a step in ~11 ms and the GPU taking ~16,
    an unbounded loop queues thousands of launches and every queued batch keeps its memory
    alive.  A few steps of slack keep the GPU fed and bound the memory of queued batches."""

    def __init__(self, depth=3, check_every=None):
        import os
        self.depth = max(1, int(depth))
        self.check_every = max(1, int(os.environ.get("WEASAL_OVERFLOW_CHECK_EVERY", "16") if check_every is None else check_every))
        self.events = []
        self._pinned = []
        self._slot = -1
        self._acc = None
        self._ticks = 0

    def tick(self, batch=None):
        """batch: the PyramidBatch this step trained on.  Its table-free backward launches (K4G, slab form) write a capacity
        flag (`SearchGrid.overflow`, set if a support ever had more incoming pairs than the slab holds).  The host already
        refuses the slab form for a grid whose search recorded rows longer than 128 (`SearchGrid.max_count`: such layers
        take the queue form or the transposed table), so a set flag means a stale or mutated grid: a pure assertion.
        Callers that want it checked pass `batch` here and call finish() after the last step.  The flags are folded into a running
        maximum on the device (two tiny launches); every `check_every`-th step that maximum travels to the host with an
        asynchronous copy and is checked `depth` steps later, when its event has completed anyway: never silent, never a
        synchronisation on the step.  (A device-to-host copy EVERY step cost 0.25 ms of idle training stream per step:
        the copy's system-scope release writes the L2 back before the next step's first kernel may start.)  Call
        finish() after the last step."""
        if not torch.cuda.is_available():
            return
        flags = None
        grids = getattr(batch, "search_grids", None) if batch is not None else None
        self._ticks += 1
        if grids:
            dev = torch.cat([g.overflow for _, g in grids])
            if self._acc is not None and self._acc.shape != dev.shape:
                flags = self._land(self._acc)        # another number of grids than before: what was folded so far goes out now
                self._acc = dev
            else:
                self._acc = dev if self._acc is None else torch.maximum(self._acc, dev)
        if flags is None and self._acc is not None and self._ticks % self.check_every == 0:
            flags = self._land(self._acc)
            self._acc = None
        e = torch.cuda.Event()
        e.record()
        self.events.append((e, flags))
        if len(self.events) > self.depth:
            ev, fl = self.events.pop(0)
            ev.synchronize()
            self._check(fl)

    def _land(self, dev):
        n = int(dev.shape[0])
        if len(self._pinned) <= self.depth + 1:              # a small ring of pinned landing buffers, allocated once
            self._pinned.append(torch.zeros(max(64, n), dtype=dev.dtype, pin_memory=True))
        self._slot = (self._slot + 1) % len(self._pinned)
        if self._pinned[self._slot].shape[0] < n:            # more grids than the buffer was sized for: a larger one
            self._pinned[self._slot] = torch.zeros(n, dtype=dev.dtype, pin_memory=True)
        flags = self._pinned[self._slot][:n]
        flags.copy_(dev, non_blocking=True)
        return flags

    @staticmethod
    def _check(fl):
        if fl is not None and int(fl.max()) != 0:
            raise RuntimeError("KPConv backward through the search grid overflowed its pair slab (%d incoming pairs): the "
                               "grid does not belong to the index matrix it was used with" % int(fl.max()))

    def finish(self):
        """after the last step: wait for everything in flight and check the flags not yet looked at"""
        if not torch.cuda.is_available():
            return
        last = self._land(self._acc) if self._acc is not None else None
        self._acc = None
        torch.cuda.current_stream().synchronize()
        for _, fl in self.events:
            self._check(fl)
        self.events = []
        self._check(last)
