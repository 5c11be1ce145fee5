"""Asynchronous input pipeline on the GPU.

The reference hides its CPU pyramid construction behind `DataLoader(num_workers=10)` worker
processes (train_DALES_PseudoLabel.py:291-296): batches are built while the previous step
trains.  Here the pyramid is built by HIP kernels, so the same overlap is one background thread
that builds the next batches on its OWN HIP stream while the training stream runs forward /
backward (the geometry kernels are scalar-unit / latency bound and co-run with the GEMMs).
The hand-over is an event + `record_stream` (PyramidBatch.activate); ctypes and torch release
the GIL inside kernels launches and stream synchronisations, so the two threads interleave.
"""
import queue
import threading

import torch

from . import pyramid


class _Replay:
    """hands out pre-drawn uniform numbers through the `rand(n)` call of pyramid.batch_grid_subsampling"""

    def __init__(self, values):
        self.values = values
        self.pos = 0

    def rand(self, n):
        out = self.values[self.pos:self.pos + n]
        if out.shape[0] != n:
            raise RuntimeError("prefetcher: more grid-orientation draws than pre-drawn for this batch")
        self.pos += n
        return out


class PyramidPrefetcher:
    """Iterates PyramidBatch objects built ahead of time from `source`, an iterable of
    (points, features, labels, lengths) with device tensors and host lengths.

    `workers` background threads, each with its OWN HIP stream and geometry workspaces, build alternate batches and hand
    them over in source order.  One pyramid needs five host round trips (the subsampled sizes of the four levels and the
    row widths of the searches must reach the host before the next level can be shaped), each of which waits for that
    stream's queued kernels: a single builder needs ~ 14 ms per DALES batch next to a training stream (its kernels take
    3.5-4.7 ms), which is the training step's own time -- measured, one builder keeps up with no loss of step time but
    with the training thread waiting 0.2-0.3 ms per step for the next batch (14.34-14.45 vs 14.38-14.51 ms with two).
    Two builders (the default, WEASAL_PREFETCH_WORKERS) overlap their chains and keep a margin.  With `seed` each worker draws the grid orientations from its own RandomState(seed + worker);
    without, the draws come from the global np.random in source order (taken under the source lock when a batch is handed to a
    worker): one process-wide stream consumed in batch order whatever the number of workers (the reference draws inside its
    forked DataLoader workers, each with its own copy of the state: there is no single reference stream to match)."""

    def __init__(self, config, source, neighborhood_limits=(), depth=2, random_grid_orient=True, device=None, seed=None,
                 workers=None, for_training=True):
        self.config = config
        self.limits = neighborhood_limits
        self.rgo = random_grid_orient
        self.for_training = bool(for_training)       # False: forward-only consumers (no tables / grids for a backward)
        self.source = iter(source)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        import os
        import numpy as np
        self.workers = max(1, int(workers if workers is not None else os.environ.get("WEASAL_PREFETCH_WORKERS", "2")))
        self.depth = max(1, depth)
        prio = -1 if os.environ.get("WEASAL_PREFETCH_PRIORITY", "0") != "0" else 0
        self.streams = [torch.cuda.Stream(device=self.device, priority=prio) for _ in range(self.workers)]
        self.stream = self.streams[0]
        self.seed = seed
        self.rngs = [None if seed is None else np.random.RandomState(seed + w) for w in range(self.workers)]
        self.error = None
        self._stop = False
        self._src_lock = threading.Lock()          # source iteration and sequence numbers
        self._cv = threading.Condition()           # finished batches, keyed by sequence number
        self._done = {}
        self._next_in = 0                          # next sequence number to hand to a worker
        self._next_out = 0                         # next sequence number the consumer takes
        self._exhausted = False
        self.threads = [threading.Thread(target=self._run, args=(w,), name="weasal-pyramid-prefetch-%d" % w, daemon=True)
                        for w in range(self.workers)]
        self.thread = self.threads[0]
        for t in self.threads:
            t.start()

    def _encoder_len(self):
        """blocks up to the first upsampling / global block: where the pyramid schedule stops (datasets/common.py:566-568)"""
        arch = self.config.architecture
        for i, b in enumerate(arch):
            if 'global' in b or 'upsample' in b:
                return i + 1
        return len(arch)

    def _take(self):
        """-> (sequence number, item) or None when the source is exhausted; waits while the pipeline is `depth` ahead"""
        with self._cv:
            while not self._stop and self._next_in - self._next_out >= self.depth + self.workers - 1:
                self._cv.wait(0.05)
        with self._src_lock:
            if self._stop or self._exhausted:
                return None
            try:
                item = next(self.source)
            except StopIteration:
                self._exhausted = True
                return None
            seq = self._next_in
            self._next_in += 1
            rng = None
            if self.rgo and self.seed is None:
                # the reference's stream: the global np.random, consumed in SOURCE order whatever the workers do -- the
                # draws of this batch (theta, phi, alpha per subsampled level, datasets/common.py:99-109) are taken here,
                # under the source lock, and replayed by the worker
                import numpy as np
                nb = len(item[3])
                # one (theta, phi, alpha) triple per sphere for every block that subsamples (pyramid.segmentation_inputs
                # calls batch_grid_subsampling exactly there), counted from the architecture itself
                pools = sum(1 for b in self.config.architecture[:self._encoder_len()] if 'pool' in b or 'strided' in b)
                rng = _Replay(np.random.rand(3 * nb * pools))
            return seq, item, rng

    def _run(self, w):
        try:
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.streams[w]):
                while not self._stop:
                    got = self._take()
                    if got is None:
                        break
                    seq, (points, features, labels, lengths), rng = got
                    batch = pyramid.build_batch(self.config, points, features, labels, lengths, self.limits, self.rgo,
                                                rng=rng if rng is not None else self.rngs[w], for_training=self.for_training)
                    with self._cv:
                        self._done[seq] = batch
                        self._cv.notify_all()
        except BaseException as e:   # surfaced to the consumer
            self.error = e
        finally:
            try:
                self.streams[w].synchronize()           # nothing of this thread's scratch is in flight any more
                from . import ops
                ops.release_thread_workspaces(self.device)   # the thread owns its geometry workspaces: freed with it
            except Exception:
                pass
            with self._cv:
                self._cv.notify_all()

    def __iter__(self):
        return self

    def __next__(self):
        with self._cv:
            while True:
                if self._next_out in self._done:
                    batch = self._done.pop(self._next_out)
                    self._next_out += 1
                    self._cv.notify_all()
                    return batch
                if self.error is not None:
                    raise self.error
                alive = any(t.is_alive() for t in self.threads)
                if (self._exhausted and self._next_out >= self._next_in) or (not alive and self._next_out not in self._done):
                    if self.error is not None:
                        raise self.error
                    raise StopIteration
                self._cv.wait(0.05)

    def close(self):
        self._stop = True
        with self._cv:
            self._done.clear()
            self._cv.notify_all()
        for t in self.threads:
            t.join(timeout=5)
