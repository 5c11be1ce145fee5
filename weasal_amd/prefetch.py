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


class PyramidPrefetcher:
    """Iterates PyramidBatch objects built ahead of time from `source`, an iterable of
    (points, features, labels, lengths) with device tensors and host lengths."""

    def __init__(self, config, source, neighborhood_limits=(), depth=2, random_grid_orient=True, device=None, seed=None):
        """seed: the background thread draws the grid orientations from its own numpy RandomState(seed); None keeps the
        global np.random stream (what the reference's workers use; then no other thread may draw from it meanwhile)"""
        self.config = config
        self.rng = None if seed is None else __import__("numpy").random.RandomState(seed)
        self.limits = neighborhood_limits
        self.rgo = random_grid_orient
        self.source = iter(source)
        self.queue = queue.Queue(maxsize=max(1, depth))
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.Stream(device=self.device)
        self.error = None
        self._stop = False
        self.thread = threading.Thread(target=self._run, name="weasal-pyramid-prefetch", daemon=True)
        self.thread.start()

    def _run(self):
        try:
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.stream):
                for item in self.source:
                    if self._stop:
                        break
                    points, features, labels, lengths = item
                    batch = pyramid.build_batch(self.config, points, features, labels, lengths, self.limits, self.rgo,
                                                rng=self.rng)
                    while not self._stop:
                        try:
                            self.queue.put(batch, timeout=0.1)
                            break
                        except queue.Full:
                            continue
        except BaseException as e:   # surfaced to the consumer
            self.error = e
        finally:
            try:
                self.stream.synchronize()               # nothing of this thread's scratch is in flight any more
                from . import ops
                ops.release_thread_workspaces(self.device)   # the thread owns its geometry workspaces: freed with it
            except Exception:
                pass
            while not self._stop:
                try:
                    self.queue.put(None, timeout=0.1)
                    break
                except queue.Full:
                    continue

    def __iter__(self):
        return self

    def __next__(self):
        batch = self.queue.get()
        if batch is None:
            if self.error is not None:
                raise self.error
            raise StopIteration
        return batch

    def close(self):
        self._stop = True
        try:
            while True:
                self.queue.get_nowait()
        except queue.Empty:
            pass
        self.thread.join(timeout=5)
