"""Device selection of the numpy facades, and the guard for the one situation they cannot serve.

The reference calls ``cpp_neighbors.batch_query`` / ``cpp_subsampling.subsample_batch`` only from DataLoader
workers (datasets/common.py:56-74,126-175,185-196; ``DataLoader(num_workers=config.input_threads)`` in
train_DALES_PseudoLabel.py:291-296) that Linux starts by ``fork`` -- after the trainer has moved the network to the
GPU.  A forked child inherits the parent's initialised HIP runtime and cannot use or re-initialise it.  The facades
detect exactly that (an ``os.register_at_fork`` hook records whether the runtime was live in the parent at the
moment of the fork) and raise the RuntimeError the reference's modules raise for every failure, with the remedy:
start the workers with ``multiprocessing_context="spawn"`` (a fresh child initialises its own runtime; tested by
tests/test_geometry_gpu.py::test_facades_from_a_spawn_worker), use ``num_workers=0``, or -- the intended GPU
pipeline -- build the pyramid on the device with ``weasal_amd.pyramid`` / ``weasal_amd.prefetch``.
"""
import os

_forked_from_gpu_parent = False


def _after_fork_in_child():
    global _forked_from_gpu_parent
    try:
        import torch
        if torch.cuda.is_initialized():
            _forked_from_gpu_parent = True
    except Exception:
        pass


os.register_at_fork(after_in_child=_after_fork_in_child)

FORK_MESSAGE = ("Error: the HIP runtime was initialised in the parent process before this worker was forked; a forked "
                "child cannot use the GPU. Start the DataLoader workers with multiprocessing_context='spawn', use "
                "num_workers=0, or build the input pyramid on the device (weasal_amd.pyramid / weasal_amd.prefetch)")


def current_device():
    """torch.device of the current HIP device, or the reference-style RuntimeError in a forked GPU child"""
    if _forked_from_gpu_parent:
        raise RuntimeError(FORK_MESSAGE)
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("Error: no HIP device is visible to this process (the facades have no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())
