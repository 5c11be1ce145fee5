"""Drop-in for the reference's ``cpp_wrappers.cpp_neighbors.radius_neighbors`` module.

``batch_query(queries, supports, q_batches, s_batches, *, radius=0.1) -> int32 [Nq, max_count]``
with the argument handling of the CPython glue (cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238):
four positional array-likes (coerced to C-contiguous float32 / int32 like PyArray_FROM_OTF,
:83-86), ``radius`` keyword-only and narrowed to float32 (format "OOOO|$f", :75), shape checks with
the reference's messages (:127-171), every failure a RuntimeError, an empty result
RuntimeError("Error") (:201-205).  The computation runs on the current HIP device
(weasal_amd/csrc/neighbors.hip); arrays go host -> HBM -> host, so use the device-tensor form
``weasal_amd.ops.radius_neighbors`` / ``weasal_amd.pyramid`` inside a training step.

HIP cannot be used in a forked child after the parent touched the GPU (the situation of the reference's
fork-started DataLoader workers): the call then raises RuntimeError with the remedy (weasal_amd/cpp_wrappers/
_device.py) -- use workers started with the "spawn" method, num_workers=0, or the device pyramid.
"""
import numpy as np

from weasal_amd.cpp_wrappers import _device


def _as(obj, dtype, what):
    try:
        return np.ascontiguousarray(np.asarray(obj), dtype=dtype)
    except Exception:
        raise RuntimeError("Error converting %s" % what)


def batch_query(queries, supports, q_batches, s_batches, *, radius=0.1):
    import torch
    from weasal_amd import ops
    q = _as(queries, np.float32, "query points to numpy arrays of type float32")
    s = _as(supports, np.float32, "support points to numpy arrays of type float32")
    qb = _as(q_batches, np.int32, "query batches to numpy arrays of type int32")
    sb = _as(s_batches, np.int32, "support batches to numpy arrays of type int32")
    if q.ndim != 2 or q.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : query.shape is not (N, 3)")
    if s.ndim != 2 or s.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : support.shape is not (N, 3)")
    if qb.ndim > 1:
        raise RuntimeError("Wrong dimensions : queries_batches.shape is not (B,) ")
    if sb.ndim > 1:
        raise RuntimeError("Wrong dimensions : supports_batches.shape is not (B,) ")
    qb, sb = qb.reshape(-1), sb.reshape(-1)
    if qb.shape[0] != sb.shape[0]:
        raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
    dev = _device.current_device()
    try:
        out = ops.radius_neighbors(torch.from_numpy(q).to(dev), torch.from_numpy(s).to(dev), qb, sb,
                                   float(np.float32(radius)), dtype=torch.int32)
    except RuntimeError as e:
        raise RuntimeError("Error" if "status 4" in str(e) else str(e))
    return out.cpu().numpy()
