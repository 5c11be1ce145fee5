"""Drop-in for the reference's ``cpp_wrappers.cpp_subsampling.grid_subsampling`` module.

``subsample(points, *, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0)``
    -> points | (points[, features][, classes])                      (wrapper.cpp:338-566)
``subsample_batch(points, batches, *, features=None, classes=None, sampleDl=0.1,
                  method="barycenters", max_p=0, verbose=0)``
    -> (points, batches[, features][, classes])                      (wrapper.cpp:62-333)
Same coercions (float32 / int32, C-contiguous), shape checks and messages as the CPython glue;
``method`` is validated and otherwise ignored like in the reference (:92-96); classes may be (N,)
or (N, d) and come back as (M, d) (:289-322).  Rows are emitted in the reference's order
(weasal_amd/csrc/subsample.hip).  Runs on the current HIP device; see radius_neighbors.py for the
fork caveat.
"""
import numpy as np

from weasal_amd.cpp_wrappers import _device


def _as(obj, dtype, what):
    try:
        return np.ascontiguousarray(np.asarray(obj), dtype=dtype)
    except Exception:
        raise RuntimeError("Error converting input %s" % what)


def _check_method(method):
    if method not in ("barycenters", "voxelcenters"):
        raise RuntimeError('Error parsing method. Valid method names are "barycenters" and "voxelcenters" ')


def _run(points, batches, features, classes, sampleDl, max_p, batched):
    import torch
    from weasal_amd import ops
    p = _as(points, np.float32, "points to numpy arrays of type float32")
    b = _as(batches, np.int32, "batches to numpy arrays of type int32") if batched else None
    f = _as(features, np.float32, "features to numpy arrays of type float32") if features is not None else None
    c = _as(classes, np.int32, "classes to numpy arrays of type int32") if classes is not None else None
    if p.ndim != 2 or p.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")
    if batched and b.ndim > 1:
        raise RuntimeError("Wrong dimensions : batches.shape is not (B,) ")
    if f is not None and (f.ndim != 2 or f.shape[0] != p.shape[0]):
        raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")
    if c is not None and (c.ndim > 2 or c.shape[0] != p.shape[0]):
        raise RuntimeError("Wrong dimensions : classes.shape is not (N,) or (N, d)")
    lens = b.reshape(-1) if batched else np.array([p.shape[0]], np.int32)
    dev = _device.current_device()
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    try:
        res = ops.grid_subsample(t(p), lens, float(np.float32(sampleDl)), max_p=int(max_p), features=t(f),
                                 labels=t(c if c is None or c.ndim == 2 else c.reshape(-1, 1)))
    except RuntimeError as e:
        raise RuntimeError("Error" if "status 4" in str(e) else str(e))
    out = [res[0].cpu().numpy(), res[1]] + [r.cpu().numpy() for r in res[2:]]
    return out


def subsample_batch(points, batches, *, features=None, classes=None, sampleDl=0.1, method="barycenters",
                    max_p=0, verbose=0):
    _check_method(method)
    return tuple(_run(points, batches, features, classes, sampleDl, max_p, True))


def subsample(points, *, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0):
    _check_method(method)
    out = _run(points, None, features, classes, sampleDl, 0, False)
    out = [out[0]] + out[2:]
    return out[0] if len(out) == 1 else tuple(out)
