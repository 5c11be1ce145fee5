"""oracle/geom.py -- TEST INFRASTRUCTURE ONLY (see oracle/ws_oracle.cpp header).

ctypes bindings for the two CPU checkers:

* ``port``  = oracle/libws_oracle.so, our own CPU restatement (ws_oracle.cpp);
* ``ref``   = oracle/_ref/libws_ref.so, the reference's unmodified C++ core
              (built by oracle/Makefile from /root/reference, prebuilt on the GPU box).

Both expose the call shapes of the reference's CPython modules
(cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238, cpp_wrappers/cpp_subsampling/wrapper.cpp:62-333)
on numpy arrays.  Nothing under weasal_amd/ imports this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = os.path.join(_HERE, "libws_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libws_ref.so")

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_u64p = C.POINTER(C.c_uint64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def have_ref():
    return os.path.exists(_REF)


def have_port():
    return os.path.exists(_PORT)


_libs = {}


def _lib(kind):
    if kind not in _libs:
        path = {"port": _PORT, "ref": _REF}[kind]
        if not os.path.exists(path):
            raise FileNotFoundError(path + " missing: run `make -C oracle`")
        _libs[kind] = C.CDLL(path)
    return _libs[kind]


def _take(ptr, n, dtype, free):
    """copy n items out of a malloc'd buffer and free it"""
    if n == 0 or not ptr:
        return np.zeros((0,), dtype=dtype)
    arr = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
    free(ptr)
    return arr


def batch_query(queries, supports, q_batches, s_batches, radius, kind="port"):
    """-> int32 [Nq, max_count]; RuntimeError("Error") on an empty result."""
    q, s = _f32(queries), _f32(supports)
    qb, sb = _i32(q_batches), _i32(s_batches)
    lib = _lib(kind)
    fn = lib.orc_radius_neighbors if kind == "port" else lib.ref_batch_query
    free = lib.orc_free if kind == "port" else lib.ref_free
    free.argtypes = [C.c_void_p]
    out = _ip()
    mc = C.c_int(0)
    fn.argtypes = [_fp, C.c_int, _fp, C.c_int, _ip, _ip, C.c_int, C.c_float, C.POINTER(_ip), _ip]
    fn.restype = C.c_int
    rc = fn(q.ctypes.data_as(_fp), q.shape[0], s.ctypes.data_as(_fp), s.shape[0],
            qb.ctypes.data_as(_ip), sb.ctypes.data_as(_ip), qb.shape[0], np.float32(radius),
            C.byref(out), C.byref(mc))
    if rc != 0:
        raise RuntimeError("Error")
    n = q.shape[0] * mc.value
    return _take(out, n, np.int32, free).reshape(q.shape[0], mc.value)


def subsample_batch(points, batches, features=None, classes=None, sampleDl=0.1, max_p=0,
                    kind="port", with_keys=False):
    """-> (points, lens[, features][, classes]) like the reference's subsample_batch.
    with_keys (port only): additionally returns (cell_keys uint64 [M], counts int32 [M])."""
    p = _f32(points)
    lens = _i32(batches)
    n, nb = p.shape[0], lens.shape[0]
    f = _f32(features) if features is not None else None
    c = _i32(classes) if classes is not None else None
    fd = f.shape[1] if f is not None else 0
    ld = (c.shape[1] if c.ndim == 2 else 1) if c is not None else 0
    lib = _lib(kind)
    op, of, oc = _fp(), _fp(), _ip()
    olens = np.zeros(nb, dtype=np.int32)
    m = C.c_int(0)
    fptr = f.ctypes.data_as(_fp) if f is not None else None
    cptr = c.ctypes.data_as(_ip) if c is not None else None
    keys = cnts = None
    if kind == "port":
        ok, on = _u64p(), _ip()
        fn = lib.orc_grid_subsample_batch
        fn.restype = C.c_int
        fn.argtypes = [_fp, C.c_int, _ip, C.c_int, _fp, C.c_int, _ip, C.c_int, C.c_float, C.c_int,
                       C.POINTER(_fp), _ip, C.POINTER(_fp), C.POINTER(_ip), C.POINTER(_u64p),
                       C.POINTER(_ip), _ip]
        rc = fn(p.ctypes.data_as(_fp), n, lens.ctypes.data_as(_ip), nb, fptr, fd, cptr, ld,
                np.float32(sampleDl), int(max_p), C.byref(op), olens.ctypes.data_as(_ip),
                C.byref(of), C.byref(oc), C.byref(ok), C.byref(on), C.byref(m))
        free = lib.orc_free
    else:
        fn = lib.ref_subsample_batch
        fn.restype = C.c_int
        fn.argtypes = [_fp, C.c_int, _ip, C.c_int, _fp, C.c_int, _ip, C.c_int, C.c_float, C.c_int,
                       C.POINTER(_fp), _ip, C.POINTER(_fp), C.POINTER(_ip), _ip]
        rc = fn(p.ctypes.data_as(_fp), n, lens.ctypes.data_as(_ip), nb, fptr, fd, cptr, ld,
                np.float32(sampleDl), int(max_p), C.byref(op), olens.ctypes.data_as(_ip),
                C.byref(of), C.byref(oc), C.byref(m))
        free = lib.ref_free
    free.argtypes = [C.c_void_p]
    if rc != 0:
        raise RuntimeError("Error")
    M = m.value
    res = [_take(op, 3 * M, np.float32, free).reshape(M, 3), olens]
    if f is not None:
        res.append(_take(of, fd * M, np.float32, free).reshape(M, fd))
    if c is not None:
        res.append(_take(oc, ld * M, np.int32, free).reshape(M, ld))
    if kind == "port":
        keys = _take(ok, M, np.uint64, free)
        cnts = _take(on, M, np.int32, free)
    if with_keys:
        res += [keys, cnts]
    return tuple(res)


def subsample(points, features=None, classes=None, sampleDl=0.1, kind="port"):
    """single-cloud form (reference: cloud_subsampling, wrapper.cpp:338-566)"""
    n = np.asarray(points).shape[0]
    r = subsample_batch(points, [n], features=features, classes=classes, sampleDl=sampleDl, kind=kind)
    out = [r[0]] + list(r[2:])
    return out[0] if len(out) == 1 else tuple(out)
