"""Device-tensor operators over the C ABI (include/weasal_hip.h).

PyTorch is used here for device memory, streams and autograd bookkeeping only: every
function below launches hand-written HIP kernels from libweasal_hip.so through ctypes, on
torch's current stream.  CPU tensors are rejected -- there is no CPU path in the product
(the CPU restatement lives in oracle/ and is test infrastructure).
"""
import collections
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream, ptr

# optional launch timer (bench.py): an object with begin(key) -> token / end(token); records HIP
# events on the current stream around selected kernel launches.  None = no overhead.
_timer = None


def set_kernel_timer(timer):
    global _timer
    _timer = timer


def _tbegin(*key):
    return _timer.begin(key) if _timer is not None else None


def _tend(tok):
    if tok is not None:
        _timer.end(tok)


INFLUENCE = {"linear": 0, "constant": 1, "gaussian": 2}
AGGREGATION = {"sum": 0, "closest": 1}


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.WeasalHipError(
                "weasal_amd operators run on the GPU only (got a %s tensor); move the batch to "
                "cuda -- there is no CPU fallback" % t.device)


def _f32c(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------
# transposed neighbour tables, cached per index tensor (one per layer and batch; every block of
# the layer and both pooling helpers reuse it)
# ------------------------------------------------------------------------------------------------
class TransposedTable:
    __slots__ = ("offsets", "pairs", "nq", "h", "ns")

    def __init__(self, inds, ns):
        _need_cuda(inds)
        assert inds.dtype == torch.int64 and inds.dim() == 2 and inds.is_contiguous()
        nq, h = inds.shape
        lib = _lib.lib()
        self.nq, self.h, self.ns = nq, h, ns
        self.offsets = torch.empty(ns + 2, dtype=torch.int32, device=inds.device)
        self.pairs = torch.empty(max(nq * h, 1), dtype=torch.int32, device=inds.device)
        scratch = torch.empty(lib.ws_transpose_scratch_bytes(nq, h, ns), dtype=torch.uint8, device=inds.device)
        check(lib.ws_transpose_build(ptr(inds), nq, h, ns, ptr(self.offsets), ptr(self.pairs), ptr(scratch),
                                     current_stream()))


    @classmethod
    def from_parts(cls, offsets, pairs, nq, h, ns):
        """a table somebody else built (ws_pyramid_build): offsets int32 [ns + 2], pairs int32 [nq * h]"""
        t = cls.__new__(cls)
        t.offsets, t.pairs, t.nq, t.h, t.ns = offsets, pairs, int(nq), int(h), int(ns)
        return t


_tables = collections.OrderedDict()
_TABLES_MAX = 32


def transposed_table(inds, ns):
    """Table for `inds` [nq,h] over `ns` supports, cached per index tensor.  The entry keeps a
    reference to the tensor, so its memory (the cache key) cannot be recycled while cached; the
    cache is a small LRU (a batch needs 3 tables per layer)."""
    key = (inds.data_ptr(), tuple(inds.shape), ns, inds._version)
    hit = _tables.get(key)
    if hit is not None:
        _tables.move_to_end(key)
        return hit[1]
    table = TransposedTable(inds, ns)
    _tables[key] = (inds, table)
    while len(_tables) > _TABLES_MAX:
        _tables.popitem(last=False)
    return table


def clear_table_cache():
    _tables.clear()
    _col0_tables.clear()


_col0_tables = {}     # (inds.data_ptr(), shape, ns) -> table of the first column (closest_pool backward)


def col0_table(inds, ns):
    """transposed table of the FIRST column of `inds` (closest_pool / nearest upsampling backward), cached per index
    tensor like the full tables"""
    key = (inds.data_ptr(), tuple(inds.shape), ns)
    hit = _col0_tables.get(key)
    if hit is not None:
        return hit[1]
    table = TransposedTable(inds[:, :1].contiguous(), ns)
    _col0_tables[key] = (inds, table)
    return table


def install_tables(full, col0):
    """pre-built tables of a prefetched batch (PyramidBatch.activate): `full` = [(inds, ns, table)],
    `col0` = [(inds, ns, table of inds[:, :1])]"""
    for inds, ns, table in full:
        _tables[(inds.data_ptr(), tuple(inds.shape), ns, inds._version)] = (inds, table)
    for inds, ns, table in col0:
        _col0_tables[(inds.data_ptr(), tuple(inds.shape), ns)] = (inds, table)


# ------------------------------------------------------------------------------------------------
# scheduling orders: a spatially coherent permutation per point set (the cell order of the
# neighbour search).  Pure performance hint -- results never depend on it.
# ------------------------------------------------------------------------------------------------
_orders = {}


def register_point_order(points, order):
    _orders[points.data_ptr()] = order


def clear_point_orders():
    _orders.clear()


def set_point_orders(pairs):
    """install the (points, order) pairs of the batch that is about to be trained on
    (PyramidBatch.point_orders); replaces whatever the previous batch registered"""
    _orders.clear()
    for points, order in pairs:
        _orders[points.data_ptr()] = order


# grids of self-query searches (table-free KPConv backward): key = the index matrix the blocks pass in
GRID_BACKWARD = os.environ.get("WEASAL_GRID_BACKWARD", "1") != "0"   # A/B switch (diagnostics, tests)
_grids = {}


class SearchGrid:
    """What ws_kpconv_gather_bwd_x_grid needs of one self-query search: the exported cell grid, the key of the
    last kept neighbour per query, the radius; `overflow` is the kernel's (never expected) capacity flag."""
    __slots__ = ("blob", "nb", "cells", "ns", "key_last", "radius", "overflow", "max_count", "cap")

    def tensors(self):
        return (self.blob, self.key_last, self.overflow)


_pool_orders = {}


def set_pool_orders(triples):
    """[(pooling index matrix [Nq, H], cell order of its queries or None, cell order of its supports or None)]: scheduling hints
    of ops.max_pool for the batch about to be trained on (PyramidBatch.activate); results never depend on them"""
    _pool_orders.clear()
    for inds, oq, osup in triples:
        if isinstance(inds, torch.Tensor) and inds.dim() == 2 and inds.shape[0] > 0:
            _pool_orders[(inds.data_ptr(), tuple(inds.shape))] = (oq, osup)


def set_search_grids(pairs):
    """install the (index matrix, SearchGrid) pairs of the batch about to be trained on (PyramidBatch.activate)"""
    _grids.clear()
    for inds, grid in pairs:
        _grids[(inds.data_ptr(), tuple(inds.shape))] = (inds, grid)


_sorted_rows = {}


def set_sorted_rows(pairs):
    """[(index matrix, search radius)]: matrices whose rows are sorted by distance from the query (the output of the radius
    search), installed by PyramidBatch.activate for the batch about to be trained on.  Where the search radius exceeds the
    reach of a layer's kernel points (the deformable radius of datasets/common.py:500-502) the linear-influence gather kernels
    stop each row at that reach (exact: the skipped influences are zeros; include/weasal_hip.h `rows_sorted`)."""
    _sorted_rows.clear()
    for m, radius in pairs:
        if isinstance(m, torch.Tensor) and m.dim() == 2 and m.shape[0] > 0:
            _sorted_rows[(m.data_ptr(), tuple(m.shape))] = float(radius)


def sorted_rows_radius(inds):
    """search radius of a registered distance-sorted matrix, None for anything else"""
    return _sorted_rows.get((inds.data_ptr(), tuple(inds.shape))) if SORTED_ROW_CUTOFF else None


def rows_cutoff_pays(inds, conv_radius):
    """the rows are sorted AND were searched well beyond the convolution's own radius (kernel points lie within ~0.7 of it,
    their influence ends `extent` = 0.4 of it further out): only then is there anything to skip"""
    r = sorted_rows_radius(inds)
    return r is not None and r > 1.2 * float(conv_radius)


SORTED_ROW_CUTOFF = os.environ.get("WEASAL_ROW_CUTOFF", "1") != "0"      # A/B switch (diagnostics, tests)
GRID_NARROW_MAX = 128      # rows up to this length: the slab form of the grid backward (ws_kpconv_gather_bwd_x_grid); wider: _wide


def _grid_for(inds):
    hit = _grids.get((inds.data_ptr(), tuple(inds.shape))) if GRID_BACKWARD else None
    return hit[1] if hit is not None else None


def _order_for(points):
    o = _orders.get(points.data_ptr())
    if o is not None and o.numel() == points.shape[0] and o.device == points.device:
        return o
    return None


# ------------------------------------------------------------------------------------------------
# KPConv gather (K3 / K4 / K6)
# ------------------------------------------------------------------------------------------------
class _KPConvGather(torch.autograd.Function):
    """wf[q,k,c] = sum_h w(q,h,k) x[inds[q,h],c]   (reference: models/blocks.py:278-367)"""

    @staticmethod
    def forward(ctx, x, deformed_kp, modulations, q_pts, s_pts, inds, kernel_points, extent, influence,
                aggregation, want_min_d2, rows_sorted=False):
        lib = _lib.lib()
        _need_cuda(x, q_pts, s_pts, inds, kernel_points)
        x = x.contiguous()
        nq, h = inds.shape
        ns, ci = x.shape
        k = kernel_points.shape[0]
        bf = x.dtype == torch.bfloat16          # bf16 feature rows (BASELINE config 5): same kernels, 8-byte row pieces
        if not bf and x.dtype != torch.float32:
            raise _lib.WeasalHipError("kpconv_gather: features must be float32 or bfloat16 (got %s)" % x.dtype)
        wf = torch.empty((nq, k, ci), dtype=x.dtype, device=x.device)
        min_d2 = torch.empty((nq, k), dtype=torch.float32, device=x.device) if want_min_d2 else None
        dkp = deformed_kp.float().contiguous() if deformed_kp is not None else None
        mod = modulations.float().contiguous() if modulations is not None else None
        tok = _tbegin("kpconv_gather_fwd", nq, h, ci)
        if rows_sorted:
            check(lib.ws_kpconv_gather_fwd_ex(ptr(q_pts), nq, ptr(s_pts), ns, ptr(inds), h, ptr(x), ci, ptr(kernel_points), k, ptr(dkp),
                                              ptr(mod), float(extent), influence, aggregation, ptr(_order_for(q_pts)), ptr(wf),
                                              ptr(min_d2), 1 if bf else 0, 1, current_stream()))
        else:
            fwd = lib.ws_kpconv_gather_fwd_bf16 if bf else lib.ws_kpconv_gather_fwd
            check(fwd(ptr(q_pts), nq, ptr(s_pts), ns, ptr(inds), h, ptr(x), ci,
                                           ptr(kernel_points), k, ptr(dkp), ptr(mod), float(extent),
                                           influence, aggregation, ptr(_order_for(q_pts)), ptr(wf), ptr(min_d2),
                                           current_stream()))
        _tend(tok)
        ctx.save_for_backward(x, dkp, mod, q_pts, s_pts, inds, kernel_points)
        ctx.cfg = (float(extent), influence, aggregation)
        return wf, min_d2

    @staticmethod
    def backward(ctx, dwf, d_min_d2):
        lib = _lib.lib()
        x, dkp, mod, q_pts, s_pts, inds, kernel_points = ctx.saved_tensors
        extent, influence, aggregation = ctx.cfg
        nq, h = inds.shape
        ns, ci = x.shape
        k = kernel_points.shape[0]
        bf = x.dtype == torch.bfloat16
        dwf = dwf.contiguous() if dwf.dtype == x.dtype else dwf.to(x.dtype).contiguous()
        f_grid = lib.ws_kpconv_gather_bwd_x_grid_bf16 if bf else lib.ws_kpconv_gather_bwd_x_grid
        f_tab = lib.ws_kpconv_gather_bwd_x_bf16 if bf else lib.ws_kpconv_gather_bwd_x
        f_geom = lib.ws_kpconv_gather_bwd_geom_bf16 if bf else lib.ws_kpconv_gather_bwd_geom
        dx = d_dkp = d_mod = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            grid = _grid_for(inds) if (nq == ns and q_pts.data_ptr() == s_pts.data_ptr()) else None
            tok = _tbegin("kpconv_gather_bwd_x", nq, h, ci)
            wide = grid is not None and grid.max_count > GRID_NARROW_MAX
            if wide and not (dkp is None and mod is None and influence == 0 and aggregation == 0):
                grid = None          # the queue form of the grid backward covers linear / sum only: transposed table
            if grid is not None and grid.ns == ns and wide:
                check(lib.ws_kpconv_gather_bwd_x_grid_wide(ptr(s_pts), ns, ptr(grid.blob), grid.nb, grid.cells, ptr(grid.key_last),
                                                           grid.radius, ptr(dwf), ci, ptr(kernel_points), k, None, None, extent,
                                                           ptr(_order_for(s_pts)), ptr(inds), h, ptr(dx), 1 if bf else 0,
                                                           current_stream()))
            elif grid is not None and grid.ns == ns:
                # self-query layer: incoming pairs re-derived from the search grid, no transposed table
                check(f_grid(ptr(s_pts), ns, ptr(grid.blob), grid.nb, grid.cells,
                                                      ptr(grid.key_last), grid.radius, ptr(dwf), ci, ptr(kernel_points), k,
                                                      ptr(dkp), ptr(mod), extent, influence, aggregation,
                                                      ptr(_order_for(s_pts)), ptr(dx), ptr(grid.overflow),
                                                      current_stream()))
            else:
                table = transposed_table(inds, ns)
                check(f_tab(ptr(q_pts), nq, ptr(s_pts), ns, ptr(inds), h, ptr(table.offsets),
                                                 ptr(table.pairs), ptr(dwf), ci, ptr(kernel_points), k, ptr(dkp),
                                                 ptr(mod), extent, influence, aggregation, ptr(_order_for(s_pts)),
                                                 ptr(dx), current_stream()))
            _tend(tok)
        if dkp is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_dkp = torch.empty_like(dkp)
            d_mod = torch.empty_like(mod) if mod is not None else None
            dmin = d_min_d2.contiguous() if d_min_d2 is not None else None
            check(f_geom(ptr(q_pts), nq, ptr(s_pts), ns, ptr(inds), h, ptr(x), ci, ptr(dwf),
                                                ptr(kernel_points), k, ptr(dkp), ptr(mod), ptr(dmin), extent,
                                                influence, aggregation, ptr(d_dkp), ptr(d_mod), current_stream()))
        return dx, d_dkp, d_mod, None, None, None, None, None, None, None, None, None


def kpconv_gather(x, q_pts, s_pts, inds, kernel_points, extent, influence="linear", aggregation="sum",
                  deformed_kp=None, modulations=None, want_min_d2=False, rows_sorted=False):
    """Fused neighbour gather + kernel-point influence + aggregate -> wf [nq, K, ci] (and min_d2).
    rows_sorted: the rows of `inds` are sorted by distance from their query (see set_sorted_rows)."""
    q_pts = _f32c(q_pts)
    s_pts = _f32c(s_pts)
    kernel_points = _f32c(kernel_points)
    inds = inds.contiguous()
    if inds.dtype != torch.int64:
        inds = inds.to(torch.int64)
    return _KPConvGather.apply(x, deformed_kp, modulations, q_pts, s_pts, inds, kernel_points, extent,
                               INFLUENCE[influence], AGGREGATION[aggregation], want_min_d2, bool(rows_sorted))


_KPCONV_GATHER_SELF = kpconv_gather      # (oracle.kpconv_ref.cpu_reference_mode swaps ops.kpconv_gather: then no fast paths)


# ------------------------------------------------------------------------------------------------
# deformable fast path (BASELINE config 5): linear influence, sum aggregation, packed kernel points
# ------------------------------------------------------------------------------------------------
class _DeformPrepare(torch.autograd.Function):
    """offset features [N, 3K | 4K] -> (kp4 [N,K,4], deformed_kp [N,K,3], modulations [N,K] | None)
    (models/blocks.py:250-267, 287-288; ws_kpconv_deform_prepare)"""

    @staticmethod
    def forward(ctx, off, kernel_points, extent, modulated):
        lib = _lib.lib()
        _need_cuda(off, kernel_points)
        off = off.float().contiguous()
        n, od = off.shape
        k = kernel_points.shape[0]
        kp4 = torch.empty((n, k, 4), dtype=torch.float32, device=off.device)
        dkp = torch.empty((n, k, 3), dtype=torch.float32, device=off.device)
        mod = torch.empty((n, k), dtype=torch.float32, device=off.device) if modulated else None
        rmax = torch.empty((1,), dtype=torch.float32, device=off.device)
        check(lib.ws_kpconv_deform_prepare(ptr(off), n, od, ptr(kernel_points), k, float(extent), 1 if modulated else 0,
                                           ptr(dkp), ptr(mod), ptr(kp4), ptr(rmax), current_stream()))
        ctx.save_for_backward(kp4)
        ctx.cfg = (n, od, k, float(extent), bool(modulated))
        ctx.mark_non_differentiable(rmax, *([mod] if mod is not None else []))
        return kp4, dkp, mod, rmax

    @staticmethod
    def backward(ctx, d_kp4, d_dkp, d_mod, d_rmax):
        lib = _lib.lib()
        kp4, = ctx.saved_tensors
        n, od, k, extent, modulated = ctx.cfg
        d_off = torch.empty((n, od), dtype=torch.float32, device=kp4.device)
        if d_kp4 is None and d_dkp is None:
            return d_off.zero_(), None, None, None
        d_kp4 = d_kp4.float().contiguous() if d_kp4 is not None else None
        d_dkp = d_dkp.float().contiguous() if d_dkp is not None else None
        check(lib.ws_kpconv_deform_prepare_bwd(ptr(d_kp4), ptr(d_dkp), ptr(kp4), n, od, k, extent, 1 if modulated else 0,
                                               ptr(d_off), current_stream()))
        return d_off, None, None, None


def deform_prepare(offset_features, kernel_points, extent, modulated):
    """-> (kp4, deformed_kp, modulations or None, kp_rmax); `modulations` is a plain copy for the module attribute (its
    gradient travels through kp4); kp_rmax [1] = max |deformed kernel point| (device scalar for the grid backward)"""
    return _DeformPrepare.apply(offset_features, _f32c(kernel_points), float(extent), bool(modulated))


def deform_fast_path_ok(x, k, influence, aggregation):
    """the packed-kernel-point kernels cover: K = 15, linear influence, sum aggregation, GPU rows of a multiple of 16 channels"""
    return (x.is_cuda and k == 15 and influence == "linear" and aggregation == "sum" and x.dim() == 2 and x.shape[1] % 16 == 0
            and x.dtype in (torch.float32, torch.bfloat16) and kpconv_gather is _KPCONV_GATHER_SELF)


class _KPConvGatherDef(torch.autograd.Function):
    """wf[q,k,c] = mod[q,k] sum_h w(q,h,k) x[inds[q,h],c] with per-query kernel points (kp4), and min_d2 [nq,K]
    (models/blocks.py:278-367 for deformable = True, KP_influence = 'linear', aggregation_mode = 'sum')"""

    @staticmethod
    def forward(ctx, x, kp4, q_pts, s_pts, inds, extent, rmax, rows_sorted):
        lib = _lib.lib()
        _need_cuda(x, kp4, q_pts, s_pts, inds)
        x = x.contiguous()
        kp4 = kp4.contiguous()
        nq, h = inds.shape
        ns, ci = x.shape
        k = kp4.shape[1]
        bf = x.dtype == torch.bfloat16
        wf = torch.empty((nq, k, ci), dtype=x.dtype, device=x.device)
        min_d2 = torch.empty((nq, k), dtype=torch.float32, device=x.device)
        tok = _tbegin("kpconv_gather_fwd_def", nq, h, ci)
        check(lib.ws_kpconv_gather_fwd_def(ptr(q_pts), nq, ptr(s_pts), ns, ptr(inds), h, ptr(x), ci, ptr(kp4), k, float(extent),
                                           ptr(_order_for(q_pts)), ptr(wf), ptr(min_d2), 1 if bf else 0, 1 if rows_sorted else 0,
                                           current_stream()))
        _tend(tok)
        ctx.save_for_backward(x, kp4, q_pts, s_pts, inds, rmax)
        ctx.extent = float(extent)
        ctx.rows_sorted = bool(rows_sorted)
        return wf, min_d2

    @staticmethod
    def backward(ctx, dwf, d_min_d2):
        lib = _lib.lib()
        x, kp4, q_pts, s_pts, inds, rmax = ctx.saved_tensors
        extent = ctx.extent
        rs = 1 if ctx.rows_sorted else 0
        nq, h = inds.shape
        ns, ci = x.shape
        k = kp4.shape[1]
        bf = 1 if x.dtype == torch.bfloat16 else 0
        dwf = dwf.contiguous() if dwf.dtype == x.dtype else dwf.to(x.dtype).contiguous()
        dx = d_kp4 = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            grid = _grid_for(inds) if (nq == ns and q_pts.data_ptr() == s_pts.data_ptr()) else None
            tok = _tbegin("kpconv_gather_bwd_x", nq, h, ci)
            if grid is not None and grid.ns == ns:
                check(lib.ws_kpconv_gather_bwd_x_grid_wide(ptr(s_pts), ns, ptr(grid.blob), grid.nb, grid.cells, ptr(grid.key_last),
                                                           grid.radius, ptr(dwf), ci, None, k, ptr(kp4), ptr(rmax), extent,
                                                           ptr(_order_for(s_pts)), ptr(inds), h, ptr(dx), bf, current_stream()))
            else:
                table = transposed_table(inds, ns)
                check(lib.ws_kpconv_gather_bwd_x_def(ptr(q_pts), nq, ptr(s_pts), ns, h, ptr(table.offsets), ptr(table.pairs),
                                                     ptr(dwf), ci, ptr(kp4), k, extent, ptr(_order_for(s_pts)), ptr(dx), bf,
                                                     current_stream()))
            _tend(tok)
        if ctx.needs_input_grad[1]:
            d_kp4 = torch.empty_like(kp4)
            dmin = d_min_d2.float().contiguous() if d_min_d2 is not None else None
            tok = _tbegin("kpconv_gather_bwd_geom", nq, h, ci)
            check(lib.ws_kpconv_gather_bwd_geom_def(ptr(q_pts), nq, ptr(s_pts), ns, ptr(inds), h, ptr(x), ci, ptr(dwf), ptr(kp4), k,
                                                    ptr(dmin), extent, ptr(_order_for(q_pts)), ptr(d_kp4), bf, rs, current_stream()))
            _tend(tok)
        return dx, d_kp4, None, None, None, None, None, None


def kpconv_gather_def(x, kp4, q_pts, s_pts, inds, extent, kp_rmax=None, rows_sorted=False):
    """deformable fast path: -> (wf [nq,K,ci], min_d2 [nq,K]); kp_rmax: deform_prepare's device scalar (bounds the grid
    backward's candidates); rows_sorted: see set_sorted_rows"""
    inds = inds.contiguous()
    if inds.dtype != torch.int64:
        inds = inds.to(torch.int64)
    return _KPConvGatherDef.apply(x, kp4, _f32c(q_pts), _f32c(s_pts), inds, float(extent), kp_rmax, bool(rows_sorted))


class _P2PRegularizer(torch.autograd.Function):
    """(fitting, repulsive) of one deformable layer (models/architectures.py:36-51): ws_p2p_regularizer_fwd / _bwd"""

    @staticmethod
    def forward(ctx, deformed_kp, min_d2, extent, repulse_extent):
        lib = _lib.lib()
        _need_cuda(deformed_kp, min_d2)
        dkp = deformed_kp.float().contiguous()
        md = min_d2.float().contiguous()
        n, k = md.shape
        out = torch.empty(2, dtype=torch.float32, device=md.device)
        scratch = torch.empty(max(lib.ws_p2p_regularizer_scratch_bytes(n), 16), dtype=torch.uint8, device=md.device)
        check(lib.ws_p2p_regularizer_fwd(ptr(dkp), None, ptr(md), n, k, float(extent), float(repulse_extent), ptr(out), ptr(scratch),
                                         current_stream()))
        ctx.save_for_backward(dkp, md)
        ctx.cfg = (float(extent), float(repulse_extent))
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.lib()
        dkp, md = ctx.saved_tensors
        n, k = md.shape
        extent, rep = ctx.cfg
        g = g.float().contiguous()
        d_md = torch.empty_like(md)
        d_dkp = torch.empty_like(dkp)
        check(lib.ws_p2p_regularizer_bwd(ptr(dkp), None, ptr(md), n, k, extent, rep, ptr(g), ptr(d_md), ptr(d_dkp), current_stream()))
        return d_dkp, d_md, None, None


def p2p_regularizer(deformed_kp, min_d2, extent, repulse_extent):
    """-> tensor [2] = (fitting loss, repulsive loss) of one deformable KPConv layer"""
    return _P2PRegularizer.apply(deformed_kp, min_d2, float(extent), float(repulse_extent))


# ------------------------------------------------------------------------------------------------
# tall-skinny GEMMs (unary MLPs and the kernel contraction) on the f32 MFMA
# ------------------------------------------------------------------------------------------------
FUSED_EPILOGUE = os.environ.get("WEASAL_FUSED_EPILOGUE", "1") != "0"   # A/B switch (diagnostics)
GEMM_MIN_ROWS = int(os.environ.get("WEASAL_GEMM_MIN_ROWS", "0"))        # A/B switch (diagnostics): fewer rows -> torch.matmul (library GEMM); default: never
XTY_MIN_ROWS = 0         # A/B switch: rows below which dW = x^T dy goes to the library GEMM (in the training step the MFMA
                         # reduction is ahead at every level that reaches it: 0.42 ms vs 0.54 ms per step at M = 10 257)


def _gemm_xb(x, b):
    lib = _lib.lib()
    m, k = x.shape
    n = b.shape[1]
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    _xb_launch(lib, x, m, k, b, n, None, None, None, y)
    return y


def _xb_launch(lib, x, m, k, b, n, bias, residual, slope, y):
    """ws_gemm_xb_epilogue, or its split-K form when the product is short and deep (scratch_bytes > 0)"""
    sb = lib.ws_gemm_xb_scratch_bytes(m, k, n)
    act, sl = (0, 0.0) if slope is None else (1, float(slope))
    ldr = residual.stride(0) if residual is not None else 0
    if sb > 0:
        scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
        check(lib.ws_gemm_xb_epilogue_splitk(ptr(x), m, k, x.stride(0), ptr(b), n, ptr(bias), ptr(residual), ldr, act, sl,
                                             ptr(y), n, ptr(scratch), sb, current_stream()))
    else:
        check(lib.ws_gemm_xb_epilogue(ptr(x), m, k, x.stride(0), ptr(b), n, ptr(bias), ptr(residual), ldr, act, sl,
                                      ptr(y), n, current_stream()))


def _gemm_xty(lib, x, dy):
    """x^T @ dy: the LDS-free MFMA reduction (rows split per workgroup: gemm.hip xty_chunk); XTY_MIN_ROWS > 0 is a
    diagnostics switch to the library GEMM for shorter operands"""
    m, k = x.shape
    n = dy.shape[1]
    if m < XTY_MIN_ROWS:
        return torch.matmul(x.t(), dy)
    db = torch.empty((k, n), dtype=torch.float32, device=x.device)
    scratch = torch.empty(max(lib.ws_gemm_xty_scratch_bytes(m, k, n), 16), dtype=torch.uint8, device=x.device)
    check(lib.ws_gemm_xty(ptr(x), m, k, x.stride(0), ptr(dy), n, dy.stride(0), ptr(db), ptr(scratch), current_stream()))
    return db


class _MatmulXB(torch.autograd.Function):
    """y = x @ b with x [M,K] tall and b [K,N] small; dx = dy @ b^T, db = x^T @ dy"""

    @staticmethod
    def forward(ctx, x, b):
        xc = x if (x.stride(1) == 1 and x.stride(0) >= x.shape[1]) else x.contiguous()
        bc = b.contiguous()
        ctx.save_for_backward(xc, bc)
        return _gemm_xb(xc, bc)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.lib()
        x, b = ctx.saved_tensors
        dy = dy if (dy.stride(1) == 1 and dy.stride(0) >= dy.shape[1]) else dy.contiguous()
        dx = db = None
        if ctx.needs_input_grad[0]:
            dx = _gemm_xb(dy, b.t().contiguous())
        if ctx.needs_input_grad[1]:
            db = _gemm_xty(lib, x, dy)
        return dx, db


# ---- bf16 rows (BASELINE config 5): X, the small matrix and Y bf16 in HBM, fp32 accumulate on the bf16 MFMA --------------
def _bf16_ok(x, k):
    return x.dtype == torch.bfloat16 and k % 32 == 0 and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0


def _bt16(b):
    """the small matrix b [K,N] (fp32 master, any strides) as contiguous bf16 [N,K]: one strided copy-cast kernel"""
    out = torch.empty((b.shape[1], b.shape[0]), dtype=torch.bfloat16, device=b.device)
    out.copy_(b.t())
    return out


def _xbt16(lib, x, bt, bias, residual, slope, out_f32):
    """act(x @ bt^T + bias + residual): x [M,K] bf16, bt [N,K] bf16 -> [M,N] bf16 (or f32)"""
    m, k = x.shape
    n = bt.shape[0]
    y = torch.empty((m, n), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    act, sl = (0, 0.0) if slope is None else (1, float(slope))
    check(lib.ws_gemm_xbt_bf16(ptr(x), m, k, x.stride(0), ptr(bt), n, bt.stride(0), ptr(bias), ptr(residual),
                               residual.stride(0) if residual is not None else 0, act, sl, ptr(y), n, 1 if out_f32 else 0,
                               current_stream()))
    return y


def _xty16(lib, x, dz):
    """x^T @ dz -> f32 [K,N] (the master gradient); x, dz bf16 rows"""
    m, k = x.shape
    n = dz.shape[1]
    db = torch.empty((k, n), dtype=torch.float32, device=x.device)
    scratch = torch.empty(max(lib.ws_gemm_xty_scratch_bytes(m, k, n), 16), dtype=torch.uint8, device=x.device)
    check(lib.ws_gemm_xty_bf16(ptr(x), m, k, x.stride(0), ptr(dz), n, dz.stride(0), ptr(db), ptr(scratch), current_stream()))
    return db


class _MatmulEpilogueBF16(torch.autograd.Function):
    """y = act(x @ b + bias + residual) with bf16 rows: x [M,K] bf16, b [K,N] the fp32 master (cast to bf16 here),
    bias f32, residual bf16, y bf16 -- or f32 (out_f32: logits, deformable offsets).  Backward: dz = dy*act'(y) (bf16) and
    the f32 bias gradient in one pass, dx = dz @ b^T on the bf16 MFMA, dW = x^T dz in fp32 (exact products).
    Shapes the bf16 kernels do not take (contraction depth not a multiple of 32: the 9 logits' dx) go through the f32
    kernels on widened operands."""

    @staticmethod
    def forward(ctx, x, b, bias, residual, slope, out_f32):
        lib = _lib.lib()
        m, k = x.shape
        n = b.shape[1]
        xc = x if (x.stride(1) == 1 and x.stride(0) >= k) else x.contiguous()
        rc = None
        if residual is not None:
            rc = residual if residual.dtype == torch.bfloat16 else residual.to(torch.bfloat16)
            rc = rc if (rc.stride(1) == 1 and rc.stride(0) >= n) else rc.contiguous()
        biasc = bias.float().contiguous() if bias is not None else None
        if _bf16_ok(xc, k):
            y = _xbt16(lib, xc, _bt16(b), biasc, rc, slope, out_f32)
        else:
            y = torch.empty((m, n), dtype=torch.float32, device=x.device)
            _xb_launch(lib, xc.float(), m, k, b.float().contiguous(), n, biasc, rc.float() if rc is not None else None, slope, y)
            y = y if out_f32 else y.to(torch.bfloat16)
        ctx.slope = slope
        ctx.has = (bias is not None, residual is not None)
        ctx.save_for_backward(xc, b, y if slope is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.lib()
        x, b, y = ctx.saved_tensors
        m, k = x.shape
        n = b.shape[1]
        dy = _rowmajor(dy)
        want_bias = ctx.has[0] and ctx.needs_input_grad[2]
        if n % 4 == 0:
            dz, dbias = _act_bwd_colsum16(lib, dy, y, ctx.slope, want_bias)          # bf16 rows
        else:                                                                           # the 9 logits: f32 rows
            dz, dbias = _act_bwd_colsum(dy.float(), y.float() if y is not None else None, ctx.slope, want_bias)
        dx = db = dres = None
        if ctx.needs_input_grad[0]:
            if dz.dtype == torch.bfloat16 and _bf16_ok(dz, n):
                dx = _xbt16(lib, dz, b.to(torch.bfloat16).contiguous() if b.is_contiguous() else _bt16(b.t()), None, None, None, False)
            else:
                # (the bf16-rounded weights the forward used, widened)
                dx = _gemm_xb(dz.float().contiguous(), b.t().to(torch.bfloat16).float().contiguous()).to(torch.bfloat16)
        dz16 = dz if dz.dtype == torch.bfloat16 else dz.to(torch.bfloat16)
        if ctx.needs_input_grad[1]:
            db = _xty16(lib, x, dz16)
        if ctx.has[1] and ctx.needs_input_grad[3]:
            dres = dz16
        return dx, db, dbias, dres, None, None


def _act_bwd_colsum16(lib, dy, y, slope, want_colsum):
    """bf16 form of _act_bwd_colsum: dy bf16 or f32, y bf16 (None: identity) -> (dz bf16, f32 column sums or None)"""
    m, n = dy.shape
    f32 = dy.dtype == torch.float32
    if y is None and not want_colsum and not f32:
        return dy, None
    need_dz = y is not None or f32
    dz = torch.empty((m, n), dtype=torch.bfloat16, device=dy.device) if need_dz else None
    colsum = scratch = None
    if want_colsum:
        colsum = torch.empty((n,), dtype=torch.float32, device=dy.device)
        scratch = torch.empty(max(lib.ws_act_bwd_colsum_bf16_scratch_bytes(m, n), 16), dtype=torch.uint8, device=dy.device)
    yb = None
    if y is not None:
        yb = y if y.dtype == torch.bfloat16 else y.to(torch.bfloat16)      # only the sign is used
    check(lib.ws_act_bwd_colsum_bf16(ptr(dy), 1 if f32 else 0, m, n, dy.stride(0), ptr(yb), yb.stride(0) if yb is not None else 0,
                                     0.0 if slope is None else float(slope), ptr(dz), n if dz is not None else 0, ptr(colsum),
                                     ptr(scratch), current_stream()))
    return (dz if need_dz else dy), colsum


def _rowmajor(t):
    return t if (t.stride(1) == 1 and t.stride(0) >= t.shape[1]) else t.contiguous()


def _act_bwd_colsum(dy, y, slope, want_colsum):
    """(dz, column sums of dz or None): LeakyReLU backward from the output y (None: identity) and the
    bias gradient in one pass (ws_act_bwd_colsum)"""
    if y is None and not want_colsum:
        return dy, None
    lib = _lib.lib()
    m, n = dy.shape
    dz = torch.empty((m, n), dtype=torch.float32, device=dy.device) if y is not None else dy
    colsum = scratch = None
    if want_colsum:
        colsum = torch.empty((n,), dtype=torch.float32, device=dy.device)
        scratch = torch.empty(max(lib.ws_act_bwd_colsum_scratch_bytes(m, n), 16), dtype=torch.uint8, device=dy.device)
    check(lib.ws_act_bwd_colsum(ptr(dy), m, n, dy.stride(0), ptr(y), y.stride(0) if y is not None else 0,
                                0.0 if slope is None else float(slope), ptr(dz) if y is not None else None, n,
                                ptr(colsum), ptr(scratch), current_stream()))
    return dz, colsum


class _MatmulEpilogue(torch.autograd.Function):
    """y = act(x @ b + bias + residual), act = LeakyReLU(slope) or identity; the activation
    backward uses the output (sign(y) == sign(pre-activation) for slope > 0)."""

    @staticmethod
    def forward(ctx, x, b, bias, residual, slope, links=None):
        # links = [incoming, outgoing] fused.GateLink or None: see fused.GateLink -- the consumer's dX product applies the
        # producer's LeakyReLU' (and dropout backward), the producer skips its activation-backward pass
        lib = _lib.lib()
        ctx.links = links if links is not None else [None, None]
        xc, bc = _rowmajor(x), b.contiguous()
        rc = _rowmajor(residual) if residual is not None else None
        biasc = bias.contiguous() if bias is not None else None
        m, k = xc.shape
        n = bc.shape[1]
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        _xb_launch(lib, xc, m, k, bc, n, biasc, rc, slope, y)
        ctx.slope = slope
        ctx.has = (bias is not None, residual is not None)
        ctx.save_for_backward(xc, bc, y if slope is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.lib()
        x, b, y = ctx.saved_tensors
        dy = _rowmajor(dy)
        want_bias = ctx.has[0] and ctx.needs_input_grad[2]
        li, lo = ctx.links
        if lo is not None and lo.pregated and ctx.slope is not None:
            dz, dbias = _act_bwd_colsum(dy, None, None, want_bias)            # (the consumer has applied this layer's LeakyReLU')
        else:
            dz, dbias = _act_bwd_colsum(dy, y if ctx.slope is not None else None, ctx.slope, want_bias)
        dx = db = dres = None
        if li is not None:
            # (on the kernels whose epilogue reads the gate rows 16 bytes per lane: the rows-on-lanes MFMA kernel and the
            # streaming form of shallow contractions -- through gemm_xb_kernel's scalar epilogue the 9-logit dX took 247
            # instead of 96 us with the gate on it)
            li.pregated = bool(ctx.needs_input_grad[0]) and (dz.shape[1] % 32 == 0 or dz.shape[1] <= 64) and b.shape[0] % 4 == 0
        if ctx.needs_input_grad[0]:
            if li is not None and li.pregated:
                # dx = dropout_bwd(dz @ b^T) * LeakyReLU'(x): x is the producer's activated (and dropped) output
                bt = b.t().contiguous()
                m, k = dz.shape
                n = bt.shape[1]
                dx = torch.empty((m, n), dtype=torch.float32, device=dz.device)
                sb = max(int(lib.ws_gemm_xb_scratch_bytes(m, k, n)), 0)
                scratch = torch.empty(max(sb, 16), dtype=torch.uint8, device=dz.device)
                p_drop, seed = li.drop if li.drop is not None else (0.0, 0)
                check(lib.ws_gemm_xb_gate_dropout(ptr(dz), m, k, dz.stride(0), ptr(bt), n, ptr(x), x.stride(0), float(li.slope),
                                                  float(p_drop), int(seed), ptr(dx), n, ptr(scratch), sb, current_stream()))
            else:
                dx = _gemm_xb(dz, b.t().contiguous())
        if ctx.needs_input_grad[1]:
            db = _gemm_xty(lib, x, dz)
        if ctx.has[1] and ctx.needs_input_grad[3]:
            dres = dz
        return dx, db, dbias, dres, None, None


def matmul_epilogue(x, b, bias=None, residual=None, slope=None, out_f32=False, links=None):
    """act(x @ b + bias + residual): one MFMA kernel (b is [K,N]; short, deep products of the deep layers split K).
    GEMM_MIN_ROWS > 0 (diagnostics) hands operands with fewer rows to the library GEMM through torch.
    bf16 rows (x.dtype bfloat16) always take the bf16 MFMA kernel; out_f32 keeps its output in f32."""
    _need_cuda(x, b)
    if x.dtype == torch.bfloat16 and x.dim() == 2:
        return _MatmulEpilogueBF16.apply(x, b, bias, residual, slope, bool(out_f32))
    if FUSED_EPILOGUE and x.dim() == 2 and x.shape[0] >= GEMM_MIN_ROWS and x.dtype == torch.float32:
        return _MatmulEpilogue.apply(x, b, bias, residual, slope, links)
    if links is not None and (links[0] is not None or links[1] is not None):
        raise _lib.WeasalHipError("matmul_epilogue: gate links need the float32 kernel path")
    if not FUSED_EPILOGUE:
        y = matmul(x, b)
        if bias is not None:
            y = y + bias
        if residual is not None:
            y = y + residual
        return y if slope is None else torch.nn.functional.leaky_relu(y, slope)
    y = torch.matmul(x, b)
    if bias is not None:
        y = y + bias
    if residual is not None:
        y = y + residual
    return y if slope is None else torch.nn.functional.leaky_relu(y, slope)


def matmul(x, b):
    """x [M,K] @ b [K,N] on the MFMA kernels of this library (GEMM_MIN_ROWS > 0: diagnostics switch to the library GEMM)"""
    _need_cuda(x, b)
    if x.dtype == torch.bfloat16 and x.dim() == 2 and b.dim() == 2:
        return _MatmulEpilogueBF16.apply(x, b, None, None, None, False)
    if x.dim() == 2 and b.dim() == 2 and x.shape[0] >= GEMM_MIN_ROWS and x.dtype == torch.float32:
        return _MatmulXB.apply(x, b)
    return torch.matmul(x, b)


def linear(x, weight):
    """x @ weight^T (nn.Linear without bias; weight is [out, in])"""
    return matmul(x, weight.t())


# ------------------------------------------------------------------------------------------------
# pooling helpers (K7)
# ------------------------------------------------------------------------------------------------
def _by_dtype(lib, name, t):
    """the f32 entry or its bf16-row form"""
    if t.dtype == torch.bfloat16:
        return getattr(lib, name + "_bf16")
    if t.dtype != torch.float32:
        raise _lib.WeasalHipError("%s: feature rows must be float32 or bfloat16 (got %s)" % (name, t.dtype))
    return getattr(lib, name)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inds):
        lib = _lib.lib()
        _need_cuda(x, inds)
        x = x.contiguous()
        ns, c = x.shape
        nq, h = inds.shape
        out = torch.empty((nq, c), dtype=x.dtype, device=x.device)
        arg = torch.empty((nq, c), dtype=torch.int32, device=x.device)
        oq, osup = _pool_orders.get((inds.data_ptr(), tuple(inds.shape)), (None, None))      # scheduling hints (PyramidBatch.activate)
        if oq is not None and oq.numel() != nq:
            oq = None
        if osup is not None and osup.numel() != ns:
            osup = None
        check(_by_dtype(lib, "ws_max_pool_fwd_ordered", x)(ptr(x), ns, c, ptr(inds), nq, h, ptr(out), ptr(arg), ptr(oq), current_stream()))
        ctx.save_for_backward(arg, inds)
        ctx.ns = ns
        ctx.order_s = osup
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.lib()
        arg, inds = ctx.saved_tensors
        nq, c = arg.shape
        h = inds.shape[1]
        table = transposed_table(inds, ctx.ns)
        dx = torch.empty((ctx.ns, c), dtype=dy.dtype, device=dy.device)
        dy = dy.contiguous()
        check(_by_dtype(lib, "ws_max_pool_bwd_ordered", dy)(ptr(dy), ptr(arg), nq, h, c, ptr(table.offsets), ptr(table.pairs), ctx.ns,
                                                            ptr(dx), ptr(ctx.order_s), current_stream()))
        return dx, None


class _ClosestPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inds):
        lib = _lib.lib()
        _need_cuda(x, inds)
        x = x.contiguous()
        ns, c = x.shape
        nq, h = inds.shape
        out = torch.empty((nq, c), dtype=x.dtype, device=x.device)
        check(_by_dtype(lib, "ws_closest_pool_fwd", x)(ptr(x), ns, c, ptr(inds), nq, h, ptr(out), current_stream()))
        # only the first column takes part (blocks.py:92): the backward needs the transposed
        # table of that column alone (lists of ~N_fine/N_coarse entries instead of ~H times that)
        ctx.table = col0_table(inds, ns) if torch.is_grad_enabled() or x.requires_grad else None
        ctx.nq = nq
        ctx.ns = ns
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.lib()
        nq = ctx.nq
        c = dy.shape[1]
        table = ctx.table
        dx = torch.empty((ctx.ns, c), dtype=dy.dtype, device=dy.device)
        dy = dy.contiguous()
        check(_by_dtype(lib, "ws_closest_pool_bwd", dy)(ptr(dy), nq, 1, c, ptr(table.offsets), ptr(table.pairs), ctx.ns, ptr(dx),
                                      current_stream()))
        return dx, None


def _as_index(inds):
    inds = inds.contiguous()
    return inds if inds.dtype == torch.int64 else inds.to(torch.int64)


def max_pool(x, inds):
    """reference: models/blocks.py:95-111"""
    return _MaxPool.apply(x, _as_index(inds))


def closest_pool(x, inds):
    """reference: models/blocks.py:80-92"""
    return _ClosestPool.apply(x, _as_index(inds))


# ------------------------------------------------------------------------------------------------
# native geometry on device tensors (K1 / K2)
# ------------------------------------------------------------------------------------------------
class _Workspaces:
    """one neighbour and one subsample workspace per (device, host thread): grow-only device scratch whose kernels
    run on the calling thread's stream -- the prefetch thread and the training thread must not share one"""

    def __init__(self):
        self.nb = {}
        self.sub = {}

    @staticmethod
    def _key(device):
        import threading
        return (torch.device(device).index or 0, threading.get_ident())

    def neighbors(self, device):
        key = self._key(device)
        if key not in self.nb:
            import ctypes as C
            h = C.c_void_p()
            check(_lib.lib().ws_neighbors_ws_create(C.byref(h)))
            self.nb[key] = h
        return self.nb[key]

    def subsample(self, device):
        key = self._key(device)
        if key not in self.sub:
            import ctypes as C
            h = C.c_void_p()
            check(_lib.lib().ws_subsample_ws_create(C.byref(h)))
            self.sub[key] = h
        return self.sub[key]


    def release_thread(self, device=None):
        """destroy the calling thread's workspaces (a prefetch thread calls this when it ends: thread identifiers are
        recycled, and a later thread must not inherit scratch that is still bound to another stream's work)"""
        import threading
        ident = threading.get_ident()
        lib = _lib.lib()
        for table, destroy in ((self.nb, lib.ws_neighbors_ws_destroy), (self.sub, lib.ws_subsample_ws_destroy)):
            for key in [k for k in table if k[1] == ident and (device is None or k[0] == (torch.device(device).index or 0))]:
                destroy(table.pop(key))


_ws = _Workspaces()


def release_thread_workspaces(device=None):
    _ws.release_thread(device)


class _DevView:
    """zero-copy int32 view of workspace memory (consumed through __cuda_array_interface__)"""

    def __init__(self, addr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (int(addr), False),
                                         "version": 2, "strides": None}


def _host_lens(lens):
    if isinstance(lens, torch.Tensor):
        lens = lens.detach().cpu().numpy()
    return np.ascontiguousarray(lens, dtype=np.int32)


def radius_neighbors(queries, supports, q_lens, s_lens, radius, limit=None, dtype=torch.int64,
                     return_counts=False, return_order=False):
    """Batched radius search on device tensors (reference: cpp_neighbors.batch_query,
    neighbors.cpp:211-332).  q_lens/s_lens: host sequences.  Returns [Nq, min(max_count, limit)]
    of `dtype` (int32 like the reference module, or int64 like datasets/common.py:551 makes it)."""
    import ctypes as C
    lib = _lib.lib()
    _need_cuda(queries, supports)
    q = _f32c(queries)
    s = _f32c(supports)
    ql, sl = _host_lens(q_lens), _host_lens(s_lens)
    if ql.shape[0] != sl.shape[0]:
        raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
    ws = _ws.neighbors(q.device)
    mc = C.c_int32(0)
    hq, hs = C.c_void_p(ql.ctypes.data), C.c_void_p(sl.ctypes.data)
    if dtype not in (torch.int32, torch.int64):
        raise ValueError("dtype must be torch.int32 or torch.int64")
    with torch.cuda.device(q.device):
        if limit is not None:
            # one query pass: write `limit` columns and count together; trim if the true width is smaller
            width = max(1, int(limit))
            out = torch.empty((q.shape[0], width), dtype=dtype, device=q.device)
            o32, o64 = (ptr(out), None) if dtype == torch.int32 else (None, ptr(out))
            check(lib.ws_radius_neighbors_search(ws, ptr(q), q.shape[0], ptr(s), s.shape[0], hq, hs, ql.shape[0],
                                                 float(np.float32(radius)), width, o32, o64, C.byref(mc),
                                                 current_stream()))
            if mc.value < width:
                out = out[:, :mc.value].contiguous()
        else:
            check(lib.ws_radius_neighbors_plan(ws, ptr(q), q.shape[0], ptr(s), s.shape[0], hq, hs, ql.shape[0],
                                               float(np.float32(radius)), C.byref(mc), current_stream()))
            width = mc.value
            out = torch.empty((q.shape[0], width), dtype=dtype, device=q.device)
            o32, o64 = (ptr(out), None) if dtype == torch.int32 else (None, ptr(out))
            check(lib.ws_radius_neighbors_fill(ws, width, o32, o64, current_stream()))
        res = [out]
        if return_counts:
            res.append(torch.as_tensor(_DevView(lib.ws_radius_neighbors_counts(ws), q.shape[0]),
                                       device=q.device).clone())
        if return_order:
            order = torch.empty(s.shape[0], dtype=torch.int32, device=q.device)
            check(lib.ws_radius_neighbors_order(ws, ptr(order), current_stream()))
            res.append(order)
    return res[0] if len(res) == 1 else tuple(res)


def widen_async_slabs(max_count):
    """a row overflowed the width-sized slab of the asynchronous search (ws_radius_neighbors_async_cap: 5/4 of the limit): this
    data has longer tails than that -- from now on the process uses the full 1024-entry slab for wide rows, so that the
    repeat of a search (synchronous, two passes) stays a one-off instead of a per-batch cost"""
    import ctypes as C
    if max_count <= 1024:
        C.c_int.in_dll(_lib.lib(), "ws_nb_wide_caps").value = 0


class DeferredSearches:
    """Batch of radius searches issued without host synchronisation (ws_radius_neighbors_search_async).
    ``add`` returns the index matrix immediately ([Nq, limit], rows padded with Ns); ``finish`` reads
    all true widths back in ONE synchronisation and returns the final matrices: trimmed to the true
    width where that is smaller than the limit (what the reference's crop yields), recomputed with the
    synchronous search in the rare case of a row with more than 128 neighbours."""

    def __init__(self, device, capacity=64):
        self.device = device
        self.slots = torch.zeros(capacity, dtype=torch.int32, device=device)
        self.calls = []

    def add(self, queries, supports, q_lens, s_lens, radius, limit, want_order=False, want_grid=False, reuse_grid=False):
        """-> index matrix; with want_order / want_grid: (matrix, cell order or None, SearchGrid or None).
        The grid is only meaningful for a self-query (queries is supports) and while `counts()` of this call
        stays <= 128 (checked by the caller after finish())."""
        import ctypes as C
        lib = _lib.lib()
        q, s = _f32c(queries), _f32c(supports)
        ql, sl = _host_lens(q_lens), _host_lens(s_lens)
        if ql.shape[0] != sl.shape[0]:
            raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
        if len(self.calls) >= self.slots.shape[0]:
            raise RuntimeError("DeferredSearches capacity exceeded")
        ws = _ws.neighbors(q.device)
        width = max(1, int(limit))
        out = torch.empty((q.shape[0], width), dtype=torch.int64, device=q.device)
        slot = len(self.calls)
        grid = None
        with torch.cuda.device(q.device):
            if reuse_grid:      # same supports (tensor, lengths, radius, contents) as the previous search of this workspace
                check(lib.ws_radius_neighbors_reuse_grid(ws, 1))
            if want_grid:
                grid = SearchGrid()
                grid.max_count = 0            # true maximum row length of the search: set by the caller after finish()
                grid.cap = int(lib.ws_radius_neighbors_async_cap(width))      # sort slab of the asynchronous pass: key_last is valid up to it
                grid.key_last = torch.empty((q.shape[0],), dtype=torch.int64, device=q.device)
                grid.radius = float(np.float32(radius))
                grid.overflow = torch.zeros((1,), dtype=torch.int32, device=q.device)
                check(lib.ws_radius_neighbors_set_key_last(ws, ptr(grid.key_last)))
            check(lib.ws_radius_neighbors_search_async(
                ws, ptr(q), q.shape[0], ptr(s), s.shape[0], C.c_void_p(ql.ctypes.data), C.c_void_p(sl.ctypes.data),
                ql.shape[0], float(np.float32(radius)), width, None, ptr(out),
                C.c_void_p(self.slots.data_ptr() + 4 * slot), current_stream()))
            order = None
            if want_order:
                order = torch.empty(s.shape[0], dtype=torch.int32, device=q.device)
                check(lib.ws_radius_neighbors_order(ws, ptr(order), current_stream()))
            if want_grid:
                nb, cells, ns_, nbytes = C.c_int32(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
                check(lib.ws_radius_neighbors_grid_info(ws, C.byref(nb), C.byref(cells), C.byref(ns_), C.byref(nbytes)))
                grid.blob = torch.empty((nbytes.value,), dtype=torch.uint8, device=q.device)
                grid.nb, grid.cells, grid.ns = nb.value, cells.value, ns_.value
                check(lib.ws_radius_neighbors_grid_export(ws, ptr(grid.blob), current_stream()))
        self.calls.append((out, (q, s, ql, sl, radius, width)))
        if want_grid:
            return out, order, grid
        return (out, order) if want_order else out

    def finish(self):
        counts = self.slots[:len(self.calls)].cpu().numpy() if self.calls else []
        self.last_counts = [int(c) for c in counts]      # true maximum row length of every call, in order
        final = []
        for (out, args), mc in zip(self.calls, counts):
            q, s, ql, sl, radius, width = args
            if mc == 0:
                raise _lib.WeasalHipError("libweasal_hip status 4: Error")
            if mc > int(_lib.lib().ws_radius_neighbors_async_cap(int(width))):      # beyond the sort slab the asynchronous pass used
                widen_async_slabs(mc)
                out = radius_neighbors(q, s, ql, sl, radius, limit=width, dtype=torch.int64)
            elif mc < width:
                out = out[:, :int(mc)].contiguous()
            final.append(out)
        self.calls = []
        return final


def grid_subsample(points, lens, dl, max_p=0, features=None, labels=None, reference_order=True,
                   return_keys=False):
    """Batched grid subsampling on device tensors (reference: cpp_subsampling.subsample_batch,
    grid_subsampling.cpp:109-211).  Returns (points [M,3], lens (numpy int32 [B]) [, features][, labels])."""
    import ctypes as C
    lib = _lib.lib()
    _need_cuda(points, features, labels)
    p = _f32c(points)
    hl = _host_lens(lens)
    nb = hl.shape[0]
    ws = _ws.subsample(p.device)
    out_lens = np.zeros(nb, dtype=np.int32)
    m = C.c_int64(0)
    f = _f32c(features)
    lab = None
    ld = 0
    if labels is not None:
        lab = labels.detach().to(torch.int32).contiguous()
        ld = 1 if lab.dim() == 1 else lab.shape[1]
    fd = f.shape[1] if f is not None else 0
    with torch.cuda.device(p.device):
        check(lib.ws_grid_subsample_plan(ws, ptr(p), p.shape[0], C.c_void_p(hl.ctypes.data), nb,
                                         float(np.float32(dl)), int(max_p), 0 if reference_order else 1,
                                         C.c_void_p(out_lens.ctypes.data), C.byref(m), current_stream()))
        M = m.value
        out_p = torch.empty((M, 3), dtype=torch.float32, device=p.device)
        out_f = torch.empty((M, fd), dtype=torch.float32, device=p.device) if f is not None else None
        out_l = torch.empty((M, ld), dtype=torch.int32, device=p.device) if lab is not None else None
        keys = torch.empty(M, dtype=torch.int64, device=p.device) if return_keys else None
        cnts = torch.empty(M, dtype=torch.int32, device=p.device) if return_keys else None
        check(lib.ws_grid_subsample_fill(ws, ptr(f), fd, ptr(lab), ld, ptr(out_p), ptr(out_f), ptr(out_l),
                                         ptr(keys), ptr(cnts), current_stream()))
    res = [out_p, out_lens]
    if f is not None:
        res.append(out_f)
    if lab is not None:
        res.append(out_l)
    if return_keys:
        res += [keys, cnts]
    return tuple(res)


def rotate_clouds_host(points, lens, rot, transpose=False):
    """rotate_clouds with HOST lengths (int32 [B]) and matrices (float32 [B,3,3]): they travel as a kernel
    argument, so the call neither copies nor synchronises (up to 64 batch elements)."""
    import ctypes as C
    lib = _lib.lib()
    _need_cuda(points)
    p = _f32c(points)
    hl = np.ascontiguousarray(lens, dtype=np.int32)
    hr = np.ascontiguousarray(rot, dtype=np.float32)
    if hl.shape[0] > 64:
        dev = p.device
        return rotate_clouds(p, torch.from_numpy(hl).to(dev), torch.from_numpy(hr).to(dev), transpose)
    out = torch.empty_like(p)
    check(lib.ws_rotate_clouds_host(ptr(p), p.shape[0], C.c_void_p(hl.ctypes.data), hl.shape[0],
                                    C.c_void_p(hr.ctypes.data), 1 if transpose else 0, ptr(out), current_stream()))
    return out


def rotate_clouds(points, lens_dev, rot, transpose=False):
    """out[i] = points[i] @ R[b] (or R[b].T) in the f32 order of datasets/common.py:116-119,131-135"""
    lib = _lib.lib()
    _need_cuda(points, lens_dev, rot)
    p = _f32c(points)
    out = torch.empty_like(p)
    check(lib.ws_rotate_clouds(ptr(p), p.shape[0], ptr(lens_dev), lens_dev.shape[0], ptr(rot.contiguous()),
                               1 if transpose else 0, ptr(out), current_stream()))
    return out


# ------------------------------------------------------------------------------------------------
# supervised contrastive loss, the [N, slc_con] part (KPFCNN.contrast_loss, architectures.py:455-497)
# ------------------------------------------------------------------------------------------------
class _ContrastRows(torch.autograd.Function):
    """loss[i] of every point against the slice rows; gradients to the normalised logits `on` (as point
    rows) and to the slice rows `xs` (= on[slc_idx], indexed by the caller so autograd adds them back)."""

    @staticmethod
    def forward(ctx, on, xs, slc_idx, certain, lbl, temperature, eps):
        lib = _lib.lib()
        on, xs = on.contiguous(), xs.contiguous()
        n, c = on.shape
        s = xs.shape[0]
        slc_idx = slc_idx.to(torch.int64).contiguous()
        certain = certain.to(torch.uint8).contiguous()
        lbl = lbl.to(torch.int64).contiguous()
        loss, rowmax, den, npos = (torch.empty((n,), dtype=torch.float32, device=on.device) for _ in range(4))
        check(lib.ws_contrast_rows_fwd(ptr(on), n, c, ptr(xs), s, ptr(slc_idx), ptr(certain), ptr(lbl),
                                       float(temperature), float(eps), ptr(loss), ptr(rowmax), ptr(den), ptr(npos),
                                       current_stream()))
        ctx.save_for_backward(on, xs, slc_idx, certain, lbl, rowmax, den, npos)
        ctx.temperature = float(temperature)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.lib()
        on, xs, slc_idx, certain, lbl, rowmax, den, npos = ctx.saved_tensors
        n, c = on.shape
        s = xs.shape[0]
        g = g.contiguous().float()
        d_on = torch.empty_like(on)
        d_xs = torch.empty_like(xs)
        scratch = torch.empty(max(lib.ws_contrast_rows_bwd_scratch_bytes(n, c, s), 16), dtype=torch.uint8, device=on.device)
        check(lib.ws_contrast_rows_bwd(ptr(on), n, c, ptr(xs), s, ptr(slc_idx), ptr(certain), ptr(lbl), ctx.temperature,
                                       ptr(rowmax), ptr(den), ptr(npos), ptr(g), ptr(d_on), ptr(d_xs), ptr(scratch),
                                       current_stream()))
        return d_on, d_xs, None, None, None, None, None


class _SoftmaxCE(torch.autograd.Function):
    """label mapping + weighted cross entropy with ignore_index -1 (ws_softmax_ce_fwd / _bwd): scalar loss"""

    @staticmethod
    def forward(ctx, logits, labels, lut, weight):
        lib = _lib.lib()
        x = logits if (logits.stride(1) == 1 and logits.stride(0) >= logits.shape[1]) else logits.contiguous()
        n, c = x.shape
        labels = labels.to(torch.int64).contiguous()
        out = torch.empty(2, dtype=torch.float32, device=x.device)          # loss, sum of weights
        scratch = torch.empty(lib.ws_softmax_ce_scratch_bytes(n), dtype=torch.uint8, device=x.device)
        check(lib.ws_softmax_ce_fwd(ptr(x), n, c, x.stride(0), ptr(labels), ptr(lut), 0 if lut is None else lut.shape[0],
                                    ptr(weight), ptr(out), out.data_ptr() + 4, ptr(scratch), current_stream()))
        ctx.save_for_backward(x, labels, lut, weight, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.lib()
        x, labels, lut, weight, out = ctx.saved_tensors
        n, c = x.shape
        g = g.to(torch.float32).contiguous()
        d = torch.empty((n, c), dtype=torch.float32, device=x.device)
        check(lib.ws_softmax_ce_bwd(ptr(x), n, c, x.stride(0), ptr(labels), ptr(lut), 0 if lut is None else lut.shape[0],
                                    ptr(weight), ptr(g), out.data_ptr() + 4, ptr(d), c, current_stream()))
        return d, None, None, None


class _Dropout(torch.autograd.Function):
    """nn.Dropout(p) in training mode as one pass each way (ws_dropout_apply): the keep mask is a function of (seed, index)
    and is recomputed by the backward instead of stored"""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = x.contiguous()
        out = torch.empty_like(x)
        check(_lib.lib().ws_dropout_apply(ptr(x), x.numel(), float(p), int(seed), ptr(out), current_stream()))
        ctx.p, ctx.seed = float(p), int(seed)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        out = torch.empty_like(g)
        check(_lib.lib().ws_dropout_apply(ptr(g), g.numel(), ctx.p, ctx.seed, ptr(out), current_stream()))
        return out, None, None


def dropout(x, p, seed=None):
    """training-mode dropout of a float32 device tensor; `seed` defaults to a draw from torch's default (CPU) generator, so
    torch.manual_seed makes a run reproducible"""
    _need_cuda(x)
    if x.dtype != torch.float32:
        raise _lib.WeasalHipError("dropout takes float32 tensors (got %s)" % x.dtype)
    if seed is None:
        seed = int(torch.randint(0, 1 << 62, (1,)).item())
    return _Dropout.apply(x, p, seed)


def cross_entropy(logits, labels, lut=None, weight=None):
    """KPFCNN.loss's criterion (models/architectures.py:362-373): `lut` [V+2] int64 maps raw label values to class
    positions (last entry: the spare -1 of out-of-table labels; None = labels are positions already, < 0 ignored), then
    CrossEntropyLoss(weight, ignore_index=-1) over the rows of logits [N, C] -- one fused pass each way"""
    _need_cuda(logits, labels)
    if logits.dim() != 2 or logits.dtype != torch.float32:
        raise _lib.WeasalHipError("cross_entropy takes float32 logits [N, C] (got %s %s)" % (logits.dtype, tuple(logits.shape)))
    lut = None if lut is None else lut.to(device=logits.device, dtype=torch.int64).contiguous()
    weight = None if weight is None else weight.to(device=logits.device, dtype=torch.float32).contiguous()
    return _SoftmaxCE.apply(logits, labels, lut, weight)


class _ContrastLoss(torch.autograd.Function):
    """the WHOLE of KPFCNN.contrast_loss (architectures.py:405-504) as one node: ws_contrast_head_fwd (softmax statistics,
    pseudo labels, normalisation, device-side slice draw) -> ws_contrast_rows_fwd -> ws_contrast_tail_fwd; five launches
    each way, no host synchronisation.  Returns (loss scalar, per_class [n_cls], slc_idx [s], num_valid int32 [1])."""

    @staticmethod
    def forward(ctx, x, labels, draw, threshold, temperature, eps, n_cls):
        lib = _lib.lib()
        x = x if (x.stride(1) == 1 and x.stride(0) >= x.shape[1]) else x.contiguous()
        n, c = x.shape
        dev = x.device
        labels = labels.to(torch.int64).contiguous()
        u = draw if draw.dtype == torch.float32 else None
        r = draw if draw.dtype == torch.int64 else None
        s = draw.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        on = torch.empty((n, c), **f32)
        inv_norm, pts, rowmax, den, npos = (torch.empty((n,), **f32) for _ in range(5))
        certain = torch.empty((n,), dtype=torch.uint8, device=dev)
        lbl = torch.empty((n,), dtype=torch.int64, device=dev)
        slc_idx = torch.empty((s,), dtype=torch.int64, device=dev)
        xs = torch.empty((s, c), **f32)
        state = torch.empty((2,), dtype=torch.int32, device=dev)
        small = torch.empty((2 * n_cls + 1,), **f32)                 # per_class | w_cls | loss
        scratch = torch.empty(max(lib.ws_contrast_head_scratch_bytes(n), lib.ws_contrast_tail_scratch_bytes(n)),
                              dtype=torch.uint8, device=dev)
        st = current_stream()
        check(lib.ws_contrast_head_fwd(ptr(x), n, c, x.stride(0), ptr(labels), float(threshold), ptr(u), ptr(r), s, ptr(on),
                                       ptr(inv_norm), ptr(certain), ptr(lbl), ptr(slc_idx), ptr(xs), ptr(state), ptr(scratch), st))
        check(lib.ws_contrast_rows_fwd(ptr(on), n, c, ptr(xs), s, ptr(slc_idx), ptr(certain), ptr(lbl), float(temperature),
                                       float(eps), ptr(pts), ptr(rowmax), ptr(den), ptr(npos), st))
        per_class, w_cls, loss = small[:n_cls], small[n_cls:2 * n_cls], small[2 * n_cls]
        check(lib.ws_contrast_tail_fwd(ptr(pts), ptr(lbl), n, n_cls, ptr(state), ptr(per_class), ptr(w_cls),
                                       small.data_ptr() + 8 * n_cls, ptr(scratch), st))
        ctx.save_for_backward(on, xs, slc_idx, certain, lbl, rowmax, den, npos, pts, inv_norm, small)
        ctx.temperature, ctx.n_cls = float(temperature), n_cls
        ctx.mark_non_differentiable(per_class, slc_idx, state)
        return loss, per_class, slc_idx, state

    @staticmethod
    def backward(ctx, g, _g_pc, _g_idx, _g_state):
        lib = _lib.lib()
        on, xs, slc_idx, certain, lbl, rowmax, den, npos, pts, inv_norm, small = ctx.saved_tensors
        n, c = on.shape
        s = xs.shape[0]
        n_cls = ctx.n_cls
        st = current_stream()
        g = g.to(torch.float32).contiguous()
        g_row = torch.empty_like(pts)
        check(lib.ws_contrast_tail_bwd(ptr(pts), ptr(lbl), n, n_cls, small.data_ptr() + 4 * n_cls, ptr(g), ptr(g_row), st))
        d_on = torch.empty_like(on)
        d_xs = torch.empty_like(xs)
        scratch = torch.empty(max(lib.ws_contrast_rows_bwd_scratch_bytes(n, c, s), 16), dtype=torch.uint8, device=on.device)
        check(lib.ws_contrast_rows_bwd(ptr(on), n, c, ptr(xs), s, ptr(slc_idx), ptr(certain), ptr(lbl), ctx.temperature,
                                       ptr(rowmax), ptr(den), ptr(npos), ptr(g_row), ptr(d_on), ptr(d_xs), ptr(scratch), st))
        d_x = torch.empty_like(on)
        check(lib.ws_contrast_head_bwd(ptr(d_on), ptr(d_xs), ptr(slc_idx), s, ptr(on), ptr(inv_norm), n, c, ptr(d_x), c, st))
        return d_x, None, None, None, None, None, None


def contrast_loss(x, labels, draw, threshold, temperature=0.1, eps=1e-8, n_cls=None):
    """KPFCNN.contrast_loss (architectures.py:405-504) on logits x [N, C] (C <= 16) and labels [N] (>= 10 = unlabelled).
    `draw`: float32 [s] uniforms in [0, 1) (slot j takes valid point floor(u_j * num_valid)) or int64 [s] positions in the list
    of valid points.  -> (loss, per_class [n_cls], slc_idx [s], state int32 [2] = (num_valid, 0))"""
    _need_cuda(x, labels, draw)
    if x.dim() != 2 or x.dtype != torch.float32 or x.shape[1] > 16:
        raise _lib.WeasalHipError("contrast_loss takes float32 logits [N, C <= 16] (got %s %s)" % (x.dtype, tuple(x.shape)))
    if draw.dtype not in (torch.float32, torch.int64) or draw.dim() != 1:
        raise _lib.WeasalHipError("contrast_loss: draw must be float32 uniforms or int64 positions [s]")
    n_cls = max(int(x.shape[1]), 10) if n_cls is None else int(n_cls)
    return _ContrastLoss.apply(x, labels, draw.contiguous(), threshold, temperature, eps, n_cls)


def contrast_rows(on, xs, slc_idx, certain, lbl, temperature, eps):
    """per-point supervised contrastive loss [N] (ws_contrast_rows_fwd / _bwd); on [N,C] normalised logits,
    xs [S,C] = on[slc_idx], certain [N] bool, lbl [N] pseudo labels"""
    _need_cuda(on, xs)
    return _ContrastRows.apply(on, xs, slc_idx, certain, lbl, temperature, eps)
