"""Whole network blocks as ONE autograd node over ONE C call each way (ws_kpblock_fwd / _bwd, ws_upunary_fwd / _bwd,
weasal_amd/csrc/blocks.hip).

The reference runs a ResnetBottleneckBlock (models/blocks.py:624-709) as ~10 torch ops and as many autograd nodes;
round 1 of this build ran it as 6 autograd.Function nodes with ~40 kernel launches issued from Python one by one
(~11 ms of host time per step, more than most of those kernels take).  Here the block's forward is one call that
launches its 6-7 kernels from C, its backward one call that launches 10-14; the gradient accumulations autograd used
to add with separate kernels (shortcut + main branch into the block input) are residual operands of GEMM epilogues.
Same arithmetic as weasal_amd.blocks' operator-by-operator path (same kernels, same order inside every sum), which
stays in place for everything these calls do not cover (deformable / modulated KPConv, bf16 rows, non-linear influence,
CPU oracle mode).
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib, ops
from ._lib import check, current_stream

FUSED_BLOCKS = os.environ.get("WEASAL_FUSED_BLOCKS", "1") != "0"      # A/B switch (diagnostics, tests)
FUSED_INFER = os.environ.get("WEASAL_FUSED_INFER", "1") != "0"        # A/B switch: forward-only 32 -> 32 layers in one launch
_timed = False


_timed_min_rows = 0


def set_timed(on, min_rows=0):
    """bench.py: bracket the K3 launch (and the contraction after it) inside the block calls with HIP events (ws_timer_*);
    min_rows: only layers with at least that many query rows (every event is a marker packet on the launch stream:
    timing all ~25 layers of a step costs more than it tells)"""
    global _timed, _timed_min_rows
    _timed = bool(on)
    _timed_min_rows = int(min_rows)


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class KPBlockDesc(C.Structure):
    """mirror of `struct ws_kpblock` (include/weasal_hip.h), field by field"""
    _fields_ = [("q_pts", _vp), ("nq", _i64), ("s_pts", _vp), ("ns", _i64), ("inds", _vp), ("h", _i32),
                ("kernel_points", _vp), ("k", _i32), ("extent", _f32), ("order_q", _vp), ("order_s", _vp),
                ("grid_blob", _vp), ("grid_nb", _i32), ("grid_cells", _i64), ("key_last", _vp), ("grid_radius", _f32),
                ("grid_overflow", _vp), ("t_offsets", _vp), ("t_pairs", _vp),
                ("in_dim", _i32), ("conv_in", _i32), ("conv_out", _i32), ("out_dim", _i32), ("strided", _i32),
                ("slope", _f32),
                ("w1", _vp), ("b1", _vp), ("wk", _vp), ("bk", _vp), ("w2", _vp), ("b2", _vp), ("ws", _vp), ("bs", _vp),
                ("feat", _vp), ("x1", _vp), ("wf", _vp), ("x2", _vp), ("pooled", _vp), ("arg", _vp), ("out", _vp),
                ("dout", _vp), ("dfeat", _vp), ("dw1", _vp), ("db1", _vp), ("dwk", _vp), ("dbk", _vp), ("dw2", _vp),
                ("db2", _vp), ("dws", _vp), ("timed", _i32), ("rows_sorted", _i32), ("infer", _i32), ("dout_pregated", _i32),
                ("gate_dfeat", _i32), ("dfeat_add", _vp)]


class UpUnaryDesc(C.Structure):
    """mirror of `struct ws_upunary`"""
    _fields_ = [("xc", _vp), ("nc", _i64), ("c_up", _i32), ("skip", _vp), ("nf", _i64), ("c_skip", _i32),
                ("ups", _vp), ("h_up", _i32), ("t_offsets", _vp), ("t_pairs", _vp),
                ("w", _vp), ("ldw", _i64), ("b", _vp), ("out_dim", _i32), ("relu", _i32), ("slope", _f32),
                ("yc", _vp), ("out", _vp), ("dout", _vp), ("dxc", _vp), ("dskip", _vp), ("dw", _vp), ("db", _vp),
                ("drop_p", _f32), ("drop_seed", C.c_uint64), ("dout_pregated", _i32), ("gate_dxc", _i32)]


_GATES_SET = False


def _bind():
    global _GATES_SET
    lib = _lib.lib()           # signatures: _lib.SIGNATURES (descriptors travel as void* = ctypes.byref(struct))
    if not _GATES_SET:         # diagnostics: WEASAL_BLOCK_GATES=0 = activation backward as separate passes
        C.c_int.in_dll(lib, "ws_block_gates").value = 0 if os.environ.get("WEASAL_BLOCK_GATES", "1") == "0" else 1
        C.c_int.in_dll(lib, "ws_block_fused_infer").value = 1 if FUSED_INFER else 0
        C.c_int.in_dll(lib, "ws_block_gather_residual").value = 0 if os.environ.get("WEASAL_BLOCK_GATHER_RESIDUAL", "1") == "0" else 1
        C.c_int.in_dll(lib, "ws_block_pool_order").value = 0 if os.environ.get("WEASAL_POOL_ORDER", "1") == "0" else 1
        if "WEASAL_BLOCK_SIDE_ROWS" in os.environ:      # 0 = weight-gradient products on the caller's stream
            C.c_int64.in_dll(lib, "ws_block_side_rows").value = int(os.environ["WEASAL_BLOCK_SIDE_ROWS"])
        _GATES_SET = True
    return lib


def timer_records(layer=False):
    """[(nq, h, ci, ms)] of the K3 launches timed inside the block calls since the last reset (synchronises);
    layer=True: [(nq, h, ci, ms, ms_layer)] with the time from the start of K3 to the end of the contraction"""
    lib = _bind()
    out = []
    nq, h, ci, ms, ml = _i64(), _i32(), _i32(), _f32(), _f32()
    for i in range(lib.ws_timer_count()):
        check(lib.ws_timer_read(i, C.byref(nq), C.byref(h), C.byref(ci), C.byref(ms)))
        if layer:
            check(lib.ws_timer_read_layer(i, C.byref(ml)))
            out.append((nq.value, h.value, ci.value, ms.value, ml.value))
        else:
            out.append((nq.value, h.value, ci.value, ms.value))
    return out


def timer_reset():
    _bind().ws_timer_reset()


def _p(t):
    return None if t is None else t.data_ptr()


def _scratch(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


class _Geom:
    """geometry + widths of one block call (plain Python object carried through the autograd node)"""
    __slots__ = ("q_pts", "s_pts", "inds", "kp", "extent", "order_q", "order_s", "grid", "table", "in_dim", "conv_in",
                 "conv_out", "out_dim", "strided", "slope", "has", "rows_sorted", "infer", "skip_slot", "link_in", "link_out")

    def fill(self, d):
        d.q_pts, d.nq = self.q_pts.data_ptr(), self.q_pts.shape[0]
        d.s_pts, d.ns = self.s_pts.data_ptr(), self.s_pts.shape[0]
        d.inds, d.h = self.inds.data_ptr(), self.inds.shape[1]
        d.kernel_points, d.k, d.extent = self.kp.data_ptr(), self.kp.shape[0], float(self.extent)
        d.order_q, d.order_s = _p(self.order_q), _p(self.order_s)
        if self.grid is not None:
            g = self.grid
            d.grid_blob, d.grid_nb, d.grid_cells = g.blob.data_ptr(), g.nb, g.cells
            d.key_last, d.grid_radius, d.grid_overflow = g.key_last.data_ptr(), g.radius, g.overflow.data_ptr()
        if self.table is not None:
            d.t_offsets, d.t_pairs = self.table.offsets.data_ptr(), self.table.pairs.data_ptr()
        d.in_dim, d.conv_in, d.conv_out, d.out_dim = self.in_dim, self.conv_in, self.conv_out, self.out_dim
        d.strided, d.slope = 1 if self.strided else 0, float(self.slope)
        d.rows_sorted = 1 if self.rows_sorted else 0
        d.infer = 1 if self.infer else 0


def _al(n):
    return (n + 63) // 64 * 64


class SkipSlot:
    """An encoder tensor read twice -- by the strided block that follows it and, through the skip connection, by a decoder
    step (architectures.py:330-341) -- gets two gradients, which autograd adds in a pass of its own (3 x 205 MB at level 0).
    With a slot the decoder's share is parked here by `skip_tap` and the strided block's backward sums it into the gradient
    it writes anyway (ws_kpblock.dfeat_add).  Whichever of the two backward nodes runs second finds out from the slot, so
    the result does not depend on the engine's order: if the block's backward came first, the tap returns its gradient the
    ordinary way."""
    __slots__ = ("grad", "armed", "taken")

    def __init__(self):
        self.grad, self.armed, self.taken = None, False, False


class GateLink:
    """Producer -> consumer link between two consecutive block calls.  The consumer's input IS the producer's LeakyReLU output
    and nothing else reads it (but a skip connection whose share arrives through a SkipSlot), so the consumer's backward can
    multiply the gradient it writes by LeakyReLU'(input) on the store (ws_kpblock.gate_dfeat / ws_upunary.gate_dxc) and the
    producer's backward skip its first pass over [rows, channels] (dout_pregated) -- blocks.py:473-507's activation backward
    without a pass of its own.  The consumer decides in EVERY backward (pregated is rewritten each time) and only when every
    contribution to the gradient is in its hands."""
    __slots__ = ("out", "pregated", "drop", "slope")

    def __init__(self):
        # out: a WEAK reference to the producer's output -- the link hangs off that tensor's own autograd node (ctx.geom /
        # ctx.links), so a strong one would close a cycle only the cyclic collector could free (0.7 GB per DALES step)
        self.out, self.pregated = None, False
        self.drop = None            # (p, seed) when the producer's output went through its fused dropout: the gate undoes it too
        self.slope = 0.1            # the producer's LeakyReLU slope


GATE_LINKS = os.environ.get("WEASAL_GATE_LINKS", "1") != "0"       # A/B switch: 0 = every block runs its own activation backward


def linear_links(batch, x, eligible):
    """[incoming, outgoing] links of a UnaryBlock called between block calls (the head); the outgoing one is completed by
    linear_links_done once the output exists"""
    if not eligible or batch is None or not GATE_LINKS:
        return None
    return [_link_in(batch, x), None]


def linear_links_done(batch, links, out, relu):
    if links is not None and batch is not None:
        links[1] = _link_out(batch, out, relu=relu)


def _link_in(batch, x):
    """the incoming link of the block about to run, if its producer ran as a block call and x is that call's output itself"""
    link = getattr(batch, "gate_link_in", None)
    if (link is None or not GATE_LINKS or link.out is None or link.out() is not x or not torch.is_grad_enabled()
            or not x.requires_grad or x.dtype != torch.float32):
        return None
    return link


def _link_out(batch, out, relu=True):
    """hand the block's output to the link its consumer will look at"""
    link = getattr(batch, "gate_link_out", None)
    if link is not None and GATE_LINKS and relu and torch.is_grad_enabled() and out.requires_grad:
        link.out = weakref.ref(out)
        return link
    return None


class _SkipTap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        slot = ctx.slot
        if slot.armed and not slot.taken and slot.grad is None:
            slot.grad = g
            return None, None
        return g, None


def skip_tap(x, slot):
    """the skip connection's view of an encoder tensor whose other reader is an armed strided block call"""
    if slot is None or not slot.armed or not torch.is_grad_enabled() or not x.requires_grad:
        return x
    return _SkipTap.apply(x, slot)


class _KPBlockFn(torch.autograd.Function):
    """[unary1 ->] KPConv -> bias -> LeakyReLU [-> unary2 + shortcut -> LeakyReLU] (blocks.py:510-564, 624-709)"""

    @staticmethod
    def forward(ctx, feat, w1, b1, wk, bk, w2, b2, wsc, bsc, geom):
        lib = _bind()
        dev = feat.device
        feat = feat.contiguous()
        g = geom
        nq, ns = g.q_pts.shape[0], g.s_pts.shape[0]
        k = g.kp.shape[0]
        # saved activations in ONE arena: x1 | wf | x2 | pooled | arg
        fused_layer = g.infer and g.conv_in == 32 and g.conv_out == 32 and not g.rows_sorted and FUSED_INFER
        sizes = [ns * g.conv_in if w1 is not None else 0, 0 if fused_layer else nq * k * g.conv_in, nq * g.conv_out if w2 is not None else 0,
                 nq * g.in_dim if (g.strided and w2 is not None) else 0, nq * g.in_dim if (g.strided and w2 is not None) else 0]
        offs, tot = [], 0
        for s in sizes:
            offs.append(tot)
            tot += _al(s)
        arena = torch.empty(max(tot, 64), dtype=torch.float32, device=dev)
        base = arena.data_ptr()
        out = torch.empty((nq, g.out_dim), dtype=torch.float32, device=dev)
        d = KPBlockDesc()
        g.fill(d)
        wkc = wk.contiguous()
        d.w1, d.b1, d.wk, d.bk, d.w2, d.b2, d.ws, d.bs = _p(w1), _p(b1), wkc.data_ptr(), _p(bk), _p(w2), _p(b2), _p(wsc), _p(bsc)
        d.feat = feat.data_ptr()
        d.x1 = base + 4 * offs[0] if sizes[0] else None
        d.wf = base + 4 * offs[1] if sizes[1] else None
        d.x2 = base + 4 * offs[2] if sizes[2] else None
        d.pooled = base + 4 * offs[3] if sizes[3] else None
        d.arg = base + 4 * offs[4] if sizes[4] else None
        d.out = out.data_ptr()
        d.timed = 1 if (_timed and g.q_pts.shape[0] >= _timed_min_rows) else 0
        nbytes = lib.ws_kpblock_fwd_scratch_bytes(C.byref(d))
        if nbytes < 0:
            check(1)
        scratch = _scratch(nbytes, dev)
        check(lib.ws_kpblock_fwd(C.byref(d), scratch.data_ptr(), scratch.numel(), current_stream()))
        ctx.geom, ctx.offs, ctx.sizes = g, offs, sizes
        ctx.save_for_backward(feat, arena, out, w1, b1, wkc, bk, w2, b2, wsc, bsc)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _bind()
        feat, arena, out, w1, b1, wk, bk, w2, b2, wsc, bsc = ctx.saved_tensors
        g, offs, sizes = ctx.geom, ctx.offs, ctx.sizes
        dev = feat.device
        dout = dout.contiguous()
        nq, ns = g.q_pts.shape[0], g.s_pts.shape[0]
        need = ctx.needs_input_grad
        d = KPBlockDesc()
        g.fill(d)
        d.w1, d.b1, d.wk, d.bk, d.w2, d.b2, d.ws, d.bs = _p(w1), _p(b1), wk.data_ptr(), _p(bk), _p(w2), _p(b2), _p(wsc), _p(bsc)
        base = arena.data_ptr()
        d.feat = feat.data_ptr()
        d.x1 = base + 4 * offs[0] if sizes[0] else None
        d.wf = base + 4 * offs[1]
        d.x2 = base + 4 * offs[2] if sizes[2] else None
        d.pooled = base + 4 * offs[3] if sizes[3] else None
        d.arg = base + 4 * offs[4] if sizes[4] else None
        d.out, d.dout = out.data_ptr(), dout.data_ptr()
        dfeat = torch.empty_like(feat) if need[0] else None
        d.dfeat = _p(dfeat)
        slot = getattr(g, "skip_slot", None)
        extra = None
        if slot is not None:
            slot.taken = True                              # (a tap that runs after this point returns its gradient itself)
            extra, slot.grad = slot.grad, None
            if extra is not None and (dfeat is None or extra.shape != feat.shape or extra.dtype != torch.float32):
                raise _lib.WeasalHipError("skip gradient %s does not match the block input %s" % (tuple(extra.shape), tuple(feat.shape)))
            if extra is not None:
                extra = extra.contiguous()
                global skip_slot_hits
                skip_slot_hits += 1
        d.dfeat_add = _p(extra)
        # gate links: this block's output gradient may arrive with the LeakyReLU' applied; its input gradient may leave so
        lo, li = getattr(g, "link_out", None), getattr(g, "link_in", None)
        d.dout_pregated = 1 if (lo is not None and lo.pregated) else 0
        if li is not None:
            # (an armed slot whose share has not arrived will be added by autograd afterwards: no gate then)
            complete = slot is None or not slot.armed or extra is not None
            li.pregated = bool(dfeat is not None and complete and (extra is None or (g.strided and w2 is not None)))
            d.gate_dfeat = 1 if li.pregated else 0
            global gate_link_hits
            gate_link_hits += d.gate_dfeat
        # parameter gradients: one tensor each (autograd adopts an unshared, contiguous gradient as .grad without a copy)
        grads = [None if p is None else torch.empty_like(p) for p in (w1, b1, wk, bk, w2, b2, wsc)]
        d.dw1, d.db1, d.dwk, d.dbk, d.dw2, d.db2, d.dws = [_p(t) for t in grads]
        nbytes = lib.ws_kpblock_bwd_scratch_bytes(C.byref(d))
        if nbytes < 0:
            check(1)
        scratch = _scratch(nbytes, dev)
        check(lib.ws_kpblock_bwd(C.byref(d), scratch.data_ptr(), scratch.numel(), current_stream()))
        dw1, db1, dwk, dbk, dw2, db2, dws = grads
        dbs = db2.clone() if (bsc is not None and db2 is not None) else None          # the shortcut bias sees the same dz as b2
        return dfeat, dw1, db1, dwk, dbk, dw2, db2, dws, dbs, None


gate_link_hits = 0      # (tests) input gradients written with the producer's LeakyReLU' so far
skip_slot_hits = 0      # (tests) skip gradients summed inside a strided block's backward so far
SKIP_SLOTS = os.environ.get("WEASAL_SKIP_SLOTS", "1") != "0"      # A/B switch: 0 = autograd adds the two gradients of a skip tensor


def _conv_ok(conv, x):
    return (FUSED_BLOCKS and x.is_cuda and x.dtype == torch.float32 and not conv.deformable and conv.K == 15
            and conv.KP_influence == 'linear' and conv.aggregation_mode == 'sum' and ops.kpconv_gather is _KPCONV_GATHER)


_KPCONV_GATHER = ops.kpconv_gather         # (oracle.kpconv_ref.cpu_reference_mode swaps ops.* : then the blocks run operator by operator)


def _geometry(conv, q_pts, s_pts, inds, strided):
    g = _Geom()
    g.q_pts, g.s_pts = ops._f32c(q_pts), ops._f32c(s_pts)
    inds = inds.contiguous()
    g.inds = inds if inds.dtype == torch.int64 else inds.to(torch.int64)
    g.kp, g.extent = conv.kernel_points, conv.KP_extent
    g.order_q, g.order_s = ops._order_for(g.q_pts), ops._order_for(g.s_pts)
    nq, ns = g.q_pts.shape[0], g.s_pts.shape[0]
    self_query = nq == ns and g.q_pts.data_ptr() == g.s_pts.data_ptr()
    grid = ops._grid_for(g.inds) if self_query else None
    # (the block calls know the slab form of the grid backward only: rows up to 128 neighbours)
    g.grid = grid if (grid is not None and grid.ns == ns and grid.max_count <= ops.GRID_NARROW_MAX) else None
    g.table = None
    g.skip_slot = None
    g.link_in = g.link_out = None
    g.strided = strided
    g.rows_sorted = ops.rows_cutoff_pays(g.inds, conv.radius)      # searched with the deformable radius: stop at the kernel's reach
    g.infer = not torch.is_grad_enabled()      # a forward pass nobody will differentiate (the testers' loops run under no_grad)
    return g


# A/B switch (diagnostics): layers with fewer rows take the operator path.  Default 0 = every layer is a block call: with
# the row-split dW reduction and the per-CU chunking of the short products (gemm.hip: xty_chunk, colsum_chunk) the deep
# levels cost the same GPU time as the library GEMM they used to be handed to (14.28 vs 14.21 ms per DALES step) and
# the host issues a step in 4.3 instead of 7.5 ms.
MIN_ROWS = int(os.environ.get("WEASAL_FUSED_MIN_ROWS", "0"))


def kpblock_eligible(block, x):
    conv = block.KPConv
    if not _conv_ok(conv, x) or x.shape[0] < MIN_ROWS:
        return False
    dims = [conv.out_channels]
    if hasattr(block, "unary2"):
        dims += [block.out_dim, block.in_dim, conv.in_channels]
    return all(v % 4 == 0 for v in dims)


def simple_block(block, x, batch, q_pts, s_pts, inds):
    """SimpleBlock / SimpleBlock2 (blocks.py:510-622): KPConv -> BatchNormBlock bias -> LeakyReLU(0.1)"""
    conv = block.KPConv
    g = _geometry(conv, q_pts, s_pts, inds, 'strided' in block.block_name)
    g.in_dim = g.conv_in = conv.in_channels
    g.conv_out = g.out_dim = conv.out_channels
    g.slope = 0.1
    if torch.is_grad_enabled() and x.requires_grad and g.grid is None:
        g.table = ops.transposed_table(g.inds, g.s_pts.shape[0])
    if getattr(batch, "skip_slot", None) is None:      # (a skip tensor has a second reader this block knows nothing about)
        g.link_in = _link_in(batch, x)
    out = _KPBlockFn.apply(x, None, None, conv.weights, block.batch_norm.epilogue_bias(), None, None, None, None, g)
    g.link_out = _link_out(batch, out)
    return out


def resnetb_block(block, x, batch, q_pts, s_pts, inds):
    """ResnetBottleneckBlock (blocks.py:624-709)"""
    conv = block.KPConv
    strided = 'strided' in block.block_name
    g = _geometry(conv, q_pts, s_pts, inds, strided)
    slot = getattr(batch, "skip_slot", None) if strided else None
    if slot is not None and SKIP_SLOTS and torch.is_grad_enabled() and x.requires_grad and x.dtype == torch.float32:
        g.skip_slot = slot
        slot.armed = True
    g.in_dim, g.conv_in, g.conv_out, g.out_dim = block.in_dim, conv.in_channels, conv.out_channels, block.out_dim
    g.slope = 0.1
    if torch.is_grad_enabled() and (g.grid is None or strided):       # (only a backward reads the table)
        g.table = ops.transposed_table(g.inds, g.s_pts.shape[0])
    u1 = block.unary1 if isinstance(block.unary1, torch.nn.Module) and hasattr(block.unary1, "mlp") else None
    us = block.unary_shortcut if hasattr(block.unary_shortcut, "mlp") else None
    if getattr(batch, "skip_slot", None) is None or g.skip_slot is not None:
        # (a skip tensor has a second reader: only with its slot armed does this block get to see that share)
        g.link_in = _link_in(batch, x)
    out = _KPBlockFn.apply(x,
                           u1.mlp.weight if u1 is not None else None, u1.batch_norm.epilogue_bias() if u1 is not None else None,
                           conv.weights, block.batch_norm_conv.epilogue_bias(),
                           block.unary2.mlp.weight, block.unary2.batch_norm.epilogue_bias(),
                           us.mlp.weight if us is not None else None, us.batch_norm.epilogue_bias() if us is not None else None,
                           g)
    g.link_out = _link_out(batch, out)
    return out


class _UpUnaryFn(torch.autograd.Function):
    """nearest_upsample -> concat(skip) -> unary as up(x @ Wx^T) + skip @ Ws^T (architectures.py:339-343)"""

    @staticmethod
    def forward(ctx, xc, skip, w, b, ups, table, relu, drop_p=0.0, drop_seed=0, links=None):
        lib = _bind()
        dev = xc.device
        xc, skip, w = xc.contiguous(), skip.contiguous(), w.contiguous()
        nc, c_up = xc.shape
        nf, c_skip = skip.shape
        out_dim = w.shape[0]
        yc = torch.empty((max(nc, 1), out_dim), dtype=torch.float32, device=dev)
        out = torch.empty((nf, out_dim), dtype=torch.float32, device=dev)
        d = UpUnaryDesc()
        d.xc, d.nc, d.c_up, d.skip, d.nf, d.c_skip = xc.data_ptr(), nc, c_up, skip.data_ptr(), nf, c_skip
        d.ups, d.h_up = ups.data_ptr(), ups.shape[1]
        d.w, d.ldw, d.b, d.out_dim, d.relu, d.slope = w.data_ptr(), w.stride(0), _p(b), out_dim, 1 if relu else 0, 0.1
        d.yc, d.out = yc.data_ptr(), out.data_ptr()
        d.drop_p, d.drop_seed = float(drop_p), int(drop_seed)      # the droplayer in front of the head, on this step's epilogue
        nbytes = lib.ws_upunary_fwd_scratch_bytes(C.byref(d))
        if nbytes < 0:
            check(1)
        scratch = _scratch(nbytes, dev)
        check(lib.ws_upunary_fwd(C.byref(d), scratch.data_ptr(), scratch.numel(), current_stream()))
        ctx.table, ctx.relu, ctx.drop = table, relu, (float(drop_p), int(drop_seed))
        ctx.links = links if links is not None else [None, None]          # [incoming (xc's producer), outgoing]
        ctx.save_for_backward(xc, skip, w, b, ups, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _bind()
        xc, skip, w, b, ups, out = ctx.saved_tensors
        dev = xc.device
        dout = dout.contiguous()
        nc, c_up = xc.shape
        nf, c_skip = skip.shape
        out_dim = w.shape[0]
        d = UpUnaryDesc()
        d.xc, d.nc, d.c_up, d.skip, d.nf, d.c_skip = xc.data_ptr(), nc, c_up, skip.data_ptr(), nf, c_skip
        d.ups, d.h_up = ups.data_ptr(), ups.shape[1]
        d.t_offsets, d.t_pairs = ctx.table.offsets.data_ptr(), ctx.table.pairs.data_ptr()
        d.w, d.ldw, d.b, d.out_dim, d.relu, d.slope = w.data_ptr(), w.stride(0), _p(b), out_dim, 1 if ctx.relu else 0, 0.1
        d.out, d.dout = out.data_ptr(), dout.data_ptr()
        d.drop_p, d.drop_seed = ctx.drop
        li, lo = ctx.links
        d.dout_pregated = 1 if (lo is not None and lo.pregated) else 0
        if li is not None:
            li.pregated = bool(ctx.needs_input_grad[0])
            d.gate_dxc = 1 if li.pregated else 0
            global gate_link_hits
            gate_link_hits += d.gate_dxc
        dxc, dskip = torch.empty_like(xc), torch.empty_like(skip)
        dw = torch.empty_like(w)
        db = torch.empty_like(b) if b is not None else None
        d.dxc, d.dskip, d.dw, d.db = dxc.data_ptr(), dskip.data_ptr(), dw.data_ptr(), _p(db)
        nbytes = lib.ws_upunary_bwd_scratch_bytes(C.byref(d))
        if nbytes < 0:
            check(1)
        scratch = _scratch(nbytes, dev)
        check(lib.ws_upunary_bwd(C.byref(d), scratch.data_ptr(), scratch.numel(), current_stream()))
        return dxc, dskip, dw, db, None, None, None, None, None, None


def upunary_eligible(x, skip, unary):
    return (FUSED_BLOCKS and x.is_cuda and x.dtype == torch.float32 and skip.dtype == torch.float32
            and ops.kpconv_gather is _KPCONV_GATHER and x.shape[1] % 32 == 0 and skip.shape[1] % 32 == 0
            and unary.out_dim % 32 == 0 and x.shape[0] > 0 and skip.shape[0] >= MIN_ROWS)


def upunary(x, skip, unary, ups, drop=None, batch=None):
    """drop = (p, seed): nn.Dropout(p) applied to the step's output inside its last epilogue (the bits of ops.dropout with that
    seed); needs the unary's LeakyReLU"""
    ups = ups.contiguous()
    ups = ups if ups.dtype == torch.int64 else ups.to(torch.int64)
    table = ops.col0_table(ups, x.shape[0]) if torch.is_grad_enabled() else None      # (only the backward reads it)
    links = [_link_in(batch, x) if batch is not None else None, None]
    if drop is not None:
        if unary.no_relu:
            raise _lib.WeasalHipError("upunary: the fused dropout follows the unary's LeakyReLU")
        out = _UpUnaryFn.apply(x, skip, unary.mlp.weight, unary.batch_norm.epilogue_bias(), ups, table, True, float(drop[0]), int(drop[1]),
                               links)
        if batch is not None:
            links[1] = _link_out(batch, out)
            if links[1] is not None:
                links[1].drop = (float(drop[0]), int(drop[1]))
        return out
    out = _UpUnaryFn.apply(x, skip, unary.mlp.weight, unary.batch_norm.epilogue_bias(), ups, table, not unary.no_relu, 0.0, 0, links)
    if batch is not None:
        links[1] = _link_out(batch, out, relu=not unary.no_relu)
    return out
