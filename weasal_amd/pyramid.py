"""Input pyramid of the KP-FCNN on the GPU.

The reference builds the multi-scale inputs on the CPU inside DataLoader workers
(datasets/common.py:461-577 ``PointCloudDataset.segmentation_inputs``): per layer one
``batch_neighbors`` for the convolution, one ``batch_grid_subsampling`` and two more
``batch_neighbors`` for pooling / upsampling, with the neighbour matrices cropped to the calibrated
``neighborhood_limits`` and cast to int64.  Here the same schedule (radii, cell sizes, crop, list
layout, dtypes) runs on device tensors with the HIP geometry kernels; nothing goes back to the host
except the per-call row widths / lengths.

``batch_grid_subsampling`` reproduces the reference's random grid orientation
(common.py:77-135, ``random_grid_orient=True`` by default): the rotation matrices are drawn from
``np.random`` on the host in the reference's order (theta, phi, alpha per call), the rotations are
applied on the device with the reference's f32 summation order.
"""
import numpy as np
import torch

from . import ops
from .kernel_points import create_3D_rotations


def batch_neighbors(queries, supports, q_batches, s_batches, radius, limit=None):
    """device form of datasets/common.py:185-196 (+ the crop of :336-346 and the int64 cast of :551)"""
    return ops.radius_neighbors(queries, supports, q_batches, s_batches, radius, limit=limit, dtype=torch.int64)


def batch_grid_subsampling(points, batches_len, sampleDl=0.1, max_p=0, random_grid_orient=True, rng=None):
    """device form of datasets/common.py:77-182 (points only, the form the pyramid uses :521).  `rng`: the
    numpy RandomState the orientations are drawn from (default: the global np.random, like the reference)"""
    lens = np.asarray(batches_len, dtype=np.int32)
    B = len(lens)
    if not random_grid_orient:
        return ops.grid_subsample(points, lens, sampleDl, max_p=max_p)
    rand = np.random.rand if rng is None else rng.rand
    theta = rand(B) * 2 * np.pi
    phi = (rand(B) - 0.5) * np.pi
    u = np.vstack([np.cos(theta) * np.cos(phi), np.sin(theta) * np.cos(phi), np.sin(phi)])
    alpha = rand(B) * 2 * np.pi
    R = create_3D_rotations(u.T, alpha).astype(np.float32)
    rotated = ops.rotate_clouds_host(points, lens, R)
    s_points, s_len = ops.grid_subsample(rotated, lens, sampleDl, max_p=max_p)
    s_points = ops.rotate_clouds_host(s_points, s_len, R, transpose=True)
    return s_points, s_len


def segmentation_inputs(config, stacked_points, stacked_features, labels, stack_lengths,
                        neighborhood_limits=(), random_grid_orient=True, point_orders=None, search_grids=None, rng=None,
                        search_radii=None):
    """-> flat list  points[L] + neighbors[L] + pools[L] + upsamples[L] + lengths[L] + [features, labels]
    (datasets/common.py:574-575), all device tensors (lengths int32, indices int64).

    With neighbourhood limits (the normal case after calibration) the searches run without host
    synchronisation and their true widths are checked once at the end (ops.DeferredSearches); without
    limits every search needs its width on the host first (two-call protocol)."""
    dev = stacked_points.device
    stacked_points = stacked_points.detach().to(torch.float32).contiguous()
    orders = [] if point_orders is None else point_orders     # (points, cell order) per layer: scheduling hints
    lens = np.asarray(stack_lengths.cpu() if isinstance(stack_lengths, torch.Tensor) else stack_lengths, dtype=np.int32)
    r_normal = config.first_subsampling_dl * config.conv_radius
    limits = list(neighborhood_limits)
    deferred = ops.DeferredSearches(dev) if len(limits) > 0 else None
    slots = []   # (list, position) of every deferred matrix
    pending_grids = []   # (slot index, SearchGrid) of the self-query searches (table-free KPConv backward)

    last = {"s": None, "r": None}     # supports / radius of the previous deferred search (grid reuse)

    radii_slots = []     # radius of every deferred search, in slot order

    def search(q, s, ql, sl, r, layer, register_order=False):
        reuse = deferred is not None and last["s"] is s and last["r"] == float(np.float32(r))
        last["s"], last["r"] = s, float(np.float32(r))
        if deferred is None:
            if not register_order:
                inds = ops.radius_neighbors(q, s, ql, sl, r, dtype=torch.int64)
            else:
                inds, order = ops.radius_neighbors(q, s, ql, sl, r, dtype=torch.int64, return_order=True)
                orders.append((s, order))
            if search_radii is not None:
                search_radii.append((inds, float(r)))
            return inds
        radii_slots.append(float(r))
        if register_order:
            want_grid = search_grids is not None and q is s
            res = deferred.add(q, s, ql, sl, r, limits[layer], want_order=True, want_grid=want_grid, reuse_grid=reuse)
            inds, order = res[0], res[1]
            orders.append((s, order))
            if want_grid:
                pending_grids.append((len(slots), res[2]))
            return inds
        return deferred.add(q, s, ql, sl, r, limits[layer], reuse_grid=reuse)

    layer_blocks = []
    input_points, input_neighbors, input_pools, input_upsamples, input_lengths = [], [], [], [], []
    empty_i = lambda: torch.zeros((0, 1), dtype=torch.int64, device=dev)
    for block in config.architecture:
        if not any(tag in block for tag in ('pool', 'strided', 'global', 'upsample')):
            layer_blocks.append(block)
            continue
        layer = len(input_points)
        if layer_blocks:
            if any('deformable' in b for b in layer_blocks):
                r = r_normal * config.deform_radius / config.conv_radius
            else:
                r = r_normal
            conv_i = search(stacked_points, stacked_points, lens, lens, r, layer, register_order=True)
            if deferred is not None:
                slots.append((input_neighbors, layer))
        else:
            conv_i = empty_i()
        if 'pool' in block or 'strided' in block:
            dl = 2 * r_normal / config.conv_radius
            pool_p, pool_b = batch_grid_subsampling(stacked_points, lens, sampleDl=dl,
                                                    random_grid_orient=random_grid_orient, rng=rng)
            r = r_normal * config.deform_radius / config.conv_radius if 'deformable' in block else r_normal
            pool_i = search(pool_p, stacked_points, pool_b, lens, r, layer)
            if deferred is not None:
                slots.append((input_pools, layer))
            up_i = search(stacked_points, pool_p, lens, pool_b, 2 * r, layer + 1)
            if deferred is not None:
                slots.append((input_upsamples, layer))
        else:
            pool_i = empty_i()
            pool_p = torch.zeros((0, 3), dtype=torch.float32, device=dev)
            pool_b = np.zeros((0,), dtype=np.int32)
            up_i = empty_i()
        input_points.append(stacked_points)
        input_neighbors.append(conv_i)
        input_pools.append(pool_i)
        input_upsamples.append(up_i)
        input_lengths.append(np.ascontiguousarray(lens))
        stacked_points, lens = pool_p, pool_b
        r_normal *= 2
        layer_blocks = []
        if 'global' in block or 'upsample' in block:
            break
    if deferred is not None:
        final = deferred.finish()
        for (lst, pos), mat in zip(slots, final):
            lst[pos] = mat
        if search_radii is not None:
            search_radii.extend(zip(final, radii_slots))
        for slot, grid in pending_grids:
            # the true maximum row length decides which grid backward the layer takes: the slab form up to 128 (the
            # in-degree of a support is bounded by the longest row), the queue form beyond (any in-degree); a search
            # that had to be redone synchronously (rows beyond the sort slab) built another grid: transposed table
            grid.max_count = deferred.last_counts[slot]
            if 0 < grid.max_count <= grid.cap:
                search_grids.append((final[slot], grid))
    # the per-layer lengths go to the device in ONE copy (views of it are handed out)
    _last_host_lengths.value = [np.ascontiguousarray(a) for a in input_lengths]
    sizes = [len(a) for a in input_lengths]
    all_lens = torch.from_numpy(np.concatenate(input_lengths).astype(np.int32)).to(dev)
    input_lengths = list(torch.split(all_lens, sizes))
    return (input_points + input_neighbors + input_pools + input_upsamples + input_lengths
            + [stacked_features, labels])


# ------------------------------------------------------------------------------------------------
# the same pyramid behind ONE library call (ws_pyramid_build, csrc/pyramid.hip)
# ------------------------------------------------------------------------------------------------
import ctypes as _C
import os as _os
import threading as _threading

NATIVE_PYRAMID = _os.environ.get("WEASAL_NATIVE_PYRAMID", "1") != "0"     # A/B switch: 0 = the per-call loop above
_ML, _MB = 8, 64          # WS_PYRAMID_MAX_LEVELS, WS_PYRAMID_MAX_BATCH


class PyramidDesc(_C.Structure):
    """mirror of `struct ws_pyramid_desc` (include/weasal_hip.h)"""
    _fields_ = ([("n_levels", _C.c_int32), ("nb", _C.c_int32), ("want_grids", _C.c_int32), ("want_tables", _C.c_int32),
                 ("points", _C.c_void_p), ("n0", _C.c_int64), ("h_rot", _C.c_void_p),
                 ("arena", _C.c_void_p), ("arena_bytes", _C.c_int64), ("scratch", _C.c_void_p), ("scratch_bytes", _C.c_int64),
                 ("conv_on", _C.c_int32 * _ML), ("pool_on", _C.c_int32 * _ML),
                 ("r_conv", _C.c_float * _ML), ("r_pool", _C.c_float * _ML), ("r_up", _C.c_float * _ML), ("dl", _C.c_float * _ML),
                 ("limit", _C.c_int32 * (_ML + 1)), ("nearest_up", _C.c_int32),
                 ("lens", (_C.c_int32 * _MB) * _ML),
                 ("needed_bytes", _C.c_int64), ("n", _C.c_int64 * _ML)]
                + [(name, _C.c_int64 * _ML) for name in ("off_points", "off_neighbors", "off_pools", "off_upsamples", "off_order",
                                                          "off_key_last", "off_blob", "blob_bytes", "grid_cells")]
                + [("off_lens", _C.c_int64), ("off_slots", _C.c_int64),
                   ("max_count", _C.c_int32 * (3 * _ML)), ("width", _C.c_int32 * (3 * _ML)),
                   ("final_width", _C.c_int32 * (3 * _ML)), ("reserved2", _C.c_int32 * (3 * _ML)),
                   ("off_toffsets", _C.c_int64 * (3 * _ML)), ("off_tpairs", _C.c_int64 * (3 * _ML))])


_arena_hint = {}          # (device index, thread, n0, limits) -> bytes the previous batch of that shape needed


def _schedule(config, limits):
    """the per-level plan of segmentation_inputs as flat arrays: (conv_on, pool_on, r_conv, r_pool, r_up, dl) per level"""
    r_normal = config.first_subsampling_dl * config.conv_radius
    levels = []
    layer_blocks = []
    for block in config.architecture:
        if not any(tag in block for tag in ('pool', 'strided', 'global', 'upsample')):
            layer_blocks.append(block)
            continue
        lv = dict(conv_on=bool(layer_blocks), pool_on=False, r_conv=0.0, r_pool=0.0, r_up=0.0, dl=0.0)
        if layer_blocks:
            deform = any('deformable' in b for b in layer_blocks)
            lv["r_conv"] = r_normal * config.deform_radius / config.conv_radius if deform else r_normal
        if 'pool' in block or 'strided' in block:
            lv["pool_on"] = True
            lv["dl"] = 2 * r_normal / config.conv_radius
            r = r_normal * config.deform_radius / config.conv_radius if 'deformable' in block else r_normal
            lv["r_pool"], lv["r_up"] = r, 2 * r
        levels.append(lv)
        r_normal *= 2
        layer_blocks = []
        if 'global' in block or 'upsample' in block:
            break
    return levels


def nearest_upsample_only(config):
    """opt-in (config.nearest_upsample_only = True or WEASAL_NEAREST_UPSAMPLE=1; one-call pyramid only): the upsampling
    matrices of the batch hold the nearest support of every point, [N, 1], instead of the full cropped rows.  KP-FCNN reads
    their first column only (models/blocks.py:80-92), so the network computes the same values; the reference's batch carries
    the full rows, which stays the default."""
    return bool(getattr(config, "nearest_upsample_only", False)) or _os.environ.get("WEASAL_NEAREST_UPSAMPLE", "0") != "0"


def native_eligible(config, points, lens, limits):
    return (NATIVE_PYRAMID and points.is_cuda and len(limits) > 0 and 1 <= len(lens) <= _MB
            and len(_schedule(config, limits)) <= _ML)


def segmentation_inputs_native(config, stacked_points, stacked_features, labels, stack_lengths, neighborhood_limits,
                               random_grid_orient=True, point_orders=None, search_grids=None, rng=None, search_radii=None,
                               tables=None):
    """segmentation_inputs for device tensors with neighbourhood limits, through ws_pyramid_build: same flat list, same
    side lists (orders / grids / radii); every output is a view of one arena tensor.  `tables`: a dict that receives the
    pre-built transposed tables {"full": [(matrix, ns, table)], "col0": [(upsampling matrix, nc, table)]}."""
    from . import _lib
    lib = _lib.lib()
    dev = stacked_points.device
    P0 = stacked_points.detach().to(torch.float32).contiguous()
    lens0 = np.asarray(stack_lengths.cpu() if isinstance(stack_lengths, torch.Tensor) else stack_lengths, dtype=np.int32)
    levels = _schedule(config, neighborhood_limits)
    L, B, n0 = len(levels), len(lens0), int(P0.shape[0])
    limits = [int(v) for v in neighborhood_limits]
    if len(limits) < L:
        raise RuntimeError("neighborhood_limits has %d entries for %d layers" % (len(limits), L))
    d = PyramidDesc()
    d.n_levels, d.nb, d.want_grids = L, B, 1 if (search_grids is not None and ops.GRID_BACKWARD) else 0
    d.want_tables = 1 if tables is not None else 0
    d.nearest_up = 1 if nearest_upsample_only(config) else 0
    d.points, d.n0 = P0.data_ptr(), n0
    for l, lv in enumerate(levels):
        d.conv_on[l], d.pool_on[l] = int(lv["conv_on"]), int(lv["pool_on"])
        d.r_conv[l], d.r_pool[l], d.r_up[l], d.dl[l] = lv["r_conv"], lv["r_pool"], lv["r_up"], lv["dl"]
        d.limit[l] = max(1, limits[l])
    d.limit[L] = max(1, limits[L]) if len(limits) > L else 1
    for b in range(B):
        d.lens[0][b] = int(lens0[b])
    rots = None
    if random_grid_orient:
        # the reference's draws, level by level (theta, phi, alpha per call: datasets/common.py:99-121)
        rand = np.random.rand if rng is None else rng.rand
        mats = []
        for lv in levels:
            if not lv["pool_on"]:
                continue
            theta = rand(B) * 2 * np.pi
            phi = (rand(B) - 0.5) * np.pi
            u = np.vstack([np.cos(theta) * np.cos(phi), np.sin(theta) * np.cos(phi), np.sin(phi)])
            alpha = rand(B) * 2 * np.pi
            mats.append(create_3D_rotations(u.T, alpha).astype(np.float32))
        if mats:
            rots = np.ascontiguousarray(np.stack(mats), dtype=np.float32)
            d.h_rot = rots.ctypes.data
    al = lambda v: (int(v) + 255) // 256 * 256
    # temporaries: two point buffers; the largest matrix (cropped copy) / the largest table's scratch (both <= n0 * widest row)
    scratch = torch.empty(2 * al(n0 * 12) + n0 * max(limits[:L + 1]) * 8 + al(n0 * 8) + (1 << 20), dtype=torch.uint8, device=dev)
    d.scratch, d.scratch_bytes = scratch.data_ptr(), scratch.numel()
    key = (dev.index or 0, _threading.get_ident(), n0, tuple(limits), L)
    hint = _arena_hint.get(key)
    if hint is None:
        # first batch of this shape: guess that every level halves the point count (the subsampling cell doubles: usually
        # a factor 4-6); a batch that needs more comes back with WS_ERR_CAPACITY and its exact size
        hint = 1 << 20
        for l in range(L):
            nxt = limits[l + 1] if l + 1 < len(limits) else 0
            hint += (n0 >> l) * ((2 * limits[l] + nxt) * 8 + 64) + 64 * B * 4
    nws, sws = ops._ws.neighbors(dev), ops._ws.subsample(dev)
    with torch.cuda.device(dev):
        for attempt in range(3):
            nbytes = (int(hint * 1.25) + (32 << 20) - 1) // (32 << 20) * (32 << 20)
            arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            d.arena, d.arena_bytes = arena.data_ptr(), nbytes
            rc = lib.ws_pyramid_build(nws, sws, _C.byref(d), _lib.current_stream())
            if rc == 5 and d.needed_bytes > nbytes:            # WS_ERR_CAPACITY: the sizes are known now
                hint = d.needed_bytes
                arena = None
                continue
            _lib.check(rc)
            break
        else:
            raise _lib.WeasalHipError("ws_pyramid_build: the arena stayed too small after three attempts")
    _arena_hint[key] = int(d.needed_bytes)

    def view(off, nbytes, dtype, shape):
        return arena[off:off + nbytes].view(dtype).view(shape)

    n = [int(d.n[l]) for l in range(L)]
    points = [P0] + [view(d.off_points[l], n[l] * 12, torch.float32, (n[l], 3)) for l in range(1, L)]
    lens_all = view(d.off_lens, L * B * 4, torch.int32, (L, B))
    lens_host = [np.array(d.lens[l][:B], dtype=np.int32) for l in range(L)]
    slots = view(d.off_slots, 4 * L * 4, torch.int32, (4 * L,))
    empty_i = lambda: torch.zeros((0, 1), dtype=torch.int64, device=dev)
    cap_of = lambda width: int(lib.ws_radius_neighbors_async_cap(int(width)))

    def finish(l, kind, off, rows, q, s, ql, sl, radius):
        """the matrix of search (l, kind) as the reference's crop leaves it: trimmed to the true width when that is smaller
        than the limit; redone with the two-call protocol in the rare case of a row beyond the asynchronous search's slab"""
        width, mc, fw = int(d.width[3 * l + kind]), int(d.max_count[3 * l + kind]), int(d.final_width[3 * l + kind])
        if mc == 0:
            raise _lib.WeasalHipError("libweasal_hip status 4: Error")
        if fw > 0:                                     # cropped in place to the widest row where that is below the limit
            mat = view(off, rows * fw * 8, torch.int64, (rows, fw))
        else:                                          # a row beyond the asynchronous search's slab
            ops.widen_async_slabs(mc)
            mat = ops.radius_neighbors(q, s, ql, sl, radius, limit=width, dtype=torch.int64)
        if search_radii is not None:
            search_radii.append((mat, float(radius)))
        return mat, mc

    neighbors, pools, upsamples = [], [], []
    orders = [] if point_orders is None else point_orders
    for l, lv in enumerate(levels):
        if lv["conv_on"]:
            mat, mc = finish(l, 0, d.off_neighbors[l], n[l], points[l], points[l], lens_host[l], lens_host[l], lv["r_conv"])
            neighbors.append(mat)
            orders.append((points[l], view(d.off_order[l], n[l] * 4, torch.int32, (n[l],))))
            if d.want_grids and int(d.final_width[3 * l]) > 0:
                grid = ops.SearchGrid()
                grid.blob = view(d.off_blob[l], int(d.blob_bytes[l]), torch.uint8, (int(d.blob_bytes[l]),))
                grid.nb, grid.cells, grid.ns = B, int(d.grid_cells[l]), n[l]
                grid.key_last = view(d.off_key_last[l], n[l] * 8, torch.int64, (n[l],))
                grid.radius = float(np.float32(lv["r_conv"]))
                grid.overflow = slots[3 * L + l:3 * L + l + 1]
                grid.max_count, grid.cap = mc, cap_of(int(d.width[3 * l]))
                search_grids.append((mat, grid))
        else:
            neighbors.append(empty_i())
        if lv["pool_on"]:
            mat, _ = finish(l, 1, d.off_pools[l], n[l + 1], points[l + 1], points[l], lens_host[l + 1], lens_host[l], lv["r_pool"])
            pools.append(mat)
            mat, _ = finish(l, 2, d.off_upsamples[l], n[l], points[l], points[l + 1], lens_host[l], lens_host[l + 1], lv["r_up"])
            upsamples.append(mat)
        else:
            pools.append(empty_i())
            upsamples.append(empty_i())
    if tables is not None:
        full, col0 = [], []

        def table(l, kind, nq, h, ns):
            o = int(d.off_toffsets[3 * l + kind])
            if o < 0:
                return None
            return ops.TransposedTable.from_parts(view(o, (ns + 2) * 4, torch.int32, (ns + 2,)),
                                                  view(int(d.off_tpairs[3 * l + kind]), max(nq * h, 1) * 4, torch.int32, (max(nq * h, 1),)),
                                                  nq, h, ns)
        for l, lv in enumerate(levels):
            if lv["conv_on"] and int(d.final_width[3 * l]) > 0:
                tb = table(l, 0, n[l], int(d.final_width[3 * l]), n[l])
                if tb is not None:
                    full.append((neighbors[l], n[l], tb))
            if lv["pool_on"]:
                if int(d.final_width[3 * l + 1]) > 0:
                    tb = table(l, 1, n[l + 1], int(d.final_width[3 * l + 1]), n[l])
                    if tb is not None:
                        full.append((pools[l], n[l], tb))
                if int(d.final_width[3 * l + 2]) > 0:
                    tb = table(l, 2, n[l], 1, n[l + 1])
                    if tb is not None:
                        col0.append((upsamples[l], n[l + 1], tb))
        tables["full"], tables["col0"] = full, col0
    _last_host_lengths.value = lens_host
    return points + neighbors + pools + upsamples + [lens_all[l] for l in range(L)] + [stacked_features, labels]


class _TLS(__import__("threading").local):
    value = None


_last_host_lengths = _TLS()    # host copy of the per-level lengths of the pyramid this thread built last (build_batch picks it up)


ACTIVATE_STALLS = None       # set to a list to collect (event, event) pairs around the hand-over wait (bench.py --stall-diag)


class PyramidBatch:
    """The `batch` object the blocks index (``.points/.neighbors/.pools/.upsamples/.lengths/
    .features/.labels``), built from the flat list like the reference's CustomBatch classes
    (datasets/DALES_PseudoLabel.py:1386-1460; a list that also carries the five bookkeeping arrays
    scales, rots, cloud_inds, center_inds, input_inds of :1410-1421 is accepted too)."""

    def __init__(self, input_list, point_orders=()):
        extra = 7 if (len(input_list) - 7) % 5 == 0 and (len(input_list) - 2) % 5 != 0 else 2
        L = (len(input_list) - extra) // 5
        self.points = list(input_list[0:L])
        self.neighbors = list(input_list[L:2 * L])
        self.pools = list(input_list[2 * L:3 * L])
        self.upsamples = list(input_list[3 * L:4 * L])
        self.lengths = list(input_list[4 * L:5 * L])
        self.features = input_list[5 * L]
        self.labels = input_list[5 * L + 1]
        if extra == 7:
            self.scales, self.rots, self.cloud_inds, self.center_inds, self.input_inds = input_list[5 * L + 2:5 * L + 7]
        self.point_orders = list(point_orders)   # [(points tensor, cell-order permutation)]: scheduling hints
        self.lengths_host = None                 # per-level lengths as host arrays (set by build_batch)
        self.search_grids = []                   # [(index matrix, ops.SearchGrid)]: table-free backward of self-query layers
        self.search_radii = []                   # [(index matrix, radius of the search that wrote it)]: rows sorted by distance
        self.tables = []                         # pre-built transposed tables [(inds, ns, table)] (build_tables)
        self.col0_tables = []
        self.ready = None                        # event recorded on the stream that built the batch

    def _tensors(self):
        for name in ("points", "neighbors", "pools", "upsamples", "lengths"):
            yield from getattr(self, name)
        yield self.features
        yield self.labels
        for _, o in self.point_orders:
            yield o
        for _, _, tb in self.tables + self.col0_tables:
            yield tb.offsets
            yield tb.pairs
        for _, grid in self.search_grids:
            for t in grid.tensors():
                yield t

    def build_tables(self):
        """Transposed neighbour tables every backward of this batch needs (KPConv dX over neighbors[l] /
        pools[l], max_pool over pools[l], closest_pool over column 0 of upsamples[l]); built here -- on the
        stream that builds the batch -- they are off the training stream's critical path."""
        self.tables, self.col0_tables = [], []
        L = len(self.points)
        have_grid = {m.data_ptr() for m, _ in self.search_grids} if ops.GRID_BACKWARD else set()
        for l in range(L):
            ns = self.points[l].shape[0]
            for mat in (self.neighbors[l], self.pools[l]):
                if mat.data_ptr() in have_grid:
                    continue         # KPConv is the only reader of neighbors[l] and goes through the grid
                if mat.shape[0] > 0 and mat.is_cuda:
                    self.tables.append((mat, ns, ops.TransposedTable(mat, ns)))
            up = self.upsamples[l]
            if up.shape[0] > 0 and up.is_cuda and l + 1 < L:
                nc = self.points[l + 1].shape[0]
                self.col0_tables.append((up, nc, ops.TransposedTable(up[:, :1].contiguous(), nc)))
        return self

    def _map(self, fn):
        for name in ("points", "neighbors", "pools", "upsamples", "lengths"):
            setattr(self, name, [fn(t) for t in getattr(self, name)])
        self.features = fn(self.features)
        self.labels = fn(self.labels)
        for name in ("scales", "rots", "cloud_inds", "center_inds", "input_inds"):
            if hasattr(self, name):
                setattr(self, name, fn(getattr(self, name)))
        return self

    def to(self, device):
        return self._map(lambda t: t.to(device))

    def pin_memory(self):
        return self._map(lambda t: t.pin_memory())

    def activate(self, stream=None):
        """Make the batch usable on `stream` (default: the current one): wait for the stream that built
        it, tell the caching allocator about the second user, install the scheduling orders.  Called by
        KPFCNN.forward; a no-op for batches built on the same stream."""
        if self.features.is_cuda:
            stream = stream or torch.cuda.current_stream(self.features.device)
            if self.ready is not None:
                if ACTIVATE_STALLS is not None:            # diagnostics: how long the consumer stream waits for the builder's stream
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    stream.wait_event(self.ready)
                    e1.record(stream)
                    ACTIVATE_STALLS.append((e0, e1))
                else:
                    stream.wait_event(self.ready)
                for t in self._tensors():
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(stream)
                self.ready = None
            ops.set_point_orders(self.point_orders)
            by_points = {p.data_ptr(): o for p, o in self.point_orders}
            ops.set_pool_orders([(self.pools[l], by_points.get(self.points[l + 1].data_ptr()), by_points.get(self.points[l].data_ptr()))
                                 for l in range(len(self.points) - 1)])
            ops.set_sorted_rows(self.search_radii)     # rows as the radius search wrote them: sorted by distance
            ops.set_search_grids(self.search_grids)
            ops.clear_table_cache()              # tables belong to one batch
            ops.install_tables(self.tables, self.col0_tables)
        return self


def build_batch(config, points, features, labels, lengths, neighborhood_limits=(), random_grid_orient=True,
                with_tables=True, rng=None, for_training=True):
    """for_training=False (forward-only use: the testers' pass): no transposed tables, no exported search grids / key_last --
    nothing a backward would need"""
    if not for_training:
        with_tables = False
    orders, grids, radii = [], [], []
    lens_np = np.asarray(lengths.cpu() if isinstance(lengths, torch.Tensor) else lengths, dtype=np.int32)
    native = native_eligible(config, points, lens_np, neighborhood_limits)
    tables = {} if (native and with_tables) else None
    if native:
        li = segmentation_inputs_native(config, points, features, labels, lens_np, neighborhood_limits, random_grid_orient,
                                        point_orders=orders, search_grids=grids if for_training else None, rng=rng,
                                        search_radii=radii, tables=tables)
    else:
        li = segmentation_inputs(config, points, features, labels, lens_np, neighborhood_limits, random_grid_orient,
                                 point_orders=orders, search_grids=grids if (points.is_cuda and for_training) else None, rng=rng,
                                 search_radii=radii)
    batch = PyramidBatch(li, orders)
    batch.search_grids = grids
    batch.search_radii = radii
    batch.lengths_host = _last_host_lengths.value        # numpy int32 [B] per level: host-side consumers need no read-back
    _last_host_lengths.value = None
    if tables is not None:
        batch.tables, batch.col0_tables = tables["full"], tables["col0"]
    elif with_tables and points.is_cuda:
        batch.build_tables()
    if points.is_cuda:
        batch.ready = torch.cuda.Event()
        batch.ready.record(torch.cuda.current_stream(points.device))
    return batch
