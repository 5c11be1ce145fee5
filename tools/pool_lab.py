"""Diagnostic: the strided block's max-pool at DALES level 0 (71 000 pooled points x 59 neighbours x 64 channels), forward and
backward, for the work assignments of csrc/pools.hip (ws_pool_interleave: 0 = contiguous chunks, n = n workgroups per XCD)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import _lib, config as wcfg, ops, pyramid, synthetic
from weasal_amd._lib import ptr, current_stream, check
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS["dales"]; cfg = wcfg.DALESPLConfig()
pts, feats, labels, lens = synthetic.make_inputs(1, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
np.random.seed(0)
b = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev), lens, wl["limits"])
lib = _lib.lib()
vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
lib.ws_priv_max_pool_fwd_u8.argtypes = [vp, i64, i32, vp, i64, i32, vp, vp, vp, vp]; lib.ws_priv_max_pool_fwd_u8.restype = C.c_int
lib.ws_priv_max_pool_bwd_u8.argtypes = [vp, vp, i64, i32, i32, vp, vp, i64, vp, vp, vp, vp]; lib.ws_priv_max_pool_bwd_u8.restype = C.c_int
orders = {p.data_ptr(): o for p, o in b.point_orders}
for lvl in (0, 1):
    c = 128 << lvl          # in_dim of the strided blocks: 128 at level 0, 256 at level 1
    inds = b.pools[lvl]; ns = b.points[lvl].shape[0]; nq, h = inds.shape
    oq, osup = orders.get(b.points[lvl + 1].data_ptr()), orders.get(b.points[lvl].data_ptr())
    x = torch.randn(ns, c, device=dev); dy = torch.randn(nq, c, device=dev)
    out = torch.empty(nq, c, device=dev); arg = torch.empty(nq, c, dtype=torch.uint8, device=dev); dx = torch.empty(ns, c, device=dev)
    tb = ops.TransposedTable(inds, ns)
    def fwd(): check(lib.ws_priv_max_pool_fwd_u8(ptr(x), ns, c, ptr(inds), nq, h, ptr(out), ptr(arg), ptr(oq), current_stream()))
    def bwd(): check(lib.ws_priv_max_pool_bwd_u8(ptr(dy), ptr(arg), nq, h, c, ptr(tb.offsets), ptr(tb.pairs), ns, ptr(dx), ptr(osup), None, current_stream()))
    def t(fn, rep=10):
        fn(); torch.cuda.synchronize(); best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(rep): fn()
            e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / rep)
        return best * 1e3
    ref_out = ref_dx = None
    for ilv, unroll in ((0, 4), (256, 4), (256, 8)):
        C.c_int.in_dll(lib, "ws_pool_interleave").value = ilv
        C.c_int.in_dll(lib, "ws_pool_unroll").value = unroll
        tf, tbw = t(fwd), t(bwd)
        if ref_out is None: ref_out, ref_dx = out.clone(), dx.clone()
        print("level %d (nq %d, h %d, c %d) interleave %3d unroll %d: fwd %6.1f us  bwd %6.1f us  same results: %s" % (lvl, nq, h, c, ilv, unroll, tf, tbw, torch.equal(out, ref_out) and torch.equal(dx, ref_dx)), flush=True)
