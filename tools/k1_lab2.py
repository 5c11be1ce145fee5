"""Diagnostic: the 13 K1 fill launches of one DALES pyramid, each timed alone (ws_radius_neighbors_fill from an existing plan:
the fill kernel without the grid build), for a few launch shapes (lab switches ws_nb_max_blocks / ws_nb_queries_per_block)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import _lib, config as wcfg, ops, pyramid, synthetic
from weasal_amd.ops import ptr, current_stream, check
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "dales"]
cfg = getattr(wcfg, wl["config"])()
pts, feats, labels, lens = synthetic.make_inputs(1, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
np.random.seed(0)
batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev), lens, wl["limits"])
lib = _lib.lib()
L = len(batch.points)
sched = pyramid._schedule(cfg, wl["limits"])          # the radii the pyramid really searches with (deformable levels: 2 r)
searches = []
for l in range(L):
    P = batch.points[l]; ll = batch.lengths[l].cpu().numpy().astype(np.int32)
    searches.append(("conv%d" % l, P, P, ll, ll, sched[l]["r_conv"], wl["limits"][l]))
    if l + 1 < L:
        Q = batch.points[l + 1]; ql = batch.lengths[l + 1].cpu().numpy().astype(np.int32)
        searches.append(("pool%d" % l, Q, P, ql, ll, sched[l]["r_pool"], wl["limits"][l]))
        searches.append(("up%d" % l, P, Q, ll, ql, sched[l]["r_up"], wl["limits"][l + 1]))
def sw(name, v):
    C.c_int.in_dll(lib, name).value = v
ws = ops._ws.neighbors(dev)
slot = torch.zeros(4, dtype=torch.int32, device=dev)
def fill_time(q, s, ql, sl, r, width, rep=6):
    out = torch.empty((q.shape[0], width), dtype=torch.int64, device=dev)
    check(lib.ws_radius_neighbors_search_async(ws, ptr(q), q.shape[0], ptr(s), s.shape[0], C.c_void_p(ql.ctypes.data), C.c_void_p(sl.ctypes.data),
                                               ql.shape[0], float(np.float32(r)), width, None, ptr(out), ptr(slot), current_stream()))
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rep):
            check(lib.ws_radius_neighbors_fill(ws, width, None, ptr(out), current_stream()))
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / rep)
    return best * 1e3, int(slot[0])
variants = [("default", 0, 0), ("cap 1024", 1024, 0), ("cap 2048", 2048, 0), ("cap 8192", 8192, 0), ("cap 16384", 16384, 0),
            ("8 q/wg cap 8192", 8192, 8), ("16 q/wg", 0, 16)]
print("%-7s %8s %8s %5s %6s | " % ("search", "nq", "ns", "width", "maxcnt") + " | ".join("%-15s" % v[0] for v in variants))
tot = [0.0] * len(variants)
for name, q, s, ql, sl, r, width in searches:
    row = []
    for i, (vn, cap, qpb) in enumerate(variants):
        sw("ws_nb_max_blocks", cap); sw("ws_nb_queries_per_block", qpb)
        t, mc = fill_time(q, s, ql, sl, r, width)
        row.append(t); tot[i] += t
    print("%-7s %8d %8d %5d %6d | " % (name, q.shape[0], s.shape[0], width, mc) + " | ".join("%12.1f us" % t for t in row), flush=True)
print("%-38s | " % "sum" + " | ".join("%12.1f us" % t for t in tot))
