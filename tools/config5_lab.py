#!/usr/bin/env python3
"""tools/config5_lab.py [level] [reps] -- isolated timings (HIP events, one stream, nothing else on the GPU) of the gather kernels
of BASELINE config 5 on one full-size batch (8 x 50 000 points, limits 422 / 519 / 472 / 193 / 34, bf16 rows):
K3 rigid / deformable, K4 through the grid and through the transposed table (rigid / deformable), K6, and the searches."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import config as wcfg, ops, pyramid, synthetic  # noqa: E402
from weasal_amd.kernel_points import load_kernels  # noqa: E402

lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dt = torch.float32 if os.environ.get("LAB_F32") else torch.bfloat16
gpu = torch.device("cuda:0")
cfg = wcfg.DALESDeformConfig()
wl = synthetic.WORKLOADS["dales_deform"]
pts, feats, labels, lens = synthetic.make_inputs(7, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
P0 = torch.from_numpy(pts).to(gpu)
np.random.seed(3)


def timed(name, fn, n=reps):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print("%-58s %8.3f ms (min %.3f)" % (name, ms[len(ms) // 2], ms[0]), flush=True)
    return ms[len(ms) // 2]


t0 = time.perf_counter()
batch = pyramid.build_batch(cfg, P0, torch.from_numpy(feats).to(gpu), torch.from_numpy(labels).to(gpu), lens, wl["limits"])
torch.cuda.synchronize()
print("pyramid (first build, with tables for the strided layers): %.1f ms" % (1e3 * (time.perf_counter() - t0)))
timed("pyramid build_batch (whole, synchronous)", lambda: pyramid.build_batch(cfg, P0, torch.from_numpy(feats).to(gpu),
                                                                             torch.from_numpy(labels).to(gpu), lens, wl["limits"]), 3)
batch.activate()
ci = 32 * 2 ** lvl
r = cfg.first_subsampling_dl * cfg.conv_radius * 2 ** lvl
extent = r * cfg.KP_extent / cfg.conv_radius
P, inds = batch.points[lvl], batch.neighbors[lvl]
n, h = inds.shape
print("level %d: N = %d, H = %d, Ci = %d, rows %s; grid max_count %s" % (lvl, n, h, ci, dt, getattr(ops._grid_for(inds), "max_count", None)))
kp = torch.from_numpy(load_kernels(r, 15, dimension=3, fixed="center").astype(np.float32)).to(gpu)
torch.manual_seed(0)
x = torch.randn(n, ci, device=gpu).to(dt)
off = 0.3 * torch.randn(n, 60, device=gpu)
kp4, dkp, mod, rmax = ops.deform_prepare(off, kp, extent, True)
kp4 = kp4.detach()
lens_l = batch.lengths[lvl].cpu().numpy()
rad = r * cfg.deform_radius / cfg.conv_radius

# ---- searches
timed("K1 self search (limit %d)" % h, lambda: ops.radius_neighbors(P, P, lens_l, lens_l, rad, limit=h))
if lvl + 1 < len(batch.points):
    Pn = batch.points[lvl + 1]
    ln = batch.lengths[lvl + 1].cpu().numpy()
    timed("K1 pool search (coarse queries, limit %d)" % h, lambda: ops.radius_neighbors(Pn, P, ln, lens_l, rad, limit=h))
    hu = wl["limits"][lvl + 1]
    timed("K1 upsample search (fine queries, coarse supports, limit %d)" % hu,
          lambda: ops.radius_neighbors(P, Pn, lens_l, ln, 2 * rad, limit=hu))

# ---- forward
xr = x.clone().requires_grad_(True)
wf_r = [None]


def k3_rigid():
    wf_r[0], _ = ops.kpconv_gather(xr, P, P, inds, kp, extent)


def k3_rigid_cut():
    wf_r[0], _ = ops.kpconv_gather(xr, P, P, inds, kp, extent, rows_sorted=True)


timed("K3 rigid forward", k3_rigid)
timed("K3 rigid forward, sorted-row cutoff", k3_rigid_cut)
xd = x.clone().requires_grad_(True)
kq = kp4.clone().requires_grad_(True)
wf_d = [None, None]


def k3_def():
    wf_d[0], wf_d[1] = ops.kpconv_gather_def(xd, kq, P, P, inds, extent)


def k3_def_cut():
    wf_d[0], wf_d[1] = ops.kpconv_gather_def(xd, kq, P, P, inds, extent, rows_sorted=True)


timed("K3 deformable forward (MODE 2)", k3_def)
timed("K3 deformable forward (MODE 2), sorted-row cutoff", k3_def_cut)
g = torch.randn(n, 15, ci, device=gpu).to(dt)
lib = ops._lib.lib()
from weasal_amd._lib import check, current_stream, ptr  # noqa: E402
grid = ops._grid_for(inds)
dx = torch.empty_like(x)
bf = 1 if dt == torch.bfloat16 else 0
order = ops._order_for(P)


def k4g(kq_, rows=True, rm=None):
    check(lib.ws_kpconv_gather_bwd_x_grid_wide(ptr(P), n, ptr(grid.blob), grid.nb, grid.cells, ptr(grid.key_last), grid.radius, ptr(g), ci,
                                               ptr(kp), 15, ptr(kq_), ptr(rm), extent, ptr(order), ptr(inds) if rows else None, h, ptr(dx),
                                               bf, current_stream()))


if grid is not None:
    timed("K4G wide rigid (rows + walk)", lambda: k4g(None))
    timed("K4G wide rigid (walk only)", lambda: k4g(None, False), 2)
    timed("K4G wide deformable (rows + walk), no reach bound", lambda: k4g(kp4))
    timed("K4G wide deformable (rows + walk), reach bound %.2f" % float(rmax), lambda: k4g(kp4, True, rmax))
t0 = time.perf_counter()
table = ops.TransposedTable(inds, n)
torch.cuda.synchronize()
print("transposed table build: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
timed("transposed table build", lambda: ops.TransposedTable(inds, n), 3)
f_tab = lib.ws_kpconv_gather_bwd_x_bf16 if bf else lib.ws_kpconv_gather_bwd_x
timed("K4 table rigid", lambda: check(f_tab(ptr(P), n, ptr(P), n, ptr(inds), h, ptr(table.offsets), ptr(table.pairs), ptr(g), ci, ptr(kp), 15,
                                            None, None, extent, 0, 0, ptr(order), ptr(dx), current_stream())))
timed("K4 table deformable (MODE 2)", lambda: check(lib.ws_kpconv_gather_bwd_x_def(ptr(P), n, ptr(P), n, h, ptr(table.offsets), ptr(table.pairs),
                                                                                     ptr(g), ci, ptr(kp4), 15, extent, ptr(order), ptr(dx), bf,
                                                                                     current_stream())))
d_kp4 = torch.empty_like(kp4)
dmin = torch.randn(n, 15, device=gpu)
timed("K6 geometry backward (matrix core)", lambda: check(lib.ws_kpconv_gather_bwd_geom_def(ptr(P), n, ptr(P), n, ptr(inds), h, ptr(x), ci, ptr(g),
                                                                                          ptr(kp4), 15, ptr(dmin), extent, ptr(order), ptr(d_kp4),
                                                                                          bf, 0, current_stream())))
timed("K6 geometry backward, sorted-row cutoff", lambda: check(lib.ws_kpconv_gather_bwd_geom_def(ptr(P), n, ptr(P), n, ptr(inds), h, ptr(x), ci,
                                                                                               ptr(g), ptr(kp4), 15, ptr(dmin), extent,
                                                                                               ptr(order), ptr(d_kp4), bf, 1, current_stream())))
