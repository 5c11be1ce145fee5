import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from weasal_amd import synthetic, config as wcfg
from weasal_amd.cpp_wrappers.cpp_neighbors import radius_neighbors as rn
from weasal_amd.cpp_wrappers.cpp_subsampling import grid_subsampling as gs
cfg = wcfg.DALESPLConfig()
pts, feats, labels, lens = synthetic.make_inputs(0, 8, 50000, 10.0, 3)
lens = lens.astype(np.int32)
r = cfg.first_subsampling_dl * cfg.conv_radius
for _ in range(2):
    out = rn.batch_query(pts, pts, lens, lens, radius=r)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 5
for _ in range(n):
    out = rn.batch_query(pts, pts, lens, lens, radius=r)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("batch_query host->host: %.1f ms per call, %d x %d int32 out (%.1f MB), %.2f M queries/s" % (1e3 * (t1 - t0) / n, out.shape[0], out.shape[1], out.nbytes / 1e6, pts.shape[0] * n / (t1 - t0) / 1e6))
for _ in range(2):
    s = gs.subsample_batch(pts, lens, sampleDl=0.8, max_p=0)
t0 = time.perf_counter()
for _ in range(n):
    s = gs.subsample_batch(pts, lens, sampleDl=0.8, max_p=0)
t1 = time.perf_counter()
print("subsample_batch host->host: %.1f ms per call (%d -> %d points)" % (1e3 * (t1 - t0) / n, pts.shape[0], s[0].shape[0]))
