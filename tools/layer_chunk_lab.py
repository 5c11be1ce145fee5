"""Diagnostic: the whole enc1 KPConv layer (K3 gather -> wf, then the contraction wf [N, 480] x W [480, 32]) launched over
all 400 000 queries at once and in row chunks, so that the contraction reads its chunk of wf while the 256 MB memory-side
cache still holds it.  GPU time by events behind a spin kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import _lib, config as wcfg, ops, pyramid, synthetic
from weasal_amd._lib import check, current_stream, ptr
dev = torch.device("cuda:0")
lib = _lib.lib()
wl = synthetic.WORKLOADS["dales"]
cfg = wcfg.DALESPLConfig()
pts, feats, labels, lens = synthetic.make_inputs(1, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
np.random.seed(0)
batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev), lens, wl["limits"])
P, inds = batch.points[0], batch.neighbors[0]
n, h, ci, co = P.shape[0], inds.shape[1], 32, 32
r = cfg.first_subsampling_dl * cfg.conv_radius
extent = r * cfg.KP_extent / cfg.conv_radius
kp = torch.from_numpy(np.load(os.path.join(os.path.dirname(ops.__file__), "data", "k_015_center_3D.npy")).astype(np.float32)).to(dev) * r
x = torch.randn(n, ci, device=dev)
w = torch.randn(15 * ci, co, device=dev) / (15 * ci) ** 0.5
wf = torch.empty(n, 15 * ci, device=dev)
out = torch.empty(n, co, device=dev)
st = current_stream()
def layer(chunks):
    per = -(-n // chunks)
    for c in range(chunks):
        a, b = c * per, min(n, (c + 1) * per)
        m = b - a
        check(lib.ws_kpconv_gather_fwd(P.data_ptr() + 12 * a, m, ptr(P), n, inds.data_ptr() + 8 * h * a, h, ptr(x), ci, ptr(kp), 15, None, None,
                                       extent, 0, 0, None, wf.data_ptr() + 4 * 15 * ci * a, None, st))
        check(lib.ws_gemm_xb_epilogue(wf.data_ptr() + 4 * 15 * ci * a, m, 15 * ci, 15 * ci, ptr(w), co, None, None, 0, 1, 0.1,
                                      out.data_ptr() + 4 * co * a, co, st))
def timeit(fn, rep=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(3_000_000)
        e0.record()
        for _ in range(rep): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / rep)
    return best
B = 4.0 * n * (h * (8 + 12 + 4 * ci) / 4.0) + 4.0 * n * (3 + co) + 4.0 * 15 * ci * co        # SURVEY 8d B_fwd
B = n * (h * (8 + 12 + 4 * ci) + 12 + 4 * co) + 4 * 15 * ci * co
layer(1); ref = out.clone()
for chunks in (1, 2, 4, 8, 16, 32):
    t = timeit(lambda: layer(chunks))
    layer(chunks)
    print("chunks %2d: %.4f ms  -> %.2f TB/s logical = %.3f of 8 TB/s   (same result: %s)" % (chunks, t, B / t / 1e9, B / t / 1e9 / 8.0, torch.equal(out, ref)), flush=True)
