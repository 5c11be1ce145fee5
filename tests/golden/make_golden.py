#!/usr/bin/env python3
"""tests/golden/make_golden.py -- generates the golden vectors in this directory.

RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference and oracle/_ref/libws_ref.so);
the GPU box only ever sees the *.npz outputs.  Nothing here is copied from the reference:
the script imports the reference's Python modules in place and calls the reference's own
compiled C++ core (oracle/_ref, built by oracle/Makefile from the reference sources).

What is produced:
  g1_neighbors.npz   batch_query, self + cross sets, 2 radii        (neighbors.cpp:211-332)
  g2_subsample.npz   subsample_batch points/lens, 2 cell sizes, max_p (grid_subsampling.cpp:109-211)
  g3_subsample_fl.npz subsample / subsample_batch with features + labels (grid_subsampling.cpp:5-106)
  g4_kpconv_*.npz    rigid KPConv forward + grads                    (models/blocks.py:238-374)
  g5_kpconv_deform*.npz deformable / modulated KPConv + regulariser  (blocks.py:244-325; architectures.py:24-57)
  g6_pools.npz       max_pool / closest_pool / global_average        (blocks.py:80-134)
  g7_pyramid.npz     PointCloudDataset.segmentation_inputs           (datasets/common.py:461-577)
  g8_kpfcnn.npz      KPFCNN logits, loss, grads, one SGD step        (architectures.py:192-403; trainer_PseudoLabel.py:80-87,199-219)
  g9_kernel_points.npz the 15x3 kernel disposition table             (kernels/dispositions/k_015_center_3D.ply)

Import aids (all local to this script, none shipped):
  * `datasets` is registered as a namespace package pointing at the reference (the HF
    `datasets` wheel in site-packages shadows it otherwise);
  * `cpp_wrappers.cpp_neighbors.radius_neighbors` / `cpp_wrappers.cpp_subsampling.grid_subsampling`
    are bound to oracle/_ref (the reference's CPython glue does not compile against NumPy 2);
  * `torch_scatter` (not installed) gets an empty placeholder so that models/architectures.py:20
    imports; KPFCNN.contrast_loss (the only user, :501) is never called -> parity unpinned there.
Neighbour fixtures are regenerated until tie-free (equal d2 inside a row makes the reference's
std::sort order implementation-defined).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
os.chdir(REF)  # kernels/kernel_points.py:410 uses a cwd-relative path

import numpy as np
import torch

from oracle import geom

torch.set_num_threads(4)

# ---- import aids ---------------------------------------------------------------------------
ds = types.ModuleType("datasets")
ds.__path__ = [os.path.join(REF, "datasets")]
sys.modules["datasets"] = ds
sys.modules["torch_scatter"] = types.ModuleType("torch_scatter")
sys.modules["torch_scatter"].scatter = None

cw = types.ModuleType("cpp_wrappers"); cw.__path__ = []
cn = types.ModuleType("cpp_wrappers.cpp_neighbors"); cn.__path__ = []
cs = types.ModuleType("cpp_wrappers.cpp_subsampling"); cs.__path__ = []
rn = types.ModuleType("cpp_wrappers.cpp_neighbors.radius_neighbors")
gs = types.ModuleType("cpp_wrappers.cpp_subsampling.grid_subsampling")
rn.batch_query = lambda q, s, qb, sb, radius=0.1: geom.batch_query(q, s, qb, sb, radius, kind="ref")


def _sub_batch(points, batches, features=None, classes=None, sampleDl=0.1, method="barycenters",
               max_p=0, verbose=0):
    return geom.subsample_batch(points, batches, features=features, classes=classes,
                                sampleDl=sampleDl, max_p=max_p, kind="ref")


def _sub(points, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0):
    return geom.subsample(points, features=features, classes=classes, sampleDl=sampleDl, kind="ref")


gs.subsample_batch = _sub_batch
gs.subsample = _sub
cn.radius_neighbors = rn
cs.grid_subsampling = gs
cw.cpp_neighbors = cn
cw.cpp_subsampling = cs
for m in (cw, cn, cs, rn, gs):
    sys.modules[m.__name__] = m

from models.blocks import KPConv, max_pool, closest_pool, global_average  # noqa: E402
from models.architectures import KPFCNN, p2p_fitting_regularizer  # noqa: E402
from datasets.common import PointCloudDataset  # noqa: E402
from utils.config import Config  # noqa: E402
from utils.ply import read_ply  # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f kB" % (name, os.path.getsize(path) / 1e3))


def sphere(rng, n, R, center=(0, 0, 0)):
    pts = np.zeros((0, 3), np.float32)
    while len(pts) < n:
        c = rng.uniform(-R, R, size=(2 * n, 3)).astype(np.float32)
        pts = np.concatenate([pts, c[(c.astype(np.float64) ** 2).sum(1) < R * R]])
    return (pts[:n] + np.asarray(center, np.float32)).astype(np.float32)


def tie_free(q, s, qb, sb, r):
    a = geom.batch_query(q, s, qb, sb, r, kind="ref")
    b = geom.batch_query(q, s, qb, sb, r, kind="port")
    return np.array_equal(a, b), a


# ---- G1 / G2 / G3 : native geometry ---------------------------------------------------------
def make_geometry():
    # self / pool index matrices are regenerated until tie-free; the long `up` rows (2r) almost
    # always contain an equal-d2 pair, so they are stored with a tiefree flag and compared
    # tie-aware (same d2 row, same index multiset) when the flag is 0.
    seed = 100
    while True:
        rng = np.random.default_rng(seed)
        pts = np.concatenate([sphere(rng, 420, 2.0), sphere(rng, 380, 2.0, (0.31, -0.17, 0.05))])
        lens = np.array([420, 380], np.int32)
        sub_p, sub_l = geom.subsample_batch(pts, lens, sampleDl=0.5, kind="ref")
        oks = []
        out = {}
        for r in (0.6, 1.0):
            ok, a = tie_free(pts, pts, lens, lens, r); oks.append(ok); out["self_r%.1f" % r] = a
            out["tiefree_self_r%.1f" % r] = np.int32(ok)
            ok, a = tie_free(sub_p, pts, sub_l, lens, r); oks.append(ok); out["pool_r%.1f" % r] = a
            out["tiefree_pool_r%.1f" % r] = np.int32(ok)
            ok, a = tie_free(pts, sub_p, lens, sub_l, 2 * r); out["up_r%.1f" % (2 * r)] = a
            out["tiefree_up_r%.1f" % (2 * r)] = np.int32(ok)
        if all(oks) or seed >= 160:
            break
        seed += 1
    print("geometry seed", seed)
    save("g1_neighbors.npz", points=pts, lens=lens, sub_points=sub_p, sub_lens=sub_l,
         radii=np.array([0.6, 1.0], np.float32), **out)

    g2 = {"points": pts, "lens": lens}
    for dl in (0.3, 0.5, 0.9):
        p, l = geom.subsample_batch(pts, lens, sampleDl=dl, kind="ref")
        g2["p_dl%.1f" % dl] = p; g2["l_dl%.1f" % dl] = l
    p, l = geom.subsample_batch(pts, lens, sampleDl=0.3, max_p=50, kind="ref")
    g2["p_dl0.3_maxp50"] = p; g2["l_dl0.3_maxp50"] = l
    # a cloud with negative/positive offsets and a degenerate one-point element
    pts2 = np.concatenate([pts[:100] * np.float32(7.3) - np.float32(40.0), pts[500:501]])
    lens2 = np.array([100, 1], np.int32)
    p, l = geom.subsample_batch(pts2, lens2, sampleDl=1.7, kind="ref")
    g2["points2"] = pts2; g2["lens2"] = lens2; g2["p2_dl1.7"] = p; g2["l2_dl1.7"] = l
    save("g2_subsample.npz", **g2)

    rng = np.random.default_rng(7)
    feats = rng.normal(size=(800, 5)).astype(np.float32)
    # labels with a dominant class per region so that histogram ties are rare; ties are
    # removed below because the reference's arg-max over unordered_map<int,int> is
    # implementation-defined on ties (grid_subsampling.cpp:99-101)
    labels = (np.floor((pts[:, 2] + 2.5) * 1.5).astype(np.int32) + (rng.random(800) < 0.1)).astype(np.int32)
    p, l, f, c = geom.subsample_batch(pts, lens, features=feats, classes=labels, sampleDl=0.5, kind="ref")
    p1, f1, c1 = geom.subsample(pts[:420], features=feats[:420], classes=labels[:420], sampleDl=0.5, kind="ref")
    pf = geom.subsample(pts[:420], features=feats[:420], sampleDl=0.5, kind="ref")
    pc = geom.subsample(pts[:420], classes=labels[:420], sampleDl=0.5, kind="ref")
    save("g3_subsample_fl.npz", points=pts, lens=lens, features=feats, labels=labels,
         b_points=p, b_lens=l, b_features=f, b_labels=c,
         s_points=p1, s_features=f1, s_labels=c1, sf_points=pf[0], sf_features=pf[1],
         sc_points=pc[0], sc_labels=pc[1])
    return pts, lens


# ---- G4 / G5 : KPConv -----------------------------------------------------------------------
def kp_inputs(rng, n_per=110, R=1.2, r=0.6, limit=None, strided=False):
    pts = np.concatenate([sphere(rng, n_per, R), sphere(rng, n_per - 10, R, (0.1, 0.05, -0.2))])
    lens = np.array([n_per, n_per - 10], np.int32)
    if strided:
        q, ql = geom.subsample_batch(pts, lens, sampleDl=2 * r / 2.5, kind="ref")
    else:
        q, ql = pts, lens
    inds = geom.batch_query(q, pts, ql, lens, r, kind="ref").astype(np.int64)
    if limit is not None:
        inds = inds[:, :limit]
    return q, pts, inds


def run_kpconv(tag, ci, co, seed, influence="linear", mode="sum", deformable=False,
               modulated=False, strided=False, limit=None, r=0.6):
    rng = np.random.default_rng(seed)
    np.random.seed(seed)           # load_kernels noise + rotation (kernel_points.py:454,478)
    torch.manual_seed(seed)
    q, s, inds = kp_inputs(rng, limit=limit, strided=strided, r=r)
    ext = r * 1.0 / 2.5            # blocks.py:523
    conv = KPConv(15, 3, ci, co, ext, r, KP_influence=influence, aggregation_mode=mode,
                  deformable=deformable, modulated=modulated)
    if deformable:
        # offset_conv.weights are kaiming-initialised; scale them down so that offsets stay
        # inside the neighbourhood, and give the bias a small non-zero value
        with torch.no_grad():
            conv.offset_conv.weights.mul_(0.1)
            conv.offset_bias.copy_(torch.from_numpy(rng.normal(scale=0.05, size=conv.offset_bias.shape).astype(np.float32)))
    x = torch.from_numpy(rng.normal(size=(s.shape[0], ci)).astype(np.float32)).requires_grad_(True)
    dy = torch.from_numpy(rng.normal(size=(q.shape[0], co)).astype(np.float32))
    out = conv(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(inds), x)
    arrs = dict(q_pts=q, s_pts=s, inds=inds, x=x.detach().numpy(), dy=dy.numpy(),
                weights=conv.weights.detach().numpy(), kernel_points=conv.kernel_points.detach().numpy(),
                KP_extent=np.float32(ext), radius=np.float32(r), out=out.detach().numpy())
    loss = (out * dy).sum()
    if deformable:
        net = types.SimpleNamespace(modules=lambda: [conv], l1=torch.nn.L1Loss(), K=15,
                                    repulse_extent=1.2, deform_fitting_power=1.0)
        reg = p2p_fitting_regularizer(net)
        loss = loss + reg
        arrs.update(reg_loss=reg.detach().numpy(), offset_features=conv.offset_features.detach().numpy(),
                    deformed_KP=conv.deformed_KP.detach().numpy(), min_d2=conv.min_d2.detach().numpy(),
                    offset_weights=conv.offset_conv.weights.detach().numpy(),
                    offset_kernel_points=conv.offset_conv.kernel_points.detach().numpy(),
                    offset_bias=conv.offset_bias.detach().numpy())
    loss.backward()
    arrs.update(grad_x=x.grad.numpy(), grad_weights=conv.weights.grad.numpy())
    if deformable:
        arrs.update(grad_offset_weights=conv.offset_conv.weights.grad.numpy(),
                    grad_offset_bias=conv.offset_bias.grad.numpy())
    save(tag, **arrs)


def make_kpconv():
    run_kpconv("g4_kpconv_3_64.npz", 3, 64, 11, limit=24)
    run_kpconv("g4_kpconv_32_32.npz", 32, 32, 12, limit=30)
    run_kpconv("g4_kpconv_32_32_strided.npz", 32, 32, 13, strided=True)
    run_kpconv("g4_kpconv_64_64.npz", 64, 64, 14)
    run_kpconv("g4_kpconv_16_16_gaussian.npz", 16, 16, 15, influence="gaussian", limit=20)
    run_kpconv("g4_kpconv_16_16_constant.npz", 16, 16, 16, influence="constant", limit=20)
    run_kpconv("g4_kpconv_16_16_closest.npz", 16, 16, 17, mode="closest", limit=20)
    run_kpconv("g5_kpconv_deform_16_16.npz", 16, 16, 21, deformable=True, r=1.2, limit=40)
    run_kpconv("g5_kpconv_deform_mod_16_32.npz", 16, 32, 22, deformable=True, modulated=True, r=1.2, limit=40)
    run_kpconv("g5_kpconv_deform_strided_16_16.npz", 16, 16, 23, deformable=True, strided=True, r=1.2)


# ---- G6 : pools -----------------------------------------------------------------------------
def make_pools():
    rng = np.random.default_rng(31)
    q, s, inds = kp_inputs(rng, strided=True)
    _, _, up = None, None, geom.batch_query(s, q, np.array([110, 100], np.int32),
                                            geom.subsample_batch(s, np.array([110, 100], np.int32), sampleDl=0.48, kind="ref")[1],
                                            1.2, kind="ref").astype(np.int64)
    x = torch.from_numpy(rng.normal(size=(s.shape[0], 24)).astype(np.float32)).requires_grad_(True)
    # make some rows all-negative so that the zero shadow row wins the max (blocks.py:104)
    with torch.no_grad():
        x[::7] = -x[::7].abs()
    dy = torch.from_numpy(rng.normal(size=(q.shape[0], 24)).astype(np.float32))
    mp = max_pool(x, torch.from_numpy(inds))
    (mp * dy).sum().backward()
    g_mp = x.grad.clone(); x.grad = None
    xc = torch.from_numpy(rng.normal(size=(q.shape[0], 24)).astype(np.float32)).requires_grad_(True)
    dyc = torch.from_numpy(rng.normal(size=(s.shape[0], 24)).astype(np.float32))
    cp = closest_pool(xc, torch.from_numpy(up))
    (cp * dyc).sum().backward()
    ga = global_average(x.detach(), [110, 100])
    save("g6_pools.npz", x=x.detach().numpy(), inds=inds, dy=dy.numpy(), max_pool=mp.detach().numpy(),
         grad_max_pool=g_mp.numpy(), xc=xc.detach().numpy(), up=up, dyc=dyc.numpy(),
         closest_pool=cp.detach().numpy(), grad_closest_pool=xc.grad.numpy(),
         global_average=ga.numpy(), lengths=np.array([110, 100], np.int32))


# ---- G7 / G8 : pyramid and network ----------------------------------------------------------
class SmallConfig(Config):
    """Vaihingen3D-PL architecture (train_Vaihingen3D_PseudoLabel.py:70-87) at reduced width."""
    dataset = "Golden"
    architecture = ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
                    'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
                    'nearest_upsample', 'unary', 'nearest_upsample', 'unary',
                    'nearest_upsample', 'unary', 'nearest_upsample', 'unary']
    num_kernel_points = 15
    first_subsampling_dl = 0.24
    conv_radius = 2.5
    deform_radius = 6.0
    KP_extent = 1.0
    KP_influence = 'linear'
    aggregation_mode = 'sum'
    first_features_dim = 16
    in_features_dim = 4
    modulated = False
    use_batch_norm = True
    batch_norm_momentum = 0.02
    deform_fitting_mode = 'point2point'
    deform_fitting_power = 1.0
    deform_lr_factor = 0.1
    repulse_extent = 1.2
    learning_rate = 0.01
    momentum = 0.98
    weight_decay = 1e-3
    grad_clip_norm = 100.0
    dropout = 0          # deterministic forward for the fixture
    class_w = []
    saving = False


def make_pyramid_and_net():
    cfg = SmallConfig()
    rng = np.random.default_rng(41)
    pts = np.concatenate([sphere(rng, 1500, 3.0), sphere(rng, 1300, 3.0)])
    lens = np.array([1500, 1300], np.int32)
    feats = np.concatenate([np.ones((2800, 1), np.float32), rng.random((2800, 1)).astype(np.float32),
                            pts[:, 2:], pts[:, 2:]], axis=1).astype(np.float32)
    labels = rng.integers(0, 9, size=2800).astype(np.int64)
    dsobj = PointCloudDataset("golden")
    dsobj.config = cfg
    limits = [20, 24, 26, 26, 20]
    dsobj.neighborhood_limits = limits
    np.random.seed(1234)   # batch_grid_subsampling draws theta, phi, alpha per level (common.py:99-106)
    li = dsobj.segmentation_inputs(pts, feats, labels, lens)
    L = 5
    arrs = dict(points=pts, lens=lens, features=feats, labels=labels, limits=np.array(limits, np.int32),
                np_seed=np.int64(1234))
    for l in range(L):
        arrs["points_%d" % l] = li[l]
        arrs["neighbors_%d" % l] = li[L + l]
        arrs["pools_%d" % l] = li[2 * L + l]
        arrs["upsamples_%d" % l] = li[3 * L + l]
        arrs["lengths_%d" % l] = li[4 * L + l]
    save("g7_pyramid.npz", **arrs)

    # ---- G8: network on that pyramid
    class Batch:
        pass
    b = Batch()
    b.points = [torch.from_numpy(a) for a in li[0:L]]
    b.neighbors = [torch.from_numpy(a) for a in li[L:2 * L]]
    b.pools = [torch.from_numpy(a) for a in li[2 * L:3 * L]]
    b.upsamples = [torch.from_numpy(a) for a in li[3 * L:4 * L]]
    b.lengths = [torch.from_numpy(a) for a in li[4 * L:5 * L]]
    b.features = torch.from_numpy(li[5 * L])
    b.labels = torch.from_numpy(li[5 * L + 1])
    np.random.seed(99)
    torch.manual_seed(99)
    label_values = np.arange(9)
    net = KPFCNN(cfg, label_values, [])
    net.train()
    sd0 = {k: v.detach().clone().numpy() for k, v in net.state_dict().items()}
    # trainer_PseudoLabel.py:80-87
    deform_params = [v for k, v in net.named_parameters() if 'offset' in k]
    other_params = [v for k, v in net.named_parameters() if 'offset' not in k]
    opt = torch.optim.SGD([{'params': other_params}, {'params': deform_params, 'lr': cfg.learning_rate * cfg.deform_lr_factor}],
                          lr=cfg.learning_rate, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
    opt.zero_grad()
    out = net(b, cfg)
    loss = net.loss(out, b.labels)
    acc = net.accuracy(out, b.labels)
    loss.backward()
    grads = {k: v.grad.detach().clone().numpy() for k, v in net.named_parameters() if v.grad is not None}
    torch.nn.utils.clip_grad_value_(net.parameters(), cfg.grad_clip_norm)
    opt.step()
    sd1 = {k: v.detach().clone().numpy() for k, v in net.state_dict().items()}
    sel = ["encoder_blocks.0.KPConv.weights", "encoder_blocks.1.KPConv.weights", "encoder_blocks.1.unary_shortcut.mlp.weight", "encoder_blocks.3.unary1.mlp.weight",
           "encoder_blocks.9.KPConv.weights", "decoder_blocks.1.mlp.weight", "decoder_blocks.7.mlp.weight",
           "head_mlp.mlp.weight", "head_mlp.batch_norm.bias", "head_softmax.mlp.weight", "head_softmax.batch_norm.bias"]
    arrs = dict(logits=out.detach().numpy(), loss=np.float32(loss.item()), acc=np.float32(acc))
    for k, v in sd0.items():
        if "num_batches_tracked" in k:
            continue
        arrs["sd0/" + k] = v
    for k in sel:
        arrs["grad/" + k] = grads[k]
        arrs["sd1/" + k] = sd1[k]
    arrs["grad_names"] = np.array(sorted(grads.keys()))
    arrs["grad_norms"] = np.array([np.linalg.norm(grads[k].astype(np.float64)) for k in sorted(grads.keys())])
    save("g8_kpfcnn.npz", **arrs)


def make_kernel_points():
    data = read_ply(os.path.join(REF, "kernels/dispositions/k_015_center_3D.ply"))
    kp = np.vstack((data['x'], data['y'], data['z'])).T
    save("g9_kernel_points.npz", kernel_points=kp)
    # load_kernels replay (kernel_points.py:407-488): same np.random seed -> same output
    from kernels.kernel_points import load_kernels
    np.random.seed(5)
    k1 = load_kernels(0.6, 15, 3, 'center')
    save("g9_load_kernels.npz", seed=np.int64(5), radius=np.float64(0.6), kernel_points=k1)


if __name__ == "__main__":
    assert geom.have_ref(), "run `make -C oracle ref` first"
    make_geometry()
    make_kpconv()
    make_pools()
    make_pyramid_and_net()
    make_kernel_points()
