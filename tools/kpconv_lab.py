"""A/B of the K3 forward forms on DALES-shaped layers (interleaved rounds in one process, HIP events):
variant 1 = entry pool + VALU accumulate, variant 2 = matrix core (ws_kpconv_variant).  Also checks that the two forms
agree bit for bit.  Usage: python tools/kpconv_lab.py [rounds]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import _lib, config as wcfg, ops, pyramid, synthetic  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    only = sys.argv[2] if len(sys.argv) > 2 else None          # e.g. "enc1": one layer, f32 only (counter runs)
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    variant = C.c_int.in_dll(lib, "ws_kpconv_variant")
    if os.environ.get("WEASAL_K3_ABLATE"):          # counter runs of an ablated kernel (diagnostics)
        C.c_int.in_dll(lib, "ws_kpconv_ablate").value = int(os.environ["WEASAL_K3_ABLATE"])
    wl = synthetic.WORKLOADS["dales"]
    cfg = wcfg.DALESPLConfig()
    pts, feats, labels, lens = synthetic.make_inputs(1, wl["spheres"], wl["points"], wl["radius"], cfg.in_features_dim)
    np.random.seed(0)
    batch = pyramid.build_batch(cfg, torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev),
                                torch.from_numpy(labels).to(dev), lens, wl["limits"])
    batch.activate()
    kp = torch.from_numpy(np.load(os.path.join(os.path.dirname(ops.__file__), "data", "k_015_center_3D.npy")).astype(np.float32)).to(dev)
    cases = [(0, 3, "enc0"), (0, 32, "enc1"), (1, 64, "enc3"), (2, 128, "enc5"), (3, 256, "enc7"), (4, 512, "enc9")]
    for dt in (torch.float32, torch.bfloat16):
        for lvl, ci, name in cases:
            if dt == torch.bfloat16 and ci < 8:
                continue
            if only is not None and (name != only or dt != torch.float32):
                continue
            P = batch.points[lvl]
            inds = batch.neighbors[lvl]
            r = cfg.first_subsampling_dl * cfg.conv_radius * 2 ** lvl
            extent = r * cfg.KP_extent / cfg.conv_radius
            x = torch.randn(P.shape[0], ci, device=dev).to(dt)
            kps = kp * r
            outs, times = {}, {1: [], 2: []}
            for rd in range(rounds):
                for v in (1, 2):
                    variant.value = v
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    wf, _ = ops.kpconv_gather(x, P, P, inds, kps, extent)
                    e1.record()
                    torch.cuda.synchronize()
                    times[v].append(e0.elapsed_time(e1))
                    outs[v] = wf
            variant.value = 2
            # ---- what a spatially sorted row layout would give: rows of points / features stored in the cell order of the
            #      level's search grid, index matrix remapped (same values, same column order -> same sums)
            order = ops._order_for(P)
            if order is not None:
                perm = order.long()
                rank = torch.empty_like(perm)
                rank[perm] = torch.arange(perm.numel(), device=dev)
                rank_pad = torch.cat([rank, torch.tensor([perm.numel()], device=dev)])
                Ps, xs = P[perm].contiguous(), x[perm].contiguous()
                inds_s = rank_pad[inds[perm]].contiguous()
                ts = []
                for rd in range(rounds):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    wfs, _ = ops.kpconv_gather(xs, Ps, Ps, inds_s, kps, extent)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                print("      sorted layout: mfma %.4f ms; equal to the unsorted result %s" % (np.median(ts), torch.equal(wfs, outs[2][perm])), flush=True)
            if name == "enc1" and dt == torch.float32:
                # ablations of the matrix-core form (diagnostics): which part of the memory traffic sets the time
                abl = C.c_int.in_dll(lib, "ws_kpconv_ablate")
                for mask, what in ((0, "full"), (1, "no wf store"), (2, "rows all = row 0"), (4, "xyz from 64 fixed points"),
                                   (8, "no index load"), (3, "no store + rows 0"), (7, "no store, rows 0, xyz fixed"),
                                   (15, "all memory ablated"), (15 + 16, "+ no MFMA"), (15 + 32, "+ no influence math"),
                                   (15 + 64, "+ no row loads"), (15 + 128, "+ no LDS reads"), (15 + 16 + 32 + 64 + 128, "+ all of these")):
                    abl.value = mask
                    ts = []
                    for rd in range(rounds):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        ops.kpconv_gather(x, P, P, inds, kps, extent)
                        e1.record()
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1))
                    print("      ablate %2d (%-28s): %.4f ms" % (mask, what, np.median(ts)), flush=True)
                abl.value = 0
                gsv = C.c_int.in_dll(lib, "ws_kpconv_gs")
                for gs in (1, 2, 4, 8):
                    gsv.value = gs
                    ts = []
                    for rd in range(rounds):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        ops.kpconv_gather(x, P, P, inds, kps, extent)
                        e1.record()
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1))
                    print("      group size %d: %.4f ms" % (gs, np.median(ts)), flush=True)
                gsv.value = 0
            same = torch.equal(outs[1], outs[2])
            md = (outs[1].float() - outs[2].float()).abs().max().item()
            n, h = inds.shape
            es = 2 if dt == torch.bfloat16 else 4
            b = n * h * (8 + 12 + es * ci) + n * (12 + es * ci) + 60 * ci * ci + 180
            t1, t2 = np.median(times[1]), np.median(times[2])
            print("%-5s %-8s N=%7d H=%3d Ci=%3d  pool %.4f ms (%.2f TB/s)  mfma %.4f ms (%.2f TB/s = %.2f of 8 TB/s)  bit-equal %s maxdiff %.2e"
                  % (name, str(dt).split(".")[1], n, h, ci, t1, b / t1 / 1e9, t2, b / t2 / 1e9, b / t2 / 1e9 / 8.0, same, md), flush=True)


if __name__ == "__main__":
    main()
