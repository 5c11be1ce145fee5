#!/usr/bin/env python3
"""tests/golden/make_golden_cache.py -- fixtures of the cached sub-sampled clouds (remainder):
what datasets/DALES_PseudoLabel.py:702-906 (load_subsampled_clouds) leaves under `input_{dl:.3f}/` for one tile, produced by
the reference's OWN calls -- datasets.common.grid_subsampling (its compiled C++ core, oracle/_ref) with labels :776-780,
utils.ply.write_ply :797-800, sklearn's KDTree for the re-projection indices :889-892 and the coarse potential points
:845 -- on a synthetic tile.  Outputs (data, not source):
  cache/tile_a.ply                 the "original" tile (x, y, z, class), written by the reference's writer
  cache/input_0.400/tile_a.ply     the cached sub-cloud file, byte for byte what the reference writes
  cache/tile_a_expected.npz        re-projection indices of the original points, coarse potential points
The KDTree pickles the reference also writes are code-bearing files: they are not produced and never read.
RUNS ONLY IN THE BUILD CONTAINER (import aids of make_golden.py)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402
import numpy as np                # noqa: E402
from sklearn.neighbors import KDTree                  # noqa: E402
from datasets.common import grid_subsampling          # noqa: E402
from utils.ply import read_ply, write_ply             # noqa: E402


def main():
    out = os.path.join(HERE, "cache")
    dl, in_radius = 0.4, 10.0
    sub_dir = os.path.join(out, "input_{:.3f}".format(dl))
    os.makedirs(sub_dir, exist_ok=True)
    rng = np.random.default_rng(2024)
    # a 12 x 12 x 4 m tile, 6 000 points, class = a function of position plus noise (so that cells hold mixed labels)
    pts = (rng.random((6000, 3)) * np.array([12.0, 12.0, 4.0])).astype(np.float32)
    lab = ((pts[:, 0] // 3).astype(np.int32) + (rng.random(6000) < 0.2).astype(np.int32) * 2) % 5
    tile = os.path.join(out, "tile_a.ply")
    write_ply(tile, [pts, lab.astype(np.int32)], ['x', 'y', 'z', 'class'])
    # ---- DALES_PseudoLabel.py:770-800
    data = read_ply(tile)
    points = np.vstack((data['x'], data['y'], data['z'])).T
    labels = data['class']
    sub_points, sub_labels = grid_subsampling(points, labels=labels, sampleDl=dl)
    sub_labels = np.squeeze(sub_labels)
    search_tree = KDTree(sub_points, leaf_size=10)
    write_ply(os.path.join(sub_dir, "tile_a.ply"), [sub_points, sub_labels], ['x', 'y', 'z', 'class'])
    # ---- :845 coarse potential points, :889-892 re-projection indices
    coarse = grid_subsampling(np.array(search_tree.data, copy=False).astype(np.float32), sampleDl=in_radius / 10)
    dist, idxs = search_tree.query(points, k=2, return_distance=True)
    proj = np.squeeze(idxs[:, 0]).astype(np.int32)
    tie_free = dist[:, 0] < dist[:, 1]                  # exact nearest-neighbour ties: implementation-defined, excluded from the comparison
    np.savez(os.path.join(out, "tile_a_expected.npz"), proj_inds=proj, proj_tie_free=tie_free, labels=labels, coarse_points=coarse,
             dl=np.float32(dl), in_radius=np.float32(in_radius), n_sub=np.int64(sub_points.shape[0]))
    print("tile", points.shape, "-> sub", sub_points.shape, "coarse", coarse.shape, "ties", int((~tie_free).sum()))


if __name__ == "__main__":
    assert mg.geom.have_ref(), "run `make -C oracle ref` first"
    main()
