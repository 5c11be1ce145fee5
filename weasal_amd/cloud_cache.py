"""The cached, grid-subsampled tiles either side of the hot path (remainder):
what the reference's dataset classes keep under ``<dataset>/input_{dl:.3f}/`` (datasets/DALES_PseudoLabel.py:702-906,
``load_subsampled_clouds``) and hand to the sphere sampler.

Per tile the reference writes three things there: ``<name>.ply`` -- the cloud subsampled at ``first_subsampling_dl`` with
the majority label of every cell (:776-800) --, a pickled sklearn KDTree of those points (:789-794) and, for validation /
test tiles, a pickle with the index of the nearest sub-point of every original point (:867-900).  Here

  * the sub-cloud PLY is the same file, byte for byte: same name and directory, points and labels from the HIP grid
    subsampling (K2: same cells, barycentres, label arg-max and row order as the reference's C++), written by
    ``weasal_amd.ply.write_ply``.  A cache directory the reference filled is read as is, and vice versa;
  * the KDTree pickles are neither written nor read (pickles are code-bearing files): what they serve -- radius queries
    around a sphere centre, nearest sub-point of every original point, the coarse potential points -- runs on K1 / K2
    (``weasal_amd.tester.nearest_projection``, ``update_potentials``);
  * the re-projection indices are cached as a plain ``<name>_proj.npz`` (two arrays, no pickle).
"""
import os

import numpy as np
import torch

from . import ops
from .ply import read_ply, write_ply


def cache_dir(root, dl):
    """``<root>/input_{dl:.3f}`` (DALES_PseudoLabel.py:708)"""
    return os.path.join(root, 'input_{:.3f}'.format(dl))


def read_tile(file_path):
    """(points float32 [N,3], labels int32 [N]) of an original tile (DALES_PseudoLabel.py:770-772)"""
    data = read_ply(file_path)
    points = np.vstack((data['x'], data['y'], data['z'])).T.astype(np.float32)
    return np.ascontiguousarray(points), np.asarray(data['class'])


def write_sub_cloud(path, sub_points, sub_labels):
    """the cached sub-cloud file (DALES_PseudoLabel.py:797-800): fields x, y, z (float32) and class (int32)"""
    return write_ply(path, [np.asarray(sub_points, np.float32), np.asarray(sub_labels, np.int32).reshape(-1)],
                     ['x', 'y', 'z', 'class'])


def read_sub_cloud(path):
    """-> (sub_points float32 [M,3], sub_labels int32 [M]) of a cached sub-cloud file (:735-737)"""
    data = read_ply(path)
    pts = np.vstack((data['x'], data['y'], data['z'])).T.astype(np.float32)
    return np.ascontiguousarray(pts), np.asarray(data['class'])


def load_subsampled_cloud(root, cloud_name, file_path, dl, device):
    """One tile of ``load_subsampled_clouds``: the cached sub-cloud if ``input_{dl:.3f}/<name>.ply`` exists, else the tile
    is read, subsampled on the GPU (K2, labels by majority per cell) and the cache file written.
    -> (sub_points [M,3] float32 device tensor, sub_labels [M] int32 device tensor, built: bool)"""
    d = cache_dir(root, dl)
    os.makedirs(d, exist_ok=True)
    sub_file = os.path.join(d, '{:s}.ply'.format(cloud_name))
    if os.path.exists(sub_file):
        pts, lab = read_sub_cloud(sub_file)
        return torch.from_numpy(pts).to(device), torch.from_numpy(lab.astype(np.int32)).to(device), False
    points, labels = read_tile(file_path)
    P = torch.from_numpy(points).to(device)
    L = torch.from_numpy(labels.astype(np.int32)).to(device)
    sub_p, _, sub_l = ops.grid_subsample(P, np.array([P.shape[0]], np.int32), dl, labels=L)
    sub_l = sub_l.reshape(-1)
    if not write_sub_cloud(sub_file, sub_p.cpu().numpy(), sub_l.cpu().numpy()):
        raise RuntimeError("could not write " + sub_file)
    return sub_p, sub_l, True


def coarse_potential_points(sub_points, in_radius):
    """the coarse cloud the sampling potentials live on: the sub-cloud subsampled again at in_radius / 10
    (DALES_PseudoLabel.py:826-845)"""
    p, _ = ops.grid_subsample(sub_points, np.array([sub_points.shape[0]], np.int32), in_radius / 10)
    return p


def reprojection_indices(root, cloud_name, file_path, sub_points, dl):
    """index of the nearest sub-cloud point of every original point and the original labels (validation / test tiles,
    DALES_PseudoLabel.py:867-900), cached as ``<name>_proj.npz`` next to the sub-cloud.
    -> (proj_inds int32 [N] numpy, labels [N] numpy)"""
    from .tester import nearest_projection
    proj_file = os.path.join(cache_dir(root, dl), '{:s}_proj.npz'.format(cloud_name))
    if os.path.exists(proj_file):
        with np.load(proj_file, allow_pickle=False) as z:
            return z['proj_inds'], z['labels']
    points, labels = read_tile(file_path)
    P = torch.from_numpy(points).to(sub_points.device)
    # every point lies within one cell diagonal of its own cell's barycentre
    proj = nearest_projection(P, sub_points, dl * 1.7321 * 1.001).cpu().numpy().astype(np.int32)
    np.savez(proj_file, proj_inds=proj, labels=labels)
    return proj, labels
