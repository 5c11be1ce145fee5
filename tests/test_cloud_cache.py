"""The cached sub-sampled tiles (`input_{dl:.3f}/`, SURVEY.md section 8f rank 4 remainder) against fixtures the reference's
own calls wrote (tests/golden/make_golden_cache.py: datasets.common.grid_subsampling with labels, utils.ply.write_ply,
sklearn KDTree.query): the cache file must be the same bytes, the re-projection indices and the coarse potential points the
same arrays.  CPU tests pin the file format and the oracle; the GPU test builds the cache through K2 / K1."""
import os
import shutil

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cache")


def _expected():
    return np.load(os.path.join(HERE, "tile_a_expected.npz"))


def test_cached_sub_cloud_reads_and_rewrites_identically(tmp_path):
    """weasal_amd.cloud_cache reads the reference-written cache file and writes the same bytes back"""
    from weasal_amd import cloud_cache
    exp = _expected()
    ref_file = os.path.join(cloud_cache.cache_dir(HERE, float(exp["dl"])), "tile_a.ply")
    assert os.path.basename(os.path.dirname(ref_file)) == "input_0.400"
    pts, lab = cloud_cache.read_sub_cloud(ref_file)
    assert pts.dtype == np.float32 and pts.shape == (int(exp["n_sub"]), 3) and lab.dtype == np.int32
    out = str(tmp_path / "again.ply")
    assert cloud_cache.write_sub_cloud(out, pts, lab)
    assert open(out, "rb").read() == open(ref_file, "rb").read()


def test_oracle_subsampling_reproduces_the_cache_file(tmp_path):
    """the CPU oracle's grid subsampling with labels (oracle/geom.py) + our writer = the reference's cache file"""
    from oracle import geom
    from weasal_amd import cloud_cache
    exp = _expected()
    points, labels = cloud_cache.read_tile(os.path.join(HERE, "tile_a.ply"))
    assert np.array_equal(labels, exp["labels"])
    sub_p, sub_l = geom.subsample(points, classes=labels.astype(np.int32), sampleDl=float(exp["dl"]))
    out = str(tmp_path / "sub.ply")
    assert cloud_cache.write_sub_cloud(out, sub_p, np.squeeze(sub_l))
    assert open(out, "rb").read() == open(os.path.join(HERE, "input_0.400", "tile_a.ply"), "rb").read()
    coarse = geom.subsample(sub_p, sampleDl=float(exp["in_radius"]) / 10)
    assert np.array_equal(coarse, exp["coarse_points"])


@pytest.mark.gpu
def test_cache_built_on_the_gpu_is_the_reference_cache(tmp_path, gpu):
    """load_subsampled_cloud on an empty cache directory: K2 with labels -> the same file bytes; then the cached file is
    read back; re-projection (K1) and coarse potential points (K2) equal the reference's KDTree / subsampling results"""
    from weasal_amd import cloud_cache
    exp = _expected()
    dl = float(exp["dl"])
    root = str(tmp_path)
    tile = os.path.join(root, "tile_a.ply")
    shutil.copy(os.path.join(HERE, "tile_a.ply"), tile)
    sub_p, sub_l, built = cloud_cache.load_subsampled_cloud(root, "tile_a", tile, dl, gpu)
    assert built and sub_p.shape[0] == int(exp["n_sub"])
    mine = os.path.join(cloud_cache.cache_dir(root, dl), "tile_a.ply")
    assert open(mine, "rb").read() == open(os.path.join(HERE, "input_0.400", "tile_a.ply"), "rb").read()
    sub_p2, sub_l2, built2 = cloud_cache.load_subsampled_cloud(root, "tile_a", tile, dl, gpu)
    assert not built2 and bool((sub_p2 == sub_p).all()) and bool((sub_l2 == sub_l).all())
    proj, labels = cloud_cache.reprojection_indices(root, "tile_a", tile, sub_p, dl)
    ok = exp["proj_tie_free"]
    assert np.array_equal(proj[ok], exp["proj_inds"][ok]) and np.array_equal(labels, exp["labels"])
    proj2, _ = cloud_cache.reprojection_indices(root, "tile_a", tile, sub_p, dl)        # from <name>_proj.npz
    assert np.array_equal(proj2, proj)
    coarse = cloud_cache.coarse_potential_points(sub_p, float(exp["in_radius"]))
    assert np.array_equal(coarse.cpu().numpy(), exp["coarse_points"])
