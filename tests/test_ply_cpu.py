"""weasal_amd.ply against files written by the reference's own utils/ply.py (tests/golden/make_golden_ply.py):
reading must give the same arrays, writing the same bytes."""
import os

import numpy as np

from weasal_amd.ply import read_ply, write_ply

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ply")


def _inputs():
    d = np.load(os.path.join(HERE, "ply_inputs.npz"))
    return d["pts"], d["cols"], d["lab"], d["conf"], d["tri"]


def test_read_matches_reference_reader():
    exp = np.load(os.path.join(HERE, "ply_expected.npz"))
    for name in ("xyz", "xyz_rgb_class", "mixed"):
        data = read_ply(os.path.join(HERE, name + ".ply"))
        want = sorted(k.split("/", 1)[1] for k in exp.files if k.startswith(name + "/"))
        assert sorted(data.dtype.names) == want
        for f in data.dtype.names:
            ref = exp[name + "/" + f]
            assert data[f].dtype == ref.dtype and np.array_equal(data[f], ref), (name, f)
    v, faces = read_ply(os.path.join(HERE, "mesh.ply"), triangular_mesh=True)
    for f in v.dtype.names:
        assert np.array_equal(v[f], exp["mesh/" + f])
    assert faces.dtype == exp["mesh/__faces__"].dtype and np.array_equal(faces, exp["mesh/__faces__"])


def test_write_is_byte_identical_to_reference_writer(tmp_path):
    pts, cols, lab, conf, tri = _inputs()
    cases = [("xyz", (pts, ['x', 'y', 'z'], None)),
             ("xyz_rgb_class.ply", ([pts, cols, lab], ['x', 'y', 'z', 'red', 'green', 'blue', 'class'], None)),
             ("mixed.ply", ((pts[:, 0], conf, lab.astype(np.int16), cols[:, :2]), ['x', 'conf', 'l16', 'r', 'g'], None)),
             ("mesh.ply", ([pts, lab], ['x', 'y', 'z', 'class'], tri))]
    for name, (fields, names, faces) in cases:
        path = str(tmp_path / name)
        assert write_ply(path, fields, names, triangular_faces=faces) is True
        path = path if path.endswith(".ply") else path + ".ply"
        ref = os.path.join(HERE, name if name.endswith(".ply") else name + ".ply")
        assert open(path, "rb").read() == open(ref, "rb").read(), name


def test_write_rejects_inconsistent_fields(tmp_path, capsys):
    pts, cols, lab, conf, tri = _inputs()
    assert write_ply(str(tmp_path / "a"), [pts, lab[:10]], ['x', 'y', 'z', 'c']) is False
    assert write_ply(str(tmp_path / "b"), [pts, lab], ['x', 'y', 'z']) is False
    assert write_ply(str(tmp_path / "c"), np.zeros((4, 2, 2)), ['x']) is False
    out = capsys.readouterr().out
    assert "wrong field dimensions" in out and "wrong number of field names" in out and "more than 2 dimensions" in out


def test_ascii_and_non_ply_are_rejected(tmp_path):
    import pytest
    p = tmp_path / "a.ply"
    p.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(ValueError):
        read_ply(str(p))
    q = tmp_path / "b.ply"
    q.write_bytes(b"nope\n")
    with pytest.raises(ValueError):
        read_ply(str(q))
