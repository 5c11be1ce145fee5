"""Generates tests/golden/ply/*.ply with the REFERENCE's own writer (utils/ply.py, pure numpy, importable in the build
container) and the arrays they must read back as (ply_expected.npz).  Run once in the build container:
    python tests/golden/make_golden_ply.py
The .ply files are data fixtures (inputs/outputs), not reference source."""
import os, sys
import numpy as np

REF = os.environ.get("WEASAL_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
from utils.ply import write_ply, read_ply     # noqa: E402

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ply")
os.makedirs(out, exist_ok=True)
rng = np.random.RandomState(7)
pts = rng.rand(257, 3).astype(np.float32) * 50
cols = rng.randint(0, 255, size=(257, 3)).astype(np.uint8)
lab = rng.randint(0, 9, size=257).astype(np.int32)
conf = rng.rand(257).astype(np.float64)
tri = rng.randint(0, 257, size=(40, 3)).astype(np.int32)
write_ply(os.path.join(out, "xyz"), pts, ['x', 'y', 'z'])
write_ply(os.path.join(out, "xyz_rgb_class.ply"), [pts, cols, lab], ['x', 'y', 'z', 'red', 'green', 'blue', 'class'])
write_ply(os.path.join(out, "mixed.ply"), (pts[:, 0], conf, lab.astype(np.int16), cols[:, :2]), ['x', 'conf', 'l16', 'r', 'g'])
write_ply(os.path.join(out, "mesh.ply"), [pts, lab], ['x', 'y', 'z', 'class'], triangular_faces=tri)
exp = {}
for name in ("xyz", "xyz_rgb_class", "mixed"):
    d = read_ply(os.path.join(out, name + ".ply"))
    for f in d.dtype.names:
        exp[name + "/" + f] = np.ascontiguousarray(d[f])
v, f = read_ply(os.path.join(out, "mesh.ply"), triangular_mesh=True)
for fn in v.dtype.names:
    exp["mesh/" + fn] = np.ascontiguousarray(v[fn])
exp["mesh/__faces__"] = f
np.savez(os.path.join(out, "ply_expected.npz"), **exp)
np.savez(os.path.join(out, "ply_inputs.npz"), pts=pts, cols=cols, lab=lab, conf=conf, tri=tri)
print("wrote", sorted(os.listdir(out)))
