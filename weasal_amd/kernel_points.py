"""Kernel-point dispositions (host side of KPConv.init_KP).

Mirrors the interface of the reference's kernels/kernel_points.py for the part the hot path
uses: ``load_kernels`` (kernel_points.py:407-488) and ``create_3D_rotations`` (:43-74).  The
cached 15-point 'center' disposition the reference ships as kernels/dispositions/
k_015_center_3D.ply is packaged as a data table (weasal_amd/data/k_015_center_3D.npy, 15x3
float64).  The optimisation-based generators (spherical_Lloyd / kernel_point_optimization_debug,
:77-404) only run in the reference when that file is missing and are out of scope: asking for a
disposition that is not packaged raises.
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def create_3D_rotations(axis, angle):
    """Rotation matrices from unit axes [N,3] and angles [N] (Rodrigues form), float64 in ->
    [N,3,3].  Same element formulas as the reference (kernel_points.py:43-74) so that seeded
    rotations are bit-identical."""
    axis = np.asarray(axis)
    angle = np.asarray(angle)
    c = np.cos(angle)
    omc = 1 - c
    s = np.sin(angle)
    ax, ay, az = axis[:, 0], axis[:, 1], axis[:, 2]
    xy = omc * ax * ay
    xz = omc * ax * az
    yz = omc * ay * az
    rows = [c + omc * (ax * ax), xy - s * az, xz + s * ay,
            xy + s * az, c + omc * (ay * ay), yz - s * ax,
            xz - s * ay, yz + s * ax, c + omc * (az * az)]
    return np.reshape(np.stack(rows, axis=1), (-1, 3, 3))


def _disposition(num_kpoints, dimension, fixed):
    name = "k_{:03d}_{:s}_{:d}D.npy".format(num_kpoints, fixed, dimension)
    path = os.path.join(_DATA, name)
    if not os.path.exists(path):
        raise NotImplementedError(
            "kernel disposition %s is not packaged (the reference would run its Lloyd / gradient "
            "optimiser, kernels/kernel_points.py:77-404, which is outside the hot path)" % name)
    return np.load(path)


def load_kernels(radius, num_kpoints, dimension, fixed, lloyd=False):
    """Kernel points for one KPConv: disposition + N(0, 0.01) noise, scaled by `radius`, rotated
    about z by a random angle.  Consumes np.random in the reference's order (one rand() for the
    angle, then normal(size=(K,dim))) so that a seeded run reproduces the reference's values."""
    kernel_points = _disposition(num_kpoints, dimension, fixed)
    if dimension != 3 or fixed == "vertical":
        raise NotImplementedError("only 3-D kernels with fixed in {'center','none'} are supported")
    theta = np.random.rand() * 2 * np.pi
    c, s = np.cos(theta), np.sin(theta)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
    kernel_points = kernel_points + np.random.normal(scale=0.01, size=kernel_points.shape)
    kernel_points = radius * kernel_points
    kernel_points = np.matmul(kernel_points, R)
    return kernel_points.astype(np.float32)
