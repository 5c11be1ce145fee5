from weasal_amd.kernel_points import load_kernels, create_3D_rotations  # noqa: F401
