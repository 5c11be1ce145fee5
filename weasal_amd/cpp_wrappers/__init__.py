"""numpy-signature drop-ins of the reference's two CPython extension modules
(cpp_wrappers/cpp_neighbors/radius_neighbors, cpp_wrappers/cpp_subsampling/grid_subsampling),
running the HIP kernels.  See INTEGRATION.md."""
