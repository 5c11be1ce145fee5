from weasal_amd.cpp_wrappers.cpp_subsampling.grid_subsampling import subsample, subsample_batch  # noqa: F401
