from weasal_amd.cpp_wrappers.cpp_neighbors.radius_neighbors import batch_query  # noqa: F401
