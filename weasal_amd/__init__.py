"""weasal_amd -- MI355X-native KPConv hot path for WeaSAL.

Layout: csrc/ (hand-written HIP kernels + the C ABI of include/weasal_hip.h), _lib.py (ctypes
binding), ops.py (device-tensor operators with autograd), blocks.py / architectures.py (the
reference's models.blocks / models.architectures module API on those operators), pyramid.py
(the input pyramid of datasets/common.py on the GPU), cpp_wrappers/ (numpy-signature drop-ins of
the reference's two CPython modules), dp.py (data-parallel step over RCCL).
"""
__version__ = "0.1.0"
