"""Diagnostic: accuracy of torch.matmul (rocBLAS / hipBLASLt) on the short deep products of the small pyramid levels
against float64, next to the MFMA kernels of this library."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for m, k, n in ((380, 3840, 256), (380, 256, 256), (70, 7680, 512), (2600, 1920, 128), (1526, 3840, 256)):
    x = torch.randn(m, k, device=dev)
    dy = torch.randn(m, n, device=dev)
    w = torch.randn(k, n, device=dev) / k ** 0.5
    ref_dw = x.double().t() @ dy.double()
    ref_y = x.double() @ w.double()
    ref_dx = dy.double() @ w.double().t()
    rel = lambda a, r: ((a.double() - r).abs().max() / r.abs().max()).item()
    lib = ops._lib.lib()
    ours_dw = ops._gemm_xty(lib, x, dy)
    ours_y = ops._gemm_xb(x, w)
    print("m=%5d k=%5d n=%4d  dW: torch %.2e ours %.2e | y: torch %.2e ours %.2e | dx: torch %.2e"
          % (m, k, n, rel(torch.matmul(x.t(), dy), ref_dw), rel(ours_dw, ref_dw), rel(torch.matmul(x, w), ref_y), rel(ours_y, ref_y),
             rel(torch.matmul(dy, w.t()), ref_dx)), flush=True)
