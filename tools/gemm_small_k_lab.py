"""Diagnostic: the HBM-bound products of DALES level 0 (M = 400 000, small K or small N) on ws_gemm_xb_epilogue for the forced
two epilogues of gemm_xb2 (ws_gemm_staged 0 = every lane stores its own row, 1 = the tile turned through LDS): time, logical bytes moved, TB/s."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weasal_amd import _lib
from weasal_amd._lib import ptr, current_stream, check
dev = torch.device("cuda:0"); lib = _lib.lib()
def t(fn, rep=10):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rep): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / rep)
    return best * 1e3
M = 400000
for k, n, res in ((32, 480, False), (32, 128, True), (128, 32, False), (64, 128, False), (32, 64, False), (480, 32, False), (128, 128, False), (128, 128, True)):
    x = torch.randn(M, k, device=dev); b = torch.randn(k, n, device=dev); y = torch.empty(M, n, device=dev)
    r = torch.randn(M, n, device=dev) if res else None
    bias = torch.randn(n, device=dev)
    row = []
    ref = None
    for st in (0, 1, 2):
        C.c_int.in_dll(lib, "ws_gemm_staged").value = min(st, 1)
        C.c_int.in_dll(lib, "ws_gemm_thin_k").value = 64 if st == 2 else 0
        us = t(lambda: check(lib.ws_gemm_xb_epilogue(ptr(x), M, k, k, ptr(b), n, ptr(bias), ptr(r), n, 1, 0.1, ptr(y), n, current_stream())))
        row.append(us)
        if ref is None: ref = y.clone()
        same = torch.equal(ref, y)
    C.c_int.in_dll(lib, "ws_gemm_staged").value = 1
    C.c_int.in_dll(lib, "ws_gemm_thin_k").value = 0
    mb = (M * k + M * n * (2 if res else 1)) * 4 / 1e6
    print("k=%4d n=%4d residual=%d: %s  (%.0f MB: %s TB/s)  same result: %s" % (k, n, res, "  ".join("staged=%d %6.1f us" % (i, u) for i, u in enumerate(row)), mb,
                                                             " / ".join("%.2f" % (mb / u) for u in row), same), flush=True)
