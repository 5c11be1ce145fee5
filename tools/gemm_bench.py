"""Diagnostic: per-shape timing of the MFMA tall-skinny GEMMs against their streaming / MFMA floors."""
import sys, torch
sys.path.insert(0, '.')
from weasal_amd import ops, _lib
from weasal_amd._lib import ptr, check, current_stream
dev = torch.device('cuda:0')
lib = _lib.lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
print("M =", M)
for (k, n) in [(64, 32), (32, 128), (64, 128), (128, 128), (480, 32), (128, 32), (45, 64), (32, 480), (128, 9), (384, 128)]:
    x = torch.randn(M, k, device=dev); b = torch.randn(k, n, device=dev); dy = torch.randn(M, n, device=dev)
    t_xb = timeit(lambda: ops._gemm_xb(x, b))
    out = torch.empty(k, n, device=dev)
    scratch = torch.empty(max(lib.ws_gemm_xty_scratch_bytes(M, k, n), 16), dtype=torch.uint8, device=dev)
    t_xty = timeit(lambda: check(lib.ws_gemm_xty(ptr(x), M, k, k, ptr(dy), n, n, ptr(out), ptr(scratch), current_stream())))
    t_torch = timeit(lambda: torch.matmul(x, b))
    byt = 4.0 * M * (k + n)
    flop = 2.0 * M * k * n
    floor = max(byt / 5.0e12, flop / 150e12) * 1e6
    print("K=%4d N=%4d  xb %7.1f us  xty %7.1f us  rocBLAS %7.1f us | floor %6.1f us (%.0f MB, %.1f GF)  xb eff %.2f xty eff %.2f" % (
        k, n, t_xb, t_xty, t_torch, floor, byt / 1e6, flop / 1e9, floor / t_xb, floor / t_xty))
