"""Diagnostic: the short products of the deep pyramid levels -- this library's MFMA kernels next to torch.matmul
(hipBLASLt), per shape and per product (y = x b, dx = dy b^T, dW = x^T dy).  GPU time by events behind a spin kernel
(the host is ahead of the GPU when the timed launches start, so launch overhead does not count)."""
import sys, torch
sys.path.insert(0, '.')
from weasal_amd import ops, _lib
dev = torch.device('cuda:0')
lib = _lib.lib()
REP = 4
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(3_000_000)
        e0.record()
        for _ in range(REP): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REP * 1e3)
    return best
levels = {int(a.split('=')[0]): 1 for a in sys.argv[1:]} if len(sys.argv) > 1 else None
shapes = []
for m, d_in, d in ((10257, 256, 512), (10257, 512, 512), (2600, 512, 1024), (2600, 1024, 1024), (380, 1024, 2048), (380, 2048, 2048)):
    q = d // 4
    shapes += [(m, d_in, q), (m, 15 * q, q), (m, q, d)] + ([(m, d_in, d)] if d_in != d else [])
shapes += [(2600, 1024 + 512, 512), (10257, 512 + 256, 256), (71070, 256 + 128, 128)]     # decoder unary blocks
tot = [0.0] * 6
for m, k, n in shapes:
    x = torch.randn(m, k, device=dev); b = torch.randn(k, n, device=dev) / k ** 0.5; dy = torch.randn(m, n, device=dev)
    bt = b.t().contiguous()
    t = [timeit(lambda: ops._gemm_xb(x, b)), timeit(lambda: torch.matmul(x, b)),
         timeit(lambda: ops._gemm_xb(dy, bt)), timeit(lambda: torch.matmul(dy, b.t())),
         timeit(lambda: ops._gemm_xty(lib, x, dy)), timeit(lambda: torch.matmul(x.t(), dy))]
    for i in range(6): tot[i] += t[i]
    fl = 2.0 * m * k * n / 150e12 * 1e6
    by = 4.0 * (m * k + k * n + m * n) / 5e12 * 1e6
    print("M=%5d K=%5d N=%5d | y ours %6.1f torch %6.1f | dx ours %6.1f torch %6.1f | dW ours %6.1f torch %6.1f us | floor %5.1f us"
          % (m, k, n, *t, max(fl, by)), flush=True)
print("totals: y %.0f / %.0f, dx %.0f / %.0f, dW %.0f / %.0f us" % tuple(tot))
