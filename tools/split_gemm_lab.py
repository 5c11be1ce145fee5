"""Diagnostic: y = x b on the f32-input MFMA (ws_gemm_split = 0) and as six bf16 MFMA partial products of three-way
splits (ws_gemm_split = 1): GPU time (events behind a spin kernel) and error against float64."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from weasal_amd import ops, _lib
dev = torch.device('cuda:0')
lib = _lib.lib()
sw = ctypes.c_int.in_dll(lib, "ws_gemm_split")
REP = 5
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
shapes = [(400000, 128, 128), (400000, 32, 128), (400000, 128, 32), (400000, 480, 32), (400000, 32, 480), (400000, 64, 128),
          (71070, 256, 256), (71070, 960, 64), (71070, 64, 960), (71070, 384, 128), (10257, 1920, 128), (10257, 512, 512), (10257, 128, 1920),
          (1526, 3840, 256), (1526, 1024, 256), (1526, 256, 1024), (380, 7680, 512), (380, 2048, 512), (380, 512, 2048)]
tot = [0.0, 0.0, 0.0]
for m, k, n in shapes:
    torch.manual_seed(1)
    x = torch.randn(m, k, device=dev) * torch.exp(torch.randn(m, 1, device=dev))       # rows of very different scale
    b = torch.randn(k, n, device=dev) / k ** 0.5
    ref = x.double() @ b.double()
    res = []
    for mode in (0, 1, 2):
        sw.value = mode
        y = ops._gemm_xb(x, b)
        err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
        rms = ((y.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        t = timeit(lambda: ops._gemm_xb(x, b))
        res.append((t, err, rms)); tot[mode] += t
    fl = 2.0 * m * k * n
    print("M=%6d K=%5d N=%5d | f32 MFMA %7.1f us err %.1e | split NT4 %7.1f us (%5.1f TF/s) err %.1e | split NT2 %7.1f us err %.1e"
          % (m, k, n, res[0][0], res[0][1], res[1][0], fl / res[1][0] / 1e6, res[1][1], res[2][0], res[2][1]), flush=True)
print("totals: f32 MFMA %.0f us, split %.0f us, split NT2 %.0f us" % tuple(tot))

print("dW = x^T dy:")
tot = [0.0, 0.0]
for m, k, n in [(400000, 128, 128), (400000, 480, 32), (400000, 32, 128), (400000, 128, 32), (71070, 256, 256), (71070, 960, 64), (10257, 1920, 128),
                (10257, 512, 512), (1526, 3840, 256), (380, 7680, 512), (380, 2048, 512)]:
    torch.manual_seed(2)
    x = torch.randn(m, k, device=dev)
    dy = torch.randn(m, n, device=dev) * torch.exp(torch.randn(m, 1, device=dev))
    ref = x.double().t() @ dy.double()
    res = []
    for mode in (0, 1):
        sw.value = mode
        o = ops._gemm_xty(lib, x, dy)
        err = ((o.double() - ref).abs().max() / ref.abs().max()).item()
        t = timeit(lambda: ops._gemm_xty(lib, x, dy))
        res.append((t, err)); tot[mode] += t
    print("M=%6d K=%5d N=%5d | f32 MFMA %7.1f us err %.1e | split %7.1f us err %.1e" % (m, k, n, res[0][0], res[0][1], res[1][0], res[1][1]), flush=True)
sw.value = 0
print("totals: f32 MFMA %.0f us, split %.0f us" % tuple(tot))
