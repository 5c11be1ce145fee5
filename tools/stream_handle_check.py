"""Diagnostic: _lib.current_stream() returns the handle torch reports (default and side streams); cost per call of both."""
import time, torch, sys
sys.path.insert(0, '.')
from weasal_amd import _lib
s = torch.cuda.Stream()
a = _lib.current_stream().value or 0; b = torch.cuda.current_stream().cuda_stream
assert a == b, (a, b)
with torch.cuda.stream(s):
    assert _lib.current_stream().value == s.cuda_stream == torch.cuda.current_stream().cuda_stream
t0 = time.perf_counter()
for _ in range(20000): _lib.current_stream()
t1 = time.perf_counter()
for _ in range(20000): torch.cuda.current_stream().cuda_stream
t2 = time.perf_counter()
print("raw %.2f us  torch %.2f us" % ((t1 - t0) / 20000 * 1e6, (t2 - t1) / 20000 * 1e6))
