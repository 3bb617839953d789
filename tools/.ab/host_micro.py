import sys, time, torch
sys.path.insert(0, "generative-turbulence_amd")
from turbdiff_amd import _lib as L
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev); y = torch.zeros(64, device=dev, dtype=torch.bfloat16)
lib = L.load()
def t(fn, n=20000):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
st = L.stream()
px, py = x.data_ptr(), y.data_ptr()
print("raw ctypes call, prebuilt args      %.2f us" % t(lambda: lib.tdx_cast(px, py, 64, 0, 1, st)))
print("L.call with L.ptr + L.stream        %.2f us" % t(lambda: L.call("tdx_cast", L.ptr(x), L.ptr(y), 64, L.F32, L.BF16, L.stream())))
print("L.ptr(x)                            %.2f us" % t(lambda: L.ptr(x), 200000))
print("L.stream()                          %.2f us" % t(lambda: L.stream(), 200000))
print("torch.empty_like(x)                 %.2f us" % t(lambda: torch.empty_like(x), 100000))
print("x.add_(1) (torch's own launch)      %.2f us" % t(lambda: x.add_(1)))
import ctypes
print("ctypes call tdx_version (no args)   %.2f us" % t(lambda: lib.tdx_version(), 200000))
