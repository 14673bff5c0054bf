"""Is a hipMemsetAsync issued inside a captured region replayed?  mfx_colsum = memset(out) + atomic accumulation."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib as L, ops
lib = L.load()
dev = "cuda"
x = torch.ones(4096, 64, device=dev)
out = torch.full((64,), -5.0, device=dev)
def run():
    L.check(lib.mfx_colsum(x.data_ptr(), out.data_ptr(), 4096, 64, 64, L.MFX_F32, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "colsum")
run(); torch.cuda.synchronize(); print("eager:", out[:3].tolist())
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    run()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run()
for i in range(3):
    g.replay(); torch.cuda.synchronize(); print("replay", i, out[:3].tolist())
for n in (8, 256, 512, 1024, 4096, 1 << 20):
    buf = torch.full((n // 4 + 1,), 7.0, device=dev)
    g2 = torch.cuda.CUDAGraph()
    hip = ctypes.CDLL("libamdhip64.so")
    with torch.cuda.graph(g2):
        hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(n), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    buf.fill_(7.0); torch.cuda.synchronize()
    g2.replay(); torch.cuda.synchronize()
    print("memset node of %8d bytes replayed: first %.1f last-in-range %.1f beyond %.1f" % (n, float(buf[0]), float(buf[n // 4 - 1]), float(buf[n // 4])))
