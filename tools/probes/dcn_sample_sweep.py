#!/usr/bin/env python
"""Sampling kernel of the project-then-sample DCN (csrc/dcn_ps.hip), 128 -> 64 @ 48 x 160 shape (projected map 70.8 MB): variants x offset spread."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib
L = lib.load()
N = 10
def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * N) * 1e3
B, H, W, Co = 8, 48, 160, 64
P = torch.randn(B, H, W, 9 * Co, device="cuda").to(torch.bfloat16)
y = torch.empty(B, H, W, Co, device="cuda", dtype=torch.bfloat16)
sc, sh = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
for std in (0.0, 1.0, 3.0, 6.0):
    om = torch.zeros(B, H, W, 32, device="cuda")
    om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * std
    om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
    row = []
    for var in (0, 1, 2, 3, 4, 5):
        lib.check(L.mfx_set_option(b"dcn_ps_var", var), "opt")
        row.append(timed(lambda: L.mfx_dcn_sample_nhwc(P.data_ptr(), om.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), B, H, W, Co, Co, 1, lib.MFX_BF16,
                                                       torch.cuda.current_stream().cuda_stream)))
    print("offset std %.1f: default(3 taps, 8x4) %.1f | 1 tap %.1f | 9 taps %.1f | 16x2 %.1f | 4x8 %.1f | 32x1 %.1f us" % (std, *row), flush=True)
