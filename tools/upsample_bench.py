#!/usr/bin/env python
"""The eight depthwise up-sample + skip-add launches of DLAUp / IDAUp (dla_dcn.py:409-425) at B = 8, bf16, 10 launches per hipGraph replay.
usage: python tools/upsample_bench.py [B=8]   (MFX_LIB_PATH=... for an A/B against another build)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops

lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [(12, 40, 256, 2, 1), (24, 80, 128, 2, 2), (48, 160, 64, 2, 4), (24, 80, 64, 4, 1)]     # H, W, C, f, count
N = 10
dt = torch.bfloat16


def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * N) * 1e3


tot = 0.0
chk = 0.0
for (H, W, C, f, cnt) in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, C, device="cuda").to(dt)
    skip = torch.randn(B, H * f, W * f, C, device="cuda").to(dt)
    w = torch.rand(4 * f * f, C, device="cuda")
    t = timed(lambda: ops.upsample_add(x, w, f, skip))
    y = ops.upsample_add(x, w, f, skip)
    chk += float(y.float().double().sum())
    mb = (x.numel() + 2 * skip.numel()) * 2 / 1e6
    tot += cnt * t
    print("%dx%d C=%d f=%d x%d: %.1f us  (%.1f MB -> %.2f TB/s)" % (H, W, C, f, cnt, t, mb, mb / t / 1e6 * 1e6 / 1e6), flush=True)
print("all 8: %.1f us   checksum %.6f" % (tot, chk))
