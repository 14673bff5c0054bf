#!/usr/bin/env python
"""Graph-replay timing of one 3x3 / stride 1 NHWC conv under the LDS-halo kernel variants (option "halo" = variant + 1).
usage: python tools/conv_bench.py B H W Cin Cout v1,v2,...   (variant 0 = automatic choice)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops

L = lib.load()
B, H, W, Cin, Cout = map(int, sys.argv[1:6])
variants = [int(v) for v in sys.argv[6].split(",")]
for kv in (sys.argv[7].split(",") if len(sys.argv) > 7 else []):      # extra library options k=v,...
    k, v = kv.split("=")
    lib.check(L.mfx_set_option(k.encode(), int(v)), "opt")
x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
p = ops.pack_conv(w, torch.bfloat16, None, None, stride=1, pad=1)
N = 20
flop = 2.0 * B * H * W * 9 * Cin * Cout
out = []
for v in variants:
    lib.check(L.mfx_set_option(b"halo", 1 if v == 0 else v + 1), "opt")
    try:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                ops.conv2d(x, p)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(N):
                    ops.conv2d(x, p)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (5 * N) * 1e3
        out.append("v%d: %.1f us (%.0f TF)" % (v, us, flop / us / 1e6))
    except Exception as ex:
        out.append("v%d: %s" % (v, str(ex)[:40]))
lib.check(L.mfx_set_option(b"halo", 1), "opt")
print("%dx%dx%d %d->%d  " % (B, H, W, Cin, Cout) + "  ".join(out), flush=True)
