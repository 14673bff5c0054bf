#!/usr/bin/env python
"""A/B of the two LDS-halo 3x3 conv kernels on the trunk's layer shapes: conv3x3_wave_kernel (option halo_cw=0) against conv3x3_cw_kernel
(halo_cw=1; 2 = its 3-waves-per-SIMD register budget), BN + residual + ReLU epilogue, 20 launches per hipGraph replay, bf16.
usage: python tools/conv_cw_bench.py [B=8] [k=v,...]   -> markdown table on stdout"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops

L = lib.load()
B0 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for kv in filter(None, (sys.argv[2] if len(sys.argv) > 2 else "").split(",")):      # extra library options, k=v,...
    lib.check(L.mfx_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1])), "opt")
SHAPES = [(96, 320, 64, 64), (48, 160, 128, 128), (24, 80, 256, 256), (12, 40, 512, 512), (96, 320, 64, 128)]
N = 20


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


print("| shape (B, H, W, Cin -> Cout) | halo_cw=0 us | halo_cw=1 us | halo_cw=2 us | TF/s at cw=1 | bit-identical |")
print("|---|---|---|---|---|---|")
for B in (B0, B0 // 2):
    for (H, W, Ci, Co) in SHAPES:
        x = torch.randn(B, H, W, Ci, device="cuda").bfloat16()
        r = torch.randn(B, H, W, Co, device="cuda").bfloat16()
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
        p = ops.pack_conv(w, torch.bfloat16, torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda"), stride=1, pad=1, act=lib.ACT_RELU)
        us, outs = [], []
        for cw in (0, 1, 2):
            lib.check(L.mfx_set_option(b"halo_cw", cw), "opt")
            outs.append(ops.conv2d(x, p, res=r).clone())
            us.append(timed(lambda: ops.conv2d(x, p, res=r)))
        lib.check(L.mfx_set_option(b"halo_cw", 1), "opt")
        same = all(torch.equal(o.view(torch.int16), outs[0].view(torch.int16)) for o in outs[1:])
        flop = 2.0 * B * H * W * 9 * Ci * Co
        print("| %d x %d x %d, %d -> %d | %.1f | %.1f | %.1f | %.0f | %s |" % (B, H, W, Ci, Co, us[0], us[1], us[2], flop / us[1] / 1e6, same), flush=True)
