#!/usr/bin/env python
"""Per-shape timing of the convolution weight gradient (C ABI mfx_conv_wgrad_oihw), bf16, B=8 layer shapes of the network.
usage: python tools/wgrad_bench.py [--opts k=v,...]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib as L, ops

ap = argparse.ArgumentParser()
ap.add_argument("--opts", default="")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--only", default="", help="substring of the shape name")
a = ap.parse_args()
lib = L.load()
for kv in filter(None, a.opts.split(",")):
    k, v = kv.split("=")
    L.check(lib.mfx_set_option(k.encode(), int(v)), "opt")
SHAPES = [  # name, B, H, W, Cin, Cout, k, stride
    ("heads 64->256 @96x320", 8, 96, 320, 64, 256, 3, 1), ("level2 64->64 @96x320", 8, 96, 320, 64, 64, 3, 1),
    ("level3 128->128 @48x160", 8, 48, 160, 128, 128, 3, 1), ("level4 256->256 @24x80", 8, 24, 80, 256, 256, 3, 1),
    ("level5 512->512 @12x40", 8, 12, 40, 512, 512, 3, 1), ("level3 64->128 s2", 8, 96, 320, 64, 128, 3, 2),
    ("root 1x1 448->128 @48x160", 8, 48, 160, 448, 128, 1, 1), ("level0 16->16 @384x1280", 8, 384, 1280, 16, 16, 3, 1),
    ("head 1x1 256->32 @96x320", 8, 96, 320, 256, 32, 1, 1),
]
dev = torch.device("cuda", 0)
ws = ops._splitk_workspace(dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, B, H, W, Ci, Co, k, s in SHAPES:
    if a.only and a.only not in name:
        continue
    Ho, Wo = H // s, W // s
    x = torch.randn(B, H, W, Ci, device=dev).bfloat16()
    dy = torch.randn(B, Ho, Wo, Co, device=dev).bfloat16()
    dw = torch.empty(Co, Ci, k, k, device=dev)

    def run():
        L.check(lib.mfx_conv_wgrad_oihw(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, Ci, Ci, k, k, s, k // 2, k // 2, Ho, Wo, Co, Co,
                                        Co, Ci, L.MFX_BF16, ws.data_ptr(), ws.numel() * 4, st), "wgrad")
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    gf = 2.0 * B * Ho * Wo * Co * Ci * k * k / 1e9
    print("%-28s %8.1f us  %7.1f GF  %7.1f TF/s" % (name, us, gf, gf / us))
