#!/usr/bin/env python
"""DCN as project-then-sample (ops.dcn_ps, csrc/dcn_ps.hip) against the fused gather kernel and against the library's fp32 DCN on the same values:
which 16-bit form is closer to fp32, at the real layer shapes.   usage: python tools/dcn_ps_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops

L = lib.load()
SHAPES = [(8, 12, 40, 512, 256), (8, 24, 80, 256, 256), (8, 24, 80, 256, 128), (8, 48, 160, 128, 64), (8, 24, 80, 256, 64), (2, 13, 37, 128, 64), (1, 7, 9, 256, 128)]
bad = 0
for dt in (torch.bfloat16, torch.float16):
    for (B, H, W, C, Co) in SHAPES:
        torch.manual_seed(1)
        x = torch.randn(B, H, W, C, device="cuda").relu().to(dt)
        w = (torch.randn(Co, C, 3, 3, device="cuda") * (1.0 / (3 * C ** 0.5))).to(dt).float()
        om = torch.zeros(B, H, W, 32, device="cuda")
        om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * 3.0
        wild = torch.rand(B, H, W, 18, device="cuda") < 0.02
        om[..., :18] = torch.where(wild, torch.randn(B, H, W, 18, device="cuda") * 30.0, om[..., :18])
        om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
        sc, sh = torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda") * 0.1
        p = ops.pack_conv(w, dt, sc, sh, stride=1, pad=1, act=1)
        p32 = ops.pack_conv(w, torch.float32, sc, sh, stride=1, pad=1, act=1)
        y32 = ops.dcn(x.float(), om, p32)
        y_g = ops.dcn(x, om, p).float()
        y_p = ops.dcn_ps(x, om, p).float()
        torch.cuda.synchronize()
        eg, ep = (y_g - y32).abs(), (y_p - y32).abs()
        tol = (2e-2 if dt == torch.bfloat16 else 4e-3) * y32.abs().clamp(min=1.0)
        nb = int((ep > tol).sum())
        bad += nb
        print("%s B%d %dx%d %d->%d: vs fp32  gather max %.4f mean %.5f | project-sample max %.4f mean %.5f | out of tol %d" %
              (str(dt).split(".")[-1], B, H, W, C, Co, float(eg.max()), float(eg.mean()), float(ep.max()), float(ep.mean()), nb), flush=True)
print("TOTAL out of tolerance:", bad)
sys.exit(1 if bad else 0)
