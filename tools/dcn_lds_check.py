#!/usr/bin/env python
"""Fourth-generation DCN kernel (csrc/dcn_lds.hip) against the generic gather kernel and the C oracle on the same inputs: in-patch samples, samples that
leave the patch (the far pass), partial tiles, both 16-bit dtypes, with the offset conv inside (module) and with given offsets.
usage: python tools/dcn_lds_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops

L = lib.load()


def opt(k, v):
    lib.check(L.mfx_set_option(k.encode(), int(v)), "opt")


ROWS = [8]


def run(B, H, W, C, Co, std, dt, seed=0, far_frac=0.0):
    torch.manual_seed(seed)
    x = torch.randn(B, H, W, C, device="cuda").relu().to(dt)
    w = torch.randn(Co, C, 3, 3, device="cuda") * (1.0 / (3 * C ** 0.5))
    om = torch.zeros(B, H, W, 32, device="cuda")
    om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * std
    if far_frac > 0:                                    # a sprinkle of wild offsets, some of them leaving the image
        wild = torch.rand(B, H, W, 18, device="cuda") < far_frac
        om[..., :18] = torch.where(wild, torch.randn(B, H, W, 18, device="cuda") * 40.0, om[..., :18])
    om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
    p = ops.pack_conv(w, dt, torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda") * 0.1, stride=1, pad=1, act=1)
    ops.add_f16_fragments(p, w)
    opt("dcn_lds", 2); opt("dcn_lds_rows", ROWS[0]); y1 = ops.dcn(x, om, p).float()
    opt("dcn_lds", 0); opt("dcn_patch", 0); opt("dcn_wave", 0); y0 = ops.dcn(x, om, p).float()
    lib.check(L.mfx_reset_options(), "reset")
    torch.cuda.synchronize()
    d = (y1 - y0).abs()
    tol = 2e-2 * y0.abs().clamp(min=1.0)
    bad = int((d > tol).sum())
    print("B%d %dx%d %d->%d std %.1f far %.3f %s: max|d| %.4f (ref max %.2f), > tol: %d of %d" %
          (B, H, W, C, Co, std, far_frac, str(dt).split(".")[-1], float(d.max()), float(y0.abs().max()), bad, d.numel()), flush=True)
    return bad


def run_module(B, H, W, C, Co, std, dt, seed=0):
    from monoflex_amd.model.backbone.dla_dcn import DeformConv
    torch.manual_seed(seed)
    x = torch.randn(B, H, W, C, device="cuda").relu().to(dt)
    m = DeformConv(C, Co).eval().cuda()
    torch.nn.init.normal_(m.conv.conv_offset_mask.weight, std=std / (0.7 * (9 * C) ** 0.5))
    with torch.no_grad():
        opt("dcn_lds", 1); opt("dcn_lds_rows", ROWS[0]); y1 = m(x).float()
        opt("dcn_lds", 0); opt("dcn_patch", 0); opt("dcn_wave", 0); y0 = m(x).float()
    lib.check(L.mfx_reset_options(), "reset")
    torch.cuda.synchronize()
    d = (y1 - y0).abs()
    tol = 3e-2 * y0.abs().clamp(min=1.0)
    bad = int((d > tol).sum())
    print("module B%d %dx%d %d->%d std %.1f %s: max|d| %.4f (ref max %.2f), > tol: %d of %d" %
          (B, H, W, C, Co, std, str(dt).split(".")[-1], float(d.max()), float(y0.abs().max()), bad, d.numel()), flush=True)
    return bad


bad = 0
for rows, dt in ((8, torch.bfloat16), (8, torch.float16), (16, torch.bfloat16), (16, torch.float16)):
    ROWS[0] = rows
    print("---- tile rows", rows)
    bad += run(2, 32, 48, 64, 64, 1.5, dt)
    bad += run(1, 20, 40, 64, 64, 2.5, dt, far_frac=0.02)          # partial tiles + far samples
    bad += run(2, 48, 64, 128, 64, 3.0, dt, far_frac=0.01)         # eight slices
    bad += run(8, 96, 320, 64, 64, 3.0, dt)
    bad += run(1, 16, 16, 64, 64, 12.0, dt)                          # mostly far
    bad += run_module(8, 96, 320, 64, 64, 2.5, dt)
    bad += run_module(1, 24, 40, 64, 64, 2.5, dt)
print("TOTAL out of tolerance:", bad)
sys.exit(1 if bad else 0)
