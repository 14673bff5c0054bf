#!/usr/bin/env python
"""Time single conv / dcn shapes under every tile/kc override (HIP events, back-to-back launches).
usage: python tools/conv_probe.py [--dtype bf16] [--only conv|dcn]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--only", default="")
ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
L = lib.load()
TILES = {1: "256x16", 2: "256x32", 3: "128x64", 4: "64x64", 5: "128x128", 6: "64x128", 7: "256x64", 8: "256x128w8", 9: "128x128w8"}


def setopt(k, v):
    lib.check(L.mfx_set_option(k.encode(), v), "opt")


def timeit(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


CONV = [  # name, B,H,W,Cin,Cout,k,s
    ("level0 16->16 @384x1280", 8, 384, 1280, 16, 16, 3, 1),
    ("level1 16->32 s2", 8, 384, 1280, 16, 32, 3, 2),
    ("level2 64->64 @96x320", 8, 96, 320, 64, 64, 3, 1),
    ("level3 128->128 @48x160", 8, 48, 160, 128, 128, 3, 1),
    ("level4 256->256 @24x80", 8, 24, 80, 256, 256, 3, 1),
    ("level5 512->512 @12x40", 8, 12, 40, 512, 512, 3, 1),
    ("offset 64->32 @96x320", 8, 96, 320, 64, 32, 3, 1),
    ("offset 256->32 @24x80", 8, 24, 80, 256, 32, 3, 1),
    ("offset 128->32 @48x160", 8, 48, 160, 128, 32, 3, 1),
    ("offset 512->32 @12x40", 8, 12, 40, 512, 32, 3, 1),
]
if args.only in ("", "conv"):
    for name, B, H, W, Ci, Co, k, s in CONV:
        x = torch.randn(B, H, W, Ci, device="cuda").to(dt)
        w = torch.randn(Co, Ci, k, k, device="cuda") * 0.05
        p = ops.pack_conv(w, dt, torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda"), stride=s, pad=k // 2, act=1)
        M = B * (H // s) * (W // s)
        fl = 2.0 * M * Co * Ci * k * k
        out = []
        if k == 3 and s == 1:
            for cg in (32, 64, 128, 256):
                if cg > Ci and cg != 32:
                    continue
                setopt("halo_cg", cg)
                for hv in range(1, 9):
                    setopt("halo", hv)
                    try:
                        us = timeit(lambda: ops.conv2d(x, p), args.reps)
                    except RuntimeError:
                        continue
                    out.append((us, "h%s/cg%d" % ("auto" if hv == 1 else "V%d" % (hv - 1), cg)))
            setopt("halo_cg", 0)
        setopt("halo", 0)
        for kc in (4, 8):
            setopt("kc", kc)
            for t in ([0] + list(TILES)):
                setopt("conv_tile", t)
                try:
                    us = timeit(lambda: ops.conv2d(x, p), args.reps)
                except RuntimeError:
                    continue
                out.append((us, "kc%d/%s" % (kc, TILES.get(t, "auto"))))
        setopt("conv_tile", 0); setopt("kc", 0); setopt("halo", 1)
        out.sort()
        print("%-28s M=%-8d best %.1f us %.0f TF | " % (name, M, out[0][0], fl / out[0][0] / 1e6) +
              "  ".join("%s:%.0f" % (n, u) for u, n in out[:9]))

DCN = [("dcn 64->64 @96x320", 8, 96, 320, 64, 64), ("dcn 128->128 @48x160", 8, 48, 160, 128, 128),
       ("dcn 128->64 @48x160", 8, 48, 160, 128, 64), ("dcn 256->256 @24x80", 8, 24, 80, 256, 256),
       ("dcn 512->256 @12x40", 8, 12, 40, 512, 256)]
if args.only in ("", "dcn"):
    for name, B, H, W, Ci, Co in DCN:
        x = torch.randn(B, H, W, Ci, device="cuda").to(dt)
        om = torch.randn(B, H, W, 32, device="cuda") * 1.5
        om[..., 18:27] = torch.sigmoid(om[..., 18:27])
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
        p = ops.pack_conv(w, dt, torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda"), stride=1, pad=1, act=1)
        M = B * H * W
        fl = 2.0 * M * Co * Ci * 9
        out = []
        for wv in range(1, 9):
            setopt("dcn_wave", wv)
            try:
                us = timeit(lambda: ops.dcn(x, om, p), args.reps)
            except RuntimeError:
                continue
            out.append((us, "w%s" % ("auto" if wv == 1 else "V%d" % (wv - 1))))
        setopt("dcn_wave", 0)
        for kc in (4, 8):
            setopt("kc", kc)
            for t in (0, 3, 4, 5, 6):
                setopt("dcn_tile", t)
                try:
                    us = timeit(lambda: ops.dcn(x, om, p), args.reps)
                except RuntimeError:
                    continue
                out.append((us, "kc%d/%s" % (kc, TILES.get(t, "auto"))))
        setopt("dcn_tile", 0); setopt("kc", 0); setopt("dcn_wave", 1)
        out.sort()
        print("%-28s M=%-8d best %.1f us %.0f TF | " % (name, M, out[0][0], fl / out[0][0] / 1e6) +
              "  ".join("%s:%.0f" % (n, u) for u, n in out[:8]))
