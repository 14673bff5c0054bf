#!/usr/bin/env python
"""Time the 3x3/s1 wave-private conv variants (option halo = variant + 1) on the DLA trunk shapes, B=8 bf16, and check each
variant's output against the automatic choice.  usage: halo_probe.py [variants comma list]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops
L = lib.load()
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 4, 5, 7, 11, 12, 14, 15, 16, 17, 18, 19, 20, 21, 22]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
SHAPES = [("level2 64->64 @96x320", 96, 320, 64, 64), ("level3 128->128 @48x160", 48, 160, 128, 128),
          ("level4 256->256 @24x80", 24, 80, 256, 256), ("level5 512->512 @12x40", 12, 40, 512, 512),
          ("dcnmain-like 128->64 @48x160", 48, 160, 128, 64)]
dt = torch.bfloat16


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, H, W, Ci, Co in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(8, H, W, Ci, device="cuda").to(dt)
    res = torch.randn(8, H, W, Co, device="cuda").to(dt)
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * (1.0 / (3 * Ci ** 0.5))
    p = ops.pack_conv(w, dt, torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda") * 0.1, stride=1, pad=1, act=1)
    fl = 2.0 * 8 * H * W * Co * Ci * 9
    lib.check(L.mfx_set_option(b"halo", 1), "opt")
    ref = ops.conv2d(x, p, res=res).float()
    out = []
    for v in variants:
        lib.check(L.mfx_set_option(b"halo", v + 1 if v else 1), "opt")
        try:
            y = ops.conv2d(x, p, res=res).float()
            err = float((y - ref).abs().max())
            us = timeit(lambda: ops.conv2d(x, p, res=res))
        except RuntimeError as e:
            out.append("V%d n/a" % v); continue
        out.append("V%d %.1f%s" % (v, us, "" if err < 0.05 else " ERR%.2g" % err))
    lib.check(L.mfx_set_option(b"halo", 1), "opt")
    print("%-30s %5.1f GF | %s" % (name, fl / 1e9, "  ".join(out)), flush=True)
