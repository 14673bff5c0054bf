#!/usr/bin/env python
"""A/B of the corner-quad DCN kernel (csrc/dcn_cq.hip) against the library's other DCN kernels on the network's layer shapes: max |diff| of the
outputs (fp32 reference of the same op: torch on the bf16-rounded operands) and microseconds per launch (10 launches per hipGraph replay).
usage: python tools/dcn_cq_probe.py [B=8] [std=3.0] [dtype=bf16|fp16] [dcn_cq=3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops

L = lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
STD = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
dt = torch.float16 if (len(sys.argv) > 3 and sys.argv[3] == "fp16") else torch.bfloat16
CQ = int(sys.argv[4]) if len(sys.argv) > 4 else 3
SHAPES = [(12, 40, 512, 256), (24, 80, 256, 256), (24, 80, 256, 128), (48, 160, 128, 128), (48, 160, 128, 64), (24, 80, 256, 64), (96, 320, 64, 64)]
N = 10


def opt(**kw):
    lib.check(L.mfx_reset_options(), "reset")
    for k, v in kw.items():
        lib.check(L.mfx_set_option(k.encode(), int(v)), "opt")


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


print("| layer (B=%d, std %.1f px, %s) | default us | cq us | max abs diff cq-default | max abs | " % (B, STD, str(dt).split(".")[-1]))
print("|---|---|---|---|---|")
for (H, W, Ci, Co) in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Ci, device="cuda").relu().to(dt)
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * (1.0 / (3 * Ci ** 0.5))
    om = torch.zeros(B, H, W, 32, device="cuda")
    om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * STD
    om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
    p = ops.pack_conv(w, dt, torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda") * 0.1, stride=1, pad=1, act=1)
    ops.add_f16_fragments(p, w)
    opt(dcn_cq=0)
    y0 = ops.dcn(x, om, p).float()
    t0 = timed(lambda: ops.dcn(x, om, p))
    opt(dcn_cq=CQ, dcn_patch=0)
    y1 = ops.dcn(x, om, p).float()
    t1 = timed(lambda: ops.dcn(x, om, p))
    print("| %dx%d %d->%d | %.1f | %.1f | %.3g | %.3g |" % (H, W, Ci, Co, t0, t1, (y1 - y0).abs().max().item(), y0.abs().max().item()), flush=True)
opt()
