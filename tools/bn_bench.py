#!/usr/bin/env python
"""Per-shape timing of the train-mode BatchNorm kernels (forward pair, backward pair) in their two forms, replayed from a
hipGraph (20 calls per replay) so that launch latency of the eager loop is not what is measured.
usage: python tools/bn_bench.py [--batch 8] [--opts k=v,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler
faulthandler.enable()
import torch
from monoflex_amd import autograd as AG
from monoflex_amd import lib as L

B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 8
if "--opts" in sys.argv:
    for kv in sys.argv[sys.argv.index("--opts") + 1].split(","):
        k, v = kv.split("=")
        L.load().mfx_set_option(k.encode(), int(v))
dev = "cuda"
shapes = [(16, 384, 1280), (32, 192, 640), (64, 96, 320), (128, 48, 160), (256, 24, 80), (512, 12, 40)]
N = 20


def graph_time(make):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn = make()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * N) * 1e3


print("| C | HxW | MB(bf16) | fwd fused us | fwd separate us | bwd fused us | bwd separate us | fwd 3-pass @5TB/s | bwd 5-pass @5TB/s |")
for C, H, W in shapes:
    x = torch.randn(B, H, W, C, device=dev).bfloat16()
    r = torch.randn(B, H, W, C, device=dev).bfloat16()
    bn = torch.nn.BatchNorm2d(C).to(dev)
    res = []
    for sep in (False, True):
        AG._BN_SEPARATE[0] = sep
        xd = x.clone().requires_grad_()

        def fwd():
            with torch.no_grad():
                AG.bn_act(xd, bn, L.ACT_RELU)
        tf = graph_time(lambda: fwd)

        def make_bwd():
            y = AG.bn_act(xd, bn, L.ACT_RELU)
            return lambda: torch.autograd.grad(y, xd, r, retain_graph=True)
        tb = graph_time(make_bwd)
        res += [tf, tb]
    AG._BN_SEPARATE[0] = False
    mb = x.numel() * 2 / 1e6
    print("| %d | %dx%d | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f |" % (C, H, W, mb, res[0], res[2], res[1], res[3], 3 * mb / 5.0, 5 * mb / 5.0), flush=True)
