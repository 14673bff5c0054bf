#!/usr/bin/env python
"""Run ONE operator of the hot path repeatedly, alone (for rocprofv3 --pmc passes and quick timing).

  one_op.py heads   [--batch 8] [--dtype bf16] [--reps 6]
  one_op.py dcn     B H W Cin Cout [--opts k=v,...] [--reps 10] [--std 1.5]          fused DCN forward (given offsets)
  one_op.py dcnmod  B H W Cin Cout [--opts ...] [--reps 10] [--std 1.5]              DCN module: offset conv + DCN + BN + ReLU
  one_op.py dcnbwd  B H W Cin Cout [--opts ...] [--reps 5]  [--std 1.5] [--dtype bf16]   NHWC DCN backward
  one_op.py conv    B H W Cin Cout [--opts ...] [--reps 10]                          3x3 / stride 1 conv + BN + residual + ReLU (trunk layer)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("op", choices=["heads", "dcn", "dcnmod", "dcnbwd", "conv"])
ap.add_argument("shape", nargs="*", type=int)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--std", type=float, default=1.5)
ap.add_argument("--opts", default="")
ap.add_argument("--eager", action="store_true", help="plain launches, no hipGraph (rocprofv3 --pmc passes: counter collection does not survive a graph capture)")
a = ap.parse_args()

from monoflex_amd import autograd as AG, lib, ops
L = lib.load()
for kv in filter(None, a.opts.split(",")):
    k, v = kv.split("=")
    lib.check(L.mfx_set_option(k.encode(), int(v)), "opt")
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
torch.manual_seed(0)


def timed(fn, what):
    """Average GPU time per call: `reps` calls captured in ONE hipGraph and replayed (no host launch gaps between the kernels)."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    if a.eager:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("%s opts[%s] (eager): %.1f us" % (what, a.opts, e0.elapsed_time(e1) * 1e3 / a.reps))
        return
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(a.reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%s opts[%s]: %.1f us" % (what, a.opts, e0.elapsed_time(e1) * 1e3 / (3 * a.reps)))


if a.op == "heads":
    import bench
    model, _, _ = bench.build_model(a.dtype, torch.device("cuda", 0))
    feat = torch.randn(a.batch, 96, 320, 64, device="cuda").relu().to(model.compute_dtype)
    pk = model.heads.predictor._pack(feat.dtype)
    timed(lambda: ops.heads_fused(feat, pk, planar_classes=3), "heads B=%d %s" % (a.batch, a.dtype))
else:
    B, H, W, Ci, Co = a.shape
    x = torch.randn(B, H, W, Ci, device="cuda").relu().to(dt)
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * (1.0 / (3 * Ci ** 0.5))
    if a.op == "conv":
        r = torch.randn(B, H, W, Co, device="cuda").to(dt)
        p = ops.pack_conv(w, dt, torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda"), stride=1, pad=1, act=1)
        timed(lambda: ops.conv2d(x, p, res=r), "conv3x3 %dx%dx%d %d->%d" % (B, H, W, Ci, Co))
    elif a.op == "dcn":
        om = torch.zeros(B, H, W, 32, device="cuda")
        om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * a.std
        om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
        p = ops.pack_conv(w, dt, torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda"), stride=1, pad=1, act=1)
        ops.add_f16_fragments(p, w)
        timed(lambda: ops.dcn(x, om, p), "dcn %dx%dx%d %d->%d std %.1f" % (B, H, W, Ci, Co, a.std))
    elif a.op == "dcnmod":
        from monoflex_amd.model.backbone.dla_dcn import DeformConv
        m = DeformConv(Ci, Co).eval().cuda()
        torch.nn.init.normal_(m.conv.conv_offset_mask.weight, std=a.std / (0.7 * (9 * Ci) ** 0.5))
        with torch.no_grad():
            timed(lambda: m(x), "dcn module %dx%dx%d %d->%d std %.1f" % (B, H, W, Ci, Co, a.std))
    else:
        xg = x.clone().requires_grad_()
        raw = torch.zeros(B, H, W, 32, device="cuda")
        raw[..., :18] = torch.randn(B, H, W, 18, device="cuda") * a.std
        raw.requires_grad_()
        wg = w.clone().requires_grad_()
        b = torch.zeros(Co, device="cuda", requires_grad=True)
        y = AG.DCNFn.apply(xg, raw, wg, b, 1, 1, 1)
        dy = torch.randn_like(y)
        timed(lambda: torch.autograd.grad(y, (xg, raw, wg, b), dy, retain_graph=True),
              "dcn bwd %dx%dx%d %d->%d %s std %.1f" % (B, H, W, Ci, Co, a.dtype, a.std))
