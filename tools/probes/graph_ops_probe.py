"""Which training-mode operator breaks under hipGraph replay after eager allocation churn?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import autograd as AG, lib as L, ops
dev = "cuda"
dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16

def churn():
    t = torch.empty(1 << 30, dtype=torch.uint8, device=dev); t.fill_(77); del t
    ts = [torch.full((1 << 20,), 3.0, device=dev) for _ in range(64)]; del ts
    torch.cuda.synchronize()

def capture(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out

def check(name, fn):
    with torch.no_grad():
        g, out = capture(fn)
    g.replay(); torch.cuda.synchronize()
    ref = [t.float().clone() for t in out]
    churn()
    g.replay(); torch.cuda.synchronize()
    ok = all(float((a - b.float()).abs().max()) <= 1e-3 * max(1.0, float(a.abs().max())) for a, b in zip(ref, out))
    print("%-40s consistent after churn: %s" % (name, ok), flush=True)

x = torch.randn(2, 32, 96, 64, device=dev).to(dt)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
bn = torch.nn.BatchNorm2d(64).to(dev).train()
img = torch.randn(2, 3, 128, 384, device=dev)
ws = torch.randn(16, 3, 7, 7, device=dev) * 0.1
check("AG.conv2d 3x3", lambda: (AG.conv2d(x, w, None, 1, 1),))
check("AG.conv2d 3x3 stride 2", lambda: (AG.conv2d(x, w, None, 2, 1),))
check("AG.conv2d 1x1 + bias fp32 out", lambda: (AG.conv2d(x, w[:, :, :1, :1].contiguous(), torch.ones(64, device=dev), 1, 0, out_dtype=torch.float32),))
check("AG.bn_act relu", lambda: (AG.bn_act(x, bn, L.ACT_RELU),))
check("AG.bn_act relu + res", lambda: (AG.bn_act(x, bn, L.ACT_RELU, x),))
check("StemConvFn", lambda: (AG.StemConvFn.apply(img, ws, dt),))
check("MaxPool2x2Fn", lambda: (AG.MaxPool2x2Fn.apply(x),))
check("CatConv1x1Fn", lambda: (AG.CatConv1x1Fn.apply(torch.randn(64, 128, 1, 1, device=dev) * 0.1, x, x),))
up = torch.rand(64, 1, 4, 4, device=dev)
skip = torch.randn(2, 64, 192, 64, device=dev).to(dt)
check("UpsampleAddFn", lambda: (AG.UpsampleAddFn.apply(x, up, skip, 2),))
raw = torch.randn(2, 32, 96, 32, device=dev)
check("DCNFn forward", lambda: (AG.DCNFn.apply(x, raw, w, torch.zeros(64, device=dev), 1, 1, 1),))
check("pack_conv_weight + conv (manual)", lambda: (ops.conv2d(x, AG._pack_weight(w, dt, 0, 64, 64, 1, 1, 1)),))
