#!/usr/bin/env python
"""The 1x1 "project" convs and Root nodes of DLA levels 2-5 (dla_dcn.py:195-203, 268-276) alone, B = 8, bf16, 10 launches per hipGraph replay: the LDS-tiled
kernels against the HBM time of the algorithmic bytes; with extra library options (k=v,...) a second column times that setting and bit-compares it.
usage: python tools/pointwise_bench.py [B=8] [k=v,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops

L = lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
OPTS = [kv.split("=") for kv in filter(None, (sys.argv[2] if len(sys.argv) > 2 else "").split(","))]
dt = torch.bfloat16
LAYERS = [("level2 project", 96, 320, [32], 64, 0), ("level2 root", 96, 320, [64, 64], 64, 1), ("level3 project", 48, 160, [64], 128, 0),
          ("level3 tree1 root", 48, 160, [128, 128], 128, 1), ("level3 root", 48, 160, [128, 128, 64, 128], 128, 1), ("level4 project", 24, 80, [128], 256, 0),
          ("level4 tree1 root", 24, 80, [256, 256], 256, 1), ("level4 root", 24, 80, [256, 256, 128, 256], 256, 1), ("level5 project", 12, 40, [256], 512, 0),
          ("level5 root", 12, 40, [512, 512, 256], 512, 1)]
N = 10


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


print("| layer (B=%d) | K -> N | default us | with options us | HBM time us | same bits |" % B)
print("|---|---|---|---|---|---|")
tot = [0.0, 0.0, 0.0]
for (name, H, W, chans, Co, act) in LAYERS:
    torch.manual_seed(0)
    xs = [torch.randn(B, H, W, c, device="cuda").relu().to(dt) for c in chans]
    K = sum(chans)
    w = torch.randn(Co, K, 1, 1, device="cuda") / K ** 0.5
    sc, sh = torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda") * 0.1
    if len(chans) == 1:
        p = ops.pack_conv(w, dt, sc, sh, stride=1, pad=0, act=act)
        fn = lambda: ops.conv2d(xs[0], p)
    else:
        p = ops.pack_cat(w, dt, sc, sh, chans, act=act)
        fn = lambda: ops.cat_conv1x1(xs, p)
    res = {}
    for pf in (0, 2):
        lib.check(L.mfx_reset_options(), "reset")
        if pf:
            for k_, v_ in OPTS:
                lib.check(L.mfx_set_option(k_.encode(), int(v_)), "opt")
        y = fn().clone()
        res[pf] = (timed(fn), y)
    floor = (B * H * W * (K + Co) * 2 + Co * K * 2) / 6.3e12 * 1e6
    tot[0] += res[0][0]; tot[1] += res[2][0]; tot[2] += floor
    print("| %s | %d -> %d | %.1f | %.1f | %.1f | %s |" % (name, K, Co, res[0][0], res[2][0], floor, torch.equal(res[0][1].view(torch.int16), res[2][1].view(torch.int16))), flush=True)
print("| all ten | | %.1f | %.1f | %.1f | |" % tuple(tot))
lib.check(L.mfx_reset_options(), "reset")
