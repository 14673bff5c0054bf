#!/usr/bin/env python
"""The nine distinct DCN layer shapes of DLAUp / IDAUp (model/backbone/dla_dcn.py:384-452), each as the kernel alone (`dcn`: offsets given) and as the
module (`mod`: offset/mask conv + DCN + BN + ReLU), bf16, 10 launches per hipGraph replay.  Offsets ~ N(0, std) px (the synthetic benchmark
weights give 2.2 .. 7 px).   usage: python tools/dcn_layers_bench.py [B=8] [std=3.0] [k=v,...library options]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops

L = lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
STD = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
for kv in filter(None, (sys.argv[3] if len(sys.argv) > 3 else "").split(",")):
    k, v = kv.split("=")
    if k == "dcn_ps":                                  # host-side switch: DCN as project-then-sample (ops.dcn_ps) in the module column
        ops.DCN_PS[0] = bool(int(v))
        continue
    lib.check(L.mfx_set_option(k.encode(), int(v)), "opt")
SHAPES = [(12, 40, 512, 256, 1), (24, 80, 256, 256, 1), (24, 80, 256, 128, 2), (48, 160, 128, 128, 2), (48, 160, 128, 64, 4), (24, 80, 256, 64, 1),
          (96, 320, 64, 64, 5)]
N = 10
dt = torch.bfloat16


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


from monoflex_amd.model.backbone.dla_dcn import DeformConv
tot_k = tot_m = 0.0
print("| layer (B=%d, std %.1f px) | count | dcn kernel us | module us | GF | TF/s (kernel) |" % (B, STD))
print("|---|---|---|---|---|---|")
for (H, W, Ci, Co, cnt) in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Ci, device="cuda").relu().to(dt)
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * (1.0 / (3 * Ci ** 0.5))
    om = torch.zeros(B, H, W, 32, device="cuda")
    om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * STD
    om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
    p = ops.pack_conv(w, dt, torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda"), stride=1, pad=1, act=1)
    ops.add_f16_fragments(p, w)
    tk = timed(lambda: ops.dcn(x, om, p))
    m = DeformConv(Ci, Co).eval().cuda()
    torch.nn.init.normal_(m.conv.conv_offset_mask.weight, std=STD / (0.7 * (9 * Ci) ** 0.5))
    with torch.no_grad():
        tm = timed(lambda: m(x))
    gf = 2.0 * B * H * W * 9 * Ci * Co / 1e9
    tot_k += cnt * tk; tot_m += cnt * tm
    print("| %dx%d %d->%d | %d | %.1f | %.1f | %.1f | %.0f |" % (H, W, Ci, Co, cnt, tk, tm, gf, gf / tk * 1e3), flush=True)
print("| all 16 | | %.0f | %.0f | | |" % (tot_k, tot_m))
