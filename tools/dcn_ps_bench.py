#!/usr/bin/env python
"""Project-then-sample DCN, the two launches timed apart (10 launches per hipGraph replay): the 1x1 projection C -> 9*Cout (mfx_conv2d_nhwc) and the
bilinear sampling of the projected map (mfx_dcn_sample_nhwc), per layer shape.  usage: python tools/dcn_ps_bench.py [B=8] [k=v,... library options]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops

L = lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for kv in filter(None, (sys.argv[2] if len(sys.argv) > 2 else "").split(",")):
    k, v = kv.split("=")
    lib.check(L.mfx_set_option(k.encode(), int(v)), "opt")
SHAPES = [(12, 40, 512, 256), (24, 80, 256, 256), (24, 80, 256, 128), (48, 160, 128, 128), (48, 160, 128, 64), (24, 80, 256, 64)]
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


print("| layer (B=%d) | projection us (tiled 1x1 kernel) | TF/s | projection us (gemm_as) | max diff | sampling us | projected map MB | fused gather kernel us | vendor GEMM (yardstick) us |" % B)
print("|---|---|---|---|---|---|---|---|---|")
for (H, W, Ci, Co) in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Ci, device="cuda").relu().to(dt)
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * (1.0 / (3 * Ci ** 0.5))
    om = torch.zeros(B, H, W, 32, device="cuda")
    om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * 3.0
    om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
    p = ops.pack_conv(w, dt, torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda"), stride=1, pad=1, act=1)
    pp = ops.dcn_ps_pack(p)
    proj = ops.conv2d(x, pp)
    y = torch.empty(B, H, W, Co, device="cuda", dtype=dt)
    tp = timed(lambda: ops.conv2d(x, pp))
    pa = torch.empty_like(proj)
    ta = timed(lambda: L.mfx_project_nhwc(x.data_ptr(), pp.w.data_ptr(), pa.data_ptr(), B * H * W, Ci, 9 * Co, Ci, 9 * Co, lib.MFX_BF16, torch.cuda.current_stream().cuda_stream))
    err = float((pa.float() - proj.float()).abs().max())
    ts = timed(lambda: L.mfx_dcn_sample_nhwc(proj.data_ptr(), om.data_ptr(), p.scale.data_ptr(), p.shift.data_ptr(), y.data_ptr(), B, H, W, Co, Co, 1,
                                             lib.MFX_BF16, torch.cuda.current_stream().cuda_stream))
    tk = timed(lambda: ops.dcn(x, om, p))
    x2, w2 = x.view(-1, Ci), pp.w[:, :Ci].t().contiguous()
    out2 = torch.empty(x2.shape[0], w2.shape[1], device="cuda", dtype=dt)
    tv = timed(lambda: torch.matmul(x2, w2, out=out2))         # what a tuned library GEMM does with the same shape (not used by the product)
    gf = 2.0 * B * H * W * 9 * Ci * Co / 1e9
    print("| %dx%d %d->%d | %.1f | %.0f | %.1f | %.4f | %.1f | %.1f | %.1f | %.1f |" % (H, W, Ci, Co, tp, gf / tp * 1e3, ta, err, ts, proj.numel() * 2 / 1e6, tk, tv), flush=True)
