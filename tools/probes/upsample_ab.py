#!/usr/bin/env python
"""Element-wise comparison of two builds' upsample_add outputs: run once per build with MFX_LIB_PATH, saves / compares gpurun_out/up_ab.pt."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops
lib.load()
outs = []
for (H, W, C, f) in [(12, 40, 256, 2), (24, 80, 64, 4), (13, 7, 64, 2)]:
    for dt in (torch.bfloat16, torch.float32):
        torch.manual_seed(0)
        x = torch.randn(2, H, W, C, device="cuda").to(dt)
        skip = torch.randn(2, H * f, W * f, C, device="cuda").to(dt)
        w = torch.rand(4 * f * f, C, device="cuda") - 0.3
        outs.append(ops.upsample_add(x, w, f, skip).float().cpu())
        outs.append(ops.upsample_add(x, w, f, None).float().cpu())
p = os.path.join(ROOT, "gpurun_out", "up_ab.pt")
if os.path.exists(p):
    ref = torch.load(p)
    for i, (a, b) in enumerate(zip(outs, ref)):
        d = (a - b).abs()
        print(i, "identical" if torch.equal(a, b) else "DIFFERENT: %d of %d elements, max %.3e" % (int((d > 0).sum()), d.numel(), float(d.max())))
else:
    torch.save(outs, p)
    print("saved")
