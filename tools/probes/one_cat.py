#!/usr/bin/env python
"""One Root-node launch (virtual concat + 1x1 conv + BN + ReLU) repeated, for rocprofv3 PMC passes.  usage: one_cat.py B H W c1,c2,... Cout [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import ops
B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
chans = [int(c) for c in sys.argv[4].split(",")]
Co = int(sys.argv[5]); reps = int(sys.argv[6]) if len(sys.argv) > 6 else 6
xs = [torch.randn(B, H, W, c, device="cuda").relu().bfloat16() for c in chans]
K = sum(chans)
w = torch.randn(Co, K, 1, 1, device="cuda") / K ** 0.5
p = ops.pack_cat(w, torch.bfloat16, torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda") * 0.1, chans, act=1)
for _ in range(reps):
    y = ops.cat_conv1x1(xs, p)
torch.cuda.synchronize()
