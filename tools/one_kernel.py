#!/usr/bin/env python
"""Run ONE conv shape repeatedly (for rocprofv3 --pmc).  usage: one_kernel.py B H W Cin Cout k s [opts k=v,...] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops
B, H, W, Ci, Co, k, s = map(int, sys.argv[1:8])
opts = sys.argv[8] if len(sys.argv) > 8 else ""
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 10
L = lib.load()
for kv in filter(None, opts.split(",")):
    a, b = kv.split("=")
    lib.check(L.mfx_set_option(a.encode(), int(b)), "opt")
dt = torch.bfloat16
x = torch.randn(B, H, W, Ci, device="cuda").to(dt)
w = torch.randn(Co, Ci, k, k, device="cuda") * 0.05
p = ops.pack_conv(w, dt, torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda"), stride=s, pad=k // 2, act=1)
for _ in range(reps):
    y = ops.conv2d(x, p)
torch.cuda.synchronize()
