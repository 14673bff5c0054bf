#!/usr/bin/env python
"""Run ONE fused-DCN shape repeatedly (for rocprofv3 --pmc / timing).  usage: one_dcn.py B H W Cin Cout [opts k=v,...] [reps] [off_std]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, ops
B, H, W, Ci, Co = map(int, sys.argv[1:6])
opts = sys.argv[6] if len(sys.argv) > 6 else ""
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
std = float(sys.argv[8]) if len(sys.argv) > 8 else 1.5
L = lib.load()
for kv in filter(None, opts.split(",")):
    a, b = kv.split("=")
    lib.check(L.mfx_set_option(a.encode(), int(b)), "opt")
dt = torch.bfloat16
torch.manual_seed(0)
x = torch.randn(B, H, W, Ci, device="cuda").to(dt)
om = torch.zeros(B, H, W, 32, device="cuda")
om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * std
om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
p = ops.pack_conv(w, dt, torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda"), stride=1, pad=1, act=1)
ops.add_f16_fragments(p, w)
for _ in range(3):
    y = ops.dcn(x, om, p)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    y = ops.dcn(x, om, p)
e1.record(); torch.cuda.synchronize()
print("dcn %dx%dx%d %d->%d opts[%s] std %.1f: %.1f us" % (B, H, W, Ci, Co, opts, std, e0.elapsed_time(e1) * 1e3 / reps))
