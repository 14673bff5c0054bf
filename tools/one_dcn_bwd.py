#!/usr/bin/env python
"""Time the NHWC DCN backward for one shape.  usage: one_dcn_bwd.py B H W Cin Cout [opts k=v,...] [reps] [off_std]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from monoflex_amd import lib, autograd as AG
B, H, W, Ci, Co = map(int, sys.argv[1:6])
opts = sys.argv[6] if len(sys.argv) > 6 else ""
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
std = float(sys.argv[8]) if len(sys.argv) > 8 else 2.5
L = lib.load()
for kv in filter(None, opts.split(",")):
    a, b = kv.split("=")
    lib.check(L.mfx_set_option(a.encode(), int(b)), "opt")
torch.manual_seed(0)
x = torch.randn(B, H, W, Ci, device="cuda").requires_grad_()
raw = torch.zeros(B, H, W, 32, device="cuda")
raw[..., :18] = torch.randn(B, H, W, 18, device="cuda") * std
raw.requires_grad_()
w = (torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05).requires_grad_()
b = torch.zeros(Co, device="cuda", requires_grad=True)
y = AG.DCNFn.apply(x, raw, w, b, 1, 1, 1)
dy = torch.randn_like(y)
for _ in range(2):
    torch.autograd.grad(y, (x, raw, w, b), dy, retain_graph=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    torch.autograd.grad(y, (x, raw, w, b), dy, retain_graph=True)
e1.record(); torch.cuda.synchronize()
print("dcn bwd %dx%dx%d %d->%d opts[%s]: %.1f us" % (B, H, W, Ci, Co, opts, e0.elapsed_time(e1) * 1e3 / reps))
