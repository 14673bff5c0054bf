#!/usr/bin/env python
"""dcn_patch variants on the 64->64 full-resolution layer (B=8, 96x320, bf16), interleaved in one process: option dcn_patch = 1
(automatic: swizzled +-7 px patch) vs 8 (padded layout, owner-computed corner base).  Same arithmetic: outputs must be bit-identical.
  usage (GPU box): python tools/probes/dcn_patch_probe.py [std] [rounds]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from monoflex_amd import lib, ops

L = lib.load()
std = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, H, W, Ci, Co = 8, 96, 320, 64, 64
torch.manual_seed(0)
x = torch.randn(B, H, W, Ci, device="cuda").relu().to(torch.bfloat16)
w = torch.randn(Co, Ci, 3, 3, device="cuda") * (1.0 / (3 * Ci ** 0.5))
om = torch.zeros(B, H, W, 32, device="cuda")
om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * std
om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
p = ops.pack_conv(w, torch.bfloat16, torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda"), stride=1, pad=1, act=1)
ops.add_f16_fragments(p, w)


def run(v):
    lib.check(L.mfx_set_option(b"dcn_patch", v), "opt")
    return ops.dcn(x, om, p)


def timed(v, reps=10):
    run(v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(v)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


a, b = run(1).clone(), run(8).clone()
print("padded variant bit-identical:", torch.equal(a, b), " max |diff| %.3g" % float((a.float() - b.float()).abs().max()))
t = {1: [], 8: []}
for _ in range(rounds):
    for v in (1, 8):
        t[v].append(timed(v))
run(1)
for v in (1, 8):
    s = sorted(t[v])
    print("dcn_patch=%d  std %.1f: median %.1f us (min %.1f max %.1f)" % (v, std, s[len(s) // 2], s[0], s[-1]))
