#!/usr/bin/env python
"""Time decode stage 1 (NMS + per-class top-K) for different strip counts.  usage: topk_probe.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoflex_amd import lib, ops
L = lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
logits = torch.randn(B, 3, 96, 320, device="cuda") * 1.5 - 2.0
hm = torch.zeros(B, 96, 320, 64, device="cuda")
hm[..., :3] = logits.permute(0, 2, 3, 1)
planar = logits.flatten(2).contiguous()
for strips in (1, 2, 4, 8, 12, 16):
    lib.check(L.mfx_set_option(b"topk_strips", strips), "opt")
    for _ in range(5):
        ops.decode_topk(hm, 0, 3, 50, planar=planar)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.decode_topk(hm, 0, 3, 50, planar=planar)
    e1.record(); torch.cuda.synchronize()
    print("strips %2d: %.1f us" % (strips, e0.elapsed_time(e1) * 1e3 / 50), flush=True)
lib.check(L.mfx_set_option(b"topk_strips", 8), "opt")
