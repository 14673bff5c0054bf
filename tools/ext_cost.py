"""What the `_ext` boundary (NCHW fp32 in / out, reference src/dcn_v2.h:9-23) costs per call against the NHWC operators the model
itself uses: layout changes, offset/mask packing and weight packing are redone on every call.  64 -> 64 @ 96x320, B = 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoflex_amd import ops, lib as L
from monoflex_amd.model.backbone.DCNv2 import _ext

dev = "cuda"
g = torch.Generator().manual_seed(1)
B, C, H, W = 8, 64, 96, 320
x = torch.randn(B, C, H, W, generator=g).to(dev)
off = (torch.randn(B, 18, H, W, generator=g) * 1.5).to(dev)
msk = torch.sigmoid(torch.randn(B, 9, H, W, generator=g)).to(dev)
w = (torch.randn(C, C, 3, 3, generator=g) / 24).to(dev)
b = torch.zeros(C, device=dev)
go = torch.randn(B, C, H, W, generator=g).to(dev)


def t(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

f_ext = t(lambda: _ext.dcn_v2_forward(x, w, b, off, msk, 3, 3, 1, 1, 1, 1, 1, 1, 1))
b_ext = t(lambda: _ext.dcn_v2_backward(x, w, b, off, msk, go, 3, 3, 1, 1, 1, 1, 1, 1, 1))
xn = x.permute(0, 2, 3, 1).contiguous()
om = torch.zeros(B, H, W, 32, device=dev); om[..., :18] = off.permute(0, 2, 3, 1); om[..., 18:27] = msk.permute(0, 2, 3, 1)
res = {}
for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
    p = ops.pack_conv(w, dt, None, b, stride=1, pad=1, act=0)
    ops.add_f16_fragments(p, w)
    xd = xn.to(dt)
    res[name] = t(lambda: ops.dcn(xd, om, p))
print("_ext.dcn_v2_forward %.0f us, _ext.dcn_v2_backward %.0f us per call (NCHW fp32, 64->64 @ 8x96x320); the NHWC operator on resident, packed "
      "operands: fp32 %.0f us, bf16 %.0f us" % (f_ext, b_ext, res["fp32"], res["bf16"]))
