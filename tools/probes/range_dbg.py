import sys, torch
sys.path.insert(0, "/root/repo")
from monoflex_amd import lib as L, ops
lib_ = L.load()
x = torch.randn(1, 12, 20, 64, device="cuda")
w3 = (torch.randn(64, 64, 3, 3, device="cuda") / 24)
p3 = ops.pack_conv(w3, ops.F16X2, None, None, stride=1, pad=1, act=1)
print("clean", lib_.mfx_f16x2_range_check(1))
y = ops.conv2d(x, p3); torch.cuda.synchronize()
print("after healthy", lib_.mfx_f16x2_range_check(0))
xb = x.clone(); xb[0, 5, 7, 3] = 7e4
y = ops.conv2d(xb, p3); torch.cuda.synchronize()
print("after bad", lib_.mfx_f16x2_range_check(0), "out max", float(y.abs().max()), "isinf", bool(torch.isinf(y).any()))
for o in (b"halo",):
    L.check(lib_.mfx_set_option(o, 0), "o")
y = ops.conv2d(xb, p3); torch.cuda.synchronize()
print("generic kernel after bad", lib_.mfx_f16x2_range_check(0), float(y.abs().max()))
