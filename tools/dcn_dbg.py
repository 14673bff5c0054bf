import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoflex_amd import lib, ops
L = lib.load()
B, H, W, C, Co, std, v = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7])
torch.manual_seed(0)
x = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
om = torch.zeros(B, H, W, 32, device="cuda")
om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * std
om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
w = torch.randn(Co, C, 3, 3, device="cuda") * 0.05
p = ops.pack_conv(w, torch.bfloat16, None, None, stride=1, pad=1, act=0)
ops.add_f16_fragments(p, w)
lib.check(L.mfx_set_option(b"dcn_patch", 0), "o"); lib.check(L.mfx_set_option(b"dcn_wave", 0), "o")
a = ops.dcn(x, om, p).float()
lib.check(L.mfx_set_option(b"dcn_patch", v), "o")
b = ops.dcn(x, om, p).float()
d = (a - b).abs()
print("max err", float(d.max()), "max ref", float(a.abs().max()))
print("err by row", d.amax(dim=(0, 2, 3)).cpu().numpy().round(2))
print("err by col", d.amax(dim=(0, 1, 3)).cpu().numpy().round(2))
print("err by ch ", d.amax(dim=(0, 1, 2)).cpu().numpy().round(2)[:16])
