import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoflex_amd import lib, ops
L = lib.load()
B, H, W, C, Co = 1, 16, 16, 64, 64
torch.manual_seed(0)
x = (torch.arange(H).view(1, H, 1, 1) * 100 + torch.arange(W).view(1, 1, W, 1) + torch.arange(C).view(1, 1, 1, C) * 0.001).expand(B, H, W, C).contiguous().cuda().to(torch.bfloat16)
x = (torch.arange(H).view(1, H, 1, 1) * 16 + torch.arange(W).view(1, 1, W, 1)).expand(B, H, W, C).float().contiguous().cuda()
if len(sys.argv) > 2 and int(sys.argv[2]):
    x = torch.arange(C).view(1, 1, 1, C).expand(B, H, W, C).float().contiguous().cuda()
x = x.to(torch.bfloat16)
om = torch.zeros(B, H, W, 32, device="cuda")
om[..., 18:27] = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tap = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w = torch.zeros(Co, C, 3, 3, device="cuda")
for n in range(Co):
    w[n, (n + (int(sys.argv[2]) if len(sys.argv) > 2 else 0)) % C, tap // 3, tap % 3] = 1.0
p = ops.pack_conv(w, torch.bfloat16, None, None, stride=1, pad=1, act=0)
ops.add_f16_fragments(p, w)
lib.check(L.mfx_set_option(b"dcn_patch", 0), "o"); lib.check(L.mfx_set_option(b"dcn_wave", 0), "o")
a = ops.dcn(x, om, p).float()
lib.check(L.mfx_set_option(b"dcn_patch", 2), "o")
b = ops.dcn(x, om, p).float()
torch.set_printoptions(linewidth=250, precision=0, sci_mode=False)
print("max err", float((a - b).abs().max()))
print("ref row0 ch0..", a[0, 0, :, 0].cpu(), "\n ref px(0,0) channels", a[0, 0, 0, :].cpu())
print("got row0 ch0..", b[0, 0, :, 0].cpu(), "\n got px(0,0) channels", b[0, 0, 0, :].cpu())
