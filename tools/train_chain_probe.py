import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monoflex_amd import autograd as AG, lib as L
def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max())
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
g = torch.Generator().manual_seed(0)
B, H, W = 2, 32, 64
x = torch.randn(B, 64, H, W, generator=g)
w1 = torch.randn(256, 64, 3, 3, generator=g) * 0.05
w2 = torch.randn(4, 256, 1, 1, generator=g) * 0.1
b2 = torch.randn(4, generator=g)
bn_r = torch.nn.BatchNorm2d(256); bn_d = torch.nn.BatchNorm2d(256).cuda()
with torch.no_grad():
    bn_r.weight.copy_(torch.rand(256, generator=g) + 0.5); bn_r.bias.copy_(torch.randn(256, generator=g) * 0.1)
bn_d.load_state_dict(bn_r.state_dict())
xr, w1r, w2r, b2r = [t.clone().requires_grad_() for t in (x, w1, w2, b2)]
h = F.conv2d(xr, w1r, padding=1); h.retain_grad()
a = F.leaky_relu(bn_r(h), 0.01); a.retain_grad()
y = F.conv2d(a, w2r, b2r)
r = torch.randn(y.shape, generator=g)
(y * r).sum().backward()
xd, w1d, w2d, b2d = [t.cuda().requires_grad_() for t in (nhwc(x), w1, w2, b2)]
hd = AG.conv2d(xd, w1d, None, 1, 1); hd.retain_grad()
ad = AG.bn_act(hd, bn_d, L.ACT_LEAKY); ad.retain_grad()
yd = AG.conv2d(ad, w2d, b2d, 1, 0)
(yd * nhwc(r).cuda()).sum().backward()
print("y", rel(yd.permute(0, 3, 1, 2), y))
print("da", rel(ad.grad.permute(0, 3, 1, 2), a.grad))
print("dh", rel(hd.grad.permute(0, 3, 1, 2), h.grad))
print("dx", rel(xd.grad.permute(0, 3, 1, 2), xr.grad))
print("dw1", rel(w1d.grad, w1r.grad), "dw2", rel(w2d.grad, w2r.grad), "db2", rel(b2d.grad, b2r.grad))
print("dgamma", rel(bn_d.weight.grad, bn_r.weight.grad), "dbeta", rel(bn_d.bias.grad, bn_r.bias.grad))
