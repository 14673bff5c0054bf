import sys, os
sys.path.insert(0, os.getcwd())
import torch
from oracle import monoflex_ref as R
from monoflex_amd import lib as L, autograd as AG
from monoflex_amd.model.backbone.DCNv2.dcn_v2 import DCN
DEV="cuda"
def _nhwc(t): return t.permute(0,2,3,1).contiguous()
def _rel(a,b): return float((a.float().cpu()-b.float().cpu()).abs().max()/b.float().abs().max())
g = torch.Generator().manual_seed(77)
dt=torch.bfloat16
rnd = lambda t: t.to(dt).float()
ref = R.DCN(64, 64)
dev = DCN(64, 64, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
with torch.no_grad():
    ref.weight.copy_(rnd(torch.randn(ref.weight.shape, generator=g) * 0.05))
    ref.bias.copy_(torch.randn(64, generator=g) * 0.1)
    ref.conv_offset_mask.weight.copy_(rnd(torch.randn(ref.conv_offset_mask.weight.shape, generator=g) * (0.3 / 576 ** 0.5)))
    b = torch.randn(27, generator=g) * 1.5
    b[18:] = torch.randn(9, generator=g)
    ref.conv_offset_mask.bias.copy_(b)
dev.load_state_dict(ref.state_dict())
dev = dev.to(DEV).train()
x = rnd(torch.randn(2, 64, 96, 320, generator=g))
xr = x.clone().requires_grad_()
yr = ref(xr)
r = rnd(torch.randn(yr.shape, generator=g))
(yr * r).sum().backward()
lib_ = L.load()
names = ["input"] + [n for n, _ in dev.named_parameters()]
want = [xr.grad] + [dict(ref.named_parameters())[n].grad for n in names[1:]]
for raw16 in ((True, False, True, False) if os.environ.get("MFX_DCN_RAW16", "1") != "0" else (False, False, False)):
    AG._RAW16[0] = raw16
    for form in ("fly", 1, 0, 0):
        L.check(lib_.mfx_set_option(b"dcn_bt_fuse_wgrad", 0 if form == 0 else 1), "opt")
        L.check(lib_.mfx_set_option(b"dcn_bt_fly", {"fly2": 2, "fly": 1}.get(form, 0)), "opt")
        dev.zero_grad(set_to_none=True)
        xd = _nhwc(x).to(DEV).to(dt).requires_grad_()
        yd = dev.forward_nhwc_train(xd)
        (yd.float() * _nhwc(r).to(DEV)).sum().backward()
        torch.cuda.synchronize()
        got = [xd.grad.float().permute(0, 3, 1, 2).cpu()] + [p.grad.float().cpu() for _, p in dev.named_parameters()]
        print(raw16, form, " ".join("%s=%.4f" % (n, _rel(a, w_)) for n, a, w_ in zip(names, got, want)), flush=True)
