"""Is the DENSE regression-head backward deterministic?  Same features, same upstream gradient, several runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_train_step as T
from monoflex_amd import autograd as AG, lib as L
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
m = T._model(dtype)
imgs, tg = T._batch(m, B=2)
pr = m.heads.predictor
with torch.no_grad():
    feat = m.backbone.forward_nhwc(imgs) if hasattr(m.backbone, "forward_nhwc") else None
x = feat.detach()
ei, el = tg.edge
torch.manual_seed(0)
B, H, W, _ = x.shape
dreg = torch.zeros(B, H, W, 50, device="cuda")
for i in range(12):
    dreg[i % B, (7 * i) % H, (13 * i) % W] = torch.randn(50, device="cuda")
t = pr.reg_features[3]; head = pr.reg_heads[3][0]
res = []
for run in range(4):
    y = AG.conv2d(x, t[0].weight, None, 1, 1)
    y.retain_grad()
    f = AG.bn_act(y, t[1], L.ACT_LEAKY)
    f.retain_grad()
    o = AG.conv2d(f, head.weight, head.bias, 1, 0, out_dtype=torch.float32)
    for p in (t[0].weight, t[1].weight, t[1].bias, head.weight, head.bias):
        p.grad = None
    (o * dreg[..., 26:29]).sum().backward()
    res.append(dict(df=f.grad.float().clone(), dy=y.grad.float().clone(), dwc=t[0].weight.grad.clone(), dg=t[1].weight.grad.clone(), db=t[1].bias.grad.clone(),
                    dw2=head.weight.grad.clone()))
for k in res[0]:
    print(k, ["%.3e" % float((r[k] - res[0][k]).norm() / res[0][k].norm().clamp(min=1e-30)) for r in res[1:]], "nnz", int((res[0][k] != 0).sum()))
