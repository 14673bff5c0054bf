"""Debug aid: per-parameter gradient error of the training path vs the CPU oracle (network order)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_train as T
from monoflex_amd import synthetic as S

out_w, out_h = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m, ref = T._models(out_w, out_h)
imgs = S.synthetic_images(B, out_h * 4, out_w * 4, seed=11)
eidx, elen = T._edges(B, out_w, out_h)
g = torch.Generator().manual_seed(12)
rc = torch.randn(B, 3, out_h, out_w, generator=g)
rr = torch.randn(B, 50, out_h, out_w, generator=g)
taps = {}
om = ref.forward_maps(imgs, eidx.long(), elen.long(), taps)
((taps['cls_logits'] * rc).sum() + (om['reg'] * rr).sum()).backward()
cls, reg = m.forward_train_maps(imgs.cuda(), eidx.cuda(), elen.cuda())
((cls * T._nhwc(rc).cuda()).sum() + (reg * T._nhwc(rr).cuda()).sum()).backward()
print("fwd", T._rel(cls.permute(0, 3, 1, 2), taps['cls_logits']), T._rel(reg.permute(0, 3, 1, 2), om['reg']))
refp = dict(ref.named_parameters())
for n, p in m.named_parameters():
    gr = refp[n].grad
    if gr is None:
        print("%-60s dead" % n); continue
    gm = float(gr.abs().max())
    print("%-60s max %10.3e  err %9.2e  rel %8.1e" % (n, gm, float((p.grad.cpu() - gr).abs().max()), float((p.grad.cpu() - gr).abs().max()) / max(gm, 1e-2)))
