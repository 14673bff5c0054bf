#!/usr/bin/env python
"""Where does the training-mode forward in bf16 leave the fp32 oracle?  Full size (B=2, 1280x384), train-mode BN: relative
error of the backbone feature map and of every head output group, HIP fp32 and HIP bf16, against oracle/monoflex_ref.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from monoflex_amd import synthetic as S
import test_gpu_train as T

B = 2
m, ref = T._models(320, 96)
tg = [S.synthetic_train_target(1000 + i) for i in range(B)]
imgs = S.synthetic_images(B, seed=1000)
ei = torch.stack([torch.as_tensor(t["edge_indices"]) for t in tg])
el = torch.as_tensor([int(t["edge_len"]) for t in tg])
taps = {}
with torch.no_grad():
    om = ref.forward_maps(imgs, ei, el, taps)
rfeat, rcls, rreg = taps["feature"], taps["cls_logits"], om["reg"]


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())


names = ["2d_dim 0:4", "3d_offset 4:6", "corner_offset 6:26", "corner_unc 26:29", "3d_dim 29:32", "ori_cls 32:40", "ori_off 40:48", "depth 48", "depth_unc 49"]
spans = [(0, 4), (4, 6), (6, 26), (26, 29), (29, 32), (32, 40), (40, 48), (48, 49), (49, 50)]
for dt in ("fp32", "bf16"):
    m.set_compute_dtype(dt)
    with torch.no_grad():
        feat = m.backbone.forward_nhwc(imgs.cuda())
        cls, reg = m.heads.predictor.forward_train(feat, ei.cuda().int(), el.cuda().int())
    f = feat.float().permute(0, 3, 1, 2).cpu()
    c = cls.float().permute(0, 3, 1, 2).cpu()
    r = reg.float().permute(0, 3, 1, 2).cpu()
    print(dt, "feature  max-rel %.3e l2-rel %.3e" % rel(f, rfeat), " | feature abs max", float(rfeat.abs().max()))
    print(dt, "cls      max-rel %.3e l2-rel %.3e" % rel(c, rcls))
    for n, (a, b) in zip(names, spans):
        d = (r[:, a:b] - rreg[:, a:b]).abs()
        idx = int(d.argmax())
        print(dt, "reg %-20s max-rel %.3e l2-rel %.3e  (ref max %.3f, worst at flat %d)" % ((n,) + rel(r[:, a:b], rreg[:, a:b]) + (float(rreg[:, a:b].abs().max()), idx)))

# ---- per-stage: the six DLA levels in bf16 training mode, default and with MFX_CONV_STATS off / deterministic reductions
from monoflex_amd import lib as L, autograd as AG
m.set_compute_dtype("bf16")
for label, det, stats_off in (("default", 0, False), ("conv-epilogue statistics off", 0, True), ("deterministic", 1, False)):
    L.set_deterministic(bool(det))
    AG._CONV_STATS_OFF[0] = stats_off
    with torch.no_grad():
        ys = m.backbone.base(imgs.cuda(), torch.bfloat16)
    print(label, " ".join("level%d l2-rel %.3e" % (i, rel(y.float().permute(0, 3, 1, 2).cpu(), taps["base"][i])[1]) for i, y in enumerate(ys)))
L.set_deterministic(False)
AG._CONV_STATS_OFF[0] = False
