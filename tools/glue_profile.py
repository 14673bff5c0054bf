#!/usr/bin/env python
"""Where do the small torch kernels of one eager training step come from?  torch.profiler grouped by python call site."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from monoflex_amd import synthetic as S
from monoflex_amd.engine.trainer import prepare_targets, train_step
from monoflex_amd.solver import build_optimizer
from monoflex_amd.structures.params_3d import make_train_target
dev = torch.device("cuda", 0)
model, _, cfg = bench.build_model("bf16", dev, train=True)
model.heads.loss_evaluator.log_as_float = False
B = 8
imgs = S.synthetic_images(B, seed=1000).to(dev)
tg = prepare_targets(model, [make_train_target(S.synthetic_train_target(1000 + i)).to(dev) for i in range(B)], dev)
opt = build_optimizer(model, cfg)
for _ in range(2):
    train_step(model, opt, imgs, tg)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step(model, opt, imgs, tg)
    torch.cuda.synchronize()
from collections import Counter
cnt = Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::") or not e.stack:
        continue
    if any(c.name.startswith("aten::") for c in e.cpu_children):            # leaf aten ops only
        continue
    if e.name in ("aten::empty", "aten::empty_like", "aten::empty_strided", "aten::as_strided", "aten::view", "aten::detach", "aten::slice",
                  "aten::select", "aten::permute", "aten::reshape", "aten::_unsafe_view", "aten::expand", "aten::unsqueeze", "aten::squeeze",
                  "aten::t", "aten::transpose", "aten::alias", "aten::resize_", "aten::lift_fresh", "aten::result_type", "aten::is_nonzero"):
        continue
    site = next((fr for fr in e.stack if "/monoflex_amd/" in fr), e.stack[0])
    site = site.split("monoflex_amd/")[-1]
    cnt[(site, e.name)] += 1
print("%-78s %-30s %6s" % ("call site", "op", "calls"))
for k, n in cnt.most_common(70):
    print("%-78s %-30s %6d" % (k[0][:78], k[1][:30], n))
print("total leaf aten ops that touch the device (approx): %d" % sum(cnt.values()))
