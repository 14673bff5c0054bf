#!/usr/bin/env python
"""Which torch ops launch the at::native kernels of the training step?  One eager step under torch.profiler (with Python stacks), the
device time of every aten op that is not one of the library's own launches, grouped by op and by the innermost monoflex_amd frame."""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from monoflex_amd import lib, synthetic as S
from monoflex_amd.engine.trainer import prepare_targets, train_step
from monoflex_amd.solver import build_optimizer
from monoflex_amd.structures.params_3d import make_train_target

lib.load()
dev = torch.device("cuda:0")
model, _, cfg = bench.build_model("bf16", dev, train=True)
model.heads.loss_evaluator.log_as_float = False
B = 8
imgs = S.synthetic_images(B, seed=1000).to(dev)
targets = prepare_targets(model, [make_train_target(S.synthetic_train_target(1000 + i)).to(dev) for i in range(B)], dev)
opt = build_optimizer(model, cfg, capturable=True)
for _ in range(3):
    train_step(model, opt, imgs, targets)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train_step(model, opt, imgs, targets)
    torch.cuda.synchronize()
by = defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::") or ev.cpu_children:
        pass
    if not ev.name.startswith("aten::"):
        continue
    t = ev.self_device_time_total
    if t <= 0:
        continue
    frame = "?"
    for fr in ev.stack:
        if ("autograd.py" in fr or "trainer.py" in fr or "/model/" in fr or "ops.py" in fr or "loss" in fr) and "torch/" not in fr:
            frame = fr.split("/")[-1][:60]
            break
    k = (ev.name, frame + " " + str(ev.input_shapes)[:70])
    by[k][0] += t
    by[k][1] += 1
tot = sum(v[0] for v in by.values())
print("torch-native device time in one step: %.0f us" % tot)
for (name, frame), (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:140]:
    print("%8.1f us %4d  %-28s %s" % (t, n, name, frame))
