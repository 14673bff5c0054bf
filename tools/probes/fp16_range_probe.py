#!/usr/bin/env python
"""Dynamic range of the training step's 16-bit tensors: one forward + backward of the detector at a given loss scale, with a hook on
every autograd node that records the largest |gradient| flowing through 16-bit tensors and the share of their non-zero entries
below fp16's normal range (6.1e-5) -- what decides the loss scale of the fp16 mode (engine.trainer.LossScaler).
usage: fp16_range_probe.py [small|full] [bf16|fp16] [scale]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_train as T

size = sys.argv[1] if len(sys.argv) > 1 else "small"
dt = sys.argv[2] if len(sys.argv) > 2 else "fp16"
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
out_w, out_h, B = (320, 96, 8) if size == "full" else (96, 32, 2)
m, _ = T._models(out_w, out_h)
m.set_compute_dtype(dt)
imgs, _, targets = T._train_batch(B, out_w, out_h)
imgs, targets = imgs.to("cuda"), [t.to("cuda") for t in targets]
loss_dict, _ = m(imgs, targets)
print({k: round(float(v), 4) for k, v in loss_dict.items()})
losses = sum(loss_dict.values()) * scale
rec = []


def walk(fn, seen):
    if fn is None or fn in seen:
        return
    seen.add(fn)
    name = fn.name()

    def hook(grad_inputs, grad_outputs, name=name):
        for g in grad_outputs:
            if g is not None and g.dtype in (torch.float16, torch.bfloat16):
                a = g.float().abs()
                nz = a[a > 0]
                rec.append((name, tuple(g.shape), float(a.max()), bool(torch.isfinite(a).all()),
                            float((nz < 6.1e-5).float().mean()) if nz.numel() else 0.0, float(nz.median()) if nz.numel() else 0.0))
    fn.register_hook(hook)
    for nf, _ in fn.next_functions:
        walk(nf, seen)


sys.setrecursionlimit(100000)
walk(losses.grad_fn, set())
losses.backward()
bad = [(n, p.grad) for n, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print("scale %g: %d 16-bit gradient tensors seen, %d with non-finite entries; %d parameter gradients non-finite" %
      (scale, len(rec), sum(1 for r in rec if not r[3]), len(bad)))
print("largest |g| (first 12 in backward order, then the 12 largest):")
for r in rec[:12]:
    print("  %-28s %-22s max %.3e finite %s  below-normal %.3f  median %.2e" % r)
for r in sorted(rec, key=lambda r: -r[2] if r[2] == r[2] else -1e30)[:12]:
    print("  %-28s %-22s max %.3e finite %s  below-normal %.3f  median %.2e" % r)
print("worst underflow share:")
for r in sorted(rec, key=lambda r: -r[4])[:8]:
    print("  %-28s %-22s max %.3e finite %s  below-normal %.3f  median %.2e" % r)
for n, g in bad[:10]:
    print("non-finite parameter gradient:", n)
