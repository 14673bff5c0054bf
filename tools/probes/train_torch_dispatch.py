#!/usr/bin/env python
"""Which lines of the package issue torch (aten) ops during one eager training step?  A TorchDispatchMode logs every aten call that touches a
device tensor together with the innermost monoflex_amd frame of the Python stack (custom autograd Functions run their backward in Python, so
they are attributed too; ops issued by autograd's own C++ nodes -- gradient accumulation -- show up as `<autograd engine>`)."""
import os
import sys
import traceback
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from monoflex_amd import lib, synthetic as S
from monoflex_amd.engine.trainer import prepare_targets, train_step
from monoflex_amd.solver import build_optimizer
from monoflex_amd.structures.params_3d import make_train_target

VIEW_OPS = ("view", "reshape", "permute", "transpose", "slice", "select", "unsqueeze", "squeeze", "expand", "as_strided", "detach", "alias", "t.default",
            "unbind", "split", "_unsafe_view", "narrow", "unfold", "size", "stride", "is_", "sym_", "_local_scalar", "empty", "lift_fresh", "_to_copy_meta")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by = defaultdict(int)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(v in name for v in VIEW_OPS):
            return out
        flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
        for a in args:
            if isinstance(a, (list, tuple)):
                flat += [t for t in a if isinstance(t, torch.Tensor)]
        if not any(t.is_cuda for t in flat) and not (isinstance(out, torch.Tensor) and out.is_cuda):
            return out
        frame = "<autograd engine>"
        for fs in reversed(traceback.extract_stack()):
            fn = fs.filename
            if "monoflex_amd" in fn and "probes" not in fn:
                frame = "%s:%d %s" % (fn.split("monoflex_amd/")[-1], fs.lineno, fs.name)
                break
        shp = ",".join(str(tuple(t.shape)) for t in flat[:2])
        self.by[(frame, name.replace("aten.", ""), shp)] += 1
        return out


lib.load()
dev = torch.device("cuda:0")
model, _, cfg = bench.build_model("bf16", dev, train=True)
model.heads.loss_evaluator.log_as_float = False
B = 8
imgs = S.synthetic_images(B, seed=1000).to(dev)
targets = prepare_targets(model, [make_train_target(S.synthetic_train_target(1000 + i)).to(dev) for i in range(B)], dev)
opt = build_optimizer(model, cfg, capturable=True)
for _ in range(2):
    train_step(model, opt, imgs, targets)
torch.cuda.synchronize()
with Log() as log:
    train_step(model, opt, imgs, targets)
torch.cuda.synchronize()
tot = sum(log.by.values())
print("aten calls on device tensors in one step (views excluded): %d" % tot)
agg = defaultdict(int)
for (frame, name, shp), n in log.by.items():
    agg[frame] += n
print("---- by source line")
for frame, n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("%4d  %s" % (n, frame))
print("---- by (line, op, shapes)")
for (frame, name, shp), n in sorted(log.by.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("%4d  %-52s %-28s %s" % (n, frame, name, shp))
