#!/usr/bin/env python
"""oracle/gen_lr_golden.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

Imports the reference's own `solver.build_optimizer` / `solver.build_scheduler` (read-only, from /root/reference) and records
the learning-rate trace of its training loop's scheduler calls (engine/trainer.py:116-126: `optimizer.step()`, then
`warmup_scheduler.step(iteration)` while `iteration < WARMUP_STEPS`, `scheduler.step(iteration)` afterwards) as
tests/golden/lr_trace.json: for LR_WARMUP on and off, the lr of a weight group and of a bias group BEFORE every iteration.
In-process patch: `collections.Iterable` (removed in Python 3.10; the reference's fastai_optim imports it)."""
import collections
import collections.abc
import json
import os
import sys
import warnings

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
collections.Iterable = collections.abc.Iterable
sys.path.insert(0, "/root/reference")
import torch                                                    # noqa: E402
from solver import build_optimizer, build_scheduler             # noqa: E402  (the reference's)

sys.path.insert(0, REPO)
from monoflex_amd.config import get_cfg                         # noqa: E402


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 4, 3)                    # conv.weight, conv.bias -> the two learning rates


def trace(warmup, iters, steps, warmup_steps):
    cfg = get_cfg(os.path.join(REPO, "runs", "monoflex.yaml"))
    cfg.SOLVER.LR_WARMUP, cfg.SOLVER.WARMUP_STEPS, cfg.SOLVER.STEPS, cfg.SOLVER.MAX_ITERATION = warmup, warmup_steps, steps, iters
    m = Tiny()
    opt = build_optimizer(m, cfg)
    sched, warm = build_scheduler(opt, total_iters_each_epoch=10, optim_cfg=cfg.SOLVER)
    warm_iters = cfg.SOLVER.WARMUP_STEPS if warmup else -1
    rows = []
    for it in range(iters):
        rows.append([g["lr"] for g in opt.param_groups])
        for p in m.parameters():
            p.grad = torch.zeros_like(p)
        opt.step()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            (warm if it < warm_iters else sched).step(it)
    return rows


if __name__ == "__main__":
    out = {"meta": "reference solver.build_scheduler driven as engine/trainer.py:116-126 drives it; rows = lr per param group "
                   "(conv.weight, conv.bias) before each iteration", "iters": 40, "steps": [20, 30], "warmup_steps": 12,
           "warmup_on": trace(True, 40, [20, 30], 12), "warmup_off": trace(False, 40, [20, 30], 12)}
    with open(os.path.join(REPO, "tests", "golden", "lr_trace.json"), "w") as f:
        json.dump(out, f)
    print(out["warmup_on"][:14], out["warmup_off"][18:22])
