"""pytest -q tools/probes/syncbn_flaky_dbg.py : the bitwise graph-vs-eager tests followed by a diagnostic copy of the SyncBN one (r06: order-dependent mismatch)."""
import copy
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_train_step import *                                           # noqa: F401,F403  (fixtures and the tests that run in front)
from test_gpu_train_step import _batch, _cfg, _model


def test_zz_syncbn_diag(nccl_world1, deterministic):
    from monoflex_amd import autograd as AG
    from monoflex_amd.engine.trainer import GraphedTrainStep, convert_sync_batchnorm, train_step, total_loss
    from monoflex_amd.solver import build_optimizer
    cfg = _cfg("bf16")
    AG._SYNC_BN_FORCE[0] = True
    try:
        b = _model("bf16")
        convert_sync_batchnorm(b)
        imgs, tg = _batch(b)
        opt_b = build_optimizer(b, cfg, capturable=True)
        step = GraphedTrainStep(b, opt_b, imgs, tg, warmup=2)
        torch.cuda.synchronize()
        a = _model("bf16", seed=5)
        convert_sync_batchnorm(a)
        a.load_state_dict({k: v.detach().clone() for k, v in b.state_dict().items()})
        opt_a = build_optimizer(a, cfg, capturable=True)
        opt_a.load_state_dict(copy.deepcopy(opt_b.state_dict()))
        sa, sb = a.state_dict(), b.state_dict()
        print("state equal before:", all(torch.equal(sa[k], sb[k]) for k in sa))
        msd = {k: v.detach().clone() for k, v in a.state_dict().items()}
        osd = copy.deepcopy(opt_a.state_dict())
        sums = []

        def hook(name):
            def h(mod, inp, out):
                t = out if torch.is_tensor(out) else None
                if t is not None:
                    sums[-1].append((name, float(t.detach().float().abs().sum()), tuple(t.shape)))
            return h
        hs = [mod.register_forward_hook(hook(n)) for n, mod in a.named_modules() if n]
        loss_b = step().clone()                                  # the failing order: b's replay first ...
        torch.cuda.synchronize()
        print("b replay loss", float(loss_b))
        for rep in range(3):                                     # ... then a's FIRST eager step, and two more from the same state
            sums.append([])
            loss_a = train_step(a, opt_a, imgs, tg)[0]
            torch.cuda.synchronize()
            print("a eager step", rep, "loss", float(loss_a))
            a.load_state_dict(msd)
            opt_a.load_state_dict(copy.deepcopy(osd))
        for h_ in hs:
            h_.remove()
        for rep in (1, 2):
            d = [(x[0], x[1], y[1], x[2]) for x, y in zip(sums[0], sums[rep]) if x[1] != y[1]]
            print("modules whose output differs between eager step 0 and", rep, ":", len(d), d[:4])
    finally:
        AG._SYNC_BN_FORCE[0] = False
