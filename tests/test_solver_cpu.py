"""Learning-rate schedule of the training loop against the REFERENCE's own solver (tests/golden/lr_trace.json, written by
oracle/gen_lr_golden.py from /root/reference/solver: build_optimizer + build_scheduler driven the way engine/trainer.py:116-126
drives them): rising cosine warm-up from BASE_LR / DIV_FACTOR, hand-over to the step decay at the right iteration, bias groups at
BIAS_LR_FACTOR x, with float and with device-scalar (capturable) learning rates."""
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "lr_trace.json")))


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 4, 3)


def _trace(warmup, tensor_lr):
    from monoflex_amd.config import get_cfg
    from monoflex_amd.engine.trainer import advance_schedule
    from monoflex_amd.solver import build_optimizer, build_scheduler, get_model_params
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.SOLVER.LR_WARMUP, cfg.SOLVER.WARMUP_STEPS = warmup, GOLD["warmup_steps"]
    cfg.SOLVER.STEPS, cfg.SOLVER.MAX_ITERATION = GOLD["steps"], GOLD["iters"]
    m = Tiny()
    if tensor_lr:                                               # what build_optimizer(capturable=True) builds on a GPU
        opt = torch.optim.AdamW(get_model_params(m, cfg, tensor_lr_device="cpu"), lr=torch.tensor(cfg.SOLVER.BASE_LR),
                                weight_decay=cfg.SOLVER.WEIGHT_DECAY, betas=(0.9, 0.99))
    else:
        opt = build_optimizer(m, cfg)
    lr_objects = [g["lr"] for g in opt.param_groups]
    sched, warm = build_scheduler(opt, total_iters_each_epoch=10, optim_cfg=cfg.SOLVER)
    assert (warm is not None) == warmup
    warm_iters = cfg.SOLVER.WARMUP_STEPS if warmup else -1
    rows = []
    for it in range(GOLD["iters"]):
        rows.append([float(g["lr"]) for g in opt.param_groups])
        for p in m.parameters():
            p.grad = torch.zeros_like(p)
        opt.step()
        advance_schedule(it, warm_iters, sched, warm)
    if tensor_lr:                                               # the schedulers fill_ the SAME device scalars a captured graph reads
        assert all(a is g["lr"] for a, g in zip(lr_objects, opt.param_groups))
    return rows


@pytest.mark.parametrize("tensor_lr", [False, True])
@pytest.mark.parametrize("warmup", [True, False])
def test_lr_trace_matches_the_reference_solver(warmup, tensor_lr):
    want = GOLD["warmup_on" if warmup else "warmup_off"]
    got = _trace(warmup, tensor_lr)
    assert len(got) == len(want)
    for it, (a, b) in enumerate(zip(got, want)):
        assert a == pytest.approx(b, rel=2e-6 if tensor_lr else 1e-12), (it, a, b)
    if warmup:
        assert want[0][0] < want[6][0] < want[13][0]            # the warm-up RISES (the fixture itself, as a sanity anchor)


def test_reference_checkpoint_keeps_this_builds_optimizer_flags():
    """A reference-layout optimizer state (one group per parameter, capturable=False, fused=None, float lr) loaded into an
    optimizer built with device-scalar learning rates: numeric hyper-parameters come from the file, the lr tensors stay the same
    objects (a captured graph reads them), construction flags stay this build's."""
    from monoflex_amd.config import get_cfg
    from monoflex_amd.solver import get_model_params
    from monoflex_amd.utils.check_point import load_optimizer_state, optimizer_state_from_reference
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    m = Tiny()
    ref_opt = torch.optim.AdamW(get_model_params(m, cfg, per_parameter_groups=True), lr=1e-4, betas=(0.9, 0.99))
    for g in ref_opt.param_groups:
        g["lr"] = g["lr"] * 0.1                                  # a decayed schedule position in the file
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    ref_opt.step()
    sd = ref_opt.state_dict()
    for g in sd["param_groups"]:
        g["capturable"], g["fused"], g["foreach"] = True, True, None          # flags that are NOT this build's
    opt = torch.optim.AdamW(get_model_params(m, cfg, tensor_lr_device="cpu"), lr=torch.tensor(cfg.SOLVER.BASE_LR), betas=(0.9, 0.99),
                            foreach=False)
    lrs = [g["lr"] for g in opt.param_groups]
    mapped = optimizer_state_from_reference(m, opt, sd)
    load_optimizer_state(opt, mapped)
    for g, lr, want in zip(opt.param_groups, lrs, (cfg.SOLVER.BASE_LR * 0.1, cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR * 0.1)):
        assert g["lr"] is lr and float(lr) == pytest.approx(want, rel=1e-6)
        assert g["foreach"] is False and g["capturable"] is False and g["fused"] is None
    assert len(opt.state) == 2 and all("exp_avg" in s for s in opt.state.values())
