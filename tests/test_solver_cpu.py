"""Learning-rate schedule of the training loop against the REFERENCE's own solver (tests/golden/lr_trace.json, written by
oracle/gen_lr_golden.py from /root/reference/solver: build_optimizer + build_scheduler driven the way engine/trainer.py:116-126
drives them): rising cosine warm-up from BASE_LR / DIV_FACTOR, hand-over to the step decay at the right iteration, bias groups at
BIAS_LR_FACTOR x, with float and with device-scalar (capturable) learning rates."""
import json
import math
import os

import numpy as np
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


class _ToyDetector(torch.nn.Module):
    """The training surface of KeypointDetector: model(images, targets) -> (loss_dict, log_loss_dict)."""

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(6, 3)

    def forward(self, images, targets=None):
        y = self.lin(images)
        return {"a_loss": y.pow(2).mean(), "b_loss": y.abs().mean() * 0.5}, {}


class _Target:
    def to(self, device):
        return self


def test_do_train_loop_on_cpu(tmp_path):
    """engine.trainer.do_train, eager path (CPU tensors never take the graphed step): the reference's loop order -- forward, summed
    loss, backward, optimizer step, THEN the schedule set to the finished iteration (warm-up below WARMUP_STEPS, the step decay after
    it) -- iteration bookkeeping from a resume point, periodic and final checkpoints from rank 0, and a falling loss."""
    from monoflex_amd.config import get_cfg
    from monoflex_amd.engine import trainer as TR
    from monoflex_amd.solver import build_optimizer, build_scheduler
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.SOLVER.MAX_ITERATION, cfg.SOLVER.LR_WARMUP, cfg.SOLVER.WARMUP_STEPS, cfg.SOLVER.STEPS = 12, True, 4, [8]
    cfg.SOLVER.SAVE_CHECKPOINT_INTERVAL, cfg.SOLVER.EVAL_INTERVAL, cfg.SOLVER.BASE_LR = 5, 0, 0.05
    torch.manual_seed(0)
    m = _ToyDetector()
    opt = build_optimizer(m, cfg)
    assert not opt.param_groups[0].get("capturable", False) and isinstance(opt.param_groups[0]["lr"], float)      # CPU: plain AdamW
    sched, warm = build_scheduler(opt, total_iters_each_epoch=4, optim_cfg=cfg.SOLVER)
    g = torch.Generator().manual_seed(1)
    batches = [{"images": torch.randn(4, 6, generator=g), "targets": [_Target()]} for _ in range(20)]
    saved, lrs = [], []

    class Ck:
        def save(self, name, **kw):
            saved.append((name, kw["iteration"]))
    real_step = opt.step
    opt.step = lambda *a, **k: (lrs.append(opt.param_groups[0]["lr"]), real_step(*a, **k))[1]
    args = {"iteration": 2}                                      # resumed after two iterations
    first = float(sum(m(batches[0]["images"])[0].values()))
    loss = TR.do_train(cfg, False, m, batches, None, opt, sched, warm, Ck(), "cpu", args)
    assert args["iteration"] == 12 and len(lrs) == 10 and np.isfinite(loss)
    assert saved == [("model_checkpoint", 5), ("model_checkpoint", 10), ("model_final", 12)]
    eta = cfg.SOLVER.BASE_LR / cfg.SOLVER.DIV_FACTOR
    cosw = lambda t: eta + (cfg.SOLVER.BASE_LR - eta) * (1 - math.cos(math.pi * t / 4)) / 2       # noqa: E731
    # lr used at iteration it (after the schedule was set to it - 1): warm-up values for it - 1 < 4, then BASE_LR, then the decay
    want = [cosw(2), cosw(3)] + [cfg.SOLVER.BASE_LR] * 4 + [cfg.SOLVER.BASE_LR * cfg.SOLVER.LR_DECAY] * 3
    assert lrs[1:] == pytest.approx(want, rel=1e-9), (lrs, want)
    assert float(sum(m(batches[0]["images"])[0].values())) < first


def test_loss_scaler_arithmetic_on_cpu():
    """engine.trainer.LossScaler (fp16 training): the scaled loss back-propagates scaled gradients, `unscale_` restores them and raises
    `found_inf` on a non-finite entry, the fused AdamW then leaves parameters / moments / counters untouched, the scale halves on an
    overflow and doubles after `growth_interval` clean steps; state_dict round trip."""
    from monoflex_amd.engine.trainer import LossScaler
    torch.manual_seed(0)
    m = torch.nn.Linear(4, 3)
    ref = torch.nn.Linear(4, 3)
    ref.load_state_dict(m.state_dict())
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2, fused=True)
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=1e-2, fused=True)
    sc = LossScaler(torch.device("cpu"), init_scale=16.0, growth_interval=2).attach(opt)
    x = torch.randn(5, 4)

    def step(poison=False):
        loss = m(x).pow(2).sum()
        opt.zero_grad()
        sc.scale_loss(loss).backward()
        if poison:
            m.weight.grad[0, 0] = float("inf")
        sc.unscale_([p.grad for p in m.parameters()])
        opt.step()
        sc.update()

    def ref_step():
        opt_ref.zero_grad()
        ref(x).pow(2).sum().backward()
        opt_ref.step()
    step(); ref_step()
    assert torch.allclose(m.weight, ref.weight, atol=1e-7) and sc.state_dict() == {"scale": 16.0, "growth_tracker": 1}
    w, st = m.weight.detach().clone(), [(s["exp_avg"].clone(), int(s["step"])) for s in opt.state.values()]
    step(poison=True)                                              # skipped: nothing moves, the scale halves
    assert torch.equal(w, m.weight) and float(sc.found_inf) == 1.0 and sc.state_dict() == {"scale": 8.0, "growth_tracker": 0}
    assert all(torch.equal(a, s["exp_avg"]) and n == int(s["step"]) for (a, n), s in zip(st, opt.state.values()))
    step(); ref_step()
    step(); ref_step()                                             # second clean step in a row: the scale doubles
    assert torch.allclose(m.weight, ref.weight, atol=1e-6) and sc.state_dict() == {"scale": 16.0, "growth_tracker": 0}
    sc2 = LossScaler(torch.device("cpu"))
    sc2.load_state_dict({"scale": 4.0, "growth_tracker": 7})
    assert sc2.state_dict() == {"scale": 4.0, "growth_tracker": 7}
    assert LossScaler.for_model(m) is None                         # (only an fp16 model gets one)
