"""Optimizer / schedule of the training-step contract (reference solver/__init__.py:10-92, engine/trainer.py:109-126).

AdamW(lr 3e-4, betas (0.9, 0.99), weight decay 1e-5); parameters whose name contains "bias" run at
BASE_LR * BIAS_LR_FACTOR.  The reference builds one param-group per parameter (280 groups -> 280 separate foreach
launches); groups with equal hyper-parameters are arithmetically identical when merged, so two groups are built
(weights / biases), which lets torch run one fused multi-tensor AdamW kernel per group."""
import math
import warnings

import torch


class CosineWarmupLR(torch.optim.lr_scheduler.LRScheduler):
    """The reference's warm-up schedule (solver/learning_schedules_fastai.py:82-91): RISES from `eta_min` at step 0 to each group's
    base learning rate at step T_max along (1 - cos)/2.  (torch's CosineAnnealingLR is the falling half and is not this.)"""

    def __init__(self, optimizer, T_max, eta_min=0.0, last_epoch=-1):
        self.T_max, self.eta_min = T_max, eta_min
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        f = (1.0 - math.cos(math.pi * self.last_epoch / self.T_max)) / 2.0
        return [self.eta_min + (base_lr - self.eta_min) * f for base_lr in self.base_lrs]


def step_scheduler(scheduler, iteration):
    """`scheduler.step(iteration)` as the reference's loop calls it (engine/trainer.py:123-126): the schedule position is SET to the
    iteration just finished (so a resumed run, or the hand-over from the warm-up schedule, lands on the right value) instead of
    advanced by one.  torch keeps that call form but warns about it."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        scheduler.step(iteration)


def get_model_params(model, cfg, per_parameter_groups=False, tensor_lr_device=None):
    base_lr = cfg.SOLVER.BASE_LR
    bias_lr = max(base_lr, base_lr * cfg.SOLVER.BIAS_LR_FACTOR)
    if tensor_lr_device is not None:
        # a captured optimizer step bakes a Python-float lr into the hipGraph; a device scalar is read at replay time, so schedulers
        # (which fill_ it) keep working under GraphedTrainStep
        mk = lambda v: torch.tensor(float(v), dtype=torch.float32, device=tensor_lr_device)      # noqa: E731
    else:
        mk = float
    if per_parameter_groups:                                         # the reference's literal layout
        return [{"params": [p], "lr": mk(bias_lr if "bias" in k else base_lr)} for k, p in model.named_parameters() if p.requires_grad]
    w = [p for k, p in model.named_parameters() if p.requires_grad and "bias" not in k]
    b = [p for k, p in model.named_parameters() if p.requires_grad and "bias" in k]
    return [{"params": w, "lr": mk(base_lr)}, {"params": b, "lr": mk(bias_lr)}]


def build_optimizer(model, cfg, per_parameter_groups=False, capturable=None):
    """`capturable=None`: capturable (device-side step counters and learning rates) whenever the model lives on a GPU, so that the
    training loop can replay the step from hipGraphs (engine/trainer.do_train); the arithmetic is AdamW's either way."""
    s = cfg.SOLVER
    dev = next((p.device for p in model.parameters() if p.is_cuda), None)
    if capturable is None:
        capturable = dev is not None and s.OPTIMIZER == "adamw"
    tensor_lr = capturable and dev is not None and s.OPTIMIZER == "adamw"
    params = get_model_params(model, cfg, per_parameter_groups, tensor_lr_device=dev if tensor_lr else None)
    on_gpu = dev is not None
    if s.OPTIMIZER == "adamw":
        lr = torch.tensor(float(s.BASE_LR), dtype=torch.float32, device=dev) if tensor_lr else s.BASE_LR
        return torch.optim.AdamW(params, lr=lr, weight_decay=s.WEIGHT_DECAY, betas=(0.9, 0.99), fused=on_gpu or None,
                                 capturable=capturable)
    if s.OPTIMIZER == "adam":
        return torch.optim.Adam(params, lr=s.BASE_LR, weight_decay=s.WEIGHT_DECAY, betas=(0.9, 0.99), fused=on_gpu or None)
    if s.OPTIMIZER == "sgd":
        return torch.optim.SGD(params, lr=s.BASE_LR, weight_decay=s.WEIGHT_DECAY, momentum=s.get("MOMENTUM", 0.9))
    raise NotImplementedError("SOLVER.OPTIMIZER %r (adam_onecycle's fastai wrapper is not part of the adamw training contract)" % s.OPTIMIZER)


def build_scheduler(optimizer, cfg=None, iters_per_epoch=1, last_epoch=-1, total_iters_each_epoch=None, optim_cfg=None):
    """Step decay by LR_DECAY (solver/__init__.py:64-92), stepped per iteration.

    Two call forms: this build's `build_scheduler(optimizer, cfg, iters_per_epoch)` -> LambdaLR decaying at
    DECAY_EPOCH_STEPS * iters_per_epoch; and the reference's `build_scheduler(optimizer, total_iters_each_epoch=...,
    optim_cfg=cfg.SOLVER)` -> `(scheduler, warmup_scheduler)` decaying at optim_cfg.STEPS (iterations, set by the entry
    script from the epochs), clipped at LR_CLIP / BASE_LR, with the cosine warm-up when LR_WARMUP is on."""
    if optim_cfg is not None:
        steps, decay = list(optim_cfg.STEPS), optim_cfg.LR_DECAY
        floor = optim_cfg.LR_CLIP / optim_cfg.BASE_LR

        def lr_lbmd(it):
            f = 1.0
            for s in steps:
                if it >= s:
                    f *= decay
            return max(f, floor)
        sched = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lbmd, last_epoch=last_epoch)
        warm = None
        if optim_cfg.LR_WARMUP:
            warm = CosineWarmupLR(optimizer, T_max=optim_cfg.WARMUP_STEPS, eta_min=optim_cfg.BASE_LR / optim_cfg.DIV_FACTOR)
        return sched, warm
    steps = [e * iters_per_epoch for e in cfg.SOLVER.DECAY_EPOCH_STEPS]
    decay = cfg.SOLVER.LR_DECAY

    def factor(it):
        f = 1.0
        for s in steps:
            if it >= s:
                f *= decay
        return f
    return torch.optim.lr_scheduler.LambdaLR(optimizer, factor, last_epoch=last_epoch)
