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


class MultiTensorAdamW:
    """`optimizer.step()` of a torch.optim.AdamW (fused, capturable, device-scalar learning rates: what `build_optimizer` makes on a GPU) as ONE HIP launch
    over all parameter tensors (csrc/adamw.hip) instead of torch's chunked multi-tensor kernels (25 launches / 345 us of the B = 8 training step).
    The optimizer object stays the owner of everything -- param_groups, `state[p] = {step, exp_avg, exp_avg_sq}` in torch's own layout -- so
    state_dict() / load_state_dict() / LR schedulers / the loss scaler's `found_inf` hook work unchanged; only the arithmetic runs elsewhere.

    The kernel reads a pointer table on the device.  Eagerly the table is rebuilt and uploaded every call (gradients are fresh tensors each step).
    Inside a stream capture the addresses of the step's gradients are known once backward has been captured, but a host-to-device copy cannot be:
    the launch is recorded against tables of its own (allocated by the preceding eager step -- NOT inside the capture: see step() -- and never
    reused by eager steps) and `finish_capture()` fills them once the capture has ended (before the first replay)."""

    def __init__(self, optimizer):
        import weakref
        self._opt = weakref.ref(optimizer)     # (the optimizer holds this object: no reference cycle, or its parameters outlive their last user until a GC pass)
        self.tables = {}          # device -> (descs tensor, prefix tensor, groups tensor, capacity)
        self.pending = []
        self.captured = []
        self.spare = {}           # device -> a table set allocated by the last eager step for the next captured launch

    @staticmethod
    def eligible(optimizer):
        if type(optimizer) is not torch.optim.AdamW:
            return False
        for g in optimizer.param_groups:
            if not g.get("capturable") or g.get("amsgrad") or g.get("maximize") or not torch.is_tensor(g["lr"]) or not g["lr"].is_cuda \
                    or g["lr"].dtype != torch.float32 or g.get("differentiable"):
                return False
            for p in g["params"]:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.is_sparse:
                    return False
        return True

    def _new_tables(self, dev, n, ngroups):
        from . import lib as L
        cap = max(n, sum(len(g["params"]) for g in self._opt().param_groups))
        return (torch.zeros(cap * ctypes_sizeof(L.AdamWDesc), dtype=torch.uint8, device=dev), torch.zeros(cap + 1, dtype=torch.int64, device=dev),
                torch.zeros(max(ngroups, len(self._opt().param_groups)) * ctypes_sizeof(L.AdamWGroup), dtype=torch.uint8, device=dev), cap)

    def _tables(self, dev, n, ngroups):
        from . import lib as L
        t = self.tables.get(dev)
        if t is None or t[3] < n or t[2].numel() < ngroups * ctypes_sizeof(L.AdamWGroup):
            t = self.tables[dev] = self._new_tables(dev, n, ngroups)
        return t

    @torch.no_grad()
    def step(self):
        import numpy as np
        from . import lib as L
        opt = self._opt()
        chunk = L.load().mfx_adamw_chunk_elems()
        by_dev = {}
        for gi, g in enumerate(opt.param_groups):
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = opt.state[p]
                if len(st) == 0:                                       # torch's lazy state initialisation (adamw.py `_init_group`)
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                gr = p.grad
                if gr.dtype != torch.float32 or not gr.is_contiguous() or gr.is_sparse:
                    raise RuntimeError("MultiTensorAdamW: dense contiguous fp32 gradients only")
                by_dev.setdefault(p.device, []).append((gi, p, gr, st))
        found_inf = getattr(opt, "found_inf", None)
        opt._opt_called = True                                         # (what torch's wrapped step() tells the LR schedulers)
        capturing = torch.cuda.is_current_stream_capturing()
        for dev, items in by_dev.items():
            n = len(items)
            descs = (L.AdamWDesc * n)()
            prefix = np.zeros(n + 1, dtype=np.int64)
            for i, (gi, p, gr, st) in enumerate(items):
                d = descs[i]
                d.p, d.g, d.m, d.v, d.step = p.data_ptr(), gr.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()
                d.numel, d.group = p.numel(), gi
                prefix[i + 1] = prefix[i] + (p.numel() + chunk - 1) // chunk
            groups = (L.AdamWGroup * len(opt.param_groups))()
            for gi, g in enumerate(opt.param_groups):
                groups[gi].lr = g["lr"].data_ptr()
                groups[gi].beta1, groups[gi].beta2 = float(g["betas"][0]), float(g["betas"][1])
                groups[gi].eps, groups[gi].weight_decay = float(g["eps"]), float(g["weight_decay"])
            host = (torch.from_numpy(np.frombuffer(bytes(descs), dtype=np.uint8).copy()), torch.from_numpy(prefix),
                    torch.from_numpy(np.frombuffer(bytes(groups), dtype=np.uint8).copy()))
            if capturing:
                # tables of their own for a captured launch: an eager step of the same optimizer later must not overwrite what the replays read.  They
                # were allocated by the last EAGER step (`spare`), outside the graph's memory pool -- a block of that pool is reused by other captured
                # temporaries, whose kernels would overwrite the table at every replay before this launch reads it
                sp = self.spare.pop(dev, None)
                if sp is None or sp[3] < n:
                    raise RuntimeError("MultiTensorAdamW: run one eager optimizer step before capturing one (it allocates the captured launch's tables)")
                dt, pt, gt = sp[:3]
                self.captured.append(sp)
                self.pending.append((dt, pt, gt, host))
            else:
                dt, pt, gt, _ = self._tables(dev, n, len(opt.param_groups))
                self._upload(dt, pt, gt, host)
                if dev not in self.spare:
                    self.spare[dev] = self._new_tables(dev, n, len(opt.param_groups))
            with torch.cuda.device(dev):
                L.check(L.load().mfx_adamw_multi(ctypes_ptr(dt), ctypes_ptr(pt), n, int(prefix[n]), ctypes_ptr(gt),
                                                 ctypes_ptr(found_inf) if found_inf is not None else None,
                                                 ctypes_stream()), "mfx_adamw_multi")

    @staticmethod
    def _upload(dt, pt, gt, host):
        dt[:host[0].numel()].copy_(host[0])
        pt[:host[1].numel()].copy_(host[1])
        gt[:host[2].numel()].copy_(host[2])

    def finish_capture(self):
        """Fill the tables of the launches recorded during a capture (call after the capture has ended, before the first replay)."""
        for dt, pt, gt, host in self.pending:
            self._upload(dt, pt, gt, host)
        self.pending = []
        if torch.cuda.is_available():
            torch.cuda.synchronize()


def ctypes_sizeof(t):
    import ctypes
    return ctypes.sizeof(t)


def ctypes_ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def ctypes_stream():
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


HIP_ADAMW = [__import__("os").environ.get("MFX_HIP_ADAMW", "1") != "0"]      # MFX_HIP_ADAMW=0: torch's own fused AdamW step (A/B)


def optimizer_step(optimizer):
    """`optimizer.step()`; a GPU AdamW built by `build_optimizer` takes the one-launch HIP form (MultiTensorAdamW) -- eagerly and inside captures alike,
    so a replayed step and an eager step stay the same arithmetic."""
    mt = getattr(optimizer, "_mfx_multi", None)
    if mt is None and HIP_ADAMW[0] and MultiTensorAdamW.eligible(optimizer):
        mt = optimizer._mfx_multi = MultiTensorAdamW(optimizer)
    if mt is None or not HIP_ADAMW[0]:
        return optimizer.step()
    return mt.step()


def finish_capture(optimizer):
    mt = getattr(optimizer, "_mfx_multi", None)
    if mt is not None:
        mt.finish_capture()
