"""Checkpoint save / resume with the reference's on-disk format (reference utils/check_point.py:11-140).

A checkpoint is one `torch.save`d dict {"model": state_dict, "optimizer": ..., "scheduler": ..., **extras} named
`<save_dir>/<name>.pth`; `<save_dir>/last_checkpoint` holds the path of the newest one and wins over an explicit
file when `use_latest` is set. Files written by the reference load here and vice versa: parameter names and OIHW weight
layout are the reference's, and the optimizer / scheduler state is written in the reference's one-group-per-parameter
layout and mapped by parameter name onto this build's merged groups when read (see below). Weights-only files (a bare state_dict) are accepted
by `DetectronCheckpointer`. There is no network in this build, so `catalog://` and `http(s)://` sources raise.
"""
import copy
import logging
import os

import torch

from .model_serialization import load_state_dict

LAST = "last_checkpoint"
FC_NAMES = ("backbone.base.fc.weight", "backbone.base.fc.bias")     # exist in the reference when MODEL.PRETRAIN builds DLA's classifier


# ---- optimizer / scheduler state: reference layout <-> this build's merged groups -------------------------------------
# The reference optimizer has ONE param group per parameter, in `model.named_parameters()` order (solver/__init__.py:10-25:
# 280 groups, 282 with backbone.base.fc.*); solver.build_optimizer here merges them into [weights, biases] so that one fused
# multi-tensor AdamW launch covers a group.  Arithmetic is identical, only the bookkeeping differs, so checkpoints are
# written in the reference's layout and either layout is accepted on load (mapped by parameter NAME).
def _trainable_names(model):
    return [n for n, p in model.named_parameters() if p.requires_grad]


def _reference_names(model, n_groups):
    names = _trainable_names(model)
    if n_groups == len(names):
        return names
    if n_groups == len(names) + 2 and not any(n in names for n in FC_NAMES):
        last_base = max(i for i, n in enumerate(names) if n.startswith("backbone.base."))
        return names[:last_base + 1] + list(FC_NAMES) + names[last_base + 1:]
    return None


def _local_index(model, optimizer):
    """parameter name -> (state index, group index) in `optimizer` (whatever its grouping)."""
    by_id = {id(p): n for n, p in model.named_parameters()}
    out, i = {}, 0
    for gi, g in enumerate(optimizer.param_groups):
        for p in g["params"]:
            out[by_id[id(p)]] = (i, gi)
            i += 1
    return out


def optimizer_state_to_reference(model, optimizer):
    """optimizer.state_dict() re-expressed with one param group per parameter in named_parameters() order."""
    sd = optimizer.state_dict()
    names = _trainable_names(model)
    if len(sd["param_groups"]) == len(names):
        return sd
    loc = _local_index(model, optimizer)
    groups, state = [], {}
    for j, n in enumerate(names):
        i, gi = loc[n]
        g = {k: _scalar(v) for k, v in sd["param_groups"][gi].items() if k != "params"}     # (device-scalar lr -> float: the reference's files hold floats)
        g["params"] = [j]
        groups.append(g)
        if i in sd["state"]:
            state[j] = sd["state"][i]
    return {"state": state, "param_groups": groups}


NUMERIC_HYPER = ("lr", "betas", "eps", "weight_decay", "amsgrad", "initial_lr", "momentum", "dampening", "nesterov", "maximize")


def _scalar(v):
    return float(v) if torch.is_tensor(v) else copy.deepcopy(v)


def _keep_build_flags(mine_group, saved_group):
    """The saved group's numeric hyper-parameters over THIS build's group: `capturable`, `fused`, `foreach`, `differentiable`
    describe how the optimizer was constructed here (a reference checkpoint carries capturable=False / fused=None, which would
    switch a capturable fused optimizer off and break GraphedTrainStep), so they are never taken from the file."""
    g = dict(mine_group)
    for k in NUMERIC_HYPER:
        if k in saved_group:
            g[k] = _scalar(saved_group[k])
    return g


def load_optimizer_state(optimizer, sd):
    """optimizer.load_state_dict(sd) that keeps device-scalar learning rates (solver.build_optimizer(capturable=True)) as the same
    tensor objects a captured graph reads, filled with the loaded values."""
    lr_tensors = [(g["lr"], g.get("initial_lr")) for g in optimizer.param_groups]
    optimizer.load_state_dict(sd)
    for g, (lr, init) in zip(optimizer.param_groups, lr_tensors):
        if torch.is_tensor(lr):
            lr.fill_(float(g["lr"]))
            g["lr"] = lr
            if torch.is_tensor(init) and "initial_lr" in g:
                init.fill_(float(g["initial_lr"]))
                g["initial_lr"] = init


def optimizer_state_from_reference(model, optimizer, sd):
    """A per-parameter-group state (the reference's layout) mapped by name onto `optimizer`'s own groups; entries of
    parameters this model does not have (backbone.base.fc.*) are dropped.  Returns None when the layout is not recognised."""
    mine = optimizer.state_dict()
    if len(sd["param_groups"]) == len(mine["param_groups"]):
        groups = [_keep_build_flags(m, g) for m, g in zip(mine["param_groups"], sd["param_groups"])]
        return {"state": sd["state"], "param_groups": groups}
    ref_names = _reference_names(model, len(sd["param_groups"]))
    if ref_names is None or any(len(g["params"]) != 1 for g in sd["param_groups"]):
        return None
    loc = _local_index(model, optimizer)
    ref_pos = {n: j for j, n in enumerate(ref_names)}
    groups = [dict(g) for g in mine["param_groups"]]
    seen = set()
    state = {}
    for n, (i, gi) in loc.items():
        j = ref_pos[n]
        src = sd["param_groups"][j]
        if gi not in seen:                                   # hyper-parameters of a merged group: those of its first member
            seen.add(gi)
            groups[gi] = _keep_build_flags(groups[gi], src)
        pid = src["params"][0]
        if pid in sd["state"]:
            state[i] = sd["state"][pid]
    return {"state": state, "param_groups": groups}


def _per_group_lists(sched_sd):
    return [k for k, v in sched_sd.items() if isinstance(v, (list, tuple)) and k in ("base_lrs", "_last_lr")]


def scheduler_state_to_reference(model, optimizer, sched_sd):
    names = _trainable_names(model)
    loc = _local_index(model, optimizer)
    out = dict(sched_sd)
    for k in _per_group_lists(sched_sd):
        if len(sched_sd[k]) == len(optimizer.param_groups) != len(names):
            out[k] = [sched_sd[k][loc[n][1]] for n in names]
        # device-scalar learning rates (solver.build_optimizer(capturable=True)) -> floats: the reference's files hold plain numbers
        # (a tensor lr would reach a non-capturable AdamW there, and the file would need map_location to load)
        out[k] = [_scalar(v) for v in out[k]]
    return out


def scheduler_state_from_reference(model, optimizer, sched_sd):
    out = dict(sched_sd)
    loc = _local_index(model, optimizer)
    for k in _per_group_lists(sched_sd):
        if len(sched_sd[k]) == len(optimizer.param_groups):
            continue
        ref_names = _reference_names(model, len(sched_sd[k]))
        if ref_names is None:
            return None
        pos = {n: j for j, n in enumerate(ref_names)}
        vals = [None] * len(optimizer.param_groups)
        for n, (_, gi) in loc.items():
            if vals[gi] is None:
                vals[gi] = sched_sd[k][pos[n]]
        out[k] = vals
    return out


class Checkpointer:
    def __init__(self, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.save_dir, self.save_to_disk = save_dir, save_to_disk
        self.logger = logger or logging.getLogger("monoflex.checkpointer")
        self.load_optimizer_scheduler = True

    # ---- writing ---------------------------------------------------------------------------------------
    def save(self, name, **extras):
        data = {"model": self.model.state_dict()}
        m = self.model.module if hasattr(self.model, "module") else self.model
        if self.optimizer is not None:
            data["optimizer"] = optimizer_state_to_reference(m, self.optimizer)
        if self.scheduler is not None and hasattr(self.scheduler, "state_dict"):
            sd = self.scheduler.state_dict()
            data["scheduler"] = scheduler_state_to_reference(m, self.optimizer, sd) if self.optimizer is not None else sd
        data.update(extras)
        path = os.path.join(self.save_dir, name + ".pth")
        self.logger.info("Saving checkpoint to %s", path)
        torch.save(data, path)
        self.tag_last_checkpoint(path)
        return path

    def tag_last_checkpoint(self, last_filename):
        with open(os.path.join(self.save_dir, LAST), "w") as f:
            f.write(last_filename)

    # ---- reading ---------------------------------------------------------------------------------------
    def has_checkpoint(self):
        return os.path.exists(os.path.join(self.save_dir, LAST))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, LAST)) as f:
                return f.read().strip()
        except IOError:                                          # removed by another process between the two calls
            return ""

    def load(self, f=None, use_latest=True):
        """Returns what is left of the checkpoint dict after model/optimizer/scheduler were consumed
        (e.g. {"iteration": n}); {} when there is nothing to load."""
        if use_latest and self.has_checkpoint():
            f = self.get_checkpoint_file()
        if not f:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        self.logger.info("Loading checkpoint from %s", f)
        ckpt = self._load_file(f)
        self._load_model(ckpt)
        if self.load_optimizer_scheduler:
            m = self.model.module if hasattr(self.model, "module") else self.model
            if "optimizer" in ckpt and self.optimizer:
                sd = optimizer_state_from_reference(m, self.optimizer, ckpt.pop("optimizer"))
                if sd is None:
                    self.logger.warning("optimizer state has an unknown parameter-group layout: not loaded (weights were)")
                else:
                    load_optimizer_state(self.optimizer, sd)
            if "scheduler" in ckpt and self.scheduler:
                sd = ckpt.pop("scheduler")
                if self.optimizer is not None:
                    sd = scheduler_state_from_reference(m, self.optimizer, sd)
                if sd is None:
                    self.logger.warning("scheduler state has an unknown parameter-group layout: not loaded")
                else:
                    self.scheduler.load_state_dict(sd)
        return ckpt

    def _load_file(self, f):
        return torch.load(f, map_location=torch.device("cpu"))

    def _load_model(self, ckpt):
        load_state_dict(self.model, ckpt.pop("model"))


class DetectronCheckpointer(Checkpointer):
    """The class the reference's train/test scripts construct (`tools/plain_train_net.py`): takes the config first and
    honours `SOLVER.LOAD_OPTIMIZER_SCHEDULER`."""

    def __init__(self, cfg, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        super().__init__(model, optimizer, scheduler, save_dir, save_to_disk, logger)
        self.cfg = cfg.clone() if hasattr(cfg, "clone") else cfg
        self.load_optimizer_scheduler = bool(getattr(self.cfg.SOLVER, "LOAD_OPTIMIZER_SCHEDULER", True))

    def _load_file(self, f):
        if f.startswith("catalog://") or f.startswith("http"):
            raise RuntimeError("%s: remote checkpoint sources are not available in this build (no network); "
                               "pass a local .pth path" % f)
        loaded = super()._load_file(f)
        if "model" not in loaded:                                # a bare state_dict
            loaded = dict(model=loaded)
        return loaded
