"""Checkpoint save / resume with the reference's on-disk format (reference utils/check_point.py:11-140).

A checkpoint is one `torch.save`d dict {"model": state_dict, "optimizer": ..., "scheduler": ..., **extras} named
`<save_dir>/<name>.pth`; `<save_dir>/last_checkpoint` holds the path of the newest one and wins over an explicit
file when `use_latest` is set. Files written by the reference load here and vice versa: parameter names, OIHW weight
layout and the optimizer's parameter order are the reference's. Weights-only files (a bare state_dict) are accepted
by `DetectronCheckpointer`. There is no network in this build, so `catalog://` and `http(s)://` sources raise.
"""
import logging
import os

import torch

from .model_serialization import load_state_dict

LAST = "last_checkpoint"


class Checkpointer:
    def __init__(self, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.save_dir, self.save_to_disk = save_dir, save_to_disk
        self.logger = logger or logging.getLogger("monoflex.checkpointer")
        self.load_optimizer_scheduler = True

    # ---- writing ---------------------------------------------------------------------------------------
    def save(self, name, **extras):
        data = {"model": self.model.state_dict()}
        if self.optimizer is not None:
            data["optimizer"] = self.optimizer.state_dict()
        if self.scheduler is not None and hasattr(self.scheduler, "state_dict"):
            data["scheduler"] = self.scheduler.state_dict()
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
            if "optimizer" in ckpt and self.optimizer:
                self.optimizer.load_state_dict(ckpt.pop("optimizer"))
            if "scheduler" in ckpt and self.scheduler:
                self.scheduler.load_state_dict(ckpt.pop("scheduler"))
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
