"""Command line and run set-up of the reference's entry script (engine/defaults.py:14-90): same flags, same `cfg` side
effects (output directory, seed, log file)."""
import argparse
import logging
import os
import random
import sys

import numpy as np
import torch

from ..utils import comm

__all__ = ["default_argument_parser", "default_setup"]


def default_argument_parser():
    p = argparse.ArgumentParser(description="MonoFlex training / evaluation (MI355X build)")
    p.add_argument("--config", dest="config_file", default="runs/monoflex.yaml", metavar="FILE", help="path to config file")
    p.add_argument("--eval", dest="eval_only", action="store_true", help="perform evaluation only")
    p.add_argument("--eval_iou", action="store_true", help="evaluate disentangling IoU")
    p.add_argument("--eval_depth", action="store_true", help="evaluate depth errors")
    p.add_argument("--eval_all_depths", action="store_true")
    p.add_argument("--eval_score_iou", action="store_true", help="evaluate the relationship between scores and IoU")
    p.add_argument("--test", action="store_true", help="test mode")
    p.add_argument("--vis", action="store_true", help="visualize when evaluating")
    p.add_argument("--ckpt", default=None, help="checkpoint for testing (default: the latest one)")
    p.add_argument("--num_gpus", type=int, default=1, help="number of gpu")
    p.add_argument("--batch_size", type=int, default=8, help="number of batch_size")
    p.add_argument("--num_work", type=int, default=8, help="number of workers for dataloader")
    p.add_argument("--output", type=str, default=None)
    p.add_argument("--vis_thre", type=float, default=0.25, help="threshold for visualize results of detection")
    p.add_argument("--num-machines", type=int, default=1)
    p.add_argument("--machine-rank", type=int, default=0, help="the rank of this machine (unique per machine)")
    p.add_argument("--dist-url", default="auto")
    p.add_argument("opts", help="Modify config options using the command-line", default=None, nargs=argparse.REMAINDER)
    return p


def default_setup(cfg, args):
    out = cfg.OUTPUT_DIR
    if out:
        os.makedirs(out, exist_ok=True)
    rank = comm.get_rank()
    logger = logging.getLogger("monoflex")
    if not logger.handlers:
        logger.setLevel(logging.INFO if rank == 0 else logging.WARNING)
        fmt = logging.Formatter("[%(asctime)s] %(name)s %(levelname)s: %(message)s", datefmt="%m/%d %H:%M:%S")
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(fmt)
        logger.addHandler(h)
        if out and rank == 0:
            fh = logging.FileHandler(os.path.join(out, "log_{}.txt".format(str(cfg.get("START_TIME", "run")).replace(" ", "_").replace(":", ""))))
            fh.setFormatter(fmt)
            logger.addHandler(fh)
    logger.info("Using %d GPUs", args.num_gpus)
    logger.info(args)
    logger.info("Loaded configuration file %s", args.config_file)
    seed = cfg.SEED if cfg.SEED >= 0 else None
    if seed is not None:                                           # engine/defaults.py:86-90 (seed + rank)
        random.seed(seed + rank); np.random.seed(seed + rank); torch.manual_seed(seed + rank)
    return logger
