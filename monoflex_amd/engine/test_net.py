"""`run_test` of the reference (engine/test_net.py:9-38): evaluate `model` on every dataset of cfg.DATASETS.TEST."""
import os

from ..data.build import build_test_loader
from ..utils import comm
from .inference import inference, inference_all_depths


def run_test(cfg, model, vis=False, eval_score_iou=False, eval_all_depths=False):
    """`eval_all_depths` selects `inference_all_depths` (engine/test_net.py:21), the evaluation once per depth-solving method.  (The reference's
    signature defaults the flag to True; its caller tools/plain_train_net.py:74 always passes args.eval_all_depths, a store_true switch.)"""
    if vis or eval_score_iou:
        raise NotImplementedError("visualisation / score-IoU statistics are not part of this build (hot path only)")
    fn = inference_all_depths if eval_all_depths else inference
    results = []
    for name, loader in zip(cfg.DATASETS.TEST, build_test_loader(cfg)):
        folder = os.path.join(cfg.OUTPUT_DIR, "inference", name) if cfg.OUTPUT_DIR else None
        if folder:
            os.makedirs(folder, exist_ok=True)
        results.append(fn(model, loader, dataset_name=name, eval_types=("detection",), device=cfg.MODEL.DEVICE,
                                 output_folder=folder, metrics=cfg.TEST.METRIC))
        comm.synchronize()
    return results
