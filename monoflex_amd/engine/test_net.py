"""`run_test` of the reference (engine/test_net.py:9-38): evaluate `model` on every dataset of cfg.DATASETS.TEST."""
import os

from ..data.build import build_test_loader
from ..utils import comm
from .inference import inference


def run_test(cfg, model, vis=False, eval_score_iou=False, eval_all_depths=True):
    if vis or eval_score_iou:
        raise NotImplementedError("visualisation / score-IoU statistics are not part of this build (hot path only)")
    results = []
    for name, loader in zip(cfg.DATASETS.TEST, build_test_loader(cfg)):
        folder = os.path.join(cfg.OUTPUT_DIR, "inference", name) if cfg.OUTPUT_DIR else None
        if folder:
            os.makedirs(folder, exist_ok=True)
        results.append(inference(model, loader, dataset_name=name, eval_types=("detection",), device=cfg.MODEL.DEVICE,
                                 output_folder=folder, metrics=cfg.TEST.METRIC))
        comm.synchronize()
    return results
