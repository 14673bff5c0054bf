"""`run_test` of the reference (engine/test_net.py:9-38): evaluate `model` on every dataset of cfg.DATASETS.TEST."""
import os

from ..data.build import build_test_loader
from ..utils import comm
from .inference import inference


def run_test(cfg, model, vis=False, eval_score_iou=False, eval_all_depths=False):
    """`eval_all_depths` (reference: `inference_all_depths`, engine/inference.py:131-198, re-runs the evaluation once per depth-solving method by mutating
    `post_processor.output_depth`) needs the decode modes other than runs/monoflex.yaml's 'soft' -- outside this build's decode contract, so
    asking for it is an error, not a silently different evaluation.  (The reference's signature defaults the flag to True; its caller
    tools/plain_train_net.py:74 always passes args.eval_all_depths, a store_true switch.)"""
    if vis or eval_score_iou:
        raise NotImplementedError("visualisation / score-IoU statistics are not part of this build (hot path only)")
    if eval_all_depths:
        raise NotImplementedError("eval_all_depths: only the experiment file's OUTPUT_DEPTH = 'soft' decode is built (device kernel csrc/decode.hip); "
                                  "the per-method re-evaluation of engine/inference.py:131-198 is not part of this build")
    results = []
    for name, loader in zip(cfg.DATASETS.TEST, build_test_loader(cfg)):
        folder = os.path.join(cfg.OUTPUT_DIR, "inference", name) if cfg.OUTPUT_DIR else None
        if folder:
            os.makedirs(folder, exist_ok=True)
        results.append(inference(model, loader, dataset_name=name, eval_types=("detection",), device=cfg.MODEL.DEVICE,
                                 output_folder=folder, metrics=cfg.TEST.METRIC))
        comm.synchronize()
    return results
