"""Evaluation loop (reference engine/inference.py:17-126): run the detector over a loader, write one KITTI result file per
image, score them.  Batches of any size are accepted (the reference is batch-1 only); the per-image (N,14) rows come back
from the device once per batch.  The disentangled-IoU statistics of the reference (`eval_score_iou`) are not produced."""
import logging
import os
import time

import torch

from ..data.evaluation import evaluate_python, generate_kitti_3d_detection
from ..parallel import barrier


def compute_on_dataset(model, data_loader, device, predict_folder, timer=None, overlap=True):
    """Returns the number of images processed; `timer`, if given, is a dict that receives the seconds spent in the model (launches + waits).
    `overlap` (default, CUDA only): ONE batch stays in flight -- the fixed-size (B,50,14) rows and validity flags of batch k are copied to
    pinned memory behind an event, batch k+1 is launched, and only then does the host wait for batch k's event and write its files
    (the reference's loop, engine/inference.py:26-56, waits for every batch before it touches the next: 21 % of the time at B = 8, bench.py
    `pipeline` leg).  Same rows, same files: the per-image selection `det[b][valid[b]]` is the one PostProcessor.forward makes."""
    model.eval()
    n, busy = 0, 0.0
    dev = torch.device(device)
    overlap = bool(overlap) and dev.type == "cuda" and hasattr(model, "detect_device")
    if getattr(getattr(getattr(model, "heads", None), "post_processor", None), "output_depth", None) == "oracle":
        overlap = False                                            # reads each image's ground truth on the host (PostProcessor.decode_oracle)
    pending = None

    def finish(p):
        ev, rows_h, valid_h, ids = p
        ev.synchronize()
        for b, image_id in enumerate(ids):
            generate_kitti_3d_detection(rows_h[b][valid_h[b].bool()], os.path.join(predict_folder, image_id + ".txt"))
        return len(ids)

    with torch.no_grad():
        for batch in data_loader:
            images, targets, image_ids = batch["images"], batch["targets"], batch["img_ids"]
            images = images.to(device)
            targets = [t.to(device) for t in targets]
            t0 = time.perf_counter()
            if overlap:
                tensors = images.tensors if hasattr(images, "tensors") else images
                det, _, valid, _ = model.detect_device(tensors, *model.device_targets(targets, dev))
                rows_h = torch.empty(det.shape, dtype=det.dtype, pin_memory=True)
                valid_h = torch.empty(valid.shape, dtype=valid.dtype, pin_memory=True)
                rows_h.copy_(det, non_blocking=True)
                valid_h.copy_(valid, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                busy += time.perf_counter() - t0
                if pending is not None:
                    n += finish(pending)
                pending = (ev, rows_h, valid_h, list(image_ids))
                continue
            output, eval_utils, _ = model(images, targets)
            outputs = [output] if torch.is_tensor(output) else list(output)
            outputs = [o.cpu() for o in outputs]                       # the host copy synchronises
            busy += time.perf_counter() - t0
            for image_id, rows in zip(image_ids, outputs):
                generate_kitti_3d_detection(rows, os.path.join(predict_folder, image_id + ".txt"))
            n += len(outputs)
        if pending is not None:
            t0 = time.perf_counter()
            pending[0].synchronize()
            busy += time.perf_counter() - t0
            n += finish(pending)
    if timer is not None:
        timer["inference_seconds"] = timer.get("inference_seconds", 0.0) + busy
    return n


def inference(model, data_loader, dataset_name, eval_types=("detections",), device="cuda", output_folder=None, metrics=("R40",)):
    """-> (ret_dicts, result text of the last metric, dis_ious) on rank 0, (None, None, None) elsewhere
    (engine/inference.py:66-126). Every rank writes the result files of its shard into `<output_folder>/data`."""
    import torch.distributed as dist
    logger = logging.getLogger("monoflex.inference")
    dataset = data_loader.dataset
    predict_folder = os.path.join(output_folder, "data")
    os.makedirs(predict_folder, exist_ok=True)
    timer = {}
    t0 = time.perf_counter()
    n = compute_on_dataset(model, data_loader, torch.device(device), predict_folder, timer)
    barrier()
    logger.info("%s: %d images in %.2f s (%.4f s / img in the model)", dataset_name, n, time.perf_counter() - t0,
                timer["inference_seconds"] / max(n, 1))
    if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
        return None, None, None
    ret_dicts, result = [], None
    for metric in metrics:
        result, ret_dict = evaluate_python(label_path=dataset.label_dir, result_path=predict_folder,
                                           label_split_file=dataset.imageset_txt, current_class=dataset.classes, metric=metric,
                                           device=device)
        logger.info("metric = %s\n%s", metric, result)
        ret_dicts.append(ret_dict)
    return ret_dicts, result, {}


EVAL_DEPTH_METHODS = ("oracle", "hard", "soft", "mean", "direct", "keypoints_center", "keypoints_02", "keypoints_13")   # engine/inference.py:154


def inference_all_depths(model, data_loader, dataset_name, eval_types=("detections",), device="cuda", output_folder=None, metrics=("R40",)):
    """`--eval_all_depths` (engine/inference.py:131-198): the evaluation once per depth-solving method, each into
    `<output_folder>/eval_all_depths/<method>`, by re-assigning `post_processor.output_depth` between passes (the decode kernel reads the mode
    per launch; 'oracle' needs the ground-truth fields of a validation split).  Logs the Car AP@0.70 bev/3d line per method and the ranking by
    3D moderate.  -> {method: ret_dict} on rank 0 (the reference returns (None, None, None); the log lines are its product), the attribute restored."""
    import numpy as np
    import torch.distributed as dist
    logger = logging.getLogger("monoflex.inference")
    dataset = data_loader.dataset
    post = model.heads.post_processor
    before = post.output_depth
    root = os.path.join(output_folder, "eval_all_depths")
    rank0 = not (dist.is_available() and dist.is_initialized() and dist.get_rank() != 0)
    ret = {}
    try:
        for method in EVAL_DEPTH_METHODS:
            logger.info("evaluation with depth method: %s", method)
            folder = os.path.join(root, method)
            os.makedirs(folder, exist_ok=True)
            if rank0:
                for f in os.listdir(folder):                               # stale predictions of an earlier run (:162-164)
                    os.remove(os.path.join(folder, f))
            barrier()
            post.output_depth = method
            compute_on_dataset(model, data_loader, torch.device(device), folder)
            barrier()
            if rank0:
                _, ret[method] = evaluate_python(label_path=dataset.label_dir, result_path=folder, label_split_file=dataset.imageset_txt,
                                                 current_class=dataset.classes, metric="R40", device=device)
    finally:
        post.output_depth = before
    if not rank0:
        return None
    cls, thresh = "Car", 0.7
    logger.info("%s AP@%.2f, %.2f:", cls, thresh, thresh)
    key = lambda kind, level: "%s_%s_%.2f/%s" % (cls, kind, thresh, level)
    for method in EVAL_DEPTH_METHODS:
        d = ret[method]
        logger.info("bev/3d AP, method %s:", method)
        logger.info("%.4f/%.4f, %.4f/%.4f, %.4f/%.4f", d[key("bev", "easy")], d[key("3d", "easy")], d[key("bev", "moderate")],
                    d[key("3d", "moderate")], d[key("bev", "hard")], d[key("3d", "hard")])
    order = np.argsort(-np.array([ret[m][key("3d", "moderate")] for m in EVAL_DEPTH_METHODS]))
    logger.info("Cls %s, Thresh %s, Sort: %s", cls, thresh, " > ".join(EVAL_DEPTH_METHODS[i] for i in order))
    return ret
