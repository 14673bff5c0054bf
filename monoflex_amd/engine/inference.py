"""Evaluation loop (reference engine/inference.py:17-126): run the detector over a loader, write one KITTI result file per
image, score them.  Batches of any size are accepted (the reference is batch-1 only); the per-image (N,14) rows come back
from the device once per batch.  The disentangled-IoU statistics of the reference (`eval_score_iou`) are not produced."""
import logging
import os
import time

import torch

from ..data.evaluation import evaluate_python, generate_kitti_3d_detection
from ..parallel import barrier


def compute_on_dataset(model, data_loader, device, predict_folder, timer=None):
    """Returns the number of images processed; `timer`, if given, is a dict that receives the model-only seconds."""
    model.eval()
    n, busy = 0, 0.0
    with torch.no_grad():
        for batch in data_loader:
            images, targets, image_ids = batch["images"], batch["targets"], batch["img_ids"]
            images = images.to(device)
            targets = [t.to(device) for t in targets]
            t0 = time.perf_counter()
            output, eval_utils, _ = model(images, targets)
            outputs = [output] if torch.is_tensor(output) else list(output)
            outputs = [o.cpu() for o in outputs]                       # the host copy synchronises
            busy += time.perf_counter() - t0
            for image_id, rows in zip(image_ids, outputs):
                generate_kitti_3d_detection(rows, os.path.join(predict_folder, image_id + ".txt"))
            n += len(outputs)
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
