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
