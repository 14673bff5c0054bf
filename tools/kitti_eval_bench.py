"""Timing of the device AP evaluator on N synthetic images next to the CPU restatement of the reference's evaluator.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monoflex_amd import synthetic as S
from monoflex_amd.data import evaluation as EV

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=3769)          # size of the KITTI val split
ap.add_argument("--cpu-images", type=int, default=24)
a = ap.parse_args()
labels = [S.synthetic_kitti_labels(5000 + i, 1242, 375, 4 + i % 14, z_range=(5, 45), occl_max=2) for i in range(a.images)]
dets = [S.synthetic_detections(9000 + i, l, 1242, 375, recall=0.85) for i, l in enumerate(labels)]
gts = [EV.parse_label_text("\n".join(l)) for l in labels]
dts = []
for d in dets:
    r = np.zeros((len(d), 16))
    r[:, 0], r[:, 3], r[:, 4:8] = d[:, 0], d[:, 1], d[:, 2:6]
    r[:, 8], r[:, 9], r[:, 10] = d[:, 8], d[:, 6], d[:, 7]      # (h,w,l) -> l,h,w
    r[:, 11:14], r[:, 14], r[:, 15] = d[:, 9:12], d[:, 12], d[:, 13]
    dts.append(r)
EV.get_official_eval_result(gts[:8], dts[:8], [0, 1, 2])       # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
text, ret = EV.get_official_eval_result(gts, dts, [0, 1, 2], metric="R40")
torch.cuda.synchronize()
gpu_s = time.perf_counter() - t0
from oracle import kitti_eval_ref as R
n = a.cpu_images
ga = [R.parse_annos("\n".join(l)) for l in labels[:n]]
da = [R.parse_annos(R.result_text(d)) for d in dets[:n]]
t0 = time.perf_counter()
R.official_result(ga, da, (0, 1, 2), "R40")
cpu_s = time.perf_counter() - t0
print(json.dumps({"what": "KITTI R40 evaluation, 3 classes x 3 levels x (bbox, bev, 3d) x 2 overlap sets",
                  "images": a.images, "gpu_seconds_incl_host": round(gpu_s, 4), "gpu_images_per_s": round(a.images / gpu_s, 1),
                  "cpu_port_images": n, "cpu_port_seconds": round(cpu_s, 2), "cpu_port_images_per_s_1core": round(n / cpu_s, 2),
                  "Car_3d_0.70/moderate": round(float(ret["Car_3d_0.70/moderate"]), 3)}))
