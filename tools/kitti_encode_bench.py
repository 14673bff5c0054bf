"""Timing of the device input pipeline (one batch of B KITTI samples) next to the CPU restatement of the reference's
per-sample __getitem__ arithmetic. Prints one JSON line; HBM roofline for the frame kernel (bytes = frames in + fp32 out)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monoflex_amd import synthetic as S
from monoflex_amd.data import encode as E
from monoflex_amd.data.datasets import kitti_utils as KU

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--objects", type=int, default=12)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
params = E.EncodeParams()
labels = [S.synthetic_kitti_labels(100 + i, 1242, 375, a.objects) for i in range(a.batch)]
recs = [KU.read_label_records(l, ("Car", "Pedestrian", "Cyclist")) for l in labels]
frames = [np.random.RandomState(i).randint(0, 256, (375, 1242, 3)).astype(np.uint8) for i in range(a.batch)]
Ps, sizes, flips = [S.KITTI_P2] * a.batch, [(1242, 375)] * a.batch, [i % 2 for i in range(a.batch)]


def run():
    img = E.preprocess_images(frames, flips, params, "cuda")
    tg = E.encode_targets(recs, Ps, sizes, flips, params, "cuda", check=False)
    return img, tg


for _ in range(5):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    run()
torch.cuda.synchronize()
wall_ms = (time.perf_counter() - t0) / a.iters * 1e3

# kernel-only: inputs resident, events around the launches
import ctypes
from monoflex_amd import lib as L
lib = L.load()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
pix = torch.from_numpy(np.concatenate([f.reshape(-1) for f in frames])).cuda()
offs = torch.tensor(np.cumsum([0] + [f.size for f in frames[:-1]]), dtype=torch.int64).cuda()
wh = torch.tensor(sizes, dtype=torch.int32).cuda()
fl = torch.tensor(flips, dtype=torch.int32).cuda()
out = torch.empty((a.batch, 3, 384, 1280), dtype=torch.float32, device="cuda")
m3, s3 = (ctypes.c_float * 3)(0.485, 0.456, 0.406), (ctypes.c_float * 3)(0.229, 0.224, 0.225)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def frame_kernel():
    L.check(lib.mfx_kitti_preprocess_u8(pix.data_ptr(), offs.data_ptr(), wh.data_ptr(), fl.data_ptr(), out.data_ptr(), a.batch, 1280, 384, m3, s3, st), "pre")


for _ in range(5):
    frame_kernel()
ev[0].record()
for _ in range(a.iters):
    frame_kernel()
ev[1].record()
torch.cuda.synchronize()
frame_ms = ev[0].elapsed_time(ev[1]) / a.iters
bytes_frame = sum(f.size for f in frames) + out.numel() * 4

from oracle import kitti_encode_ref as K
t0 = time.perf_counter()
n_cpu = 0
while time.perf_counter() - t0 < 5.0:
    i = n_cpu % a.batch
    K.encode_sample(labels[i], S.KITTI_P2, 1242, 375, do_flip=bool(flips[i]))
    K.transform_image(frames[i], do_flip=bool(flips[i]))
    n_cpu += 1
cpu_ms = (time.perf_counter() - t0) / n_cpu * 1e3
print(json.dumps({"what": "KITTI input pipeline, batch %d x 1242x375, %d label lines/image" % (a.batch, a.objects),
                  "gpu_ms_per_batch_incl_host_packing_and_h2d": round(wall_ms, 3), "gpu_images_per_s": round(a.batch / wall_ms * 1e3, 1),
                  "frame_kernel_ms": round(frame_ms, 4), "frame_kernel_GBps": round(bytes_frame / frame_ms / 1e6, 1),
                  "frame_kernel_hbm_frac": round(bytes_frame / frame_ms / 1e6 / 8000, 3),
                  "cpu_port_ms_per_image": round(cpu_ms, 3), "cpu_port_images_per_s_1core": round(1e3 / cpu_ms, 1)}))
