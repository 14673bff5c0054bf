#!/usr/bin/env python
"""Bitwise repeatability of the inference step (no atomics on that path: every launch of the same inputs must give the same bits) per compute mode;
a difference is a data race or a hardware hazard in some kernel.  Prints, per mode, how many of `REPS` repeats differ from the first in the head map /
detections, and the worst difference."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
import bench
from monoflex_amd import lib, synthetic as S
from monoflex_amd.structures.params_3d import make_test_target

lib.load()
dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", "20"))
for dtype in os.environ.get("MODES", "bf16 fp16 fp16x2 fp32").split():
    for B in (8, 1):
        model, _, _ = bench.build_model(dtype, dev)
        images = bench.bench_images(B, 0, dev)
        tg = model.device_targets([make_test_target(S.synthetic_target(320, 96)) for _ in range(B)], dev)
        with torch.no_grad():
            det0, topk0, valid0, hm0 = [t.clone() for t in model.detect_device(images, *tg)]
            bad, worst = 0, 0.0
            for r in range(REPS):
                det, topk, valid, hm = model.detect_device(images, *tg)
                torch.cuda.synchronize()
                # (the head map's padding channels 3..7 / 58..63 are never written)
                d = torch.cat(((hm[..., :3] - hm0[..., :3]).abs().flatten(), (hm[..., 8:58] - hm0[..., 8:58]).abs().flatten()))
                same = float(d.max()) == 0.0 and torch.equal(det[valid0.bool()], det0[valid0.bool()]) and torch.equal(valid, valid0) and torch.equal(topk, topk0)
                if not same:
                    bad += 1
                    worst = max(worst, float(d.max()))
        print("%-6s B=%d: %d of %d repeats differ from the first (max |d head map| %.3e)" % (dtype, B, bad, REPS, worst), flush=True)
