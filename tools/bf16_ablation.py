#!/usr/bin/env python
"""Where does the bf16 mode's deviation from the reference come from?  (VERDICT r02 "weak" #4 / task 6)

The benchmarked configuration (B = 8, 1280 x 384, synthetic weights seed 0, image seed 1000 = the image of tests/golden/e2e_full.npz)
is run with ONE stage at a time in fp32 and everything else in bf16; every variant is compared with the REFERENCE's outputs stored in
the golden file (class logits / regression values at 562 pixels, the top-50 index set) and timed (hipGraph replay, whole step).
Writes a markdown table (stdout): profiles/r03_bf16_ablation.md.

  python tools/bf16_ablation.py [--batch 8]"""
import argparse
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from monoflex_amd import synthetic as S
from monoflex_amd.model.backbone.DCNv2 import dcn_v2 as DV
from monoflex_amd.structures.params_3d import make_test_target

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda", 0)
model, _, _ = bench.build_model("bf16", dev)
B = a.batch
images = S.synthetic_images(B, 384, 1280, seed=1000).to(dev)
targets = [make_test_target(S.synthetic_target(320, 96)) for _ in range(B)]
ei, el, pad, calib, size, rowmap = model.device_targets(targets, dev)
F32, BF = torch.float32, torch.bfloat16
bb, heads = model.backbone, model.heads
g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_full.npz"))
assert ast.literal_eval(str(g["meta"]))["seeds"][0] == 1000


from monoflex_amd import lib as L, ops
from monoflex_amd.model.backbone.dla_dcn import _conv_bn


def base_levels(dts):
    """DLA.forward (eval) with a compute dtype per stage: dts = (stem, level0, ..., level5)."""
    base = bb.base
    packs = base.__dict__.setdefault("_packs", {})
    d0 = dts[0]
    if ("stem", d0) not in packs:
        scale, shift = ops.fold_bn(base.base_layer[1])
        packs[("stem", d0)] = ops.pack_stem(base.base_layer[0].weight, d0, scale, shift)
    _, _, H, W = images.shape
    if d0 == BF and packs[("stem", d0)].Cout == 16:
        x = ops.stem_conv(images, packs[("stem", d0)])
    else:
        x = ops.conv2d(ops.pack_image(images, d0), packs[("stem", d0)], out_hw=(H, W))
    y = []
    for i in range(6):
        lvl = getattr(base, "level%d" % i)
        x = x.to(dts[i + 1])
        if i < 2:
            for j in range(0, len(lvl), 3):
                x = ops.conv2d(x, _conv_bn(lvl, "c%d" % j, lvl[j], lvl[j + 1], dts[i + 1], L.ACT_RELU))
        else:
            x = lvl(x)
        y.append(x)
    return y


def step(base_dt, up_dt, head_dt, off32):
    DV.OFFSET_CONV_FP32[0] = off32
    x = [t.to(up_dt) if t is not None else None for t in (base_levels(base_dt) if isinstance(base_dt, tuple) else bb.base(images, base_dt))]
    x = bb.dla_up(list(x))
    y = [x[i] for i in range(bb.last_level - bb.first_level)]
    bb.ida_up(y, 0, len(y))
    hm = heads.predictor.forward_nhwc(y[-1].to(head_dt), ei, el, rowmap)
    det, topk, valid = heads.post_processor.decode_device(hm, pad, calib, size, heads.predictor.last_cls_planar)
    return det, topk, valid, hm


def measure(fn):
    with torch.no_grad():
        for _ in range(2):
            out = fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = fn()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / 10


VARIANTS = [("all bf16 (the benchmarked mode)", BF, BF, BF, False),
            ("DLA base (stem .. level5) in fp32", F32, BF, BF, False),
            ("DLAUp + IDAUp (16 DCN modules, up-samplers) in fp32", BF, F32, BF, False),
            ("only the 16 offset/mask convs in fp32 (fp32 copy of their bf16 input, fp32 weights)", BF, BF, BF, True),
            ("heads in fp32 (bf16 feature map converted)", BF, BF, F32, False),
            ("base + DLAUp/IDAUp in fp32, heads bf16", F32, F32, BF, False),
            ("base in fp32 + offset convs in fp32", F32, BF, BF, True),
            ("stem + level0 + level1 in fp32 (the full-resolution, HBM-bound layers)", (F32, F32, F32, BF, BF, BF, BF), BF, BF, False),
            ("level2 in fp32", (BF, BF, BF, F32, BF, BF, BF), BF, BF, False),
            ("level3 in fp32", (BF, BF, BF, BF, F32, BF, BF), BF, BF, False),
            ("level4 in fp32", (BF, BF, BF, BF, BF, F32, BF), BF, BF, False),
            ("level5 in fp32", (BF, BF, BF, BF, BF, BF, F32), BF, BF, False),
            ("level3 + level4 + level5 in fp32", (BF, BF, BF, BF, F32, F32, F32), BF, BF, False),
            ("all fp32 (the parity mode)", F32, F32, F32, False)]
print("# bf16 deviation by stage (B = %d, 1280 x 384, vs the reference's outputs in tests/golden/e2e_full.npz, image seed 1000)\n" % B)
print("| variant | max abs d(class logit) | max abs d(regression) | top-50 index agreement | identical order | ms / step (B = %d) |" % B)
print("|---|---|---|---|---|---|")
base_ms = None
for name, bd, ud, hd, off in VARIANTS:
    out, ms = measure(lambda: step(bd, ud, hd, off))
    d = bench.deviation_vs_reference(tuple(t if isinstance(t, torch.Tensor) else t for t in out), "mixed")
    base_ms = base_ms or ms
    print("| %s | %.4f | %.4f | %.2f | %s | %.3f (%+.1f %%) |" % (name, d["max_abs_dlogit"], d["max_abs_dreg"], d["topk_index_agreement"],
                                                                d["topk_identical_order"], ms, 100 * (ms / base_ms - 1)))
DV.OFFSET_CONV_FP32[0] = False
