#!/usr/bin/env python
"""Per-operator timing of one forward+decode on the GPU (HIP events around every C-ABI call).
usage: python tools/layer_bench.py [--batch 8] [--dtype bf16] [--opts conv_tile=5,kc=4] [--iters 5]"""
import argparse
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--opts", default="")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--summary", action="store_true")
args = ap.parse_args()

from monoflex_amd import lib, ops, synthetic as S
from monoflex_amd.structures.params_3d import make_test_target
sys.path.insert(0, ROOT)
import bench

L = lib.load()
for kv in filter(None, args.opts.split(",")):
    k, v = kv.split("=")
    lib.check(L.mfx_set_option(k.encode(), int(v)), "set_option")

model, _, _ = bench.build_model(args.dtype, torch.device("cuda", 0))
B = args.batch
images = S.synthetic_images(B, 384, 1280, seed=1000).cuda()
targets = [make_test_target(S.synthetic_target(320, 96)) for _ in range(B)]
tg = model.device_targets(targets, "cuda")

records = OrderedDict()
state = {"on": False, "idx": 0}


def wrap(name, fn, describe):
    def inner(*a, **k):
        if not state["on"]:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        key = (state["idx"], name)
        state["idx"] += 1
        d, flops, bytes_ = describe(r, *a, **k)
        rec = records.setdefault(key, dict(desc=d, flops=flops, bytes=bytes_, us=[]))
        rec["us"].append(e0.elapsed_time(e1) * 1e3)
        return r
    return inner


def nbytes(*ts):
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


def d_conv(y, x, p, res=None, **k):
    M = y.numel() // y.shape[-1]
    return ("conv k%dx%d s%d Ck%d -> %d  M=%d" % (p.kh, p.kw, p.stride, p.Ck, p.Cout, M), 2.0 * M * p.Cout_pad * p.kh * p.kw * p.Ck,
            nbytes(x, y, res, p.w))


def d_cat(y, srcs, p):
    M = y.numel() // y.shape[-1]
    return ("root cat K=%d -> %d  M=%d" % (p.K_pad, p.Cout, M), 2.0 * M * p.Cout_pad * p.K_pad, nbytes(y, p.w, *srcs))


def d_dcn(y, x, om, p):
    M = y.numel() // y.shape[-1]
    return ("dcn %d -> %d  M=%d" % (x.shape[-1], p.Cout, M), 2.0 * M * p.Cout_pad * p.K_pad, nbytes(x, y, om, p.w))


def d_heads(y, x, p, **k):
    M = x.numel() // 64
    return ("heads fused M=%d" % M, 2.0 * M * 9 * (256 * 576 + 32 * 256), nbytes(x, y[0]))


def d_mem(name):
    return lambda y, *a, **k: (name, 0.0, nbytes(y, *[t for t in a if torch.is_tensor(t)]))


ops.conv2d = wrap("conv2d", ops.conv2d, d_conv)
ops.cat_conv1x1 = wrap("cat", ops.cat_conv1x1, d_cat)
ops.dcn = wrap("dcn", ops.dcn, d_dcn)
ops.heads_fused = wrap("heads", ops.heads_fused, d_heads)
ops.maxpool2x2 = wrap("maxpool", ops.maxpool2x2, d_mem("maxpool"))
ops.upsample_add = wrap("upsample", ops.upsample_add, d_mem("upsample+add"))
ops.pack_image = wrap("pack_image", ops.pack_image, d_mem("pack image"))
ops.decode_topk = wrap("topk", ops.decode_topk, lambda y, *a, **k: ("decode topk", 0.0, 0))
ops.decode_boxes = wrap("boxes", ops.decode_boxes, lambda y, *a, **k: ("decode boxes", 0.0, 0))
ops.edge_scatter_add = wrap("edge_scatter", ops.edge_scatter_add, lambda y, *a, **k: ("edge scatter", 0.0, 0))

with torch.no_grad():
    for _ in range(2):
        model.detect_device(images, *tg)
    torch.cuda.synchronize()
    state["on"] = True
    for _ in range(args.iters):
        state["idx"] = 0
        model.detect_device(images, *tg)
    state["on"] = False
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        model.detect_device(images, *tg)
    e1.record()
    torch.cuda.synchronize()
    eager_ms = e0.elapsed_time(e1) / 10

tot = 0.0
groups = {}
print("%-4s %-10s %-46s %9s %9s %9s" % ("#", "op", "shape", "us", "TFLOP/s", "GB/s"))
for (i, name), r in records.items():
    us = sorted(r["us"])[len(r["us"]) // 2]
    tot += us
    g = groups.setdefault(name, [0.0, 0.0])
    g[0] += us
    g[1] += r["flops"]
    if not args.summary:
        print("%-4d %-10s %-46s %9.1f %9.1f %9.0f" % (i, name, r["desc"], us, r["flops"] / us / 1e6 if r["flops"] else 0,
                                                       r["bytes"] / us / 1e3 if r["bytes"] else 0))
print("---- sum of op medians %.1f us ; eager step %.3f ms ; opts=%s" % (tot, eager_ms, args.opts))
for name, (us, fl) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print("  %-12s %9.1f us  %5.1f%%  %8.1f TFLOP/s" % (name, us, 100 * us / tot, fl / us / 1e6 if fl else 0))
