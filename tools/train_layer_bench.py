#!/usr/bin/env python
"""Live timing of the dominant kernel group of the training step (used by `bench.py --mode train` for its `roofline` object).

The group is the NHWC DCNv2 backward of the largest DCN shape (64 -> 64 @ 96x320, five of the sixteen DCN layers;
reference dcn_v2_cuda.cu:206-335): d(columns) GEMM, offset/mask/input gradients and the weight gradient.  Algorithmic
work per launch: two GEMMs of M x 9C x Cout each (data gradient through the columns, weight gradient)
= 4 * M * 9C * Cout FLOP, M = B*96*320; algorithmic HBM bytes: x, dy read + dx written + offsets read/written
(the columns never need to exist in HBM).  Timed with HIP events on the launch stream.

  python tools/train_layer_bench.py [--batch 8] [--dtype bf16]      prints the roofline object as JSON
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}          # dense MFMA TFLOP/s (fp16 = the bf16 rate)


def dominant_kernel_roofline(dtype, B, device, H=96, W=320, C=64, Cout=64, reps=10):
    import torch
    from monoflex_amd import autograd as AG
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(dtype, torch.float32)
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(B, H, W, C, generator=g).relu().to(dt).to(device).requires_grad_()
    raw = torch.zeros(B, H, W, 32)
    raw[..., :18] = torch.randn(B, H, W, 18, generator=g) * 1.5          # the synthetic weights' offset spread (synthetic.py)
    raw[..., 18:27] = torch.randn(B, H, W, 9, generator=g)
    raw = raw.to(device).requires_grad_()
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(device).requires_grad_()
    b = torch.zeros(Cout, device=device, requires_grad=True)
    y = AG.DCNFn.apply(x, raw, w, b, 1, 1, 1)
    dy = torch.randn(y.shape, generator=g).to(dt).to(device)

    def run():
        torch.autograd.grad(y, (x, raw, w, b), dy, retain_graph=True)
    for _ in range(3):
        run()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / reps
    M = B * H * W
    flops = 4.0 * M * 9 * C * Cout
    es = 4 if dtype == "fp32" else 2
    alg_bytes = M * (C * es + Cout * es + C * 4 + 32 * 4 * 2)
    # roofline of the group: the larger of its MFMA floor and its HBM floor (bytes that MUST move: x, dy, offsets in; dx, d(offsets),
    # dW out -- the d(columns) / columns intermediates the implementation materialises are not algorithmic)
    mfma_floor_ms = flops / (PEAK[dtype] * 1e9)
    hbm_floor_ms = alg_bytes / 8e12 * 1e3
    if hbm_floor_ms >= mfma_floor_ms:
        achieved = alg_bytes / ms / 1e6                       # GB/s
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4)}
    else:
        achieved = flops / ms / 1e9
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK[dtype], "unit": "TFLOP/s", "frac": round(achieved / PEAK[dtype], 4)}
    traffic = traffic_source = None
    try:                                                       # PMC pass of the same group (tools/pmc_dcnbwd.sh), committed with its raw CSV
        for tag in ("r06", "r05", "r04", "r03", "r02"):               # the newest committed measurement
            f = os.path.join(ROOT, "profiles", tag + "_dcnbwd_traffic.json")
            if not os.path.exists(f):
                continue
            t = json.load(open(f))
            if t.get("dtype") == dtype and t.get("batch") == B and (H, W, C, Cout) == (96, 320, 64, 64):
                traffic, traffic_source = int(t["traffic_bytes"]), t["source"]
            break
    except (OSError, ValueError, KeyError):
        pass
    roof.update({"kernel": "DCNv2 backward group (dcn_bwd_sample_wgrad_fly, dcn_bwd_tile_fly, dcn_bwd_far_fly, slab sum, bias sums), %d->%d @ %dx%d, B=%d"
                           % (C, Cout, H, W, B),
                 "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": round(ms, 4), "algorithmic_flops_per_launch": flops,
                 "algorithmic_bytes_per_launch": alg_bytes, "hbm_floor_ms": round(hbm_floor_ms, 4), "mfma_floor_ms": round(mfma_floor_ms, 4),
                 "materialised_bytes_per_launch": 0,
                 "note": "r04: neither d(columns) nor the columns are materialised on this shape -- both consumers rebuild d(columns) from dy on the "
                         "matrix cores (csrc/dcn_bwd_tile.hip, third generation; option dcn_bt_fly = 0 restores the d(columns) GEMM of r03: 2.05 GB per call)"})
    return roof


if __name__ == "__main__":
    import argparse
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    print(json.dumps(dominant_kernel_roofline(a.dtype, a.batch, torch.device("cuda", 0))))
