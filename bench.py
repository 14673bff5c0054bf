#!/usr/bin/env python
"""bench.py -- images/sec of the MonoFlex hot path on MI355X.

One "step" = DLA-34 + DCNv2 + all heads forward + decode over one batch of synthetic 1280x384
images already resident in HBM (BASELINE.json configs[1]: batch 8 per GPU, bf16).  N>1: one
process per GPU (torchrun), independent replicas over disjoint image shards -- the inference path
has no exchange step, so there is no collective in the timed region (weak scaling).

Prints ONE JSON line on rank 0 (see README/DESIGN for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

FWD_GFLOP_PER_IMG = 177.99          # SURVEY 8d / BASELINE.md section 3 (2 x 88.997 GMAC)
HEADS_GFLOP_PER_IMG = 2 * 41.185    # Appendix A: 9 x (3x3 64->256 + 1x1) per image
PEAK_BF16_TFLOPS = 2500.0           # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_F32_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-images", type=int, default=6)
    return ap.parse_args()


def build_model(dtype, device):
    from monoflex_amd import synthetic as S
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.detector import KeypointDetector
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.MODEL.PRETRAIN = False
    cfg.DATASETS.TEST_SPLIT = "test"
    cfg.MODEL.COMPUTE_DTYPE = dtype
    m = KeypointDetector(cfg).eval()
    sd = S.synthetic_state_dict(m.state_dict(), seed=0, cls_bias=-1.0)      # -1.0: all 50 slots pass 0.2 (worst-case decode)
    m.load_state_dict(sd)
    return m.to(device), sd


def cpu_baseline(sd, n_images):
    """The oracle (CPU port of the reference path) timed on this host's cores: forward+decode, B=1 per call."""
    from monoflex_amd import synthetic as S
    from oracle import monoflex_ref as R
    ref = R.KeypointDetectorRef().eval()
    ref.load_state_dict(sd)
    tgt = S.synthetic_target(320, 96)
    tgt = dict(tgt, calib=R.Calib(tgt["P"]))
    t0 = time.time()
    for i in range(n_images):
        img = S.synthetic_images(1, 384, 1280, seed=2000 + i)
        ref.detect(img, [tgt])
    dt = time.time() - t0
    return {"value": round(n_images / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d x (1,3,384,1280) image, forward+decode, oracle/monoflex_ref.py (torch fp32 convs + C DCN, "
                      "OpenMP), %.1f s" % (n_images, dt)}


def main():
    args = parse()
    from monoflex_amd import parallel
    rank, world, local_rank = parallel.init_from_env(backend="nccl")
    dist = None
    if world > 1:
        import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d (launch with torchrun for N>1)" % (args.gpus, world), file=sys.stderr)

    from monoflex_amd import lib, synthetic as S
    from monoflex_amd.structures.params_3d import make_test_target
    lib.load()
    model, sd = build_model(args.dtype, device)
    B = args.batch
    images = S.synthetic_images(B, 384, 1280, seed=parallel.shard_seed(1000, rank, B)).to(device)   # resident in HBM
    targets = [make_test_target(S.synthetic_target(320, 96)) for _ in range(B)]
    tg = model.device_targets(targets, device)

    def step():
        return model.detect_device(images, *tg)

    with torch.no_grad():
        for _ in range(max(args.warmup, 2)):
            out = step()
        torch.cuda.synchronize()
        graph, mode = None, "eager"
        if not args.no_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    step()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = step()
                graph.replay()
                torch.cuda.synchronize()
                mode = "hipGraph"
            except Exception as e:                                         # noqa: BLE001
                print("hipGraph capture failed (%s); timing eager launches" % e, file=sys.stderr)
                graph, mode = None, "eager"
        run = graph.replay if graph is not None else step

        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
            # the (B,50,14) rows + validity mask go back to the host like engine/inference.py:39
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        det, topk, valid, hm = out
        det_host = det.cpu()

        rate, elapsed, n_img_total = parallel.aggregate_throughput(elapsed, B * args.steps, device=device)

        # ---- roofline of the dominant kernel (fused heads: 46 % of the forward FLOPs), HIP events on the launch stream
        feat = model.backbone.forward_nhwc(images)
        pk = model.heads.predictor._pack(feat.dtype)
        from monoflex_amd import ops
        for _ in range(3):
            ops.heads_fused(feat, pk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.heads_fused(feat, pk)
        e1.record()
        torch.cuda.synchronize()
        heads_ms = e0.elapsed_time(e1) / reps

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    traffic = None
    tj = os.path.join(ROOT, "profiles", "r01_heads_traffic.json")       # PMC pass of the same kernel/config (rocprofv3 cannot run inside the bench)
    if os.path.exists(tj) and args.dtype == "bf16" and B == 8:
        traffic = json.load(open(tj))["traffic_bytes"]
    n_img = n_img_total
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
    achieved = HEADS_GFLOP_PER_IMG * B / heads_ms                # GFLOP / ms = TFLOP/s
    res = {
        "metric": "images/sec at 1280x384, DLA-34+DCNv2 forward+decode",
        "value": round(n_img / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "DLA-34+DCNv2+9 heads+edge fusion forward + NMS/top-K/3D decode, batch %d per GPU, "
                               "1280x384, %s (BASELINE.json configs[1])" % (B, args.dtype),
                   "batch_per_gpu": B, "launch": mode, "parallelism": "replicas x%d (no collective on the inference path)" % world,
                   "model_tflops_per_s": round(FWD_GFLOP_PER_IMG * n_img / elapsed / 1e3, 2),
                   "detections_last_step": int(valid.sum().item())},
        "roofline": {"kernel": "heads_fused_kernel", "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                     "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                     "avg_launch_ms": round(heads_ms, 4),
                     "algorithmic_flops_per_launch": HEADS_GFLOP_PER_IMG * B * 1e9},
    }
    if not args.no_cpu_baseline and world == 1:
        try:
            res["cpu_baseline"] = cpu_baseline(sd, args.cpu_baseline_images)
        except Exception as e:                                             # noqa: BLE001
            res["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": "failed: %s" % e}
    else:
        res["cpu_baseline"] = None
    print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
