#!/usr/bin/env python
"""bench.py -- images/sec of the MonoFlex hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode infer|train] [--batch B] [--dtype bf16|fp16|fp32|fp16x2]

--mode infer (default; BASELINE configs[1] / C5 with --batch 32): one "step" = DLA-34 + DCNv2 + all heads forward +
    NMS / top-K / 3D decode over one batch of synthetic 1280x384 images already resident in HBM, replayed from one hipGraph.
    N > 1: one process per GPU, independent replicas over disjoint image shards -- the inference path has no exchange
    step, so there is no collective in the timed region (weak scaling).
--mode train (configs[2]/[3], the second half of the metric): one "step" = forward + 11 losses + backward + AdamW at
    `--batch` images per GPU; N > 1 = data parallel, gradients averaged by an RCCL all-reduce of one flat fp32 buffer
    between the captured forward/backward graph and the captured optimizer graph (engine/trainer.GraphedTrainStep).

The default run (`python bench.py --gpus N`, what the driver records) carries the whole BASELINE.json metric in ONE line: the
top level is the forward+decode number (bf16, configs[1]); `"train"` is a short graphed run of the training step (fwd + loss +
bwd + AdamW, data parallel for N > 1: the metric's "fwd+bwd img/s") with its own roofline object; `"fp32_parity"` is the mode
that meets the north-star tolerance (<= 1e-3 on logits, identical top-K against the reference's goldens) timed the same way;
`"fp16x2_parity"` meets the same tolerance on the fp16 matrix pipe (fp32 activations, every MFMA operand an fp16 (hi, lo) pair);
`"pipeline"` feeds the captured step from host frames every step (H2D, device pre-processing and the D2H of the rows inside the timed region:
what the reference's evaluation loop times, engine/inference.py:35-43, plus the feeding); N > 1 adds `"train_local_bn"` (rank-local BN
statistics) before `"train"` (SyncBN like the reference, its collectives captured in the step's graphs);
`"fp16"` is the same measurement with IEEE-half activations (same kernels instantiated for fp16, same MFMA rate; its deviation from
the reference is 4-8x smaller than bf16's -- the headline stays bf16 because BASELINE.json configs[1] names bf16);
`"train_fp16"` is the training step with fp16 activations under the dynamic loss scaler (BASELINE configs[3]'s "fp16 MFMA path";
`config.loss_scale` / `optimizer_steps_applied` say what the scaler did).
`--legs none` prints the top level only.  Every timed region is exactly `--steps` steps between barrier + synchronize; it is
repeated `--repeats` times and the MEDIAN repeat is reported (all repeats are listed in `config.timing`).

N > 1 without a torchrun environment: this script re-executes itself under `python -m torch.distributed.run` (one rank
per GPU, rendezvous on 127.0.0.1), so `python bench.py --gpus 8` is a complete command.  Rank 0 prints ONE JSON line.
"""
import argparse
import ast
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_PER_IMG = 177.99          # SURVEY 8d / BASELINE.md section 3 (2 x 88.997 GMAC)
TRAIN_GFLOP_PER_IMG = 3 * FWD_GFLOP_PER_IMG          # dgrad + wgrad each ~ forward
HEADS_GFLOP_PER_IMG = 2 * 41.185    # Appendix A: 9 x (3x3 64->256 + 1x1) per image
PEAK_BF16_TFLOPS = 2500.0           # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_F32_TFLOPS = 157.3
# PMC passes of the heads kernel (tools/pmc_heads.sh <tag>): the newest committed measurement (the bf16 kernel is unchanged since r03)
HEADS_TRAFFIC_JSON = next((f for f in (os.path.join(ROOT, "profiles", t + "_heads_traffic.json") for t in ("r06", "r05", "r04", "r03")) if os.path.exists(f)),
                          os.path.join(ROOT, "profiles", "r03_heads_traffic.json"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", default="infer", choices=["infer", "train"])
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp16", "fp16x2"],
                    help="activation dtype (parameters, gradients and accumulators are fp32); fp16 training runs under the dynamic loss scaler; "
                         "fp16x2 (inference) = fp32 activations, every MFMA operand an fp16 (hi, lo) pair: fp32-grade results on the fp16 matrix pipe")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of captured hipGraphs")
    ap.add_argument("--streams", type=int, default=0, help="inference: run the batch as this many sub-batches on forked streams inside the one "
                                                           "captured step (independent sub-batches overlap their under-filled launches and tails). "
                                                           "0 = per mode: 1 for the 16-bit modes (r05, B=8: 2.635 ms with 1, 2.680 with 2 -- the round's "
                                                           "faster 3x3 / DCN kernels fill the chip from one 8-image launch), 2 for fp32 / fp16x2 "
                                                           "(6.18 vs 6.70 ms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-images", type=int, default=6)
    ap.add_argument("--split", action="store_true", help="train mode: the segmented data-parallel form (cut backward, flat gradient buffer, "
                                                        "per-segment exchange, optimizer graph) also on one GPU, to price the segmentation")
    ap.add_argument("--sync-bn", default="auto", choices=["auto", "on", "off"],
                    help="train mode, N > 1: synchronised BatchNorm statistics (the reference trains with USE_SYNC_BN True: runs/monoflex.yaml:59). "
                         "auto = on whenever N > 1; the statistics collectives are captured inside the step's hipGraphs")
    ap.add_argument("--legs", default="all", choices=["all", "none"], help="infer mode: add the `train` and `fp32_parity` legs to the line")
    ap.add_argument("--repeats", type=int, default=None, help="timed regions of exactly --steps steps; the median is reported")
    ap.add_argument("--leg-timeout", type=int, default=420, help="seconds after which unfinished legs are reported as errors and the line is printed")
    ap.add_argument("--train-steps", type=int, default=50, help="timed steps per repeat of the `train` legs (SURVEY 8d: >= 50 timed iterations)")
    ap.add_argument("--train-warmup", type=int, default=10)
    ap.add_argument("--train-repeats", type=int, default=3, help="timed regions of the `train` legs; the median is reported")
    ap.add_argument("--train-batches", type=int, default=4, help="distinct synthetic batches rotated through the training step")
    ap.add_argument("--opts", default="", help="library tuning options k=v,... (mfx_set_option), for experiments")
    ap.add_argument("--no-families", action="store_true", help="skip the per-family stage timings (profiling runs: their stand-alone replays would end the trace)")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 30 if a.mode == "infer" else 50
    if a.warmup is None:
        a.warmup = 5 if a.mode == "infer" else 10
    if a.repeats is None:
        a.repeats = 5 if a.mode == "infer" else 3
    return a


def timed_repeats(run, steps, repeats, images_per_repeat, device):
    """`repeats` timed regions of exactly `steps` steps, each bracketed by barrier + synchronize on both sides; per repeat the MAX
    elapsed over ranks and the SUM of images over ranks.  Returns (median elapsed s, total images of one repeat, [elapsed s])."""
    import statistics
    import torch
    from monoflex_amd import parallel
    all_s, n_img = [], images_per_repeat
    for _ in range(max(1, repeats)):
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        parallel.barrier()
        dt = time.perf_counter() - t0
        _, dt, n_img = parallel.aggregate_throughput(dt, images_per_repeat, device=device)
        all_s.append(dt)
    return statistics.median(all_s), n_img, all_s


def respawn_under_torchrun(args):
    """`bench.py --gpus N` outside a torchrun environment: start N ranks (reference engine/launch.py:23-89 does the
    same with mp.spawn + NCCL init) and hand over the exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def build_model(dtype, device, train=False):
    from monoflex_amd import synthetic as S
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.detector import KeypointDetector
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.MODEL.PRETRAIN = False
    if not train:
        cfg.DATASETS.TEST_SPLIT = "test"
    cfg.MODEL.COMPUTE_DTYPE = dtype
    m = KeypointDetector(cfg)
    m = m.train() if train else m.eval()
    # cls_bias -1.0: all 50 slots pass 0.2 (worst-case decode)
    sd = S.synthetic_state_dict(m.state_dict(), seed=0, **({} if train else {"cls_bias": -1.0}))
    m.load_state_dict(sd)
    return m.to(device), sd, cfg


def cpu_baseline(sd, n_images):
    """The oracle (CPU port of the reference path) timed on this host's cores.  Two bounded samples: forward+decode at
    B=1 (the bench's own workload, `value`) and BASELINE configs[0] (C1: backbone forward on 4 images)."""
    import torch
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
    out = {"value": round(n_images / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "%d x (1,3,384,1280) image, forward+decode, oracle/monoflex_ref.py (torch fp32 convs + C DCN, "
                     "OpenMP), %.1f s" % (n_images, dt)}
    try:
        imgs = S.synthetic_images(4, 384, 1280, seed=1000)
        t0 = time.time()
        with torch.no_grad():
            ref.backbone(imgs)
        dt = time.time() - t0
        out["c1_backbone_b4"] = {"value": round(4 / dt, 4), "unit": "images/s",
                                 "sample": "BASELINE configs[0]: DLA-34+DCNv2 backbone forward on 4 x 1280x384, %.1f s" % dt}
    except Exception as e:                                             # noqa: BLE001
        out["c1_backbone_b4"] = {"value": None, "sample": "failed: %s" % e}
    return out


def golden_meta():
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_full.npz"))
    return g, ast.literal_eval(str(g["meta"]))


def bench_images(B, rank, device):
    """Rank r's synthetic batch.  Rank 0's first images are the frames of tests/golden/e2e_full.npz (the reference's own outputs exist for them: every
    timed mode reports its deviation from the reference on exactly what it was timed on); all other frames are drawn from disjoint seed streams."""
    import torch
    from monoflex_amd import parallel, synthetic as S
    seeds = [parallel.shard_seed(3000, rank, B) + i for i in range(B)]
    if rank == 0:
        gold = list(golden_meta()[1]["seeds"])[:B]
        seeds[:len(gold)] = gold
    return torch.cat([S.synthetic_images(1, 384, 1280, seed=s_) for s_ in seeds]).to(device)


def deviation_vs_reference(out, dtype):
    """Images 0 .. 3 of rank 0's batch are the frames of tests/golden/e2e_full.npz (same weights): how far the benchmarked mode is from the
    REFERENCE's own outputs (logits at ~560 pixels per image, the top-K sequence, (N,14) rows), worst case over the golden images in the batch."""
    import numpy as np
    import torch
    g, meta = golden_meta()
    if meta["cls_bias"] != -1.0:
        return None
    nb = int(out[0].shape[0])
    ngold = min(nb, len(meta["seeds"]))
    dl = dr = 0.0
    agree, same_order, row_delta = 1.0, True, 0.0
    for n in range(ngold):
        det, topk, valid, hm = [t[n].float().cpu() for t in out]
        p_ = "img%d_" % n
        pix = torch.as_tensor(g[p_ + "pix"])
        dl = max(dl, float(np.abs(hm[..., :3].permute(2, 0, 1).reshape(3, -1)[:, pix].numpy() - g[p_ + "cls_logits_at"]).max()))
        dr = max(dr, float(np.abs(hm[..., 8:58].permute(2, 0, 1).reshape(50, -1)[:, pix].numpy() - g[p_ + "reg_at"]).max()))
        # a peak = (class, pixel): one pixel can rank for two classes
        mine = topk[:, 2].numpy().astype(np.int64) * (1 << 20) + topk[:, 1].numpy().astype(np.int64)
        ref = g[p_ + "topk_cls"].astype(np.int64) * (1 << 20) + g[p_ + "topk_index"].astype(np.int64)
        agree = min(agree, len(set(mine.tolist()) & set(ref.tolist())) / float(len(ref)))
        same = bool(np.array_equal(mine, ref))
        same_order = same_order and same
        rows, want = det[valid.bool()].numpy(), g[p_ + "result"]
        if same and rows.shape == want.shape and row_delta is not None:
            row_delta = max(row_delta, float(np.abs(rows - want).max())) if len(want) else row_delta
        else:
            row_delta = None                                   # rows are compared only where both sides decoded the same peaks in the same order
    return {"golden": "tests/golden/e2e_full.npz (reference KeypointDetector, image seeds %s; no two of a frame's top-51 peaks closer than 4e-4 in logit units)"
                      % meta["seeds"][:ngold], "dtype": dtype, "golden_images_in_batch": ngold,
            "max_abs_dlogit": round(dl, 6), "max_abs_dreg": round(dr, 6), "topk_index_agreement": round(agree, 4),
            "topk_identical_order": same_order, "max_abs_row_delta": row_delta,
            "north_star_bar": "fp32 / fp16x2 modes: <=1e-3 on logits, identical top-K sequence (tests/test_gpu_e2e.py)"}


def _hip_event_ms(fn, reps):
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# SURVEY section 8d / Appendix A, GMAC per 1280x384 image: F1 (stem + level0 + level1) 2.85, DLA levels 2-5 (3x3, 1x1 project, Root) 27.25,
# the 16 DCN modules (3x3 main GEMM 13.3 + 27-channel offset/mask conv 4.36 + depthwise up-sampling 0.05) 17.71, the nine heads 41.185
FAMILY_GMAC_PER_IMG = {"f1": 2.85, "trunk_levels_2_5": 30.097 - 2.85, "dcn_modules": 17.666 + 0.049, "heads": 41.185}


def _graph_replay_ms(fn, reps=10):
    """Average GPU time of `fn` (a capturable sequence of launches): ONE capture, `reps` replays between two HIP events on the replay stream."""
    import torch
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def family_rooflines(model, images, tg, dtype, B, heads_ms):
    """Per kernel FAMILY of the forward step, timed live: each stage of the network is captured alone (single stream, whole batch) and
    replayed between HIP events; a family's time is the difference of two nested stages.  `frac` = algorithmic FLOPs / time /
    the dense 16-bit MFMA peak (fp32 mode: the f32 MFMA peak) -- the figures VERDICT r4 asked to see beside the heads kernel's."""
    import torch
    from monoflex_amd import lib as L, ops
    from monoflex_amd.model.backbone import dla_dcn as D
    bb = model.backbone
    base = bb.base
    cd = bb.compute_dtype
    with torch.no_grad():
        feat = bb.forward_nhwc(images)
        pk = model.heads.predictor._pack(ops.compute_tag(model.heads.predictor, feat.dtype))
        heads_ms = _graph_replay_ms(lambda: ops.heads_fused(feat, pk))       # (same clock as the other stages: replayed, not eager launches)
        t_total = _graph_replay_ms(lambda: model.detect_device(images, *tg))
        t_backbone = _graph_replay_ms(lambda: bb.forward_nhwc(images))
        t_base = _graph_replay_ms(lambda: base(images, cd))
        tag = ops.compute_tag(base, cd)
        packs = base.__dict__.get("_packs", {})
        t_f1 = None
        if D.FUSE_F1[0] and ("stem", tag) in packs and (tag == ops.F16X2 or tag in (torch.bfloat16, torch.float16)):
            p0 = D._conv_bn(base.level0, "c0", base.level0[0], base.level0[1], cd, L.ACT_RELU)
            p1 = D._conv_bn(base.level1, "c0", base.level1[0], base.level1[1], cd, L.ACT_RELU)
            t_f1 = _graph_replay_ms(lambda: ops.f1_fused(images, packs[("stem", tag)], p0, p1))
    peak = PEAK_F32_TFLOPS if dtype == "fp32" else PEAK_BF16_TFLOPS
    rows = []

    def row(name, ms, what):
        gf = 2.0 * FAMILY_GMAC_PER_IMG[name] * B if name in FAMILY_GMAC_PER_IMG else None
        rows.append({"family": name, "kernels": what, "us_per_step": round(1e3 * ms, 1), "gflop": None if gf is None else round(gf, 1),
                     "tflops": None if gf is None else round(gf / ms, 1), "frac": None if gf is None else round(gf / ms / peak, 4)})
    row("heads", heads_ms, "heads_fused_kernel (nine 3x3 64->256 + ABN + 1x1 branches)")
    row("dcn_modules", t_backbone - t_base, "16 x (offset/mask conv + DCNv2 + BN + ReLU), 8 x up-sample + skip add (DLAUp / IDAUp)")
    if t_f1 is not None:
        row("trunk_levels_2_5", t_base - t_f1, "3x3 / stride-2 / 1x1 / Root concat convs + BN + ReLU + residual, max-pools of DLA levels 2-5")
        row("f1", t_f1, "f1_fused_kernel (stem 7x7 -> level0 -> level1)")
    else:
        rows.append({"family": "base", "kernels": "stem + DLA levels 0-5", "us_per_step": round(1e3 * t_base, 1), "gflop": round(2 * 30.097 * B, 1),
                     "tflops": round(2 * 30.097 * B / t_base, 1), "frac": round(2 * 30.097 * B / t_base / peak, 4)})
    row("edge_fusion_and_decode", t_total - t_backbone - heads_ms, "edge row convs + scatter, NMS / top-K / 3D decode (latency-bound, no FLOP figure)")
    return {"timing": "each stage captured alone on one stream (whole batch) and replayed 10x between HIP events; nested stages subtracted",
            "single_stream_step_us": round(1e3 * t_total, 1), "peak_tflops": peak, "families": rows}


def heads_traffic(dtype, B):
    """HBM bytes per heads launch from the committed PMC pass (rocprofv3 cannot run inside the bench)."""
    if os.path.exists(HEADS_TRAFFIC_JSON):
        t = json.load(open(HEADS_TRAFFIC_JSON))
        if t.get("dtype") == dtype and t.get("batch") == B:
            return t["traffic_bytes"], t.get("source", os.path.relpath(HEADS_TRAFFIC_JSON, ROOT))
    return None, None


def run_infer(args, rank, world, device, dtype=None, leg=False):
    """The forward+decode measurement in `dtype` (default: --dtype).  `leg=True`: the short form used for the `fp32_parity` leg
    (one repeat, no roofline / CPU baseline)."""
    import torch
    from monoflex_amd import lib, ops, parallel, synthetic as S
    from monoflex_amd.structures.params_3d import make_test_target
    dtype = dtype or args.dtype
    lib.load()
    model, sd, _ = build_model(dtype, device)
    B = args.batch
    # sub-batches on forked streams inside the one graph: 16-bit modes fill the chip from one 8-image stream (r05: 2.635 vs 2.680 ms; r06 kernels: 2.468 / 2.474
    # vs 2.470 / 2.459 ms -- a wash), but at B = 32 two 16-image streams overlap one another's launch tails: 8.37 -> 7.83 ms (3825 -> 4085 img/s; four
    # streams 7.99), profiles/r06_streams.md
    nstreams = args.streams if args.streams > 0 else ((2 if B >= 16 else 1) if dtype in ("bf16", "fp16") else 2)
    images = bench_images(B, rank, device)                   # resident in HBM; rank 0: the golden frames first
    targets = [make_test_target(S.synthetic_target(320, 96)) for _ in range(B)]
    tg = model.device_targets(targets, device)

    def step():
        return model.detect_device(images, *tg)

    if nstreams > 1 and B % nstreams == 0 and B // nstreams >= 2 and not args.no_graph:
        # the batch as `streams` sub-batches on forked streams inside ONE step / one graph: independent sub-batches overlap their
        # under-filled launches (level 4/5 convs and DCNs run 120-240 workgroups on 256 CUs at B = 8) and their tails
        ns = nstreams
        if B % ns:
            raise SystemExit("--streams must divide --batch")
        per = B // ns
        parts = [(images[i * per:(i + 1) * per].contiguous(), model.device_targets(targets[i * per:(i + 1) * per], device)) for i in range(ns)]
        side_streams = [torch.cuda.Stream() for _ in range(ns - 1)]

        def step():                                               # noqa: F811
            cur = torch.cuda.current_stream()
            outs = [None] * ns
            for i, st in enumerate(side_streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    outs[i + 1] = model.detect_device(parts[i + 1][0], *parts[i + 1][1])
            outs[0] = model.detect_device(parts[0][0], *parts[0][1])
            for st in side_streams:
                cur.wait_stream(st)
            return tuple(torch.cat([o[k] for o in outs]) for k in range(3)) + ([o[3] for o in outs],)

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

        elapsed, n_img, all_s = timed_repeats(run, args.steps, 1 if leg else args.repeats, B * args.steps, device)
        det, topk, valid, hm = out
        if isinstance(hm, list):
            hm = torch.cat(hm)
            out = (det, topk, valid, hm)
        if leg:
            feat = model.backbone.forward_nhwc(images)
            pk = model.heads.predictor._pack(ops.compute_tag(model.heads.predictor, feat.dtype))
            for _ in range(3):
                ops.heads_fused(feat, pk)
            heads_ms = _hip_event_ms(lambda: ops.heads_fused(feat, pk), 20)
            fam = None if args.no_families else family_rooflines(model, images, tg, dtype, B, heads_ms)
            if rank != 0:
                return None
            peak = PEAK_F32_TFLOPS if dtype == "fp32" else PEAK_BF16_TFLOPS
            achieved = HEADS_GFLOP_PER_IMG * B / heads_ms
            leg_roof = {"kernel": "heads_fused_kernel", "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": None, "avg_launch_ms": round(heads_ms, 4),
                        "algorithmic_flops_per_launch": HEADS_GFLOP_PER_IMG * B * 1e9,
                        **({"mfma_products_per_multiply": 3, "frac_of_issued_mfma_work": round(3 * achieved / peak, 4)} if dtype == "fp16x2" else {})}
            return {"roofline": leg_roof, "roofline_families": fam,
                    "metric": "images/sec at 1280x384, DLA-34+DCNv2 forward+decode (%s)"
                              % ("the mode inside the north-star tolerance, f32 MFMA" if dtype == "fp32" else
                                 "inside the north-star tolerance: fp32 activations, split-precision fp16 (hi, lo) MFMA operands, fp32 accumulate"
                                 if dtype == "fp16x2" else
                                 "BASELINE.json configs[4] (C5): batch 32 per GPU, hipGraph-captured DLA+DCN+heads+top-K" if B == 32 and dtype == "bf16" else
                                 "IEEE-half activations: the bf16 mode's kernels and speed, 4-8x closer to the reference"),
                    "value": round(n_img / elapsed, 2), "unit": "images/s", "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                    "steps": args.steps, "dtype": dtype, "batch_per_gpu": B, "launch": mode,
                    **({"operand_range_ok": bool(lib.f16x2_range_ok())} if dtype == "fp16x2" else {}),     # the split mode's one precondition: |activation| <= 65504
                    "vs_reference": deviation_vs_reference(out, dtype)}

        # ---- roofline of the dominant kernel (fused heads: 46 % of the forward FLOPs), HIP events on the launch stream
        feat = model.backbone.forward_nhwc(images)
        pk = model.heads.predictor._pack(ops.compute_tag(model.heads.predictor, feat.dtype))
        for _ in range(3):
            ops.heads_fused(feat, pk)
        heads_ms = _hip_event_ms(lambda: ops.heads_fused(feat, pk), 20)

        fam = None if args.no_families else family_rooflines(model, images, tg, dtype, B, heads_ms)
    if rank != 0:
        return None
    peak = PEAK_F32_TFLOPS if dtype == "fp32" else PEAK_BF16_TFLOPS          # (dense fp16 MFMA = the bf16 rate; fp16x2 issues 3 fp16 products per multiply)
    achieved = HEADS_GFLOP_PER_IMG * B / heads_ms                # GFLOP / ms = TFLOP/s
    traffic, traffic_source = heads_traffic(dtype, B)
    res = {
        "metric": "images/sec at 1280x384, DLA-34+DCNv2 forward+decode",
        "value": round(n_img / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": "DLA-34+DCNv2+9 heads+edge fusion forward + NMS/top-K/3D decode, batch %d per GPU, "
                               "1280x384, %s (BASELINE.json configs[%d])" % (B, dtype, 4 if B == 32 else 1),
                   "timing": {"repeats": len(all_s), "reported": "median repeat", "steps_per_repeat": args.steps,
                              "ms_per_step_each": [round(1e3 * t / args.steps, 4) for t in all_s]},
                   "batch_per_gpu": B, "launch": mode, "parallelism": "replicas x%d (no collective on the inference path)" % world,
                   "sub_batch_streams": nstreams if (nstreams > 1 and B % nstreams == 0 and B // nstreams >= 2 and not args.no_graph) else 1,
                   "model_tflops_per_s": round(FWD_GFLOP_PER_IMG * n_img / elapsed / 1e3, 2),
                   "detections_last_step": int(valid.sum().item()),
                   "h2d_excluded": True, "d2h_excluded": True,
                   "timed_region": "graph replays only: the fp32 image batch is already in HBM and the (B,50,14) rows stay "
                                   "on the device (the reference's timer includes the D2H, engine/inference.py:35-43)",
                   **({"operand_range_ok": bool(lib.f16x2_range_ok())} if dtype == "fp16x2" else {}),
                   "vs_reference": deviation_vs_reference(out, dtype)},
        "roofline": {"kernel": "heads_fused_kernel", "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                     "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "avg_launch_ms": round(heads_ms, 4),
                     "algorithmic_flops_per_launch": HEADS_GFLOP_PER_IMG * B * 1e9,
                     **({"mfma_products_per_multiply": 3, "frac_of_issued_mfma_work": round(3 * achieved / peak, 4)} if dtype == "fp16x2" else {})},
        "roofline_families": fam,
    }
    if not args.no_cpu_baseline and world == 1:
        try:
            res["cpu_baseline"] = cpu_baseline(sd, args.cpu_baseline_images)
        except Exception as e:                                             # noqa: BLE001
            res["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": "failed: %s" % e}
        c1 = (res["cpu_baseline"] or {}).get("c1_backbone_b4")
        if c1:                                                             # BASELINE configs[0] at the top level too (a summary of the line keeps it)
            res["cpu_baseline_c1"] = dict(c1, cores=res["cpu_baseline"].get("cores"), kind="port")
    else:
        res["cpu_baseline"] = None
    return res


def run_pipeline(args, rank, world, device, dtype=None):
    """What the reference's evaluation loop times per batch (engine/inference.py:26-56: `model(images, targets)` + the D2H copy of the
    result rows + synchronize), PLUS the host side of feeding it: a fresh batch of uint8 camera frames every step -> pinned staging
    buffer -> H2D -> device normalise / centre-pad (csrc/kitti_encode.hip, what data.DeviceLoader runs per batch) written straight into
    the captured step's input -> graph replay (DLA + DCN + heads + top-K + decode) -> D2H of the (B,50,14) rows and validity flags ->
    synchronize.  Strictly sequential, like the reference's loop (no overlap between consecutive batches).  File reading and PNG decoding
    are DataLoader-worker work outside the reference's timer too and are not part of the loop: the frames are synthetic 375x1242 RGB."""
    import ctypes
    import numpy as np
    import torch
    from monoflex_amd import lib, parallel, synthetic as S
    from monoflex_amd.structures.params_3d import make_test_target
    dtype = dtype or args.dtype
    L = lib.load()
    model, _, _ = build_model(dtype, device)
    B, fh, fw, pool = args.batch, 375, 1242, 4
    rng = np.random.RandomState(1000 + rank)
    frames = [rng.randint(0, 256, size=(B, fh, fw, 3), dtype=np.uint8) for _ in range(pool)]
    nbytes = fh * fw * 3
    stride = (nbytes + 15) // 16 * 16
    host = torch.empty(B * stride, dtype=torch.uint8, pin_memory=True)
    hview = host.numpy().reshape(B, stride)[:, :nbytes]
    pixels = torch.empty(B * stride, dtype=torch.uint8, device=device)
    offsets = torch.arange(B, dtype=torch.int64, device=device) * stride
    img_wh = torch.tensor([[fw, fh]] * B, dtype=torch.int32, device=device)
    flip = torch.zeros(B, dtype=torch.int32, device=device)
    images = torch.zeros((B, 3, 384, 1280), dtype=torch.float32, device=device)              # the captured step's static input
    mean, std = (ctypes.c_float * 3)(0.485, 0.456, 0.406), (ctypes.c_float * 3)(0.229, 0.224, 0.225)
    tg = model.device_targets([make_test_target(S.synthetic_target(320, 96)) for _ in range(B)], device)
    rows_host = torch.empty((B, 50, 14), dtype=torch.float32, pin_memory=True)
    valid_host = torch.empty((B, 50), dtype=torch.int32, pin_memory=True)

    def feed(i):
        hview[:] = frames[i % pool].reshape(B, nbytes)
        pixels.copy_(host, non_blocking=True)
        lib.check(L.mfx_kitti_preprocess_u8(pixels.data_ptr(), offsets.data_ptr(), img_wh.data_ptr(), flip.data_ptr(), images.data_ptr(), B, 1280, 384,
                                            mean, std, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "mfx_kitti_preprocess_u8")

    with torch.no_grad():
        feed(0)
        for _ in range(2):
            out = model.detect_device(images, *tg)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model.detect_device(images, *tg)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = model.detect_device(images, *tg)
        det, _, valid, _ = out
        it = [0]

        def run():
            feed(it[0])
            it[0] += 1
            graph.replay()
            rows_host.copy_(det, non_blocking=True)
            valid_host.copy_(valid, non_blocking=True)
            torch.cuda.synchronize()
        for _ in range(3):
            run()
        elapsed, n_img, all_s = timed_repeats(run, args.steps, 3, B * args.steps, device)
        # the same loop without the host side (the reference's own timed region: model + D2H + sync on a batch already on the device)
        def run_model_d2h():
            graph.replay()
            rows_host.copy_(det, non_blocking=True)
            valid_host.copy_(valid, non_blocking=True)
            torch.cuda.synchronize()
        el2, n2, _ = timed_repeats(run_model_d2h, args.steps, 3, B * args.steps, device)

        # the overlapped feeder a serving loop would run: two pinned staging buffers and two device pixel buffers; while the graph of batch k
        # replays, the host packs batch k + 1 and a copy stream uploads it; the host waits only for the rows of batch k - 1 (two batches in flight)
        host2 = [host, torch.empty_like(host).pin_memory()]
        pix2 = [pixels, torch.empty_like(pixels)]
        rows2 = [rows_host, torch.empty_like(rows_host).pin_memory()]
        val2 = [valid_host, torch.empty_like(valid_host).pin_memory()]
        copy_stream = torch.cuda.Stream()
        uploaded = [torch.cuda.Event(), torch.cuda.Event()]
        done = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]
        cur = torch.cuda.current_stream()
        for e in done + consumed:
            e.record(cur)
        it2 = [0]

        def run_overlapped():
            k = it2[0]; it2[0] += 1
            s_ = k & 1
            done[s_].synchronize()                                # the rows of batch k - 2 (same slot) are on the host: its buffers are free
            host2[s_].numpy().reshape(B, stride)[:, :nbytes][:] = frames[k % pool].reshape(B, nbytes)
            copy_stream.wait_event(consumed[s_])                  # (the pre-processing kernel that read pix2[s_] two batches ago has run)
            with torch.cuda.stream(copy_stream):
                pix2[s_].copy_(host2[s_], non_blocking=True)
                uploaded[s_].record(copy_stream)
            cur.wait_event(uploaded[s_])
            lib.check(L.mfx_kitti_preprocess_u8(pix2[s_].data_ptr(), offsets.data_ptr(), img_wh.data_ptr(), flip.data_ptr(), images.data_ptr(), B, 1280, 384,
                                                mean, std, ctypes.c_void_p(cur.cuda_stream)), "mfx_kitti_preprocess_u8")
            consumed[s_].record(cur)
            graph.replay()
            rows2[s_].copy_(det, non_blocking=True)
            val2[s_].copy_(valid, non_blocking=True)
            done[s_].record(cur)
        for _ in range(4):
            run_overlapped()
        el3, n3, all3 = timed_repeats(run_overlapped, args.steps, 3, B * args.steps, device)
    if rank != 0:
        return None
    return {"metric": "images/sec at 1280x384, forward+decode fed from the host every step (fresh uint8 frames -> H2D -> device "
                      "normalise/pad -> graph replay -> D2H of the rows -> sync; sequential, one batch in flight)",
            "value": round(n_img / elapsed, 2), "unit": "images/s", "ms_per_step": round(1e3 * elapsed / args.steps, 4), "steps": args.steps,
            "dtype": dtype, "batch_per_gpu": B, "launch": "hipGraph", "h2d_included": True, "d2h_included": True,
            "h2d_bytes_per_step": int(B * stride), "d2h_bytes_per_step": int(rows_host.numel() * 4 + valid_host.numel() * 4),
            "detections_last_step": int(valid_host.sum()),
            "reference_timed_region": {"what": "model + D2H + synchronize per batch, input already on the device (engine/inference.py:35-43)",
                                       "value": round(n2 / el2, 2), "unit": "images/s", "ms_per_step": round(1e3 * el2 / args.steps, 4)},
            "overlapped": {"what": "the same work with two batches in flight: the host packs and a copy stream uploads batch k + 1 while the graph of batch k "
                                   "replays; the host waits for the rows of batch k - 1 only",
                           "value": round(n3 / el3, 2), "unit": "images/s", "ms_per_step": round(1e3 * el3 / args.steps, 4),
                           "ms_per_step_each": [round(1e3 * t / args.steps, 4) for t in all3]},
            "timing": {"repeats": len(all_s), "reported": "median repeat", "ms_per_step_each": [round(1e3 * t / args.steps, 4) for t in all_s]}}


def cpu_train_baseline(n_steps=1):
    """Reference-shaped training step on the host cores: the oracle network in training mode + the loss module,
    forward + 11 losses + backward at B=1 (a bounded sample of the same workload)."""
    import torch
    from monoflex_amd import synthetic as S
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.head.detector_loss import make_loss_evaluator
    from monoflex_amd.structures.params_3d import make_train_target
    from oracle import monoflex_ref as R
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    ref = R.KeypointDetectorRef().train()
    ref.load_state_dict(S.synthetic_state_dict(ref.state_dict(), seed=0))
    tgt = S.synthetic_train_target(1000)
    ei = torch.as_tensor(tgt["edge_indices"]).long()[None]
    el = torch.as_tensor([int(tgt["edge_len"])]).long()
    evaluator = make_loss_evaluator(cfg)
    t0 = time.time()
    for _ in range(n_steps):
        maps = ref.forward_maps(S.synthetic_images(1, seed=1000), ei, el)
        loss_dict, _ = evaluator(maps, [make_train_target(tgt)])
        sum(loss_dict.values()).backward()
    dt = time.time() - t0
    return {"value": round(n_steps / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d x (B=1 forward + 11 losses + backward), oracle/monoflex_ref.py, %.1f s" % (n_steps, dt)}


def run_train(args, rank, world, device, steps=None, warmup=None, leg=False, dtype=None, sync_bn=None):
    import torch
    from monoflex_amd import lib, parallel, synthetic as S
    from monoflex_amd.engine.trainer import (GraphedTrainStep, LossScaler, convert_sync_batchnorm, prepare_targets, train_step,
                                             wrap_data_parallel)
    from monoflex_amd.solver import build_optimizer
    from monoflex_amd.structures.params_3d import make_train_target
    lib.load()
    dtype = dtype or args.dtype
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    model, _, cfg = build_model(dtype, device, train=True)
    model.heads.loss_evaluator.log_as_float = False                  # no host sync inside the step
    if sync_bn is None:
        sync_bn = args.sync_bn != "off"
    sync_bn = bool(sync_bn and world > 1)
    if sync_bn:
        convert_sync_batchnorm(model)
    B = args.batch
    seed = parallel.shard_seed(1000, rank, B)
    # `nbatch` DISTINCT synthetic batches (images, objects, edge sequences), rotated step by step: the captured step's static buffers are
    # overwritten in place before every replay (GraphedTrainStep.load_batch -- what do_train does with the loader's batches), inside the
    # timed region
    nbatch = max(1, int(args.train_batches))
    batches = []
    for k in range(nbatch):
        sk = seed + 7919 * k
        im_k = S.synthetic_images(B, seed=sk).to(device)
        tg_k = prepare_targets(model, [make_train_target(S.synthetic_train_target(sk + i)).to(device) for i in range(B)], device)
        batches.append((im_k, tg_k))
    from monoflex_amd.engine.trainer import _clone_targets
    imgs, targets = batches[0][0].clone(), _clone_targets(batches[0][1])       # the step's private static buffers (captured addresses)
    graphed = not args.no_graph                                        # (SyncBN's statistics collectives are captured with the step)
    opt = build_optimizer(model, cfg, capturable=graphed)
    scaler = LossScaler.for_model(model, device)                       # fp16 activations: dynamic loss scaling (None otherwise)
    if scaler is not None:
        scaler.attach(opt)
    if graphed:
        # (the bench opts in to the CAPTURED SyncBN collectives -- library default: eager -- because this leg runs behind the watchdog and
        # after `train_local_bn`, whose graphs hold no collective)
        step = GraphedTrainStep(model, opt, imgs, targets, split=True if args.split else None, scaler=scaler, graph_sync_bn=True)
        mode = "hipGraph (fwd+loss+bwd+AdamW)" if not step.split else \
            "%d hipGraphs (fwd+loss+bwd piece 0 | bwd pieces 1..%d), RCCL all-reduce of piece k's slice of the flat fp32 gradient buffer on a " \
            "comm stream while piece k+1 runs | hipGraph AdamW" % (len(step.graphs), len(step.graphs) - 1)
    else:
        net = wrap_data_parallel(model, device_ids=[device.index]) if world > 1 else model
        mode = "eager" + (" + torch DDP (bucketed all-reduce overlapped with backward)" if world > 1 else "")

        def step():
            return train_step(net, opt, imgs, targets, scaler=scaler)[0]
    it = [0]
    if graphed:
        def run_one():
            if nbatch > 1:
                step.load_batch(*batches[it[0] % nbatch])
            it[0] += 1
            return step()
    else:
        def run_one():
            im_k, tg_k = batches[it[0] % nbatch]
            it[0] += 1
            return train_step(net, opt, im_k, tg_k, scaler=scaler)[0]
    for _ in range(warmup):
        loss = run_one()
    last = [loss]

    def run():
        last[0] = run_one()
    elapsed, n_img, all_s = timed_repeats(run, steps, args.train_repeats if leg else max(args.repeats, 1), B * steps, device)
    loss_v = float(last[0])
    overlap = bool(getattr(step, "overlap", False))
    loss_scale_info = {}
    if scaler is not None:                                             # fp16: the dynamic loss scale after the run and the steps AdamW applied
        applied = min(int(st["step"]) for st in opt.state.values()) if opt.state else 0
        total = warmup + steps * len(all_s)                            # (GraphedTrainStep's capture warm-up is rolled back: it counts no steps)
        loss_scale_info = {"loss_scale": float(scaler.scale), "optimizer_steps_applied": applied, "optimizer_steps_run": total}

    # ---- roofline of the dominant training kernel family, timed live on one representative layer
    from tools.train_layer_bench import dominant_kernel_roofline
    roof = dominant_kernel_roofline(dtype, B, device)
    if rank != 0:
        return None
    res = {
        "metric": "images/sec at 1280x384, DLA-34+DCNv2 forward+loss+backward+AdamW (training step)",
        "value": round(n_img / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(1e3 * elapsed / steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": "MonoFlex training step (fwd + 11 losses + bwd + AdamW), batch %d per GPU, 1280x384, %s activations, "
                               "fp32 parameters/gradients (BASELINE.json configs[2]/[3] per-GPU shape)" % (B, dtype),
                   "batch_per_gpu": B, "global_batch": B * world, "launch": mode, "distinct_batches_rotated": nbatch,
                   "parallelism": "dp%d" % world if world > 1 else "single GPU", "sync_bn": sync_bn,
                   "overlap": overlap,
                   "timing": {"repeats": len(all_s), "reported": "median repeat", "steps_per_repeat": steps,
                              "ms_per_step_each": [round(1e3 * t / steps, 4) for t in all_s]},
                   "model_tflops_per_s": round(TRAIN_GFLOP_PER_IMG * n_img / elapsed / 1e3, 2), "loss_last_step": loss_v,
                   "h2d_excluded": True, "bn_one_launch_barriers_ok": bool(lib.bn_onepass_ok()), **loss_scale_info},
        "roofline": roof,
    }
    if leg:
        return res
    if not args.no_cpu_baseline and world == 1:
        try:
            res["cpu_baseline"] = cpu_train_baseline()
        except Exception as e:                                             # noqa: BLE001
            res["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": "failed: %s" % e}
    else:
        res["cpu_baseline"] = None
    return res


def order_line(res, seen, world):
    """The ONE line in the order a truncated capture keeps best: the contract's keys, the roofline / CPU-baseline objects, then every leg's headline
    number as a top-level scalar (so a driver that parses the line has them even when it drops nested objects: VERDICT r5 item 8), the ranks that met,
    the legs themselves, and the long per-family blob last."""
    head = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline"]
    out = {k: res[k] for k in head if k in res}

    def leg_val(name, key="value"):
        v = res.get(name)
        return v.get(key) if isinstance(v, dict) and "error" not in v else None
    scal = {"b32_images_per_s": leg_val("b32"), "fp16x2_images_per_s": leg_val("fp16x2_parity"), "fp32_images_per_s": leg_val("fp32_parity"),
            "fp16_images_per_s": leg_val("fp16"), "train_images_per_s": leg_val("train"), "train_ms_per_step": leg_val("train", "ms_per_step"),
            "train_fp16_images_per_s": leg_val("train_fp16"), "train_local_bn_images_per_s": leg_val("train_local_bn")}
    out.update({k: v for k, v in scal.items() if v is not None})
    out["rccl_ranks"] = len(seen)
    out["ranks_seen"] = seen
    tail = ("roofline_families",)
    for k, v in res.items():
        if k not in out and k not in tail:
            out[k] = v
    for k in tail:
        if k in res:
            out[k] = res[k]
    return out


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(respawn_under_torchrun(args))
    import torch
    from monoflex_amd import parallel
    rank, world, local_rank = parallel.init_from_env(backend="nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; reporting n_gpus=%d" % (args.gpus, world, world), file=sys.stderr)
    if args.opts:
        from monoflex_amd import lib
        for kv in filter(None, args.opts.split(",")):
            k, v = kv.split("=")
            lib.check(lib.load().mfx_set_option(k.encode(), int(v)), "set_option")
    if args.mode == "train":
        res = run_train(args, rank, world, device)
    else:
        res = run_infer(args, rank, world, device)
        if args.legs == "all" and args.dtype == "bf16" and not args.no_graph:
            import gc
            import threading
            legs = {}
            # Inference legs first, training legs last: the data-parallel training legs are the only ones with collectives, so a hang in
            # one (first multi-rank RCCL run of a path) costs the line nothing that was already measured.  N > 1: `train_local_bn` is the
            # data-parallel step with rank-local BN statistics (no collective inside the graphs), `train` / `train_fp16` synchronise
            # them like the reference (captured RCCL all-reduces); N = 1: the three coincide and `train_local_bn` is not run.
            import copy
            args_b32 = copy.copy(args)
            args_b32.batch = 32                                          # BASELINE.json configs[4] (C5): batch 32 per GPU, hipGraph, bf16
            leg_list = [("b32", lambda: run_infer(args_b32, rank, world, device, dtype="bf16", leg=True)),
                        ("fp32_parity", lambda: run_infer(args, rank, world, device, dtype="fp32", leg=True)),
                        ("fp16x2_parity", lambda: run_infer(args, rank, world, device, dtype="fp16x2", leg=True)),
                        ("pipeline", lambda: run_pipeline(args, rank, world, device)),
                        ("fp16", lambda: run_infer(args, rank, world, device, dtype="fp16", leg=True))]
            if world > 1:
                leg_list.append(("train_local_bn", lambda: run_train(args, rank, world, device, steps=args.train_steps, warmup=args.train_warmup,
                                                                     leg=True, sync_bn=False)))
            leg_list += [("train", lambda: run_train(args, rank, world, device, steps=args.train_steps, warmup=args.train_warmup, leg=True)),
                         ("train_fp16", lambda: run_train(args, rank, world, device, steps=args.train_steps, warmup=args.train_warmup, leg=True,
                                                          dtype="fp16"))]

            seen_early = parallel.ranks_seen(device)                    # (before the legs: the watchdog's line carries it too)

            def bail():
                # a leg that hangs (first multi-rank RCCL run of the segmented exchange, a wedged capture) must not take the headline with
                # it: after --leg-timeout seconds rank 0 prints the line with what is finished and every rank leaves
                if rank == 0:
                    out = dict(res, **legs)
                    for k, _ in leg_list:
                        out.setdefault(k, {"error": "leg did not finish within %d s" % args.leg_timeout})
                    print(json.dumps(order_line(out, seen_early, world)), flush=True)
                os._exit(3)                                               # the line is complete as far as it goes; a hung leg is NOT a clean run
            dog = threading.Timer(args.leg_timeout, bail)
            dog.daemon = True
            dog.start()
            for name, fn in leg_list:
                gc.collect()
                torch.cuda.empty_cache()
                try:
                    legs[name] = fn()
                except Exception as e:                                         # noqa: BLE001  (a failed leg is reported, not fatal)
                    legs[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            dog.cancel()
            if rank == 0:
                res.update(legs)
    seen = parallel.ranks_seen(device)                                     # all-gather inside the job: which ranks really met (every rank calls it)
    if rank == 0:
        print(json.dumps(order_line(res, seen, world)))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
