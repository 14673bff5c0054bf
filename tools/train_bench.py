#!/usr/bin/env python
"""Training-step throughput: forward + loss + backward + AdamW on synthetic data,
fp32, one process per GPU (launch under torch.distributed.run for N > 1: DDP over RCCL, optional SyncBN).
Prints one JSON line; this is a secondary measurement, bench.py carries the headline (inference) metric."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"], help="activation dtype (parameters and their gradients stay fp32)")
    ap.add_argument("--opts", default="", help="library options k=v,... (mfx_set_option)")
    ap.add_argument("--graph", action="store_true", help="capture forward+loss+backward+optimizer in one hipGraph and replay it")
    a = ap.parse_args()
    from monoflex_amd import parallel as par
    from monoflex_amd import synthetic as S
    from monoflex_amd.config import get_cfg
    from monoflex_amd.engine.trainer import convert_sync_batchnorm, prepare_targets, train_step, wrap_data_parallel
    from monoflex_amd.model.detector import KeypointDetector
    from monoflex_amd.solver import build_optimizer
    from monoflex_amd.structures.params_3d import make_train_target
    from monoflex_amd import lib as _lib
    for kv in filter(None, a.opts.split(",")):
        k_, v_ = kv.split("=")
        _lib.check(_lib.load().mfx_set_option(k_.encode(), int(v_)), "set_option")
    rank, world, local_rank = par.init_from_env()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_cfg(os.path.join(root, "runs", "monoflex.yaml"))
    cfg.MODEL.PRETRAIN = False
    cfg.MODEL.COMPUTE_DTYPE = a.dtype
    model = KeypointDetector(cfg)
    model.load_state_dict(S.synthetic_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).train()
    model.heads.loss_evaluator.log_as_float = False                  # no host sync inside the step
    if a.sync_bn and world > 1:
        convert_sync_batchnorm(model)
    opt = build_optimizer(model, cfg, capturable=a.graph)
    net = wrap_data_parallel(model, device_ids=[local_rank]) if world > 1 else model
    seed = par.shard_seed(1000, rank, a.batch)
    imgs = S.synthetic_images(a.batch, seed=seed).to(dev)
    targets = [make_train_target(S.synthetic_train_target(seed + i)).to(dev) for i in range(a.batch)]
    targets = prepare_targets(model, targets, dev)                   # stacked once: static inputs of the step

    def step():
        return train_step(net, opt, imgs, targets)

    for _ in range(a.warmup):
        step()
    graph = None
    if a.graph:
        if world > 1:
            raise SystemExit("--graph: single-process only (DDP's bucket hooks are not captured here)")
        opt.zero_grad(set_to_none=False)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            total, _, _ = step()
    par.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_host = 0.0
    for _ in range(a.steps):
        h0 = time.perf_counter()
        if graph is not None:
            graph.replay()
        else:
            total, _, _ = step()
        t_host += time.perf_counter() - h0
    par.barrier(); torch.cuda.synchronize()
    rate, dt, n_img = par.aggregate_throughput(time.perf_counter() - t0, a.batch * a.steps, device=dev)
    if rank == 0:
        print(json.dumps({"metric": "train_images_per_sec", "value": rate, "unit": "images/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
                          "host_enqueue_ms_per_step": 1e3 * t_host / a.steps, "graph": bool(a.graph),
                          "dtype": "f32" if a.dtype == "fp32" else "bf16", "data": "synthetic", "loss": float(total),
                          "config": {"workload": "MonoFlex DLA-34 1280x384 fwd+loss+bwd+AdamW", "batch_per_gpu": a.batch,
                                     "sync_bn": bool(a.sync_bn and world > 1)}}))


if __name__ == "__main__":
    main()
