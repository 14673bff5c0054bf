"""What does SyncBN cost on the graphed step?  B=8, 1280x384, bf16, one GPU, one-rank RCCL group: the step with rank-local BN statistics
(fused two-launch BN, no collective) against the step with every BN synchronised (separate statistics / finalize / apply kernels and the
2 x 57 statistics all-reduces captured in the graph; on one rank a collective is RCCL's launch + copy, i.e. the latency floor without
the xGMI hops).  usage: python tools/sync_bn_price.py"""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from monoflex_amd import autograd as AG, synthetic as S
from monoflex_amd.engine.trainer import GraphedTrainStep, convert_sync_batchnorm, prepare_targets
from monoflex_amd.solver import build_optimizer
from monoflex_amd.structures.params_3d import make_train_target

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
dev = torch.device("cuda", 0)
out = {}
for sync in (False, True):
    AG._SYNC_BN_FORCE[0] = sync
    model, _, cfg = bench.build_model("bf16", dev, train=True)
    model.heads.loss_evaluator.log_as_float = False
    if sync:
        convert_sync_batchnorm(model)
    B = 8
    imgs = S.synthetic_images(B, seed=1000).to(dev)
    targets = prepare_targets(model, [make_train_target(S.synthetic_train_target(1000 + i)).to(dev) for i in range(B)], dev)
    opt = build_optimizer(model, cfg, capturable=True)
    step = GraphedTrainStep(model, opt, imgs, targets)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    out["sync" if sync else "local"] = (time.perf_counter() - t0) / 10 * 1e3
    del step, model, opt
    torch.cuda.empty_cache()
print("train step B=8 bf16: rank-local BN %.2f ms, SyncBN (captured collectives, one-rank group) %.2f ms" % (out["local"], out["sync"]))
dist.destroy_process_group()
