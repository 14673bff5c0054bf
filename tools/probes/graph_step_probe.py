"""Crash bisect: graphed training step on the small test configuration.  usage: graph_step_probe.py dtype v1|v2 [tr0]"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_train_step as T
from monoflex_amd import autograd as AG, lib as L
from monoflex_amd.engine.trainer import GraphedTrainStep
from monoflex_amd.solver import build_optimizer
dtype, gen = sys.argv[1], sys.argv[2]
split = "split" in sys.argv
if "nccl" in sys.argv:
    import torch.distributed as dist
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29513", rank=0, world_size=1)
    if "coll" in sys.argv:
        t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize(); print("collective done", flush=True)
AG._DCN_BWD_V1[0] = gen == "v1"
if "tr0" in sys.argv:
    L.check(L.load().mfx_set_option(b"wgrad_tr", 0), "opt")
m = T._model(dtype)
imgs, tg = T._batch(m)
opt = build_optimizer(m, T._cfg(dtype), capturable=True)
step = GraphedTrainStep(m, opt, imgs, tg, warmup=2, split=split)
print("captured", flush=True)
for i in range(3):
    l = step(); torch.cuda.synchronize(); print("replay", i, float(l), flush=True)
if "interleave" in sys.argv:
    from monoflex_amd.engine.trainer import train_step
    a = T._model(dtype, seed=5)
    opt_a = build_optimizer(a, T._cfg(dtype), capturable=True)
    print("eager twin step", float(train_step(a, opt_a, imgs, tg)[0]), flush=True)
    if "empty" in sys.argv:
        torch.cuda.empty_cache(); print("emptied cache", flush=True)
    for i in range(2):
        l = step(); torch.cuda.synchronize(); print("replay after eager work", i, float(l), flush=True)

def _replays(tag):
    for i in range(2):
        l = step(); torch.cuda.synchronize(); print("replay after", tag, i, float(l), flush=True)
if "alloc" in sys.argv:
    t = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); t.zero_(); del t
    ts = [torch.randn(1 << 20, device="cuda") for _ in range(50)]; del ts
    torch.cuda.synchronize(); _replays("alloc/free")
if "fwdonly" in sys.argv:
    a = T._model(dtype, seed=5)
    ld, _ = a(imgs, tg); print("twin forward", float(sum(ld.values())), flush=True); del ld
    _replays("twin forward")
if "evalfwd" in sys.argv:
    a = T._model(dtype, seed=5).eval()
    with torch.no_grad():
        f = a.backbone.forward_nhwc(imgs)
    torch.cuda.synchronize(); print("twin eval backbone", float(f.float().abs().mean()), flush=True)
    _replays("twin eval backbone")
