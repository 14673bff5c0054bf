import os, sys, ctypes
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from test_gpu_train_step import _batch, _cfg, _model
from monoflex_amd import lib as L
from monoflex_amd.engine.trainer import GraphedTrainStep
from monoflex_amd.solver import build_optimizer
cfg = _cfg("bf16")
b = _model("bf16")
imgs, tg = _batch(b)
opt = build_optimizer(b, cfg, capturable=True)
step = GraphedTrainStep(b, opt, imgs, tg, warmup=2)
torch.cuda.synchronize()
mt = opt._mfx_multi
print("captured table sets:", len(mt.captured), "pending:", len(mt.pending), flush=True)
dt, pt, gt = mt.captured[0]
n = dt.numel() // ctypes.sizeof(L.AdamWDesc)
raw = dt.cpu().numpy().tobytes()
descs = (L.AdamWDesc * n).from_buffer_copy(raw)
print("n", n, "first desc p %x g %x m %x v %x step %x numel %d group %d" % (descs[0].p or 0, descs[0].g or 0, descs[0].m or 0, descs[0].v or 0, descs[0].step or 0, descs[0].numel, descs[0].group), flush=True)
zeros = sum(1 for d in descs if not d.step)
print("descs with null step:", zeros, "prefix head", pt[:4].tolist(), "prefix tail", pt[-2:].tolist(), flush=True)
p0 = [p for g_ in opt.param_groups for p in g_["params"] if p.grad is not None or True][0]
print("param0 ptr %x state step ptr %x" % (p0.data_ptr(), opt.state[p0]["step"].data_ptr()), flush=True)
loss = step(); torch.cuda.synchronize(); print("replay ok", float(loss), flush=True)
