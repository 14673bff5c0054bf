import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
import test_gpu_train_step as T
from monoflex_amd.engine.trainer import train_step
from monoflex_amd.solver import build_optimizer
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
for dtype in ("fp32", "bf16"):
    cfg = T._cfg(dtype); m = T._model(dtype); imgs, tg = T._batch(m)
    opt = build_optimizer(m, cfg, capturable=True)
    out = train_step(m, opt, imgs, tg)
    torch.cuda.synchronize()
    bad = [n for n, p in m.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    print(dtype, "loss", float(out[0]), "params with non-finite grads:", len(bad), bad[:8])
