import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
import test_gpu_train_step as T
from monoflex_amd.engine.trainer import GraphedTrainStep
from monoflex_amd.solver import build_optimizer
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import socket
s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
dtype = "fp32"
cfg = T._cfg(dtype); m = T._model(dtype); imgs, tg = T._batch(m)
opt = build_optimizer(m, cfg, capturable=True)
sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
step = GraphedTrainStep(m, opt, imgs, tg, warmup=3, split=False)
torch.cuda.synchronize()
assert all(torch.equal(sd0[k], v) for k, v in m.state_dict().items())
assert len(opt.state) > 250 and all(float(st["step"]) == 0 and not bool(st["exp_avg"].any()) and not bool(st["exp_avg_sq"].any()) for st in opt.state.values())
step(); torch.cuda.synchronize()
from monoflex_amd import gram_heads as GH
print("DEBUG", {k: v.tolist() for k, v in (GH._DEBUG or {}).items()})
sdx = m.state_dict()
print("running stats nan:", [k for k, v in sdx.items() if "running" in k and not bool(torch.isfinite(v).all())][:10])
print("grad finite:", [n for n, p in m.named_parameters() if p.grad is not None and bool(torch.isfinite(p.grad).all())][:10])
print("grad nan:", [n for n, p in m.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())][:6])
sd1 = {k: v.detach().clone() for k, v in m.state_dict().items()}
print("nan keys after step:", [k for k, v in sd1.items() if v.is_floating_point() and not bool(torch.isfinite(v).all())][:8])
GraphedTrainStep(m, opt, imgs, tg, warmup=2, split=False)
torch.cuda.synchronize()
bad = [k for k, v in m.state_dict().items() if not torch.equal(sd1[k], v)]
print(len(bad), bad[:12])
for k in bad[:4]:
    print(k, sd1[k].flatten()[:4], m.state_dict()[k].flatten()[:4])
