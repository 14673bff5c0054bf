import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from monoflex_amd import ops, synthetic as S
from monoflex_amd.structures.params_3d import make_test_target
model, _ = bench.build_model("bf16", "cuda")
B = 8
images = S.synthetic_images(B, 384, 1280, seed=1000).cuda()
targets = [make_test_target(S.synthetic_target(320, 96)) for _ in range(B)]
tg = model.device_targets(targets, "cuda")
orig = ops.dcn
def rec(x, om, p):
    o = om[..., :18].float()
    print("dcn %4d->%4d %3dx%3d  offset std %.2f  P(|d|>=3) %.3f  P(|d|>=5) %.3f  max %.1f" % (x.shape[3], p.Cout, x.shape[1], x.shape[2], float(o.std()), float((o.abs() >= 3).float().mean()), float((o.abs() >= 5).float().mean()), float(o.abs().max())))
    return orig(x, om, p)
ops.dcn = rec
with torch.no_grad():
    model.detect_device(images, *tg)
torch.cuda.synchronize()
