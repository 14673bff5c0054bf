import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from monoflex_amd import synthetic as S
from monoflex_amd.structures.params_3d import make_test_target
from torch.profiler import profile, ProfilerActivity
model, _ = bench.build_model("bf16", "cuda")
B = 8
images = S.synthetic_images(B, 384, 1280, seed=1000).cuda()
targets = [make_test_target(S.synthetic_target(320, 96)) for _ in range(B)]
tg = model.device_targets(targets, "cuda")
with torch.no_grad():
    for _ in range(2):
        model.detect_device(images, *tg)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model.detect_device(images, *tg)
        torch.cuda.synchronize()
seen = {}
for e in prof.events():
    n = e.name
    if n.startswith("aten::copy_") or n.startswith("aten::clone") or n.startswith("aten::contiguous") or n.startswith("aten::to") or "emcpy" in n:
        st = [s for s in (e.stack or []) if "monoflex_amd" in s or "bench" in s]
        key = (n, tuple(st[:3]))
        seen[key] = seen.get(key, 0) + 1
for (n, st), c in sorted(seen.items(), key=lambda kv: -kv[1])[:25]:
    print(c, n, " <- ".join(s.split("/")[-1] for s in st))
