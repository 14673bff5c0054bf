import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from monoflex_amd import lib, ops
L = lib.load()
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (B,H,W,Ci) in [(8,96,320,64),(8,48,160,128),(8,24,80,256),(8,12,40,512)]:
    x = torch.randn(B,H,W,Ci, device="cuda").bfloat16()
    w = torch.randn(27, Ci, 3, 3, device="cuda")*0.05
    for act in (0, 3):
        p = ops.pack_conv(w, torch.bfloat16, None, torch.zeros(27, device="cuda"), stride=1, pad=1, act=act, cout=32)
        for od in (torch.bfloat16, torch.float32):
            r = []
            for hv in (1, 2, 3, 0):
                lib.check(L.mfx_set_option(b"halo", hv), "o")
                r.append("halo=%d:%.1f" % (hv, timeit(lambda: ops.conv2d(x, p, out_dtype=od))))
            print(H, W, Ci, "act", act, od, "  ".join(r))
