#!/usr/bin/env python
"""Run-to-run repeatability of the tile-owned DCN backward (mfx_dcn_backward_v2_rt) on fixed inputs, per form (options dcn_bt_fly / dcn_bt_fuse_wgrad):
which outputs differ between launches, by how much and where."""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from monoflex_amd import lib as L

lib_ = L.load()
DEV = "cuda"
dt, code = torch.bfloat16, L.MFX_BF16
g = torch.Generator().manual_seed(5)
B, C, Cout, H, W = 2, 64, 64, 96, 320
x = torch.randn(B, H, W, C, generator=g).to(DEV).to(dt)
om = torch.zeros(B, H, W, 32)
om[..., :18] = torch.randn(B, H, W, 18, generator=g) * float(os.environ.get("STD", "1.5"))
om[..., 18:27] = torch.sigmoid(torch.randn(B, H, W, 9, generator=g))
om = om.to(DEV)
w = (torch.randn(Cout, C, 3, 3, generator=g) * 0.05).to(DEV)
dy = torch.randn(B, H, W, Cout, generator=g).to(DEV).to(dt)
nws = lib_.mfx_dcn_backward_v2_workspace_bytes(B, C, H, W, Cout, code)
ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
ptr = lambda t: ctypes.c_void_p(t.data_ptr())


def run(raw16):
    dx = torch.empty_like(x)
    draw = torch.zeros((B, H, W, 32), dtype=dt if raw16 else torch.float32, device=DEV)
    dw, db = torch.empty(Cout, C, 3, 3, device=DEV), torch.empty(Cout, device=DEV)
    L.check(lib_.mfx_dcn_backward_v2_rt(ptr(x), ptr(om), ptr(w), ptr(dy), ptr(dx), ptr(draw), raw16, ptr(dw), ptr(db), B, C, H, W, Cout, code,
                                        ptr(ws), nws, None), "v2")
    torch.cuda.synchronize()
    return {"dx": dx.float(), "draw": draw.float(), "dw": dw, "db": db}


for form in ("fly", 1, 0):
    L.check(lib_.mfx_set_option(b"dcn_bt_fuse_wgrad", 0 if form == 0 else 1), "opt")
    L.check(lib_.mfx_set_option(b"dcn_bt_fly", 1 if form == "fly" else 0), "opt")
    for raw16 in ((0, 1) if not os.environ.get("RAW0") else (0,)):
        base = run(raw16)
        for rep in range(4):
            o = run(raw16)
            msg = []
            for k in base:
                d = (o[k] - base[k]).abs()
                n = int((d > 0).sum())
                if n:
                    rel = float(d.max() / base[k].abs().max())
                    where = ""
                    if k in ("dx", "draw"):
                        nz = (d > 0).any(dim=-1).nonzero()
                        where = " b %d..%d y %d..%d x %d..%d" % (nz[:, 0].min(), nz[:, 0].max(), nz[:, 1].min(), nz[:, 1].max(), nz[:, 2].min(), nz[:, 2].max())
                    msg.append("%s: %d differ, rel %.2e%s" % (k, n, rel, where))
                    if k == "draw" and rep == 0:
                        nz = (d > 0).nonzero()[:24]
                        for b_, y_, x_, c_ in nz.tolist():
                            print("    draw[b %d y %3d x %3d (x%%32 = %2d) ch %2d] %.5f -> %.5f" % (b_, y_, x_, x_ % 32, c_, float(base[k][b_, y_, x_, c_]), float(o[k][b_, y_, x_, c_])))
            print("form", form, "raw16", raw16, "rep", rep, "|", "; ".join(msg) or "identical", flush=True)
