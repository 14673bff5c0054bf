#!/usr/bin/env python
"""Phase time stamps of dcn_lds_kernel (probe build: tools/build_variant.sh probe dcn_lds.hip -DMFX_PROBES; MFX_LIB_PATH=build_variants/lib_probe.so).
usage: MFX_LIB_PATH=... python tools/probes/dcn_lds_probe.py [module|kernel] [B H W C]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from monoflex_amd import lib, ops

L = lib.load()
lib.check(L.mfx_set_option(b"dcn_lds_rows", int(os.environ.get("DCN_LDS_ROWS", "8"))), "opt")
mode = sys.argv[1] if len(sys.argv) > 1 else "module"
B, H, W, C = [int(v) for v in sys.argv[2:6]] if len(sys.argv) > 5 else (8, 96, 320, 64)
dt = torch.bfloat16
torch.manual_seed(0)
x = torch.randn(B, H, W, C, device="cuda").relu().to(dt)
if mode == "module":
    from monoflex_amd.model.backbone.dla_dcn import DeformConv
    m = DeformConv(C, 64).eval().cuda()
    torch.nn.init.normal_(m.conv.conv_offset_mask.weight, std=2.5 / (0.7 * (9 * C) ** 0.5))
    fn = lambda: m(x)
else:
    w = torch.randn(64, C, 3, 3, device="cuda") * (1.0 / (3 * C ** 0.5))
    om = torch.zeros(B, H, W, 32, device="cuda")
    om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * 2.5
    om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
    p = ops.pack_conv(w, dt, torch.ones(64, device="cuda"), torch.zeros(64, device="cuda"), stride=1, pad=1, act=1)
    ops.add_f16_fragments(p, w)
    lib.check(L.mfx_set_option(b"dcn_lds", 2), "opt")
    fn = lambda: ops.dcn(x, om, p)
with torch.no_grad():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
ROWS = int(os.environ.get("DCN_LDS_ROWS", "8"))
nwg = min(1024, B * ((H + ROWS - 1) // ROWS) * ((W + 15) // 16))
n = 1024 * 4 * 16
buf = (ctypes.c_ulonglong * n)()
rc = L.mfx_dcn_lds_probe_read(buf, n)
assert rc == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4, 16)[:nwg].astype(np.float64)
span = t[:, :, 15].max() - t[:, :, 0].min()
print("launch %.1f us (events, incl. the module's other launches); kernel span %.0f ticks; %d workgroups" % (us, span, nwg))
names = {0: "start", 1: "0a nbhd loaded", 2: "0b conv done", 3: "0c transposed", 4: "G table (GW)", 5: "barrier A", 6: "GB+patch0+barrier B",
         7: "slice0", 8: "patch1+barriers", 9: "slice1", 10: "patch2+barriers", 11: "slice2", 12: "patch3+barriers", 13: "slice3", 14: "far pass", 15: "epilogue"}
prev = t[:, :, 0]
life = (t[:, :, 15] - t[:, :, 0]).mean()
print("mean wave lifetime %.0f ticks (= %.1f us if the span is the kernel)" % (life, life / span * us))
for k in range(1, 16):
    if not (t[:, :, k] > 0).all():
        continue
    d = t[:, :, k] - prev
    print("%-22s mean %8.0f  p10 %8.0f  p90 %8.0f  (%.1f %% of lifetime)" % (names[k], d.mean(), np.percentile(d, 10), np.percentile(d, 90), 100 * d.mean() / life))
    prev = t[:, :, k]
# first-round vs second-round workgroups
starts = t[:, 0, 0] - t[:, :, 0].min()
print("workgroup start offsets (ticks): p50 %.0f p90 %.0f max %.0f" % (np.percentile(starts, 50), np.percentile(starts, 90), starts.max()))
