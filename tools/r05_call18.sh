#!/bin/bash
# two-ahead gather of the fused DCN kernel: bit-identity with the one-ahead form and per-layer time, KC = 8 and 4
cd /root/repo
mkdir -p gpurun_out
( python - <<'PY'
import torch, sys
sys.path.insert(0, "/root/repo")
from monoflex_amd import lib, ops
L = lib.load()
def opt(**kw):
    lib.check(L.mfx_reset_options(), "r")
    for k, v in kw.items(): lib.check(L.mfx_set_option(k.encode(), int(v)), k)
for (B, H, W, Ci, Co) in [(2, 12, 40, 512, 256), (2, 24, 80, 256, 128), (2, 47, 33, 128, 64), (1, 24, 80, 256, 64)]:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Ci, device="cuda").relu().bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda") / (3 * Ci ** 0.5)
    om = torch.zeros(B, H, W, 32, device="cuda"); om[..., :18] = torch.randn(B, H, W, 18, device="cuda") * 3; om[..., 18:27] = torch.rand(B, H, W, 9, device="cuda")
    p = ops.pack_conv(w, torch.bfloat16, torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda"), stride=1, pad=1, act=1)
    outs = {}
    for name, kw in [("pf0", dict(dcn_wave=0, dcn_patch=0)), ("pf2", dict(dcn_wave=0, dcn_patch=0, dcn_pf=2)), ("pf2kc4", dict(dcn_wave=0, dcn_patch=0, dcn_pf=2, kc=4)),
                     ("pf2ks", dict(dcn_wave=0, dcn_patch=0, dcn_pf=2, dcn_ksplit=5)), ("pf0ks", dict(dcn_wave=0, dcn_patch=0, dcn_ksplit=5))]:
        opt(**kw); outs[name] = ops.dcn(x, om, p).float()
    print((H, W, Ci, Co), "pf2==pf0", torch.equal(outs["pf0"], outs["pf2"]), "kc4 maxdiff", (outs["pf2kc4"] - outs["pf0"]).abs().max().item(),
          "ksplit pf2==pf0", torch.equal(outs["pf0ks"], outs["pf2ks"]))
opt()
PY
  for o in "dcn_wave=0" "dcn_wave=0,dcn_pf=2" "dcn_wave=0,dcn_pf=2,kc=4" "dcn_wave=0,kc=4"; do echo "== $o"; timeout 300 python tools/dcn_layers_bench.py 8 3.0 $o 2>&1 | grep -v amdgpu.ids | grep -v "^|---" | awk -F'|' '{print $2, $4}' | tr '\n' ';'; echo; done ) > gpurun_out/dcn_pf2.md 2>&1
cat gpurun_out/dcn_pf2.md
