"""GPU parity tests of the split-precision mode (compute tag ops.F16X2 / MFX_F16X2, csrc/common.h f32s_t): fp32 activations, every MFMA
operand an fp16 (hi, lo) pair.  The mode claims fp32-grade results -- the reference computes in fp32 (src/cuda/dcn_v2_cuda.cu:58) -- so every
operator is held to the SAME tolerance as the fp32 kernels against the same torch fp32 / C-oracle references (tests/test_gpu_ops.py), and,
operator by operator, it must sit as close to an fp64 reference as the f32-MFMA kernels do (within 4x)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import CONV_CASES, DEV, _close, _dcn_case, _from_nhwc, _g, _ops, _to_nhwc

pytestmark = pytest.mark.gpu
F32 = torch.float32


def test_split_chunks_layout_and_value():
    """Host packing: each 16-byte chunk = [4 hi halves | 4 lo halves] of its 4 values; hi + lo reproduces the value to ~2^-22."""
    ops, L = _ops()
    w = torch.randn(8, 64, generator=_g(1)) * torch.logspace(-6, 2, 64)
    s = ops.split_chunks(w)
    assert s.dtype == torch.float32 and s.shape == w.shape
    h = s.view(torch.float16).view(8, 16, 8)                       # [row][chunk][8 halves]
    hi, lo = h[..., :4].float().reshape(8, 64), h[..., 4:].float().reshape(8, 64)
    assert torch.equal(hi, w.half().float())
    err = (hi.double() + lo.double() - w.double()).abs()
    # |x| >= 0.25: the lo half is a normal fp16 number (22 significant bits together); below it is an fp16 subnormal: absolute 2^-25
    assert bool((err <= torch.maximum(w.double().abs() * 2.0 ** -22, torch.tensor(2.0 ** -25, dtype=torch.float64)) * 1.0001).all())
    s_ = ops.split_weight_scale(w)
    assert 2 ** 11 <= float(w.abs().max()) * s_ < 2 ** 12 and math.log2(s_) == int(math.log2(s_))


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_split(case):
    ops, L = _ops()
    B, Cin, Cout, H, W, k, s, use_res, act = case
    g = _g(hash(case) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), None, s, k // 2) * scale.view(1, -1, 1, 1).double() + shift.view(1, -1, 1, 1).double()
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res.double()
    ref = F.relu(ref) if act == 1 else F.leaky_relu(ref, 0.01) if act == 2 else ref
    outs = {}
    for tag in (F32, ops.F16X2):
        p = ops.pack_conv(w.to(DEV), tag, scale.to(DEV), shift.to(DEV), stride=s, pad=k // 2, act=act)
        assert p.split == (tag == ops.F16X2)
        y = ops.conv2d(_to_nhwc(x, F32), p, res=_to_nhwc(res, F32) if use_res else None)
        torch.cuda.synchronize()
        assert y.dtype == torch.float32
        outs[tag] = _from_nhwc(y)
    _close(outs[ops.F16X2], ref.float(), F32, Cin * k * k, "split conv2d %s" % (case,))
    e32, e16 = float((outs[F32].double() - ref).abs().max()), float((outs[ops.F16X2].double() - ref).abs().max())
    assert e16 <= 4 * e32 + 1e-6, "split-precision error %.3e vs f32-MFMA error %.3e" % (e16, e32)


@pytest.mark.parametrize("variant", [0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15])
def test_conv3x3_halo_variants_split(variant):
    """Every LDS-halo kernel variant (option "halo" = variant + 1; 0 = the generic implicit-GEMM kernel) in split precision."""
    ops, L = _ops()
    lib_ = L.load()
    g = _g(40 + variant)
    bn = {0: 64, 2: 16, 3: 32, 4: 64, 5: 128, 6: 256, 7: 64, 8: 128, 9: 32, 10: 32, 11: 16, 12: 128, 13: 64, 14: 64, 15: 128}[variant]
    Cin, Cout, H, W = 64, bn, 19, 37
    x = torch.randn(2, Cin, H, W, generator=g).relu()
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 24.0
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(2, Cout, H, W, generator=g) if Cout >= 64 else None
    ref = F.conv2d(x, w, None, 1, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref = F.relu(ref + res if res is not None else ref)
    p = ops.pack_conv(w.to(DEV), ops.F16X2, scale.to(DEV), shift.to(DEV), stride=1, pad=1, act=1)
    L.check(lib_.mfx_set_option(b"halo", variant), "opt")
    try:
        y = ops.conv2d(_to_nhwc(x, F32), p, res=_to_nhwc(res, F32) if res is not None else None)
        torch.cuda.synchronize()
    finally:
        lib_.mfx_set_option(b"halo", 1)
    _close(_from_nhwc(y), ref, F32, Cin * 9, "split halo variant %d" % variant)


@pytest.mark.parametrize("shape", [(32, 32, 3), (32, 64, 7), (128, 128, 8), (128, 64, 13), (256, 256, 12), (256, 32, 11), (16, 16, 2)])
def test_conv3x3_halo_pair_walk_vs_step_walk(shape):
    """The LDS-halo kernel's pair-walking K loop (three products per step pair, mfx_conv_desc.w_frag_pair; Cin >= 32) against the same
    kernel walking single steps (option halo_pair = 0: four products per pair) and against fp64 torch -- one and several channel groups,
    K-split variants.  Cin = 16 has no pairs (a pair would straddle taps) and must keep the step walk."""
    ops, L = _ops()
    lib_ = L.load()
    Cin, Cout, variant = shape
    g = _g(900 + Cin + Cout)
    H, W = 21, 35
    x = torch.randn(2, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), None, 1, 1) * scale.view(1, -1, 1, 1).double() + shift.view(1, -1, 1, 1).double()
    p = ops.pack_conv(w.to(DEV), ops.F16X2, scale.to(DEV), shift.to(DEV), stride=1, pad=1, act=0)
    assert (p.w_frag_pair is not None) == (Cin >= 32)
    outs = []
    L.check(lib_.mfx_set_option(b"halo", variant), "opt")
    try:
        for pair in (1, 0):
            L.check(lib_.mfx_set_option(b"halo_pair", pair), "opt")
            outs.append(_from_nhwc(ops.conv2d(_to_nhwc(x, F32), p)))
            torch.cuda.synchronize()
    finally:
        lib_.mfx_set_option(b"halo", 1)
        lib_.mfx_set_option(b"halo_pair", 1)
    for y in outs:
        _close(y, ref.float(), F32, Cin * 9, "split halo pair walk %s" % (shape,))
    e_pair, e_step = (float((y.double().cpu() - ref).abs().max()) for y in outs)
    assert e_pair <= 2 * e_step + 1e-6, "pair walk error %.3e vs step walk %.3e" % (e_pair, e_step)
    if Cin < 32:
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("B,H,W", [(2, 38, 74), (1, 384, 1280)])
def test_f1_fused_split_vs_three_launches_and_torch(B, H, W):
    """csrc/f1_fused.hip in split precision: stem 7x7 -> level0 3x3 -> level1 3x3 / s2 in one kernel (fp32 image in, fp32 level1 map out, both
    full-resolution maps as (hi, lo) fp16 pairs in LDS) against the three separate split-precision launches on the same packs and against
    fp64 torch: the fp32-grade bound of the mode, ragged tiles and image borders included."""
    ops, L = _ops()
    g = _g(62)
    img = torch.randn(B, 3, H, W, generator=g)
    w7 = torch.randn(16, 3, 7, 7, generator=g) / 147 ** 0.5
    w0 = torch.randn(16, 16, 3, 3, generator=g) / 12.0
    w1 = torch.randn(32, 16, 3, 3, generator=g) / 12.0
    bn = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1) for c in (16, 16, 32)]
    ps = ops.pack_stem(w7.to(DEV), ops.F16X2, bn[0][0].to(DEV), bn[0][1].to(DEV))
    p0 = ops.pack_conv(w0.to(DEV), ops.F16X2, bn[1][0].to(DEV), bn[1][1].to(DEV), stride=1, pad=1, act=L.ACT_RELU)
    p1 = ops.pack_conv(w1.to(DEV), ops.F16X2, bn[2][0].to(DEV), bn[2][1].to(DEV), stride=2, pad=1, act=L.ACT_RELU)
    x = img.to(DEV)
    got = ops.f1_fused(x, ps, p0, p1)
    want_hip = ops.conv2d(ops.conv2d(ops.stem_conv(x, ps), p0), p1)
    torch.cuda.synchronize()
    assert got.shape == want_hip.shape == (B, H // 2, W // 2, 32) and got.dtype == torch.float32
    aff = lambda t, i: F.relu(t * bn[i][0].view(1, -1, 1, 1).double() + bn[i][1].view(1, -1, 1, 1).double())      # noqa: E731
    r = aff(F.conv2d(img.double(), w7.double(), None, 1, 3), 0)
    r = aff(F.conv2d(r, w0.double(), None, 1, 1), 1)
    ref = aff(F.conv2d(r, w1.double(), None, 2, 1), 2).permute(0, 2, 3, 1)
    a, b_ = got.double().cpu(), want_hip.double().cpu()
    scale = float(ref.abs().max())
    e_f, e_3 = float((a - ref).abs().max()) / scale, float((b_ - ref).abs().max()) / scale
    assert e_f < 2e-6, "fused split F1: max err / max |ref| = %.3e (three launches: %.3e)" % (e_f, e_3)
    assert float((a - b_).abs().max()) / scale < 2e-6


def test_stem_and_cat_split():
    ops, L = _ops()
    g = _g(11)
    x = torch.randn(2, 3, 20, 36, generator=g)
    w = torch.randn(16, 3, 7, 7, generator=g) / 147 ** 0.5
    scale, shift = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, w, None, 1, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    p = ops.pack_stem(w.to(DEV), ops.F16X2, scale.to(DEV), shift.to(DEV))
    y = ops.stem_conv(x.to(DEV), p)                                # the dedicated split-precision stem kernel (csrc/stem.hip)
    torch.cuda.synchronize()
    assert y.dtype == torch.float32
    _close(_from_nhwc(y), ref, F32, 147, "split stem")
    xb = torch.randn(1, 3, 37, 150, generator=g)                   # ragged tiles (8 x 64 blocks), three column tiles
    refb = F.relu(F.conv2d(xb, w, None, 1, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    _close(_from_nhwc(ops.stem_conv(xb.to(DEV), p)), refb, F32, 147, "split stem, ragged")
    chans = [128, 128, 64, 128]
    xs = [torch.randn(2, c, 6, 10, generator=g) for c in chans]
    w = torch.randn(128, sum(chans), 1, 1, generator=g) / sum(chans) ** 0.5
    scale, shift = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    ref = F.relu(F.conv2d(torch.cat(xs, 1), w) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    y = ops.cat_conv1x1([_to_nhwc(t, F32) for t in xs], ops.pack_cat(w.to(DEV), ops.F16X2, scale.to(DEV), shift.to(DEV), chans))
    torch.cuda.synchronize()
    _close(_from_nhwc(y), ref, F32, sum(chans), "split root cat conv")


@pytest.mark.parametrize("shape", [(1, 64, 64, 24, 40), (2, 128, 64, 12, 20), (1, 256, 128, 6, 10), (1, 64, 128, 40, 56)])
@pytest.mark.parametrize("wave", [0, 1])
def test_dcn_split_vs_oracle(shape, wave):
    """Fused DCNv2 (first-generation tile kernel and the wave kernel) in split precision against oracle/dcn_v2_ref.c, incl. the -1 boundary
    and +-30 px samples of `_dcn_case`."""
    from oracle import dcn_ref
    ops, L = _ops()
    lib_ = L.load()
    x, off, msk, w, b = _dcn_case(21, *shape)
    want = dcn_ref.dcn_v2_forward(x, w, b, off, msk, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    B, C, Co, H, W = shape
    om = torch.zeros(B, H, W, 32)
    om[..., :18] = off.permute(0, 2, 3, 1)
    om[..., 18:27] = msk.permute(0, 2, 3, 1)
    p = ops.pack_conv(w.to(DEV), ops.F16X2, None, b.to(DEV), stride=1, pad=1, act=0)
    L.check(lib_.mfx_set_option(b"dcn_wave", 2 + 5 * wave if wave else 0), "opt")     # 0: tile kernel; 7 -> variant 6 (2 waves x FN 2, BN 64)
    try:
        y = ops.dcn(_to_nhwc(x, F32), om.to(DEV), p)
        torch.cuda.synchronize()
    finally:
        lib_.mfx_set_option(b"dcn_wave", 1)
    got = _from_nhwc(y)
    assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


def test_heads_fused_split_vs_torch():
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.head.detector_predictor import _predictor, REG_OFF
    from monoflex_amd import synthetic as S
    from oracle import monoflex_ref as R
    ops, L = _ops()
    cfg = get_cfg(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "runs", "monoflex.yaml"))
    ref = R.Predictor().eval()
    sd = S.synthetic_state_dict({"heads.predictor." + k: v for k, v in ref.state_dict().items()}, seed=3)
    ref.load_state_dict({k[len("heads.predictor."):]: v for k, v in sd.items()})
    m = _predictor(cfg, 64).eval()
    m.load_state_dict(ref.state_dict())
    m.to(DEV)
    tgt = S.synthetic_target(40, 24)
    x = torch.randn(2, 64, 24, 40, generator=_g(9)).relu()
    taps = {}
    ei = torch.stack([tgt["edge_indices"]] * 2)
    el = torch.tensor([tgt["edge_len"]] * 2)
    with torch.no_grad():
        maps = ref(x, ei, el, taps)
    want32 = m.forward_nhwc(_to_nhwc(x, F32), ei.to(DEV, torch.int32), el.to(DEV, torch.int32)).cpu()
    for mod in m.modules():
        mod.__dict__["_mfx_split"] = True
    hm = m.forward_nhwc(_to_nhwc(x, F32), ei.to(DEV, torch.int32), el.to(DEV, torch.int32)).cpu()
    assert m._pack(ops.F16X2).split and ("heads", ops.F16X2) in m._packs
    got_cls = hm[..., :3].permute(0, 3, 1, 2)
    got_reg = hm[..., REG_OFF:REG_OFF + 50].permute(0, 3, 1, 2)
    assert float((got_cls - taps["cls_logits"]).abs().max()) < 1e-4
    assert float((got_reg - maps["reg"]).abs().max()) < 1e-4 * max(1.0, float(maps["reg"].abs().max()))
    assert float((hm[..., :3] - want32[..., :3]).abs().max()) < 1e-4


def test_deformconv_module_split():
    """DeformConv = offset/mask conv + DCN + BN + ReLU in split precision against the oracle module (fp32 tolerance)."""
    from monoflex_amd.model.backbone.dla_dcn import DeformConv
    from oracle import monoflex_ref as R
    torch.manual_seed(4)
    ref = R.DeformConv(64, 64).eval()
    torch.nn.init.normal_(ref.conv.conv_offset_mask.weight, std=1.5 / 24)
    torch.nn.init.normal_(ref.conv.conv_offset_mask.bias, std=0.2)
    ref.actf[0].running_mean.normal_(0, 0.1); ref.actf[0].running_var.uniform_(0.8, 1.2)
    ref.actf[0].weight.data.uniform_(0.8, 1.2); ref.actf[0].bias.data.normal_(0, 0.1)
    x = torch.randn(2, 64, 24, 40).relu()
    with torch.no_grad():
        want = ref(x)
    m = DeformConv(64, 64).eval()
    m.load_state_dict(ref.state_dict())
    m.to(DEV)
    for mod in m.modules():
        mod.__dict__["_mfx_split"] = True
    got = _from_nhwc(m(_to_nhwc(x, F32)))
    assert m.conv.packed_main(__import__("monoflex_amd").ops.F16X2, m.actf[0], 1).split
    assert float((got - want).abs().max()) < 5e-5 * max(1.0, float(want.abs().max()))
