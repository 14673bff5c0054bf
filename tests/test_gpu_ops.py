"""GPU parity tests, operator level: every HIP operator (through the C ABI) against a plain torch fp32
CPU reference of the same op, or the C DCN oracle.  fp32 mode: tight tolerances (f32 MFMA is an exact
fmaf chain); bf16 mode: inputs/weights rounded to bf16 first, tolerance ~ bf16 epsilon x sqrt(K)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from monoflex_amd import lib as L, ops
    L.load()                                   # fail loudly if the HIP library is missing
    return ops, L


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _to_nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def _from_nhwc(y):
    return y.float().cpu().permute(0, 3, 1, 2)


def _tol(dtype, K):
    return (2e-5 * max(1.0, K ** 0.5 / 8), 1e-5) if dtype == torch.float32 else (2e-2, 2e-2)


def _close(a, b, dtype, K, what):
    rtol, atol = _tol(dtype, K)
    scale = float(b.abs().max().clamp(min=1.0))
    err = float((a - b).abs().max())
    assert err <= atol * scale + rtol * scale, "%s: max abs err %.3e (scale %.3e)" % (what, err, scale)


CONV_CASES = [
    # B, Cin, Cout, H, W, k, stride, residual, act
    (2, 16, 16, 24, 40, 3, 1, False, 1),
    (1, 16, 32, 24, 40, 3, 2, False, 1),
    (2, 32, 64, 16, 24, 3, 2, False, 1),
    (2, 64, 64, 12, 20, 3, 1, True, 1),
    (1, 64, 128, 12, 20, 3, 2, False, 0),
    (1, 128, 128, 64, 72, 3, 1, True, 1),       # > 512 tiles of 128x128 -> big-tile path
    (1, 32, 64, 6, 10, 1, 1, False, 0),
    (1, 256, 512, 6, 10, 3, 1, False, 1),
    (3, 64, 64, 5, 7, 3, 1, False, 2),          # ragged M (105 rows), leaky
    # 64 bytes of K: the pointwise packing keeps K at 4 chunks instead of padding to 8 (ops.pack_conv, r05): Cin 32 in 16 bits, Cin 16 in fp32
    (2, 32, 64, 48, 80, 1, 1, True, 1),
    (2, 16, 64, 9, 13, 1, 1, False, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case, dtype):
    ops, L = _ops()
    B, Cin, Cout, H, W, k, s, use_res, act = case
    g = _g(hash(case) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x, w, None, s, k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        if dtype == torch.bfloat16:
            res = res.bfloat16().float()
        ref = ref + res
    ref = F.relu(ref) if act == 1 else F.leaky_relu(ref, 0.01) if act == 2 else ref
    p = ops.pack_conv(w.to(DEV), dtype, scale.to(DEV), shift.to(DEV), stride=s, pad=k // 2, act=act)
    if k == 1 and Cin * (4 if dtype == torch.float32 else 2) == 64:
        assert p.K_pad == Cin                              # the 64-byte K stays 64 bytes wide
    y = ops.conv2d(_to_nhwc(x, dtype), p, res=_to_nhwc(res, dtype) if use_res else None)
    torch.cuda.synchronize()
    _close(_from_nhwc(y), ref, dtype, Cin * k * k, "conv2d %s" % (case,))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv2d_a_identity_asymmetric(dtype):
    # transpose-detecting check: identity weights on an asymmetric input must return the input
    ops, L = _ops()
    x = torch.arange(2 * 64 * 5 * 9, dtype=torch.float32).reshape(2, 64, 5, 9) % 251 / 16.0
    w = torch.zeros(64, 64, 1, 1)
    w[torch.arange(64), torch.arange(64), 0, 0] = 1.0
    p = ops.pack_conv(w.to(DEV), dtype, None, None, stride=1, pad=0, act=0)
    y = ops.conv2d(_to_nhwc(x, dtype), p)
    assert torch.equal(_from_nhwc(y), x.to(dtype).float())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_conv7x7(dtype):
    ops, L = _ops()
    g = _g(11)
    x = torch.randn(2, 3, 20, 36, generator=g)
    w = torch.randn(16, 3, 7, 7, generator=g) / 147 ** 0.5
    scale, shift = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.relu(F.conv2d(x, w, None, 1, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    p = ops.pack_stem(w.to(DEV), dtype, scale.to(DEV), shift.to(DEV))
    y = ops.conv2d(ops.pack_image(x.to(DEV), dtype), p, out_hw=(20, 36))
    torch.cuda.synchronize()
    _close(_from_nhwc(y), ref, dtype, 147, "stem")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_root_cat_conv1x1(dtype):
    ops, L = _ops()
    g = _g(12)
    chans = [128, 128, 64, 128]                     # level3.tree2.root: 448 -> 128
    xs = [torch.randn(2, c, 6, 10, generator=g) for c in chans]
    w = torch.randn(128, sum(chans), 1, 1, generator=g) / sum(chans) ** 0.5
    scale, shift = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    if dtype == torch.bfloat16:
        xs, w = [t.bfloat16().float() for t in xs], w.bfloat16().float()
    ref = F.relu(F.conv2d(torch.cat(xs, 1), w) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    p = ops.pack_cat(w.to(DEV), dtype, scale.to(DEV), shift.to(DEV), chans)
    y = ops.cat_conv1x1([_to_nhwc(t, dtype) for t in xs], p)
    torch.cuda.synchronize()
    _close(_from_nhwc(y), ref, dtype, sum(chans), "root cat conv")


def test_split_precision_range_sentinel_trips_on_an_overflowing_activation():
    """mfx_f16x2_range_check: the split-precision mode turns fp32 activations into fp16 (hi, lo) operand pairs, hi = fp16(x) overflows above 65504.  A healthy
    input leaves the flag clear; one activation of 7e4 (or a NaN) anywhere in a conv's input sets it, for the LDS-halo kernel and for the generic one; reset clears."""
    ops, L = _ops()
    lib_ = L.load()
    g = _g(181)
    x = torch.randn(1, 12, 20, 64, generator=g).to(DEV)
    w3 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(DEV)
    w1 = (torch.randn(64, 64, 1, 1, generator=g) / 8).to(DEV)
    p3 = ops.pack_conv(w3, ops.F16X2, None, None, stride=1, pad=1, act=L.ACT_RELU)
    p1 = ops.pack_conv(w1, ops.F16X2, None, None, stride=1, pad=0, act=L.ACT_RELU)
    L.f16x2_range_ok()                                        # clear whatever earlier tests left
    for p in (p3, p1):
        ops.conv2d(x, p)
        assert L.f16x2_range_ok()
        for bad in (7.0e4, float("nan"), -float("inf")):
            xb = x.clone(); xb[0, 5, 7, 3] = bad
            ops.conv2d(xb, p)
            assert not L.f16x2_range_ok(reset=False)
            assert not L.f16x2_range_ok()                     # still set; this call resets
            assert L.f16x2_range_ok()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_and_upsample(dtype):
    ops, L = _ops()
    g = _g(13)
    x = torch.randn(2, 64, 8, 12, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    y = ops.maxpool2x2(_to_nhwc(x, dtype))
    assert torch.equal(_from_nhwc(y), F.max_pool2d(x, 2, 2))
    for f in (2, 4):
        w = torch.rand(64, 1, 2 * f, 2 * f, generator=g)
        skip = torch.randn(2, 64, 8 * f, 12 * f, generator=g)
        if dtype == torch.bfloat16:
            skip = skip.bfloat16().float()
        ref = F.conv_transpose2d(x, w, None, stride=f, padding=f // 2, groups=64) + skip
        y = ops.upsample_add(_to_nhwc(x, dtype), ops.pack_upsample(w.to(DEV)), f, skip=_to_nhwc(skip, dtype))
        torch.cuda.synchronize()
        _close(_from_nhwc(y), ref, dtype, 4, "upsample f=%d" % f)


def _dcn_case(seed, B, C, Co, H, W, off_scale=2.0):
    g = _g(seed)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, 18, H, W, generator=g) * off_scale
    msk = torch.sigmoid(torch.randn(B, 9, H, W, generator=g))
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    off[0, :, 0, 0] = 30.0                           # samples far outside the map
    off[-1, :, H // 2, W // 3] = -30.0
    off[0, 0::2, 1, 1] = -1.0                        # exactly on the -1 boundary (must be excluded: h > -1)
    return x, off, msk, w, b


@pytest.mark.parametrize("shape", [(2, 2, 2, 4, 4), (2, 8, 8, 12, 20), (1, 64, 64, 24, 40), (2, 128, 64, 12, 20), (1, 256, 128, 6, 10)])
def test_ext_dcn_v2_forward_vs_oracle(shape):
    """The `_ext.dcn_v2_forward` boundary (NCHW fp32) against oracle/dcn_v2_ref.c."""
    from monoflex_amd.model.backbone.DCNv2 import _ext
    from oracle import dcn_ref
    x, off, msk, w, b = _dcn_case(21, *shape)
    want = dcn_ref.dcn_v2_forward(x, w, b, off, msk, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    got = _ext.dcn_v2_forward(x.to(DEV), w.to(DEV), b.to(DEV), off.to(DEV), msk.to(DEV), 3, 3, 1, 1, 1, 1, 1, 1, 1).cpu()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.detach().abs().max()))


@pytest.mark.parametrize("shape", [(2, 8, 8, 12, 20), (1, 64, 64, 24, 40), (2, 128, 64, 12, 20)])
def test_ext_dcn_v2_vs_library_bilinear_sampler(shape):
    """The HIP kernels behind `_ext.dcn_v2_forward / _backward` against a formulation that shares nothing with the oracle: the modulated deformable
    convolution rebuilt around torch's own grid_sample (float64 on the host: the library's bilinear rule with zeros beyond the map, and the library's
    input / coordinate derivatives; tests/test_oracle_dcn.py pins the C oracle with the same form).  Forward and all five gradients."""
    from monoflex_amd.model.backbone.DCNv2 import _ext
    from oracle import dcn_ref
    x, off, msk, w, b = _dcn_case(31, *shape)
    off[0, 0::2, 1, 1] = -0.75                      # (exactly -1 is a kink of the bilinear rule: the two forms may take either one-sided derivative there)
    leaves = [t.double().clone().requires_grad_() for t in (x, off, msk, w, b)]
    want = dcn_ref.dcn_v2_grid_sample(*leaves)
    go = torch.randn(want.shape, generator=_g(32))
    want.backward(go.double())
    dev = [t.to(DEV) for t in (x, w, b, off, msk)]
    got = _ext.dcn_v2_forward(*dev, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    assert float((got.cpu().double() - want.detach()).abs().max()) < 3e-5 * max(1.0, float(want.detach().abs().max()))
    gi, goff, gm, gw, gb = _ext.dcn_v2_backward(*dev, go.to(DEV), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    for name, g_, leaf in (("input", gi, leaves[0]), ("offset", goff, leaves[1]), ("mask", gm, leaves[2]), ("weight", gw, leaves[3]), ("bias", gb, leaves[4])):
        scale = max(1.0, float(leaf.grad.abs().max()))
        assert float((g_.cpu().double() - leaf.grad).abs().max()) / scale < 1e-4, name


def test_ext_dcn_zero_offset_known_answer():
    # reference testcpu.py:32-67: zero offsets, mask 0.5, identity weight -> |input - 2*output| < 1e-10
    from monoflex_amd.model.backbone.DCNv2.dcn_v2 import DCNv2
    torch.manual_seed(0)
    x = torch.randn(2, 2, 4, 4)
    m = DCNv2(2, 2, (3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(DEV)
    m.weight.data.zero_(); m.bias.data.zero_()
    m.weight.data[0, 0, 1, 1] = 1.0; m.weight.data[1, 1, 1, 1] = 1.0
    out = m(x.to(DEV), torch.zeros(2, 18, 4, 4, device=DEV), torch.full((2, 9, 4, 4), 0.5, device=DEV)).cpu()
    assert float((x - 2 * out).abs().max()) < 1e-10


@pytest.mark.parametrize("shape", [(1, 64, 64, 20, 36), (2, 128, 64, 9, 17), (1, 64, 128, 16, 16)])
def test_ext_dcn_v2_backward_fast_route_equals_the_scatter_route(shape):
    """`_ext.dcn_v2_backward` behind the C boundary takes the tile-owned second-generation kernels where the geometry is the model's own (3x3 / stride 1 /
    pad 1, power-of-two channel counts >= 64; option ext_bwd_fast) -- with d_raw's mask channels as the gradient of the MASK (the boundary takes the mask as
    an input: no sigmoid derivative, src/dcn_v2.h:48-59) -- and the first-generation scatter kernels elsewhere: the same five gradients on both routes,
    also for a mask that is no sigmoid output (values outside (0, 1), zeros) and offsets that leave the map."""
    from monoflex_amd import lib as L
    from monoflex_amd.model.backbone.DCNv2 import _ext
    x, off, msk, w, b = _dcn_case(24, *shape, off_scale=3.0)
    msk = torch.randn(msk.shape, generator=_g(6)) * 0.8                        # any real mask, not sigmoid(...)
    msk[:, ::3] = 0.0
    go = torch.randn(shape[0], shape[2], shape[3], shape[4], generator=_g(5))
    args = [t.to(DEV) for t in (x, w, b, off, msk, go)]
    lib_ = L.load()
    L.check(lib_.mfx_set_option(b"ext_bwd_fast", 0), "opt")
    slow = [t.cpu() for t in _ext.dcn_v2_backward(*args, 3, 3, 1, 1, 1, 1, 1, 1, 1)]
    L.check(lib_.mfx_set_option(b"ext_bwd_fast", 1), "opt")
    fast = [t.cpu() for t in _ext.dcn_v2_backward(*args, 3, 3, 1, 1, 1, 1, 1, 1, 1)]
    L.check(lib_.mfx_reset_options(), "reset")
    for f_, s_, name in zip(fast, slow, ["grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"]):
        scale = max(1.0, float(s_.abs().max()))
        assert float((f_ - s_).abs().max()) < 2e-4 * scale, name


@pytest.mark.parametrize("shape", [(2, 2, 2, 4, 4), (2, 8, 8, 12, 20), (1, 64, 64, 12, 20), (2, 128, 64, 6, 10)])
def test_ext_dcn_v2_backward_vs_oracle(shape):
    from monoflex_amd.model.backbone.DCNv2 import _ext
    from oracle import dcn_ref
    x, off, msk, w, b = _dcn_case(22, *shape)
    go = torch.randn(shape[0], shape[2], shape[3], shape[4], generator=_g(5))
    want = dcn_ref.dcn_v2_backward(x, w, b, off, msk, go, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    got = _ext.dcn_v2_backward(*[t.to(DEV) for t in (x, w, b, off, msk, go)], 3, 3, 1, 1, 1, 1, 1, 1, 1)
    for g_, w_, name in zip(got, want, ["grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"]):
        scale = max(1.0, float(w_.abs().max()))
        assert float((g_.cpu() - w_).abs().max()) < 1e-4 * scale, name


def test_dcn_autograd_gradcheck_reference_tolerances():
    # reference testcpu.py:69-97 on the HIP autograd Function (fp32, eps 1e-3, atol 1e-4, rtol 1e-2)
    from torch.autograd import gradcheck
    from monoflex_amd.model.backbone.DCNv2.dcn_v2 import dcn_v2_conv
    torch.manual_seed(3)
    inp = (torch.rand(2, 2, 4, 4) * 0.01).to(DEV).requires_grad_()
    offset = (torch.randn(2, 18, 4, 4) * 2).to(DEV).requires_grad_()
    mask = torch.sigmoid(torch.rand(2, 9, 4, 4)).to(DEV).requires_grad_()
    weight = torch.randn(2, 2, 3, 3).to(DEV).requires_grad_()
    bias = torch.rand(2).to(DEV).requires_grad_()
    assert gradcheck(dcn_v2_conv, (inp, offset, mask, weight, bias, 1, 1, 1, 1), eps=1e-3, atol=1e-4, rtol=1e-2,
                     nondet_tol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dcn_module_fused_bn_relu(dtype):
    """DeformConv = DCN + BN + ReLU (dla_dcn.py:384-396), NHWC fused path, against the oracle module."""
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
    got = _from_nhwc(m(_to_nhwc(x, dtype)))
    if dtype == torch.float32:
        assert float((got - want).abs().max()) < 5e-5 * max(1.0, float(want.detach().abs().max()))
    else:   # bf16: offsets carry ~3 significant digits -> sampled values move; check relative L2 error
        rel = float((got - want).norm() / want.norm())
        assert rel < 3e-2, rel


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_heads_fused_vs_torch(dtype):
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.head.detector_predictor import _predictor, REG_OFF
    from monoflex_amd import synthetic as S
    from oracle import monoflex_ref as R
    import os
    cfg = get_cfg(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "runs", "monoflex.yaml"))
    ref = R.Predictor().eval()
    sd = S.synthetic_state_dict({"heads.predictor." + k: v for k, v in ref.state_dict().items()}, seed=3)
    ref.load_state_dict({k[len("heads.predictor."):]: v for k, v in sd.items()})
    m = _predictor(cfg, 64).eval()
    m.load_state_dict(ref.state_dict())
    m.to(DEV)
    tgt = S.synthetic_target(40, 24)
    x = torch.randn(2, 64, 24, 40, generator=_g(9)).relu()
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    taps = {}
    ei = torch.stack([tgt["edge_indices"]] * 2)
    el = torch.tensor([tgt["edge_len"]] * 2)
    with torch.no_grad():
        maps = ref(x, ei, el, taps)
    hm = m.forward_nhwc(_to_nhwc(x, dtype), ei.to(DEV, torch.int32), el.to(DEV, torch.int32)).cpu()
    assert torch.equal(m.last_cls_planar.cpu().view(2, 3, 24, 40), hm[..., :3].permute(0, 3, 1, 2))   # planar copy stays in sync
    got_cls = hm[..., :3].permute(0, 3, 1, 2)
    got_reg = hm[..., REG_OFF:REG_OFF + 50].permute(0, 3, 1, 2)
    tol = 1e-4 if dtype == torch.float32 else 5e-2
    assert float((got_cls - taps["cls_logits"]).abs().max()) < tol
    assert float((got_reg - maps["reg"]).abs().max()) < tol * max(1.0, float(maps["reg"].abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_heads_fused_mfma_32x32x16_form_equals_the_16x16x32_form(dtype):
    """csrc/heads.hip heads_fused32_kernel (option heads_mfma32 = 1: the same tile on v_mfma_f32_32x32x16, its own weight packs w1_32 / w2_32, the trunk
    accumulators feeding GEMM2 in the 32x32 D layout) against the production kernel on the same input: the same products summed in fp32 in another
    order, so equal to a few fp32 ulps of the 576-term sums -- and the same rounding of the trunk activations to 16 bits except where a value sits on a
    rounding boundary.  Ragged map (partial tiles), persistent and per-tile launches, every branch incl. the 32-output pass."""
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.head.detector_predictor import _predictor
    import os
    ops, L = _ops()
    lib_ = L.load()
    cfg = get_cfg(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "runs", "monoflex.yaml"))
    m = _predictor(cfg, 64).eval().to(DEV)
    pk = m._pack(dtype)
    assert pk.w1_32 is not None and pk.w2_32 is not None
    written = torch.zeros(pk.ld_out, dtype=torch.bool)
    for o, c in zip(pk.ch_off, pk.c_out):
        written[o:o + c] = True
    written = written.to(DEV)
    x = torch.randn(3, 21, 37, 64, generator=_g(4)).relu().to(DEV, dtype)
    outs = {}
    for persist in (0, 7):
        for m32 in (0, 1):
            L.check(lib_.mfx_set_option(b"heads_persist", persist), "opt"); L.check(lib_.mfx_set_option(b"heads_mfma32", m32), "opt")
            hm, planar = ops.heads_fused(x, pk, planar_classes=3)
            outs[(persist, m32)] = (hm[..., written].float().cpu(), planar.float().cpu())
    L.check(lib_.mfx_reset_options(), "reset")
    ref = outs[(0, 0)]
    scale = float(ref[0].abs().max())
    for key, (hm, planar) in outs.items():
        assert torch.isfinite(hm).all()
        err = float((hm - ref[0]).abs().max())
        assert err <= (2e-2 if dtype == torch.bfloat16 else 3e-3) * max(1.0, scale), (key, err)           # (a flipped 16-bit rounding of a trunk activation)
        assert float((hm - ref[0]).abs().mean()) <= 2e-4 * max(1.0, scale), key
        assert float((planar - ref[1]).abs().max()) <= (2e-2 if dtype == torch.bfloat16 else 3e-3) * max(1.0, scale)
    assert torch.equal(outs[(7, 1)][0], outs[(0, 1)][0])                                                   # schedules agree bit for bit in the new form too


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_heads_fused_schedules_agree(dtype):
    """The persistent launch (contiguous ranges of (tile, branch) units per workgroup, patch reloaded when the tile changes inside a
    range) computes every output with the same operations in the same order as the one-workgroup-per-tile launch: bit-identical maps."""
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.head.detector_predictor import _predictor
    import os
    ops, L = _ops()
    lib_ = L.load()
    cfg = get_cfg(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "runs", "monoflex.yaml"))
    m = _predictor(cfg, 64).eval().to(DEV)
    pk = m._pack(dtype)
    written = torch.zeros(pk.ld_out, dtype=torch.bool)                              # the row's gaps between branches are never written
    for o, c in zip(pk.ch_off, pk.c_out):
        written[o:o + c] = True
    written = written.to(DEV)
    x = torch.randn(3, 21, 37, 64, generator=_g(4)).relu().to(DEV, dtype)           # ragged: 3 x 3 tiles per image, 243 units

    def run(**opts):
        for k, v in opts.items():
            L.check(lib_.mfx_set_option(k.encode(), v), "opt")
        try:
            hm, planar = ops.heads_fused(x, pk, planar_classes=3)
            torch.cuda.synchronize()
            return hm[..., written].clone(), planar.clone()
        finally:
            lib_.mfx_set_option(b"heads_persist", 1)      # the defaults
            lib_.mfx_set_option(b"heads_planes", 0)

    want = run(heads_persist=0)
    assert bool(torch.isfinite(want[0]).all())
    for opts in ({"heads_persist": 7}, {"heads_persist": 26}, {"heads_persist": 1}, {"heads_persist": 11, "heads_planes": 1}):
        got = run(**opts)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), opts


def test_decode_vs_reference_goldens(golden_dir):
    """Device decode against fixtures captured from the reference's own PostProcessor."""
    import os
    from monoflex_amd import synthetic as S
    from monoflex_amd.structures.params_3d import Calibration
    ops, L = _ops()
    g = np.load(os.path.join(golden_dir, "decode_only.npz"))
    tgt = S.synthetic_target(320, 96)
    n = 0
    while "case%d_seed" % n in g:
        gen = _g(int(g["case%d_seed" % n]))
        logits = torch.randn(1, 3, 96, 320, generator=gen) * 0.8 - 2.0 + float(g["case%d_shift" % n])
        reg = torch.randn(1, 50, 96, 320, generator=gen) * 0.7
        hm = torch.zeros(1, 96, 320, 64)
        hm[..., :3] = logits.permute(0, 2, 3, 1)
        hm[..., 8:58] = reg.permute(0, 2, 3, 1)
        hm = hm.to(DEV)
        scores, index = ops.decode_topk(hm, 0, 3, 50)
        calib = torch.from_numpy(Calibration(tgt["P"]).as_f32()).view(1, 6).to(DEV)
        pad = tgt["pad_size"].view(1, 2).to(DEV, torch.int32)
        size = torch.tensor(list(tgt["size"]), dtype=torch.int32, device=DEV)
        det, topk, valid = ops.decode_boxes(hm, 8, scores, index, calib, pad, size, 0.2)
        res = det[0][valid[0].bool()].cpu().numpy()
        want = g["case%d_result" % n]
        assert res.shape == want.shape, (n, res.shape, want.shape)
        if want.shape[0]:
            assert np.allclose(res, want, rtol=1e-4, atol=2e-3), (n, np.abs(res - want).max())
        n += 1
    assert n >= 4


@pytest.mark.parametrize("mode", ["hard", "mean", "direct", "keypoints_avg", "keypoints_center", "keypoints_02", "keypoints_13", "oracle"])
def test_decode_depth_modes_vs_reference_goldens(golden_dir, mode):
    """The reference's other `output_depth` settings (detector_infer.py:149-198) through mfx_decode_boxes_mode, and 'oracle' (get_oracle_depths,
    :238-277) through PostProcessor.decode_oracle, against rows captured from the reference's PostProcessor on the same maps."""
    import os
    from monoflex_amd import synthetic as S
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.head.detector_infer import make_post_processor
    from monoflex_amd.structures.params_3d import Calibration, ParamsList
    ops, L = _ops()
    g = np.load(os.path.join(golden_dir, "decode_only.npz"))
    tgt = S.synthetic_target(320, 96)
    cfg = get_cfg(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "runs", "monoflex.yaml"), [])
    post = make_post_processor(cfg)
    post.output_depth = mode
    for n in (0, 1):
        gen = _g(int(g["case%d_seed" % n]))
        logits = torch.randn(1, 3, 96, 320, generator=gen) * 0.8 - 2.0 + float(g["case%d_shift" % n])
        reg = torch.randn(1, 50, 96, 320, generator=gen) * 0.7
        hm = torch.zeros(1, 96, 320, 64)
        hm[..., :3] = logits.permute(0, 2, 3, 1)
        hm[..., 8:58] = reg.permute(0, 2, 3, 1)
        t = ParamsList(image_size=tgt["size"], is_train=False)
        t.add_field("pad_size", tgt["pad_size"])
        t.add_field("calib", Calibration(tgt["P"]))
        if mode == "oracle":
            G = g["case%d_gt_boxes" % n].shape[0]
            pad_rows = lambda a: torch.cat((torch.from_numpy(a), torch.zeros((2,) + a.shape[1:], dtype=torch.from_numpy(a).dtype)))
            t.add_field("reg_mask", torch.cat((torch.ones(G, dtype=torch.uint8), torch.zeros(2, dtype=torch.uint8))))
            t.add_field("cls_ids", pad_rows(g["case%d_gt_cls" % n]))
            t.add_field("gt_bboxes", pad_rows(g["case%d_gt_boxes" % n]))
            t.add_field("locations", pad_rows(np.stack((np.zeros(G, np.float32), np.zeros(G, np.float32), g["case%d_gt_depth" % n]), axis=1)))
            with pytest.raises(ValueError):
                post.decode_device(hm.to(DEV), *post.prepare_targets([t], DEV))
        res, _, _ = post({"hm_nhwc": hm.to(DEV), "cls": None}, [t])
        want = g["case%d_result_%s" % (n, mode)]
        assert tuple(res.shape) == want.shape, (n, res.shape, want.shape)
        assert np.allclose(res.cpu().numpy(), want, rtol=1e-4, atol=2e-3), (n, mode, np.abs(res.cpu().numpy() - want).max())
    with pytest.raises(ValueError):
        ops.decode_boxes(hm.to(DEV), 8, *ops.decode_topk(hm.to(DEV), 0, 3, 50), torch.zeros(1, 6, device=DEV), torch.zeros(1, 2, dtype=torch.int32, device=DEV),
                         torch.tensor([1280, 384], dtype=torch.int32, device=DEV), 0.2, depth_mode="combine")


def _topk_reference(logits, K):
    """torch restatement of nms_hm + per-class top-K with ties toward the lower flat index (layers/utils.py:39-77)."""
    heat = torch.sigmoid(logits).clamp(1e-4, 1 - 1e-4)
    keep = torch.nn.functional.max_pool2d(heat, 3, 1, 1) == heat
    flat = (heat * keep).flatten(2)                                              # (B, C, H*W)
    # a stable descending sort on the value alone keeps equal values in ascending index order
    order = torch.sort(flat, dim=-1, descending=True, stable=True).indices[..., :K]
    return torch.gather(flat, 2, order), order


@pytest.mark.parametrize("B,H,W,K,kind", [(2, 96, 320, 50, "dense"), (1, 96, 320, 50, "sparse"), (2, 37, 50, 50, "ties"),
                                          (1, 96, 320, 100, "plateau"), (3, 20, 24, 50, "dense")])
def test_decode_topk_strip_kernel_equals_single_workgroup(B, H, W, K, kind):
    """Stage 1 of the decode, strip-parallel form (row strips + merge) against the single-workgroup kernel and a torch
    restatement: identical scores and indices for every strip count, with ties, plateaus and fewer than K survivors."""
    ops, L = _ops()
    lib_ = L.load()
    g = _g(5)
    logits = torch.randn(B, 3, H, W, generator=g) * 1.5 - 2.0
    if kind == "sparse":                                                        # fewer than K non-zero survivors in one class
        logits[:, 0] = -30.0
        logits[:, 0, 10, 11], logits[:, 0, 50, 200], logits[:, 0, 95, 319] = 2.0, 1.0, 3.0
    elif kind == "ties":                                                        # many exactly equal heat values
        logits = (logits * 2).round() / 2
    elif kind == "plateau":                                                     # constant regions survive the NMS as a whole
        logits[:, 1, 20:40, 100:140] = 1.25
        logits[:, 2] = 0.5
    hm = torch.zeros(B, H, W, 64)
    hm[..., :3] = logits.permute(0, 2, 3, 1)
    hm = hm.to(DEV)
    want_s, want_i = _topk_reference(logits.to(DEV), K)
    try:
        outs = {}
        for strips in (1, 2, 3, 8, 16):
            L.check(lib_.mfx_set_option(b"topk_strips", strips), "opt")
            s_, i_ = ops.decode_topk(hm, 0, 3, K)
            outs[strips] = (s_.clone(), i_.clone())
        planar = logits.to(DEV).flatten(2).contiguous()
        L.check(lib_.mfx_set_option(b"topk_strips", 8), "opt")
        outs["planar"] = ops.decode_topk(hm, 0, 3, K, planar=planar)
    finally:
        L.check(lib_.mfx_set_option(b"topk_strips", 8), "opt")
    for k, (s_, i_) in outs.items():
        assert torch.equal(s_, outs[1][0]) and torch.equal(i_, outs[1][1]), k
    assert torch.allclose(outs[8][0], want_s, rtol=0, atol=1e-6) and torch.equal(outs[8][1].long(), want_i)


@pytest.mark.parametrize("B,C,Cout,H,W,off_std", [(2, 64, 64, 20, 40, 1.5), (1, 128, 64, 33, 48, 1.0), (2, 64, 128, 16, 16, 4.0),
                                                 (1, 256, 256, 12, 24, 2.0)])
def test_dcn_patch_kernel_matches_first_generation(B, C, Cout, H, W, off_std):
    """LDS-patch DCN kernel (dcn_patch.hip, all three tile heights) against the first-generation global-gather kernel on
    the same bf16 inputs: same blend arithmetic and K order, so the results agree to accumulation rounding.  Ragged
    tiles (H, W not multiples of the tile), two channel slices, several n-tiles, and offsets far outside the patch
    (off_std 4: most samples take the global fallback; some leave the image)."""
    from monoflex_amd import lib as L, ops
    g = _g(31)
    x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).to(DEV)
    om = torch.zeros(B, H, W, 32)
    om[..., :18] = torch.randn(B, H, W, 18, generator=g) * off_std
    om[..., 18:27] = torch.rand(B, H, W, 9, generator=g)
    om = om.to(DEV)
    w = torch.randn(Cout, C, 3, 3, generator=g) * (1.0 / (3 * C ** 0.5))
    p = ops.pack_conv(w.to(DEV), torch.bfloat16, torch.rand(Cout, generator=g).to(DEV) + 0.5, torch.randn(Cout, generator=g).to(DEV),
                      stride=1, pad=1, act=L.ACT_RELU)
    ops.add_f16_fragments(p, w)
    lib_ = L.load()
    try:
        L.check(lib_.mfx_set_option(b"dcn_patch", 0), "opt"); L.check(lib_.mfx_set_option(b"dcn_wave", 0), "opt")
        want = ops.dcn(x, om, p).float().cpu()
        for v in (2, 3, 4, 5, 6, 7):                         # 5..7: +-7 pixel margin, 32-channel slices
            L.check(lib_.mfx_set_option(b"dcn_patch", v), "opt")
            got = ops.dcn(x, om, p).float().cpu()
            err = float((got - want).abs().max())
            assert err <= 2e-2 * max(1.0, float(want.abs().max())), (v, err)
            assert float((got - want).abs().mean()) <= 1e-3 * max(1.0, float(want.abs().mean())), v
    finally:
        L.check(lib_.mfx_set_option(b"dcn_patch", 1), "opt"); L.check(lib_.mfx_set_option(b"dcn_wave", 1), "opt")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 12, 40, 512, 512, 3, 1, True), (2, 24, 80, 256, 256, 3, 1, False), (2, 24, 80, 128, 256, 3, 2, False),
                                   (1, 12, 40, 256, 512, 1, 1, False)])
def test_conv_split_k_matches_single_pass(dtype, shape):
    """Split-K path of the generic conv (small-M / long-K layers of DLA level4/level5): every legal split count against
    the single-pass kernel with the same epilogue (BN scale/shift, residual, ReLU)."""
    from monoflex_amd import lib as L, ops
    B, H, W, Ci, Co, k, s, with_res = shape
    g = _g(41)
    x = torch.randn(B, H, W, Ci, generator=g).to(dtype).to(DEV)
    w = (torch.randn(Co, Ci, k, k, generator=g) / (k * Ci ** 0.5)).to(DEV)
    p = ops.pack_conv(w, dtype, (torch.rand(Co, generator=g) + 0.5).to(DEV), torch.randn(Co, generator=g).to(DEV), stride=s, pad=k // 2,
                      act=L.ACT_RELU)
    p.w_frag = None                                            # keep the layer on the generic kernel
    res = torch.randn(B, H // s, W // s, Co, generator=g).to(dtype).to(DEV) if with_res else None
    lib_ = L.load()
    try:
        L.check(lib_.mfx_set_option(b"ksplit", 1), "opt")
        want = ops.conv2d(x, p, res=res).float().cpu()
        for ks in (0, 2, 3, 8):
            L.check(lib_.mfx_set_option(b"ksplit", ks), "opt")
            got = ops.conv2d(x, p, res=res).float().cpu()
            tol = 2e-5 if dtype == torch.float32 else 2e-2
            assert float((got - want).abs().max()) <= tol * max(1.0, float(want.abs().max())), ks
    finally:
        L.check(lib_.mfx_set_option(b"ksplit", 0), "opt")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C", [(2, 20, 40, 64), (1, 12, 40, 512), (2, 9, 17, 128)])
def test_offset_conv_k_split_waves(dtype, B, H, W, C):
    """3x3 conv with a narrow output (the 27-channel DCN offset/mask conv, fp32 out, sigmoid on the mask channels):
    halo-kernel variants whose waves split K (8: 4 waves, 9: 2 waves, 10: BN16 x 4 waves) against the generic kernel."""
    from monoflex_amd import lib as L, ops
    g = _g(51)
    x = torch.randn(B, H, W, C, generator=g).to(dtype).to(DEV)
    w = (torch.randn(27, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
    bias = torch.randn(27, generator=g).to(DEV)
    p = ops.pack_conv(w, dtype, None, bias, stride=1, pad=1, act=L.ACT_DCN_OFFMASK, cout=32)
    lib_ = L.load()
    try:
        L.check(lib_.mfx_set_option(b"halo", 0), "opt")
        want = ops.conv2d(x, p, out_dtype=torch.float32).cpu()
        for v in (8, 9, 10):
            L.check(lib_.mfx_set_option(b"halo", v + 1), "opt")
            got = ops.conv2d(x, p, out_dtype=torch.float32).cpu()
            tol = 2e-5 if dtype == torch.float32 else 2e-3
            assert float((got - want).abs().max()) <= tol * max(1.0, float(want.abs().max())), v
    finally:
        L.check(lib_.mfx_set_option(b"halo", 1), "opt")


def _dcn_nhwc_case(B, C, Cout, H, W, off_std, dtype, far_frac=0.0, seed=71):
    from monoflex_amd import lib as L, ops
    g = _g(seed)
    x = torch.randn(B, H, W, C, generator=g).relu().to(dtype).to(DEV)
    om = torch.zeros(B, H, W, 32)
    om[..., :18] = torch.randn(B, H, W, 18, generator=g) * off_std
    if far_frac > 0:                                           # a sprinkle of wild offsets: samples that leave the LDS patch, some of them the image
        wild = torch.rand(B, H, W, 18, generator=g) < far_frac
        om[..., :18] = torch.where(wild, torch.randn(B, H, W, 18, generator=g) * 40.0, om[..., :18])
    om[..., 18:27] = torch.rand(B, H, W, 9, generator=g)
    om = om.to(DEV)
    w = torch.randn(Cout, C, 3, 3, generator=g) * (1.0 / (3 * C ** 0.5))
    p = ops.pack_conv(w.to(DEV), dtype, torch.rand(Cout, generator=g).to(DEV) + 0.5, torch.randn(Cout, generator=g).to(DEV) * 0.1,
                      stride=1, pad=1, act=L.ACT_RELU)
    ops.add_f16_fragments(p, w)
    p32 = ops.pack_conv(w.to(dtype).float().to(DEV), torch.float32, p.scale, p.shift, stride=1, pad=1, act=L.ACT_RELU)
    return x, om, p, p32


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows", [16, 8])
@pytest.mark.parametrize("B,C,Cout,H,W,off_std,far", [(2, 64, 64, 32, 48, 1.5, 0.0), (1, 64, 64, 20, 40, 2.5, 0.02), (2, 128, 64, 48, 64, 3.0, 0.01),
                                                      (1, 64, 64, 16, 16, 12.0, 0.0), (1, 256, 64, 9, 21, 2.0, 0.05)])
def test_dcn_lds_kernel_matches_the_gather_kernel(dtype, rows, B, C, Cout, H, W, off_std, far):
    """csrc/dcn_lds.hip (r06, fourth generation: LDS patch + geometry table + branch-free sampling loop, far pass from global memory), both tile
    heights, against the library's gather kernel and its fp32 kernel on the same values: in-patch samples, samples that leave the patch or the
    image (the far pass: `far` = fraction of wild offsets; std 12 on a 16 x 16 map = mostly far), partial tiles, 4 to 16 channel slices."""
    from monoflex_amd import lib as L, ops
    x, om, p, p32 = _dcn_nhwc_case(B, C, Cout, H, W, off_std, dtype, far)
    assert p.w_pair_f16 is not None
    lib_ = L.load()
    L.check(lib_.mfx_set_option(b"dcn_lds", 0), "opt"); L.check(lib_.mfx_set_option(b"dcn_patch", 0), "opt"); L.check(lib_.mfx_set_option(b"dcn_wave", 0), "opt")
    ref = ops.dcn(x.float(), om, p32).cpu()                   # fp32 kernel on the same 16-bit-rounded operands
    want = ops.dcn(x, om, p).float().cpu()
    L.check(lib_.mfx_set_option(b"dcn_lds", 2), "opt"); L.check(lib_.mfx_set_option(b"dcn_lds_rows", rows), "opt")
    got = ops.dcn(x, om, p).float().cpu()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert float((got - want).abs().max()) <= tol * max(1.0, float(want.detach().abs().max()))
    assert float((got - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    # (the LDS kernels blend the four corners in packed fp16 -- weights and sums round to 11 bits -- where the gather kernel blends in fp32: measured 1.7x
    # the gather kernel's mean distance from fp32 on fp16 maps, equal on bf16 maps whose own 8-bit rounding dominates)
    assert float((got - ref).abs().mean()) <= 2.5 * float((want - ref).abs().mean()) + 1e-6


@pytest.mark.parametrize("B,C,Cout,H,W,off_std,far", [(2, 64, 64, 32, 48, 1.5, 0.0), (1, 64, 64, 20, 40, 2.5, 0.02), (2, 128, 64, 48, 64, 3.0, 0.01),
                                                      (1, 64, 64, 16, 16, 12.0, 0.0), (8, 64, 64, 96, 320, 2.5, 0.001)])
def test_dcn_lds_split_kernel_matches_the_fp32_kernel(B, C, Cout, H, W, off_std, far):
    """Split precision (MFX_F16X2: the mode that carries the north-star gate): csrc/dcn_lds.hip dcn_lds_split_kernel -- fp32 patch in LDS, fp32 blend, (hi, lo)
    fp16 operand pairs, three MFMAs per product -- against the library's fp32 kernel (v_mfma_f32_16x16x4_f32) and its split-precision gather kernel on
    the same fp32 values: fp32-grade agreement (the reference computes in fp32: src/cuda/dcn_v2_cuda.cu:58), far pass and partial tiles included, and the
    range sentinel stays clear."""
    from monoflex_amd import lib as L, ops
    g = _g(79)
    x = torch.randn(B, H, W, C, generator=g).relu().to(DEV)
    om = torch.zeros(B, H, W, 32)
    om[..., :18] = torch.randn(B, H, W, 18, generator=g) * off_std
    if far > 0:
        wild = torch.rand(B, H, W, 18, generator=g) < far
        om[..., :18] = torch.where(wild, torch.randn(B, H, W, 18, generator=g) * 40.0, om[..., :18])
    om[..., 18:27] = torch.rand(B, H, W, 9, generator=g)
    om = om.to(DEV)
    w = torch.randn(Cout, C, 3, 3, generator=g) * (1.0 / (3 * C ** 0.5))
    sc, sh = torch.rand(Cout, generator=g).to(DEV) + 0.5, torch.randn(Cout, generator=g).to(DEV) * 0.1
    p32 = ops.pack_conv(w.to(DEV), torch.float32, sc, sh, stride=1, pad=1, act=L.ACT_RELU)
    ps = ops.pack_conv(w.to(DEV), ops.F16X2, sc, sh, stride=1, pad=1, act=L.ACT_RELU)
    ops.add_f16_fragments(ps, w)
    assert ps.split and ps.w_pair_f16 is not None and ps.w_frag_f16 is not None
    lib_ = L.load()
    L.f16x2_range_ok()                                         # clear the flag
    L.check(lib_.mfx_set_option(b"dcn_lds", 0), "opt"); L.check(lib_.mfx_set_option(b"dcn_patch", 0), "opt"); L.check(lib_.mfx_set_option(b"dcn_wave", 0), "opt")
    ref = ops.dcn(x, om, p32).cpu()
    gather = ops.dcn(x, om, ps).cpu()
    L.check(lib_.mfx_set_option(b"dcn_lds", 2), "opt")
    got = ops.dcn(x, om, ps).cpu()
    assert L.f16x2_range_ok()
    scale = max(1.0, float(ref.abs().max()))
    e_new, e_old = float((got - ref).abs().max()), float((gather - ref).abs().max())
    assert e_new <= 2e-5 * scale, (e_new, e_old)
    assert e_new <= 3.0 * e_old + 2e-6 * scale, (e_new, e_old)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,C,Cout,H,W", [(2, 512, 256, 12, 40), (2, 256, 64, 24, 80), (1, 128, 64, 13, 37), (1, 256, 128, 7, 9)])
def test_dcn_project_then_sample_matches_the_gather_kernel(dtype, B, C, Cout, H, W):
    """DCN as two launches (ops.dcn_ps: the 1x1 projection C -> 9 Cout of the whole map, then csrc/dcn_ps.hip samples the projected map -- bilinear
    interpolation commutes with the contraction over channels) against the fused gather kernel and the fp32 kernel: the same op up to the 16-bit
    rounding of the projected map; odd map sizes (partial pixel blocks), offsets that leave the image."""
    from monoflex_amd import lib as L, ops
    x, om, p, p32 = _dcn_nhwc_case(B, C, Cout, H, W, 3.0, dtype, 0.02, seed=73)
    lib_ = L.load()
    L.check(lib_.mfx_set_option(b"dcn_patch", 0), "opt"); L.check(lib_.mfx_set_option(b"dcn_wave", 0), "opt")
    ref = ops.dcn(x.float(), om, p32).cpu()
    want = ops.dcn(x, om, p).float().cpu()
    got = ops.dcn_ps(x, om, p).float().cpu()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert float((got - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().mean()) <= 1.25 * float((want - ref).abs().mean()) + 1e-6
    assert ops.dcn_ps_applies(x, p) == (B * H * W * 9 * Cout * 2 <= ops.DCN_PS_MAX_BYTES[0])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K,N", [(3840, 512, 2304), (1001, 128, 576), (15360, 256, 2304), (130, 256, 576), (61440, 128, 576), (61440, 64, 1152),
                                   (999, 64, 2304)])
def test_project_gemm_equals_the_tiled_1x1_kernel(dtype, M, K, N):
    """csrc/gemm_as.hip (mfx_project_nhwc: activation-stationary GEMM, weight rows of a fragment pair permuted for 16-byte stores, channel chunks split
    over workgroups on small maps) against mfx_conv2d_nhwc's 1x1 kernel on the same operands -- the same fp32 sums, rounded once: equal bits except
    where the two summation orders straddle a 16-bit rounding boundary -- and against torch; ragged M."""
    ops, L = _ops()
    g = _g(161)
    x = torch.randn(1, 1, M, K, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, K, 1, 1, generator=g) / K ** 0.5).to(DEV)
    p = ops.pack_conv(w, dtype, None, None, stride=1, pad=0, act=L.ACT_NONE)
    assert p.K_pad == K and p.Cout_pad == N
    want = ops.conv2d(x, p).view(M, N)
    got = torch.empty(M, N, dtype=dtype, device=DEV)
    L.check(L.load().mfx_project_nhwc(x.data_ptr(), p.w.data_ptr(), got.data_ptr(), M, K, N, K, N, L.MFX_BF16 if dtype == torch.bfloat16 else L.MFX_F16,
                                      torch.cuda.current_stream().cuda_stream), "mfx_project_nhwc")
    ref = x.view(M, K).float() @ w.view(N, K).to(dtype).float().t()
    tol = (2e-2 if dtype == torch.bfloat16 else 3e-3) * max(1.0, float(ref.abs().max()))
    assert float((got.float() - ref).abs().max()) <= tol
    assert float((got.float() - want.float()).abs().max()) <= tol and float((got != want).float().mean()) <= 2e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows", [16, 8])
def test_dcn_lds_module_with_the_offset_conv_inside(dtype, rows):
    """The 64 -> 64 module as ONE launch of dcn_lds_kernel<OF> (offset / mask conv of the tile inside, phase 0) against the two-launch form
    (offset conv, then the gather kernel), at a size the automatic choice takes (B H W >= 65536) and with partial tiles."""
    from monoflex_amd import lib as L
    from monoflex_amd.model.backbone.dla_dcn import DeformConv
    lib_ = L.load()
    torch.manual_seed(5)
    for (B, H, W) in ((3, 100, 232), (8, 96, 320)):
        x = torch.randn(B, H, W, 64, device=DEV).relu().to(dtype)
        m = DeformConv(64, 64).eval().to(DEV)
        torch.nn.init.normal_(m.conv.conv_offset_mask.weight, std=2.5 / (0.7 * (9 * 64) ** 0.5))
        with torch.no_grad():
            L.check(lib_.mfx_set_option(b"dcn_lds", 1), "opt"); L.check(lib_.mfx_set_option(b"dcn_lds_rows", rows), "opt")
            got = m(x).float().cpu()
            L.check(lib_.mfx_set_option(b"dcn_lds", 0), "opt"); L.check(lib_.mfx_set_option(b"dcn_patch", 0), "opt"); L.check(lib_.mfx_set_option(b"dcn_wave", 0), "opt")
            want = m(x).float().cpu()
        L.check(lib_.mfx_reset_options(), "reset")
        tol = 3e-2 if dtype == torch.bfloat16 else 6e-3         # (the offsets themselves come from two differently rounded convs)
        frac_bad = float(((got - want).abs() > tol * want.abs().clamp(min=1.0)).float().mean())
        assert frac_bad <= 1e-4, frac_bad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,Cout,H,W", [(2, 512, 256, 12, 40), (2, 256, 128, 24, 40), (1, 128, 64, 17, 23)])
def test_dcn_split_k_matches_single_pass(dtype, B, C, Cout, H, W):
    """Split-K of the fused DCN kernel (small maps): forced split counts incl. ones that start in the middle of a tap,
    against the single-pass kernel with the same BN/ReLU epilogue."""
    from monoflex_amd import lib as L, ops
    g = _g(61)
    x = torch.randn(B, H, W, C, generator=g).to(dtype).to(DEV)
    om = torch.zeros(B, H, W, 32)
    om[..., :18] = torch.randn(B, H, W, 18, generator=g) * 2.0
    om[..., 18:27] = torch.rand(B, H, W, 9, generator=g)
    om = om.to(DEV)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
    p = ops.pack_conv(w, dtype, (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV), stride=1, pad=1,
                      act=L.ACT_RELU)
    lib_ = L.load()
    try:
        for o in (b"dcn_patch", b"dcn_wave"):
            L.check(lib_.mfx_set_option(o, 0), "opt")
        L.check(lib_.mfx_set_option(b"dcn_ksplit", 1), "opt")
        want = ops.dcn(x, om, p).float().cpu()
        for ks in (0, 2, 5, 9):
            L.check(lib_.mfx_set_option(b"dcn_ksplit", ks), "opt")
            got = ops.dcn(x, om, p).float().cpu()
            tol = 2e-5 if dtype == torch.float32 else 2e-2
            assert float((got - want).abs().max()) <= tol * max(1.0, float(want.abs().max())), ks
    finally:
        L.check(lib_.mfx_set_option(b"dcn_ksplit", 0), "opt")
        for o in (b"dcn_patch", b"dcn_wave"):
            L.check(lib_.mfx_set_option(o, 1), "opt")


@pytest.mark.parametrize("B,H,W", [(2, 16, 64), (1, 37, 70), (2, 9, 130)])
def test_stem_kernel_matches_generic_path(B, H, W):
    """Dedicated bf16 stem kernel (reads the NCHW planes, weights in registers) against the generic implicit-GEMM path
    over the padded NHWC4 image, same packed weights, BN + ReLU; and against F.conv2d in fp32."""
    from monoflex_amd import lib as L, ops
    g = _g(71)
    img = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(16, 3, 7, 7, generator=g) * 0.1
    scale, shift = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g)
    p = ops.pack_stem(w.to(DEV), torch.bfloat16, scale.to(DEV), shift.to(DEV), act=L.ACT_RELU)
    want = ops.conv2d(ops.pack_image(img.to(DEV), torch.bfloat16), p, out_hw=(H, W)).float().cpu()
    got = ops.stem_conv(img.to(DEV), p).float().cpu()
    assert got.shape == (B, H, W, 16)
    assert float((got - want).abs().max()) <= 1e-2 * max(1.0, float(want.detach().abs().max()))
    ref = torch.relu(torch.nn.functional.conv2d(img, w, padding=3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    assert float((got - ref).abs().max()) <= 3e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,Cout,variants", [(64, 64, (6, 12, 13)), (128, 128, (7, 11, 14)), (256, 256, (5, 11)), (512, 512, (11,))])
def test_halo_conv_k_split_wide_variants(dtype, C, Cout, variants):
    """Halo-kernel variants for wide outputs, with and without K-split waves (11: BN128 x 2-way, 12/13: BN64 x 2/4-way,
    14: 2 x FN4 x 2-way), forced one by one against the generic kernel, BN + residual + ReLU epilogue, ragged tile edges."""
    from monoflex_amd import lib as L, ops
    g = _g(91)
    B, H, W = 2, 13, 37
    x = torch.randn(B, H, W, C, generator=g).to(dtype).to(DEV)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
    p = ops.pack_conv(w, dtype, (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV), stride=1, pad=1,
                      act=L.ACT_RELU)
    res = torch.randn(B, H, W, Cout, generator=g).to(dtype).to(DEV)
    lib_ = L.load()
    try:
        L.check(lib_.mfx_set_option(b"halo", 0), "opt")
        want = ops.conv2d(x, p, res=res).float().cpu()
        for v in variants:
            L.check(lib_.mfx_set_option(b"halo", v + 1), "opt")
            got = ops.conv2d(x, p, res=res).float().cpu()
            tol = 2e-5 if dtype == torch.float32 else 2e-2
            assert float((got - want).abs().max()) <= tol * max(1.0, float(want.abs().max())), v
    finally:
        L.check(lib_.mfx_set_option(b"halo", 1), "opt")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W", [(2, 48, 96), (1, 38, 74), (1, 384, 1280)])
def test_f1_fused_equals_three_launches_and_torch(dtype, B, H, W):
    """csrc/f1_fused.hip: stem 7x7 -> level0 3x3 -> level1 3x3 / s2 (each + folded BN + ReLU) in one kernel, both full-resolution maps in LDS,
    against (a) the three separate launches on the same packed operands (same 16-bit roundings of both intermediate maps; the sums are
    ordered differently, so single values move by an ulp of the storage type) and (b) torch fp32 convs with the intermediates rounded to the
    storage type.  Ragged level1 tiles (19 x 37 outputs), image borders (zero padding of all three convs), and the bench's frame size."""
    ops, L = _ops()
    g = _g(61)
    img = torch.randn(B, 3, H, W, generator=g)
    rnd = lambda t: t.to(dtype).float()                                    # noqa: E731
    w7 = rnd(torch.randn(16, 3, 7, 7, generator=g) / 147 ** 0.5)
    w0 = rnd(torch.randn(16, 16, 3, 3, generator=g) / 12.0)
    w1 = rnd(torch.randn(32, 16, 3, 3, generator=g) / 12.0)
    bn = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1) for c in (16, 16, 32)]
    ps = ops.pack_stem(w7.to(DEV), dtype, bn[0][0].to(DEV), bn[0][1].to(DEV))
    p0 = ops.pack_conv(w0.to(DEV), dtype, bn[1][0].to(DEV), bn[1][1].to(DEV), stride=1, pad=1, act=L.ACT_RELU)
    p1 = ops.pack_conv(w1.to(DEV), dtype, bn[2][0].to(DEV), bn[2][1].to(DEV), stride=2, pad=1, act=L.ACT_RELU)
    x = img.to(DEV)
    got = ops.f1_fused(x, ps, p0, p1)
    want_hip = ops.conv2d(ops.conv2d(ops.stem_conv(x, ps), p0), p1)
    torch.cuda.synchronize()
    assert got.shape == want_hip.shape == (B, H // 2, W // 2, 32) and got.dtype == dtype
    a, b_ = got.float().cpu(), want_hip.float().cpu()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert float((a - b_).norm() / b_.norm()) < 3 * eps, float((a - b_).norm() / b_.norm())
    assert float((a - b_).abs().max()) <= 8 * eps * max(1.0, float(b_.abs().max()))
    aff = lambda t, i: F.relu(t * bn[i][0].view(1, -1, 1, 1) + bn[i][1].view(1, -1, 1, 1))      # noqa: E731
    r = rnd(aff(F.conv2d(rnd(img), w7, None, 1, 3), 0))
    r = rnd(aff(F.conv2d(r, w0, None, 1, 1), 1))
    r = aff(F.conv2d(r, w1, None, 2, 1), 2)
    ref = r.permute(0, 2, 3, 1)
    assert float((a - ref).norm() / ref.norm()) < 4 * eps, float((a - ref).norm() / ref.norm())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,Cout,B,H,W", [(64, 64, 2, 24, 48), (64, 64, 1, 13, 37), (64, 128, 1, 19, 33), (128, 128, 2, 16, 32), (128, 128, 1, 13, 37),
                                          (256, 256, 2, 16, 32), (256, 256, 1, 11, 21), (512, 512, 2, 12, 40), (512, 512, 1, 9, 17), (512, 512, 1, 18, 21)])
def test_conv_cw_kernel_is_bit_identical_to_the_halo_kernel(dtype, C, Cout, B, H, W):
    """csrc/conv_cw.hip (compile-time geometry, software-pipelined K loop, branch-free patch load, residual prefetch) keeps the K order, the
    accumulation order and the epilogue arithmetic of conv3x3_wave_kernel: same bits, with and without the residual, ReLU / LeakyReLU / no
    activation, full and ragged tiles, one and two channel groups, K-split waves (reference layers: dla_dcn.py:84-98)."""
    ops, L = _ops()
    g = _g(131)
    x = torch.randn(B, H, W, C, generator=g).to(dtype).to(DEV)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
    res = torch.randn(B, H, W, Cout, generator=g).to(dtype).to(DEV)
    lib_ = L.load()
    try:
        for act in (L.ACT_RELU, L.ACT_LEAKY, L.ACT_NONE):
            p = ops.pack_conv(w, dtype, (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV), stride=1, pad=1, act=act)
            for r in (res, None):
                L.check(lib_.mfx_set_option(b"halo_cw", 0), "opt")
                want = ops.conv2d(x, p, res=r)
                L.check(lib_.mfx_set_option(b"halo_cw", 1), "opt")
                got = ops.conv2d(x, p, res=r)
                assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (act, r is not None)
                L.check(lib_.mfx_set_option(b"halo", 0), "opt")              # and the generic implicit-GEMM kernel agrees to rounding
                gen = ops.conv2d(x, p, res=r)
                L.check(lib_.mfx_set_option(b"halo", 1), "opt")
                assert float((got.float() - gen.float()).abs().max()) <= 2e-2 * max(1.0, float(gen.float().abs().max()))
    finally:
        L.check(lib_.mfx_set_option(b"halo_cw", 1), "opt")
        L.check(lib_.mfx_set_option(b"halo", 1), "opt")


@pytest.mark.parametrize("v,C,Cout,B,H,W", [(6, 64, 64, 2, 24, 48), (6, 64, 64, 1, 13, 37), (7, 64, 128, 1, 19, 33), (7, 128, 128, 2, 16, 32), (7, 128, 128, 1, 13, 37),
                                            (11, 128, 128, 1, 10, 20), (11, 256, 256, 2, 16, 32), (11, 256, 256, 1, 11, 21), (11, 512, 512, 2, 12, 40), (11, 512, 512, 1, 9, 17)])
def test_conv_cws_split_precision_kernel_is_bit_identical_to_the_halo_kernel(v, C, Cout, B, H, W):
    """csrc/conv_cws.hip (r06: the split-precision pair-walking 3x3 kernel with compile-time geometry, a software-pipelined K loop and a branch-free patch
    load) keeps the K order, the accumulation order (hi.hi, lo.hi, hi.lo per pair and row) and the epilogue arithmetic of
    conv3x3_wave_kernel<f32s_t, .., PR>: same bits, with and without the residual, every activation, full and ragged tiles, one to eight channel
    groups, 1- and 2-way K splits; and fp32-grade agreement with torch (the reference computes in fp32)."""
    ops, L = _ops()
    g = _g(151)
    x = torch.randn(B, H, W, C, generator=g).to(DEV)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
    res = torch.randn(B, H, W, Cout, generator=g).to(DEV)
    lib_ = L.load()
    L.f16x2_range_ok()
    for act in (L.ACT_RELU, L.ACT_LEAKY, L.ACT_NONE):
        sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
        p = ops.pack_conv(w, ops.F16X2, sc, sh, stride=1, pad=1, act=act)
        assert p.split and p.w_frag_pair is not None
        for r in (res, None):
            L.check(lib_.mfx_set_option(b"halo", v + 1), "opt")
            L.check(lib_.mfx_set_option(b"halo_cws", 0), "opt")
            want = ops.conv2d(x, p, res=r)
            L.check(lib_.mfx_set_option(b"halo_cws", 1), "opt")
            got = ops.conv2d(x, p, res=r)
            assert got.dtype == torch.float32 and torch.equal(got.view(torch.int32), want.view(torch.int32)), (act, r is not None)
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().cpu(), None, padding=1).permute(0, 2, 3, 1) * sc.double().cpu() + sh.double().cpu()
        ref = torch.relu(ref) if act == L.ACT_RELU else torch.nn.functional.leaky_relu(ref, 0.01) if act == L.ACT_LEAKY else ref
        assert float((got.double().cpu() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    L.check(lib_.mfx_reset_options(), "reset")
    assert L.f16x2_range_ok()


@pytest.mark.parametrize("C,B,H,W", [(64, 2, 24, 48), (128, 2, 16, 32), (128, 1, 13, 37), (256, 1, 11, 21), (512, 2, 12, 40), (512, 1, 9, 17)])
def test_conv_cws_offset_mask_conv_is_bit_identical_to_the_halo_kernel(C, B, H, W):
    """The DCN modules' 27-channel offset / mask conv in split precision on conv3x3_cws_kernel's one-slice / four-way K-split instantiation (18 pairs over
    four waves: the fifth pair only on two of them): same bits as conv3x3_wave_kernel's variant 8."""
    ops, L = _ops()
    g = _g(153)
    x = torch.randn(B, H, W, C, generator=g).to(DEV)
    w = (torch.randn(27, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
    bias = torch.randn(27, generator=g).to(DEV)
    p = ops.pack_conv(w, ops.F16X2, None, bias, stride=1, pad=1, act=L.ACT_DCN_OFFMASK, cout=32)
    lib_ = L.load()
    L.check(lib_.mfx_set_option(b"halo_cws", 0), "opt")
    want = ops.conv2d(x, p, out_dtype=torch.float32)
    L.check(lib_.mfx_set_option(b"halo_cws", 1), "opt")
    got = ops.conv2d(x, p, out_dtype=torch.float32)
    L.check(lib_.mfx_reset_options(), "reset")
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, bias, padding=1).permute(0, 2, 3, 1)
    ref = torch.cat((ref[..., :18], torch.sigmoid(ref[..., 18:27])), -1)
    assert float((got[..., :27] - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("v,C,Cout,stride,exact", [(6, 32, 64, 1, True), (7, 32, 128, 1, True), (11, 32, 256, 1, False), (12, 128, 64, 1, True), (13, 256, 64, 1, True),
                                                   (5, 64, 256, 1, True)])
def test_conv_cw_training_shapes_equal_the_halo_kernel(dtype, v, C, Cout, stride, exact):
    """conv3x3_cw_kernel's instantiations for the layers only the training step has (r05: `MFX_TRACE_CW=1` lists what stays on the run-time-geometry
    kernel): 32 input channels (data gradients of the DCN modules' 27-channel offset / mask convs, accumulated into an existing gradient = the
    residual input), the K-split 128 -> 64 and 256 -> 64 forms, 64 -> 256 with four fragments per wave.  Variant forced with option `halo`;
    same bits as conv3x3_wave_kernel except where the two kernels split K differently (variant 11 at 32 channels: one step per tap)."""
    ops, L = _ops()
    g = _g(161)
    B, H, W = 2, 21, 35
    x = torch.randn(B, H, W, C, generator=g).to(dtype).to(DEV)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
    res = torch.randn(B, H, W, Cout, generator=g).to(dtype).to(DEV)
    p = ops.pack_conv(w, dtype, None, None, stride=stride, pad=1, act=L.ACT_NONE)
    lib_ = L.load()
    L.check(lib_.mfx_set_option(b"halo", v + 1), "opt")
    for r in (res, None):
        L.check(lib_.mfx_set_option(b"halo_cw", 0), "opt")
        want = ops.conv2d(x, p, res=r)
        L.check(lib_.mfx_set_option(b"halo_cw", 1), "opt")
        got = ops.conv2d(x, p, res=r)
        if exact:
            assert torch.equal(got.view(torch.int16), want.view(torch.int16)), r is not None
        else:
            assert float((got.float() - want.float()).abs().max()) <= 1.6e-2 * max(1.0, float(want.float().abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("v,C,Cout,stride,H,W", [(6, 64, 64, 1, 21, 35), (7, 128, 128, 1, 16, 48), (6, 32, 64, 2, 45, 75), (7, 64, 128, 2, 48, 64)])
def test_conv_cw_statistics_epilogue_equals_the_halo_kernel(dtype, v, C, Cout, stride, H, W):
    """Train-mode BN statistics of the conv's output from conv3x3_cw_kernel's own epilogue (mfx_conv_desc.stats): the stored map has the same bits as
    conv3x3_wave_kernel's, the per-channel sums and sums of squares agree with it to fp32 summation order and with a torch reduction of the stored map."""
    ops, L = _ops()
    g = _g(171)
    B = 2
    x = torch.randn(B, H, W, C, generator=g).to(dtype).to(DEV)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
    p = ops.pack_conv(w, dtype, None, None, stride=stride, pad=1, act=L.ACT_NONE)
    lib_ = L.load()
    ncopy = lib_.mfx_bn_ncopy(Cout)
    L.check(lib_.mfx_set_option(b"halo", v + 1), "opt")
    outs = {}
    for cw in (0, 1):
        L.check(lib_.mfx_set_option(b"halo_cw", cw), "opt")
        st = torch.zeros(ncopy, 2 * Cout, device=DEV)
        y = ops.conv2d(x, p, stats=st)
        assert ops.conv2d.last_stats_done
        outs[cw] = (y, st.sum(0).cpu())
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    yf = outs[1][0].float().reshape(-1, Cout)
    ref = torch.cat([yf.sum(0), (yf * yf).sum(0)]).cpu()
    for cw in (0, 1):
        assert torch.allclose(outs[cw][1], ref, rtol=2e-4, atol=2e-3), (cw, float((outs[cw][1] - ref).abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,B,H,W,variant", [(128, 2, 16, 32, 8), (128, 1, 13, 37, 10), (256, 2, 16, 32, 10), (256, 1, 11, 21, 8), (512, 2, 12, 40, 10),
                                             (512, 1, 9, 17, 8)])
def test_conv_cw_offset_mask_conv_is_bit_identical_to_the_halo_kernel(dtype, C, B, H, W, variant):
    """The 27-channel DCN offset / mask conv of the wide layers (dcn_v2.py:104-122: fp32 out, bias, sigmoid on channels 18..26, N padded to 32) on
    conv3x3_cw_kernel's one-slice / four-way K-split instantiations: same bits as conv3x3_wave_kernel's variants 8 and 10."""
    ops, L = _ops()
    g = _g(141)
    x = torch.randn(B, H, W, C, generator=g).to(dtype).to(DEV)
    w = (torch.randn(27, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
    bias = torch.randn(27, generator=g).to(DEV)
    p = ops.pack_conv(w, dtype, None, bias, stride=1, pad=1, act=L.ACT_DCN_OFFMASK, cout=32)
    lib_ = L.load()
    try:
        L.check(lib_.mfx_set_option(b"halo", variant + 1), "opt")
        L.check(lib_.mfx_set_option(b"halo_cw", 0), "opt")
        want = ops.conv2d(x, p, out_dtype=torch.float32)
        L.check(lib_.mfx_set_option(b"halo_cw", 1), "opt")
        got = ops.conv2d(x, p, out_dtype=torch.float32)
        assert got.dtype == torch.float32 and torch.equal(got.view(torch.int32), want.view(torch.int32))
        ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(dtype).float(), bias, padding=1).permute(0, 2, 3, 1)
        ref = torch.cat((ref[..., :18], torch.sigmoid(ref[..., 18:27])), -1)
        assert float((got[..., :27] - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    finally:
        L.check(lib_.mfx_set_option(b"halo_cw", 1), "opt")
        L.check(lib_.mfx_set_option(b"halo", 1), "opt")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,Cout,B,H,W,exact", [(32, 64, 2, 96, 160, True), (32, 64, 1, 45, 75, True), (64, 128, 2, 48, 64, True), (64, 128, 1, 27, 39, True),
                                                (128, 256, 2, 24, 40, False), (128, 256, 1, 21, 35, False), (256, 512, 2, 24, 40, False)])
def test_conv_cw_stride2_equals_the_halo_kernel(dtype, C, Cout, B, H, W, exact):
    """The stride-2 convs that open DLA levels 2-5 (dla_dcn.py:84-98) on conv3x3_cw_kernel's 17 x 33-patch instantiations against
    conv3x3_wave_kernel: same bits where no wave splits K (same channel groups, same step order); the K-split variants hand other steps to
    each wave (two 64-channel steps per tap instead of a round-robin over 32-channel groups), i.e. the same products in another fp32
    summation order -- equal to rounding.  Even and odd map sizes, ragged tiles, BN + ReLU epilogue, and the generic kernel as a third opinion."""
    ops, L = _ops()
    g = _g(151)
    x = torch.randn(B, H, W, C, generator=g).to(dtype).to(DEV)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
    p = ops.pack_conv(w, dtype, (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV), stride=2, pad=1, act=L.ACT_RELU)
    lib_ = L.load()
    try:
        L.check(lib_.mfx_set_option(b"halo_cw", 0), "opt")
        want = ops.conv2d(x, p)
        L.check(lib_.mfx_set_option(b"halo_cw", 1), "opt")
        got = ops.conv2d(x, p)
        assert got.shape == (B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cout)
        if exact:
            assert torch.equal(got.view(torch.int16), want.view(torch.int16))
        else:
            assert float((got.float() - want.float()).abs().max()) <= 1.6e-2 * max(1.0, float(want.float().abs().max()))      # one 16-bit ulp of the largest value
        L.check(lib_.mfx_set_option(b"halo", 0), "opt")
        gen = ops.conv2d(x, p)
        assert float((got.float() - gen.float()).abs().max()) <= 2e-2 * max(1.0, float(gen.float().abs().max()))
    finally:
        L.check(lib_.mfx_set_option(b"halo_cw", 1), "opt")
        L.check(lib_.mfx_set_option(b"halo", 1), "opt")
