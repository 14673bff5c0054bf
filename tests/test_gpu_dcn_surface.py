"""GPU tests of the reference's DCNv2 operator surface as the reference itself exercises it (model/backbone/DCNv2/dcn_v2.py:97-128,
testcuda.py:69-97,169-180, src/dcn_v2.h:9-59): `DCN.forward(input)` in NCHW with autograd recording, deformable groups > 1 through
`_ext` and `DCN`, the train-mode DeformConv(DCN -> BN -> ReLU) block, and the error behaviour of what this build narrows."""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import DEV, _dcn_case, _g

pytestmark = pytest.mark.gpu


def _rand_offset_conv(m, seed, std_w=1.5 / 24, std_b=0.2):
    g = _g(seed)
    m.conv_offset_mask.weight.data.copy_(torch.randn(m.conv_offset_mask.weight.shape, generator=g) * std_w)
    m.conv_offset_mask.bias.data.copy_(torch.randn(m.conv_offset_mask.bias.shape, generator=g) * std_b)


def _close(a, b, tol, what):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, "%s: max abs err %.3e (scale %.3e)" % (what, err, scale)


def test_dcn_forward_nchw_is_differentiable_vs_oracle():
    """`DCN(64, 64)(input)` with requires_grad, the reference's call form (dla_dcn.py:391-396): output and all five gradients against
    the oracle module (torch offset conv + C DCNv2 forward / backward, oracle/monoflex_ref.py:50-67)."""
    from monoflex_amd.model.backbone.DCNv2.dcn_v2 import DCN
    from oracle import monoflex_ref as R
    ref = R.DCN(64, 64)
    _rand_offset_conv(ref, 3)
    ref.bias.data.normal_(0, 0.1, generator=_g(4))
    hip = DCN(64, 64, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    hip.load_state_dict(ref.state_dict())
    hip.to(DEV).train()
    x = torch.randn(2, 64, 20, 28, generator=_g(5))
    tgt = torch.randn(2, 64, 20, 28, generator=_g(6))
    xr = x.clone().requires_grad_()
    (ref(xr) * tgt).sum().backward()
    xh = x.to(DEV).requires_grad_()
    out = hip(xh)
    assert out.shape == (2, 64, 20, 28) and out.requires_grad and out.is_contiguous()
    (out * tgt.to(DEV)).sum().backward()
    with torch.no_grad():
        _close(out, ref(x), 5e-5, "output")
    _close(xh.grad, xr.grad, 2e-4, "grad_input")
    for n, p in hip.named_parameters():
        _close(p.grad, dict(ref.named_parameters())[n].grad, 3e-4, "grad " + n)
    # no graph recorded -> the fused eval kernels, same values
    with torch.no_grad():
        _close(hip.eval()(x.to(DEV)), out, 5e-5, "eval path")


def test_example_dconv_two_deformable_groups():
    """testcuda.py:169-180 `example_dconv`: DCN(64, 64, deformable_groups=2), forward + `error.backward()`; checked against the C oracle
    (which implements groups, oracle/dcn_v2_ref.c:99-124) fed by a torch offset/mask conv."""
    from monoflex_amd.model.backbone.DCNv2.dcn_v2 import DCN
    from oracle import dcn_ref
    dcn = DCN(64, 64, kernel_size=(3, 3), stride=1, padding=1, deformable_groups=2)
    _rand_offset_conv(dcn, 7)
    assert dcn.conv_offset_mask.weight.shape[0] == 54
    w_off, b_off, w, b = [t.detach().clone() for t in (dcn.conv_offset_mask.weight, dcn.conv_offset_mask.bias, dcn.weight, dcn.bias)]
    dcn.to(DEV)
    x = torch.randn(2, 64, 24, 24, generator=_g(8))
    xh = x.to(DEV).requires_grad_()
    output = dcn(xh)
    target = torch.empty_like(output).uniform_(-0.01, 0.01)
    error = (target - output).mean()
    error.backward()
    assert output.shape == (2, 64, 24, 24)

    xr = x.clone().requires_grad_()
    wr, br, wor, bor = [t.requires_grad_() for t in (w, b, w_off, b_off)]
    o = F.conv2d(xr, wor, bor, 1, 1)
    off, msk = o[:, :36], torch.sigmoid(o[:, 36:54])
    ref = dcn_ref.dcn_v2_conv(xr, off, msk, wr, br, 1, 1, 1, 2)
    (target.cpu() - ref).mean().backward()
    n = float(output.numel())                                         # (the example's loss is a MEAN: gradients carry 1 / numel)
    _close(output, ref, 5e-5, "output dg=2")
    _close(xh.grad * n, xr.grad * n, 3e-4, "grad_input dg=2")
    _close(dcn.weight.grad * n, wr.grad * n, 3e-4, "grad_weight dg=2")
    _close(dcn.conv_offset_mask.weight.grad * n, wor.grad * n, 1e-3, "grad offset-conv weight dg=2")
    _close(dcn.conv_offset_mask.bias.grad * n, bor.grad * n, 1e-3, "grad offset-conv bias dg=2")
    _close(dcn.bias.grad * n, br.grad * n, 1e-4, "grad_bias dg=2")


@pytest.mark.parametrize("dg", [2, 4])
def test_ext_forward_backward_deformable_groups_vs_oracle(dg):
    from monoflex_amd.model.backbone.DCNv2 import _ext
    from oracle import dcn_ref
    B, C, Co, H, W = 2, 32, 16, 10, 14
    g = _g(30 + dg)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, 18 * dg, H, W, generator=g) * 2
    msk = torch.sigmoid(torch.randn(B, 9 * dg, H, W, generator=g))
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    go = torch.randn(B, Co, H, W, generator=g)
    want = dcn_ref.dcn_v2_forward(x, w, b, off, msk, 3, 3, 1, 1, 1, 1, 1, 1, dg)
    got = _ext.dcn_v2_forward(*[t.to(DEV) for t in (x, w, b, off, msk)], 3, 3, 1, 1, 1, 1, 1, 1, dg)
    _close(got, want, 2e-5, "forward dg=%d" % dg)
    wantb = dcn_ref.dcn_v2_backward(x, w, b, off, msk, go, 3, 3, 1, 1, 1, 1, 1, 1, dg)
    gotb = _ext.dcn_v2_backward(*[t.to(DEV) for t in (x, w, b, off, msk, go)], 3, 3, 1, 1, 1, 1, 1, 1, dg)
    for a, r, name in zip(gotb, wantb, ["grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"]):
        assert a.shape == r.shape, name
        _close(a, r, 1e-4, "%s dg=%d" % (name, dg))


def test_check_gradient_on_the_dcn_module():
    """The reference's `check_gradient_dconv` (testcuda.py:69-97: gradcheck eps 1e-3, atol 1e-4, rtol 1e-2) applied to the MODULE's own
    forward (offset/mask conv + sigmoid + DCNv2 as one differentiable NCHW function of the input and the DCN bias; 32 -> 64 channels is
    the narrowest DCN the NHWC kernels take: K = 9 C must fill whole 128-byte k-iterations, outputs come in 64-channel tiles)."""
    from torch.autograd import gradcheck
    from monoflex_amd.model.backbone.DCNv2.dcn_v2 import DCN
    torch.manual_seed(3)
    m = DCN(32, 64, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    _rand_offset_conv(m, 11, std_w=0.05, std_b=0.3)
    m.to(DEV)
    inp = (torch.rand(1, 32, 3, 3) * 0.01).to(DEV).requires_grad_()

    def f(i, b):
        from monoflex_amd import autograd as AG
        c = m.conv_offset_mask
        x = i.permute(0, 2, 3, 1).contiguous()
        return AG.dcn_module(x, c.weight.detach(), c.bias.detach(), m.weight.detach(), b, 1, 1, 1).permute(0, 3, 1, 2)
    assert torch.allclose(f(inp, m.bias), m(inp), atol=1e-6)              # the module's forward IS this function
    b = m.bias.detach().clone().requires_grad_()
    assert gradcheck(f, (inp, b), eps=1e-3, atol=1e-4, rtol=1e-2, nondet_tol=1e-5)


def test_deformconv_block_train_mode_vs_oracle():
    """DeformConv = DCN -> BatchNorm(batch statistics) -> ReLU (dla_dcn.py:384-396) in TRAIN mode against oracle/monoflex_ref.py:
    output, input gradient, all parameter gradients and the BN running statistics of one step."""
    from monoflex_amd.model.backbone.dla_dcn import DeformConv
    from oracle import monoflex_ref as R
    torch.manual_seed(5)
    ref = R.DeformConv(64, 64).train()
    _rand_offset_conv(ref.conv, 12)
    ref.actf[0].weight.data.uniform_(0.8, 1.2); ref.actf[0].bias.data.normal_(0, 0.1)
    hip = DeformConv(64, 64)
    hip.load_state_dict(ref.state_dict())
    hip.to(DEV).train()
    x = torch.randn(2, 64, 16, 24, generator=_g(13)).relu()
    tgt = torch.randn(2, 64, 16, 24, generator=_g(14))
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    (yr * tgt).sum().backward()
    xh = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_()                   # the block's own interface is NHWC
    yh = hip(xh)
    (yh * tgt.permute(0, 2, 3, 1).to(DEV)).sum().backward()
    _close(yh.permute(0, 3, 1, 2), yr, 1e-4, "output")
    _close(xh.grad.permute(0, 3, 1, 2), xr.grad, 5e-4, "grad_input")
    rp = dict(ref.named_parameters())
    for n, p in hip.named_parameters():
        _close(p.grad, rp[n].grad, 1e-3, "grad " + n)
    _close(hip.actf[0].running_mean, ref.actf[0].running_mean, 1e-5, "running_mean")
    _close(hip.actf[0].running_var, ref.actf[0].running_var, 1e-5, "running_var")


def test_unsupported_geometry_surfaces_as_runtime_error():
    """What this build narrows relative to src/dcn_v2.h:9-23 fails the way the reference's AT_ASSERTM failures do: RuntimeError, with
    the reason, never a silent wrong answer or a CPU fallback."""
    from monoflex_amd.model.backbone.DCNv2 import _ext
    from monoflex_amd.model.backbone.DCNv2.dcn_v2 import DCN, DCNv2
    x, off, msk, w, b = [t.to(DEV) for t in _dcn_case(1, 1, 16, 16, 8, 8)]
    with pytest.raises(RuntimeError, match="kernel shape"):
        _ext.dcn_v2_forward(x, w, b, off, msk, 5, 5, 1, 1, 1, 1, 1, 1, 1)                # reference: "Input shape and kernel shape wont match"
    with pytest.raises(RuntimeError, match="kernel channels"):
        _ext.dcn_v2_forward(x[:, :8].contiguous(), w, b, off, msk, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):                                                    # (the MODULE with its own offset conv stays square; `_ext` / DCNv2 are general)
        DCN(16, 16, kernel_size=(3, 3), stride=(1, 2), padding=1).to(DEV)(x)
    m = DCNv2(16, 16, (3, 3), 1, 1).to(DEV)
    with pytest.raises(AssertionError):                                                  # dcn_v2.py:84-87
        m(x, off[:, :16], msk)
    with pytest.raises(RuntimeError, match="CPU"):
        _ext.dcn_v2_forward(x.cpu(), w.cpu(), b.cpu(), off.cpu(), msk.cpu(), 3, 3, 1, 1, 1, 1, 1, 1, 1)


@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("B,H,W,std", [(8, 96, 320, 2.0), (3, 150, 150, 5.0)])
def test_dcn_module_with_the_offset_conv_inside_the_kernel(B, H, W, std, half):
    """`ops.dcn_module` on the shapes the LDS-patch kernel takes (64 -> 64 on large 16-bit maps): the kernel runs the module's 27-channel
    offset/mask conv itself (csrc/dcn_patch.hip, OF = true).  Against the two-launch form on the same operands: the offset rows it writes
    (fp32 (B,H,W,32): offsets, sigmoid(mask), zero padding) agree to the rounding of a differently ordered fp32 sum, the outputs to what that
    offset difference moves; and the module against the oracle (torch offset conv + C DCNv2) at the bf16 / fp16 bar of the unfused path.
    Ragged tiles (150 x 150) and samples leaving the patch (sigma 5 px) included."""
    from monoflex_amd import lib as L, ops
    from monoflex_amd.model.backbone.DCNv2.dcn_v2 import DCN
    from oracle import monoflex_ref as R
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[half]
    g = _g(91)
    ref = R.DCN(64, 64)
    with torch.no_grad():
        ref.weight.copy_((torch.randn(ref.weight.shape, generator=g) * 0.05).to(dt).float())
        ref.bias.copy_(torch.randn(64, generator=g) * 0.1)
        ref.conv_offset_mask.weight.copy_((torch.randn(ref.conv_offset_mask.weight.shape, generator=g) * (0.3 / 24)).to(dt).float())
        bb = torch.randn(27, generator=g) * std
        bb[18:] = torch.randn(9, generator=g)
        ref.conv_offset_mask.bias.copy_(bb)
    m = DCN(64, 64, kernel_size=(3, 3), stride=1, padding=1)
    m.load_state_dict(ref.state_dict())
    m.to(DEV).eval()
    x = torch.randn(B, 64, H, W, generator=g).to(dt).float()
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)
    lib_ = L.load()
    p_off, p_main = m.packed_offset(dt), m.packed_main(dt)
    assert p_off.w_frag_f16 is not None
    before = None
    try:
        L.check(lib_.mfx_set_option(b"dcn_fuse_off", 0), "opt")
        y0, om0 = ops.dcn_module(xd, p_off, p_main, need_offmask=True)
        L.check(lib_.mfx_set_option(b"dcn_fuse_off", 1), "opt")
        d = ops._dcn_desc(xd, None, p_main, torch.empty_like(y0), off=p_off)
        assert lib_.mfx_dcn_fuses_offset_conv(__import__("ctypes").byref(d)) == 1
        y1, om1 = ops.dcn_module(xd, p_off, p_main, need_offmask=True)
        y2, om2 = ops.dcn_module(xd, p_off, p_main)                       # inference form: no offset map leaves the kernel
        torch.cuda.synchronize()
    finally:
        lib_.mfx_set_option(b"dcn_fuse_off", 1)
    assert om2 is None and torch.equal(y1, y2)
    assert float((om1 - om0).abs().max()) < 2e-3 * max(1.0, float(om0.abs().max())), float((om1 - om0).abs().max())
    assert float(om1[..., 27:].abs().max()) == 0.0 and bool(((om1[..., 18:27] > 0) & (om1[..., 18:27] < 1)).all())
    rel01 = float((y1.float() - y0.float()).norm() / y0.float().norm())
    assert rel01 < 5e-3, rel01
    with torch.no_grad():
        want = ref(x)
    rel = float((y1.float().permute(0, 3, 1, 2).cpu() - want).norm() / want.norm())
    rel0 = float((y0.float().permute(0, 3, 1, 2).cpu() - want).norm() / want.norm())
    assert rel < 3e-2 and rel < 1.5 * rel0 + 1e-3, (rel, rel0)


@pytest.mark.parametrize("geom", [(3, 3, 1, 2, 1, 1, 1, 1), (3, 3, 2, 1, 1, 2, 1, 1), (3, 3, 1, 1, 2, 1, 2, 1), (3, 2, 1, 1, 1, 0, 1, 1), (1, 3, 2, 1, 0, 1, 1, 2)])
@pytest.mark.parametrize("dg", [1, 2])
def test_ext_boundary_per_axis_geometry_and_groups_in_one_c_call(geom, dg):
    """`mfx_dcn_v2_forward / _backward` are as general as src/dcn_v2.h:9-23, 48-59: stride_h != stride_w, pad_h != pad_w, dil_h != dil_w,
    non-square kernels (<= 9 taps) and deformable groups -- ONE call of the C entry each (the binding passes `deformable_group` through) --
    against the C oracle (src/cpu/dcn_v2_im2col_cpu.cpp:27-329 restated), forward and all five gradients."""
    from monoflex_amd.model.backbone.DCNv2 import _ext
    from oracle import dcn_ref
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    g = _g(40 + dg)
    B, C, Cout, H, W = 2, 16 * dg, 24, 11, 14
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    kk = kh * kw
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Cout, C, kh, kw, generator=g) * 0.1
    b = torch.randn(Cout, generator=g) * 0.1
    off = torch.randn(B, 2 * dg * kk, Ho, Wo, generator=g) * 1.5
    msk = torch.sigmoid(torch.randn(B, dg * kk, Ho, Wo, generator=g))
    go = torch.randn(B, Cout, Ho, Wo, generator=g)
    want = dcn_ref.dcn_v2_forward(x, w, b, off, msk, kh, kw, sh, sw, ph, pw, dh, dw, dg)
    got = _ext.dcn_v2_forward(*[t.to(DEV) for t in (x, w, b, off, msk)], kh, kw, sh, sw, ph, pw, dh, dw, dg)
    assert got.shape == want.shape == (B, Cout, Ho, Wo)
    _close(got, want, 2e-5, "forward %s dg=%d" % (geom, dg))
    wantb = dcn_ref.dcn_v2_backward(x, w, b, off, msk, go, kh, kw, sh, sw, ph, pw, dh, dw, dg)
    gotb = _ext.dcn_v2_backward(*[t.to(DEV) for t in (x, w, b, off, msk, go)], kh, kw, sh, sw, ph, pw, dh, dw, dg)
    for a, r, name in zip(gotb, wantb, ["grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"]):
        assert a.shape == r.shape, name
        _close(a, r, 1e-4, "%s %s dg=%d" % (name, geom, dg))
