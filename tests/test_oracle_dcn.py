"""Pins oracle/dcn_v2_ref.c with the reference's own known-answer tests for DCNv2
(/root/reference/model/backbone/DCNv2/testcpu.py) and cross-checks it against an independently
written pure-torch formulation (oracle/dcn_ref.py: dcn_v2_torch)."""
import torch
from torch import nn
from torch.autograd import gradcheck

from oracle import dcn_ref


def test_zero_offset_identity():
    # testcpu.py:32-67 check_zero_offset: zero offsets, mask = sigmoid(0) = 0.5, identity weight
    # -> output == 0.5 * input ; reference tolerance 1e-10 on |input - 2*output|
    torch.manual_seed(0)
    N, C, H, W = 2, 2, 4, 4
    x = torch.randn(N, C, H, W)
    offset = torch.zeros(N, 18, H, W)
    mask = torch.sigmoid(torch.zeros(N, 9, H, W))
    weight = torch.zeros(C, C, 3, 3)
    for c in range(C):
        weight[c, c, 1, 1] = 1.0
    bias = torch.zeros(C)
    out = dcn_ref.dcn_v2_conv(x, offset, mask, weight, bias, 1, 1, 1, 1)
    assert (x - 2 * out).abs().max() < 1e-10
    out_c = dcn_ref.dcn_v2_forward(x, weight, bias, offset, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1, use_torch_gemm=False)
    assert (x - 2 * out_c).abs().max() < 1e-10


def test_zero_offset_equals_half_conv():
    # at init every DCN equals 0.5*conv3x3 + bias
    torch.manual_seed(1)
    x = torch.randn(2, 8, 12, 20)
    w = torch.randn(6, 8, 3, 3) * 0.1
    b = torch.randn(6)
    out = dcn_ref.dcn_v2_conv(x, torch.zeros(2, 18, 12, 20), torch.full((2, 9, 12, 20), 0.5), w, b, 1, 1, 1, 1)
    ref = 0.5 * torch.nn.functional.conv2d(x, w, None, 1, 1) + b.view(1, -1, 1, 1)
    assert (out - ref).abs().max() < 2e-6


def test_gradcheck_reference_tolerances():
    # testcpu.py:69-97 check_gradient_dconv: N=2, C=2->2, 4x4, offsets randn*2, fp32,
    # gradcheck(eps=1e-3, atol=1e-4, rtol=1e-2)
    torch.manual_seed(3)
    N, inC, inH, inW, outC = 2, 2, 4, 4, 2
    inp = (torch.rand(N, inC, inH, inW) * 0.01).requires_grad_()
    offset = (torch.randn(N, 18, inH, inW) * 2).requires_grad_()
    mask = torch.sigmoid(torch.rand(N, 9, inH, inW)).detach().requires_grad_()
    weight = torch.randn(outC, inC, 3, 3).requires_grad_()
    bias = torch.rand(outC).requires_grad_()
    assert gradcheck(dcn_ref.dcn_v2_conv, (inp, offset, mask, weight, bias, 1, 1, 1, 1),
                     eps=1e-3, atol=1e-4, rtol=1e-2)


def _rand_case(seed, B, C, Co, H, W, off_scale=2.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, 18, H, W, generator=g) * off_scale
    msk = torch.sigmoid(torch.randn(B, 9, H, W, generator=g))
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    return x, off, msk, w, b


def test_c_vs_torch_forward_and_grads():
    # two independent restatements of dcn_v2_im2col_cpu.cpp must agree, including samples
    # that fall outside the map (offsets with std 2 on a 12x20 map; a few at +-30)
    x, off, msk, w, b = _rand_case(5, 2, 8, 8, 12, 20)
    off[0, :, 0, 0] = 30.0
    off[1, :, 5, 7] = -30.0
    a = dcn_ref.dcn_v2_conv(x, off, msk, w, b, 1, 1, 1, 1)
    t = dcn_ref.dcn_v2_torch(x, off, msk, w, b)
    assert (a - t).abs().max() < 1e-5
    c = dcn_ref.dcn_v2_forward(x, w, b, off, msk, 3, 3, 1, 1, 1, 1, 1, 1, 1, use_torch_gemm=False)
    assert (a - c).abs().max() < 1e-5

    go = torch.randn_like(a)
    leaves = [t_.clone().requires_grad_() for t_ in (x, off, msk, w, b)]
    dcn_ref.dcn_v2_torch(*leaves).backward(go)
    grads_c = dcn_ref.dcn_v2_backward(x, w, b, off, msk, go, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    # C order: grad_input, grad_offset, grad_mask, grad_weight, grad_bias
    for gc, leaf, name in zip(grads_c, [leaves[0], leaves[1], leaves[2], leaves[3], leaves[4]],
                              ["input", "offset", "mask", "weight", "bias"]):
        scale = leaf.grad.abs().max().clamp(min=1.0)
        assert ((gc - leaf.grad).abs().max() / scale) < 2e-5, name


def test_c_oracle_vs_library_bilinear_sampler():
    """the fractional-offset arithmetic of oracle/dcn_v2_ref.c (forward, and all five gradients) against a formulation whose sampler
    is PyTorch's own grid_sample in float64 -- offsets with std 2 on a 12 x 20 map, so samples cross the border band (-1, 0) / (H - 1, H) where the
    reference guards corner by corner (dcn_v2_im2col_cpu.cpp:27-55, 178-189), and a few at +-30 that miss the map."""
    x, off, msk, w, b = _rand_case(11, 2, 8, 8, 12, 20)
    off[0, :, 0, 0] = 30.0
    off[1, :, 5, 7] = -30.0
    a = dcn_ref.dcn_v2_conv(x, off, msk, w, b, 1, 1, 1, 1)
    leaves = [t_.double().clone().requires_grad_() for t_ in (x, off, msk, w, b)]
    t = dcn_ref.dcn_v2_grid_sample(*leaves)
    assert (a.double() - t).abs().max() < 2e-5
    band = ((torch.arange(12.).view(1, 1, 12, 1) - 1 + off[:, 0:1]) < 0) & ((torch.arange(12.).view(1, 1, 12, 1) - 1 + off[:, 0:1]) > -1)
    assert int(band.sum()) > 5                                    # the case does exercise the guarded band
    go = torch.randn(a.shape, generator=torch.Generator().manual_seed(3))
    t.backward(go.double())
    grads_c = dcn_ref.dcn_v2_backward(x, w, b, off, msk, go, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    for gc, leaf, name in zip(grads_c, leaves, ["input", "offset", "mask", "weight", "bias"]):
        scale = leaf.grad.abs().max().clamp(min=1.0)
        assert ((gc.double() - leaf.grad).abs().max() / scale) < 5e-5, name


def test_module_split_convention():
    # dcn_v2.py:118-128: offset = first 18 of the 27 conv outputs, mask = sigmoid(last 9)
    from oracle.monoflex_ref import DCN
    torch.manual_seed(0)
    m = DCN(4, 5)
    nn.init.normal_(m.conv_offset_mask.weight, std=0.2)
    x = torch.randn(1, 4, 6, 7)
    out27 = m.conv_offset_mask(x)
    o1, o2, mk = torch.chunk(out27, 3, dim=1)
    ref = dcn_ref.dcn_v2_torch(x, torch.cat((o1, o2), 1), torch.sigmoid(mk), m.weight, m.bias)
    assert (m(x) - ref).abs().max() < 1e-5
