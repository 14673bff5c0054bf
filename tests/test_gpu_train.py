"""GPU gradient parity of the training path: every differentiable HIP operator against torch CPU autograd of the
reference's layer (nn.Conv2d / BatchNorm2d / MaxPool2d / ConvTranspose2d / the oracle's DCNv2 restatement), then the
whole network (train-mode BN) against the CPU oracle under a fixed linear surrogate loss.  fp32; tolerances are
relative to the largest gradient entry of each tensor."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}      # activation dtypes of the training path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-20))


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("cin,cout,k,stride,bias", [(16, 32, 3, 1, False), (32, 64, 3, 2, False), (64, 128, 1, 1, False),
                                                    (64, 27, 3, 1, True), (256, 3, 1, 1, True), (256, 20, 1, 1, True)])
def test_conv_grads(cin, cout, k, stride, bias):
    from monoflex_amd import autograd as AG
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, cin, 20, 36, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.1
    b = torch.randn(cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=k // 2)
    r = torch.randn(yr.shape, generator=g)
    (yr * r).sum().backward()
    xd = _nhwc(x).to(DEV).requires_grad_()
    wd = w.to(DEV).requires_grad_()
    bd = b.to(DEV).requires_grad_() if bias else None
    yd = AG.conv2d(xd, wd, bd, stride, k // 2)
    assert _rel(yd.permute(0, 3, 1, 2), yr) < 1e-5
    (yd * _nhwc(r).to(DEV)).sum().backward()
    assert _rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < 1e-5
    assert _rel(wd.grad, wr.grad) < 1e-5
    if bias:
        assert _rel(bd.grad, br.grad) < 1e-5


def test_conv1d_as_conv_grads():
    """Edge-fusion Conv1d k=3 over an explicitly replicate-padded sequence (pad 0 in the kernel)."""
    from monoflex_amd import autograd as AG
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 64, 1, 50, generator=g)
    w = torch.randn(64, 64, 1, 3, generator=g) * 0.1
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = F.conv2d(xr, wr)
    r = torch.randn(yr.shape, generator=g)
    (yr * r).sum().backward()
    xd, wd = _nhwc(x).to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    yd = AG.conv2d(xd, wd, None, 1, 0)
    (yd * _nhwc(r).to(DEV)).sum().backward()
    assert _rel(yd.permute(0, 3, 1, 2), yr) < 1e-5
    assert _rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < 1e-5 and _rel(wd.grad, wr.grad) < 1e-5


def test_stem_conv_wgrad():
    from monoflex_amd import autograd as AG
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 32, 64, generator=g)
    w = (torch.randn(16, 3, 7, 7, generator=g) * 0.1)
    wr = w.clone().requires_grad_()
    yr = F.conv2d(x, wr, padding=3)
    r = torch.randn(yr.shape, generator=g)
    (yr * r).sum().backward()
    wd = w.to(DEV).requires_grad_()
    yd = AG.StemConvFn.apply(x.to(DEV), wd, torch.float32)
    (yd * _nhwc(r).to(DEV)).sum().backward()
    assert _rel(yd.permute(0, 3, 1, 2), yr) < 1e-5 and _rel(wd.grad, wr.grad) < 1e-5


@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("B,H,W", [(2, 32, 64), (3, 37, 70), (1, 8, 32), (2, 96, 320)])
def test_stem_conv_wgrad_bf16_toeplitz_kernel(B, H, W, half):
    """mfx_stem_wgrad_bf16 (flat image rows in LDS, Toeplitz operand through transposed reads) against the generic dilated-tap
    kernel on the same bf16 operands and against torch autograd of F.conv2d; ragged tiles included."""
    from monoflex_amd import autograd as AG
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 3, H, W, generator=g).to(DT[half]).float()
    w = (torch.randn(16, 3, 7, 7, generator=g) * 0.1)
    wr = w.clone().requires_grad_()
    yr = F.conv2d(x, wr, padding=3)
    r = torch.randn(yr.shape, generator=g).to(DT[half]).float()
    (yr * r).sum().backward()
    got = {}
    for generic in (False, True):
        AG._STEM_WGRAD_GENERIC[0] = generic
        try:
            wd = w.to(DEV).requires_grad_()
            yd = AG.StemConvFn.apply(x.to(DEV), wd, DT[half])
            (yd.float() * _nhwc(r).to(DEV)).sum().backward()
            got[generic] = wd.grad.detach().cpu()
        finally:
            AG._STEM_WGRAD_GENERIC[0] = False
    assert _rel(got[False], got[True]) < 2e-3, _rel(got[False], got[True])
    assert _rel(got[False], wr.grad) < 1.5e-2, _rel(got[False], wr.grad)


@pytest.mark.parametrize("act,res", [("relu", False), ("relu", True), ("leaky", False), ("none", False)])
def test_bn_act_grads(act, res):
    from monoflex_amd import autograd as AG
    from monoflex_amd import lib as L
    g = torch.Generator().manual_seed(4)
    C = 64
    x = torch.randn(3, C, 12, 20, generator=g) * 2 + 0.5
    rs = torch.randn(3, C, 12, 20, generator=g) if res else None
    bn_r, bn_d = torch.nn.BatchNorm2d(C, momentum=0.1), torch.nn.BatchNorm2d(C, momentum=0.1).to(DEV)
    with torch.no_grad():
        bn_r.weight.copy_(torch.rand(C, generator=g) + 0.5); bn_r.bias.copy_(torch.randn(C, generator=g))
    bn_d.load_state_dict(bn_r.state_dict())
    xr = x.clone().requires_grad_()
    rr = rs.clone().requires_grad_() if res else None
    t = bn_r(xr)
    if res:
        t = t + rr
    yr = {"relu": F.relu, "leaky": lambda v: F.leaky_relu(v, 0.01), "none": lambda v: v}[act](t)
    r = torch.randn(yr.shape, generator=g)
    (yr * r).sum().backward()
    xd = _nhwc(x).to(DEV).requires_grad_()
    rd = _nhwc(rs).to(DEV).requires_grad_() if res else None
    yd = AG.bn_act(xd, bn_d, {"relu": L.ACT_RELU, "leaky": L.ACT_LEAKY, "none": L.ACT_NONE}[act], rd)
    (yd * _nhwc(r).to(DEV)).sum().backward()
    assert _rel(yd.permute(0, 3, 1, 2), yr) < 1e-5
    assert _rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < 1e-4
    assert _rel(bn_d.weight.grad, bn_r.weight.grad) < 1e-4 and _rel(bn_d.bias.grad, bn_r.bias.grad) < 1e-4
    assert _rel(bn_d.running_mean, bn_r.running_mean) < 1e-5 and _rel(bn_d.running_var, bn_r.running_var) < 1e-5
    if res:
        assert _rel(rd.grad.permute(0, 3, 1, 2), rr.grad) < 1e-5


@pytest.mark.parametrize("dt,C,extra", [("fp32", 64, False), ("fp32", 128, True), ("bf16", 64, True), ("bf16", 256, False)])
def test_basic_block_identity_residual_through_the_conv_node(dt, C, extra):
    """A BasicBlock with the identity residual (dla_dcn.py:84-98, tree2 of every Tree): x feeds conv1 and the add behind bn2.  conv1's node returns x as
    a second output, so the residual's gradient reaches that node and is added by the data-gradient conv's epilogue (no autograd add).  Same
    forward, same gradients as torch's block, and as the two-consumer form (MFX_RESIDUAL_ALIAS=0); `extra` gives x a third consumer that autograd
    still adds."""
    from monoflex_amd import autograd as AG
    from monoflex_amd.model.backbone.dla_dcn import BasicBlock
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, C, 12, 20, generator=g)
    blk = BasicBlock(C, C).train()
    with torch.no_grad():
        for bn in (blk.bn1, blk.bn2):
            bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
    ref = {k: v.clone() for k, v in blk.state_dict().items()}

    def torch_block(xr):
        c1 = F.conv2d(xr, pr["conv1.weight"], None, 1, 1)
        b1 = F.relu(F.batch_norm(c1, None, None, pr["bn1.weight"], pr["bn1.bias"], True, 0.1, 1e-5))
        c2 = F.conv2d(b1, pr["conv2.weight"], None, 1, 1)
        return F.relu(F.batch_norm(c2, None, None, pr["bn2.weight"], pr["bn2.bias"], True, 0.1, 1e-5) + xr)
    pr = {k: v.clone().requires_grad_() for k, v in ref.items() if v.is_floating_point() and "running" not in k}
    xr = x.clone().requires_grad_()
    yr = torch_block(xr)
    r = torch.randn(yr.shape, generator=g)
    ((yr * r).sum() + ((xr * xr).sum() if extra else 0.0)).backward()

    got = {}
    for alias in (True, False):
        AG.RESIDUAL_ALIAS[0] = alias
        try:
            b = BasicBlock(C, C).to(DEV).train()
            b.load_state_dict(ref)
            xd = _nhwc(x).to(DEV).to(DT[dt]).requires_grad_()
            yd = b(xd)
            ((yd.float() * _nhwc(r).to(DEV)).sum() + ((xd.float() * xd.float()).sum() if extra else 0.0)).backward()
            got[alias] = (yd.detach().float().permute(0, 3, 1, 2), xd.grad.float().permute(0, 3, 1, 2), b.conv1.weight.grad.clone(), b.conv2.weight.grad.clone(),
                          b.bn2.weight.grad.clone())
        finally:
            AG.RESIDUAL_ALIAS[0] = True
    tol = 2e-4 if dt == "fp32" else 3e-2
    names = ("y", "dx", "dconv1", "dconv2", "dgamma2")
    errs = {al: [_rel(a, w) for a, w in zip(got[al], (yr, xr.grad, pr["conv1.weight"].grad, pr["conv2.weight"].grad, pr["bn2.weight"].grad))] for al in (True, False)}
    for n, e_alias, e_two in zip(names, errs[True], errs[False]):
        # against torch's fp32 block: inside the mode's tolerance, and never worse than the two-consumer form by more than rounding
        assert e_alias < max(tol, 1.5 * e_two), (n, e_alias, e_two)
    assert _rel(got[True][0], got[False][0]) < (1e-5 if dt == "fp32" else 1e-2)      # the forward is the same launches (statistics summed by atomics: not bitwise)
    for a, o in zip(got[True][1:], got[False][1:]):
        assert _rel(a, o) < (1e-5 if dt == "fp32" else 2e-2)


@pytest.mark.parametrize("C,shape,dt", [(16, (2, 40, 64), "fp32"), (64, (3, 12, 20), "bf16"), (512, (2, 6, 10), "fp32"), (128, (8, 48, 160), "bf16"),
                                         (64, (3, 12, 20), "fp16"), (128, (8, 48, 160), "fp16")])
def test_bn_two_launch_form_reuses_its_scratch(C, shape, dt):
    """mfx_bn_train_fwd / mfx_bn_train_bwd keep their sums in a persistent scratch that every call must leave zero: repeated
    forwards, a forward without a backward, and two backwards through one forward all agree with the separate-kernel form
    (stats / finalize / apply with freshly zeroed buffers), and num_batches_tracked counts in the kernel."""
    from monoflex_amd import autograd as AG
    from monoflex_amd import lib as L
    dtype = DT[dt]
    g = torch.Generator().manual_seed(C)
    B, H, W = shape
    bn_a, bn_b = torch.nn.BatchNorm2d(C).to(DEV), torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn_a.weight.copy_(torch.rand(C, generator=g) + 0.5); bn_a.bias.copy_(torch.randn(C, generator=g))
    bn_b.load_state_dict(bn_a.state_dict())
    tol = 2e-2 if dt != "fp32" else 1e-4
    for it in range(3):
        x = (torch.randn(B, H, W, C, generator=g) * (1 + it) + 0.3 * it).to(DEV).to(dtype)
        r = torch.randn(B, H, W, C, generator=g).to(DEV).to(dtype)
        outs = []
        for bn, separate in ((bn_a, False), (bn_b, True)):
            AG._BN_SEPARATE[0] = separate
            try:
                xd = x.clone().requires_grad_()
                if it == 1:
                    with torch.no_grad():
                        AG.bn_act(xd, bn, L.ACT_RELU)                      # a forward that never sees a backward
                y = AG.bn_act(xd, bn, L.ACT_RELU)
                g1 = torch.autograd.grad(y, xd, r, retain_graph=True)[0]
                bn.zero_grad()
                (y.float() * r.float()).sum().backward()                   # second backward through the same forward
                outs.append((y, g1, xd.grad, bn.weight.grad.clone(), bn.bias.grad.clone()))
            finally:
                AG._BN_SEPARATE[0] = False
        for i, (a, b) in enumerate(zip(*outs)):
            if dt != "fp32" and i < 3:
                # 16-bit inputs sit on a grid, and when a grid value lands within rounding of a channel's ReLU threshold, the few dozen
                # elements holding it switch sides with the last bits of the batch statistics (atomics order differs between the forms):
                # a handful of whole gradient entries, not an error of the kernels -- bound the share of such elements instead of the max
                d = (a.float() - b.float()).abs()
                assert float((d > tol * b.float().abs().max()).float().mean()) < 2e-5, (it, i, _rel(a.float(), b.float()))
                continue
            assert _rel(a.float(), b.float()) < tol, (it, i, _rel(a.float(), b.float()))
        assert _rel(outs[0][1].float(), outs[0][2].float()) < (1e-2 if dt != "fp32" else 1e-4)             # the two backwards of the fused form agree with each other
    assert _rel(bn_a.running_mean, bn_b.running_mean) < 1e-5 and _rel(bn_a.running_var, bn_b.running_var) < 1e-5
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 4
    assert float(AG._bn_scratch(bn_a.weight).abs().max()) == 0.0


@pytest.mark.parametrize("C,shape,dt,act,res", [
    (512, (8, 12, 40), "bf16", "relu", False),        # DLA level 5 (2 slots per thread)
    (256, (8, 24, 80), "bf16", "relu", True),         # level 4, residual (4 slots)
    (128, (8, 48, 160), "bf16", "relu", False),       # level 3 (8 slots)
    (64, (8, 96, 320), "bf16", "relu", False),        # level 2 / the DCN modules (16 slots)
    (64, (8, 96, 320), "bf16", "relu", True),         # forward one pass (16 slots x 2 operands), backward falls back (3 operands x 16 slots)
    (64, (8, 96, 320), "fp16", "leaky", False),
    (128, (3, 13, 17), "bf16", "none", False),        # ragged: the last slot of most threads is empty
    (16, (2, 40, 64), "fp32", "relu", True),
    (32, (1, 7, 9), "fp16", "leaky", True),           # fewer chunks than one workgroup has threads
    (32, (8, 192, 640), "bf16", "relu", False),       # 63 MB: does not fit one resident grid -> the two-launch form on both sides
])
def test_bn_one_pass_kernels_equal_the_two_launch_form(C, shape, dt, act, res):
    """r06: statistics + element-wise pass in one launch with a grid barrier (csrc/train_kernels.hip bn_fwd_onepass_kernel /
    bn_bwd_onepass_kernel).  Same expressions as the two-launch form, only the summation order of the column sums differs: outputs,
    every gradient and the running statistics agree to rounding; the scratch (sums, tickets, barrier words, stuck flag) is zero
    after every call; three repetitions re-use it."""
    from monoflex_amd import autograd as AG
    from monoflex_amd import lib as L
    dtype = DT[dt]
    code = {"relu": L.ACT_RELU, "leaky": L.ACT_LEAKY, "none": L.ACT_NONE}[act]
    g = torch.Generator().manual_seed(C + shape[1])
    B, H, W = shape
    bn_a, bn_b = torch.nn.BatchNorm2d(C).to(DEV), torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn_a.weight.copy_(torch.rand(C, generator=g) + 0.5); bn_a.bias.copy_(torch.randn(C, generator=g) * 0.5)
    bn_b.load_state_dict(bn_a.state_dict())
    tol = 2e-2 if dt != "fp32" else 1e-4
    for it in range(3):
        x = (torch.randn(B, H, W, C, generator=g) * (1 + 0.5 * it) + 0.3 * it).to(DEV).to(dtype)
        rs = torch.randn(B, H, W, C, generator=g).to(DEV).to(dtype) if res else None
        r = torch.randn(B, H, W, C, generator=g).to(DEV).to(dtype)
        outs = []
        for bn, onepass in ((bn_a, 3), (bn_b, 0)):             # 3 = backward and forward; every size (the defaults keep small maps and the forward on two launches)
            L.check(L.load().mfx_set_option(b"bn_onepass", onepass), "set_option")
            L.check(L.load().mfx_set_option(b"bn_onepass_min_chunks", 0), "set_option")
            try:
                xd = x.clone().requires_grad_()
                rd = rs.clone().requires_grad_() if res else None
                bn.zero_grad()
                y = AG.bn_act(xd, bn, code, rd)
                y.backward(r)
                outs.append((y.detach(), xd.grad, rd.grad if res else xd.grad, bn.weight.grad.clone(), bn.bias.grad.clone()))
            finally:
                L.check(L.load().mfx_reset_options(), "reset_options")
        for i, (a, b) in enumerate(zip(*outs)):
            if dt != "fp32" and i < 3:         # (see test_bn_two_launch_form_reuses_its_scratch: values on the activation threshold may switch sides)
                d = (a.float() - b.float()).abs()
                assert float((d > tol * b.float().abs().max()).float().mean()) < 2e-5, (it, i, _rel(a.float(), b.float()))
                continue
            assert _rel(a.float(), b.float()) < (tol if i < 3 else (2e-3 if dt != "fp32" else 2e-4)), (it, i, _rel(a.float(), b.float()))
        assert float(AG._bn_scratch(bn_a.weight).abs().max()) == 0.0, "the one-pass kernels must leave sums, tickets and barrier words zero"
    assert _rel(bn_a.running_mean, bn_b.running_mean) < 1e-5 and _rel(bn_a.running_var, bn_b.running_var) < 1e-5
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 3
    # and against torch (fp32 arithmetic on the same 16-bit inputs)
    bn_t = torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn_t.weight.copy_(bn_a.weight); bn_t.bias.copy_(bn_a.bias)
    xt = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_()
    t = bn_t(xt)
    if res:
        t = t + rs.float().permute(0, 3, 1, 2)
    yt = {"relu": F.relu, "leaky": lambda v: F.leaky_relu(v, 0.01), "none": lambda v: v}[act](t)
    yt.backward(r.float().permute(0, 3, 1, 2))
    ya, dxa = outs[0][0], outs[0][1]
    d = (dxa.float().permute(0, 3, 1, 2) - xt.grad).abs()
    assert float((d > tol * xt.grad.abs().max()).float().mean()) < 1e-4
    assert _rel(ya.float().permute(0, 3, 1, 2), yt.detach()) < (1e-2 if dt != "fp32" else 1e-5)
    assert _rel(outs[0][3], bn_t.weight.grad) < 5e-3 and _rel(outs[0][4], bn_t.bias.grad) < 5e-3


@pytest.mark.parametrize("act", ["relu", "leaky"])
@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
def test_bn_backward_recomputes_the_activation_sign_from_its_input(act, dt):
    """Without a residual the BN backward does not read the forward output: the sign of x*scale + shift is recomputed with the
    forward's expression.  Same gradients as the form that reads the stored output (bitwise for the data gradient's ReLU mask)."""
    from monoflex_amd import autograd as AG
    from monoflex_amd import lib as L
    dtype = DT[dt]
    g = torch.Generator().manual_seed(11)
    C = 128
    x = (torch.randn(4, 24, 40, C, generator=g) * 1.5 + 0.2).to(DEV).to(dtype)
    r = torch.randn(4, 24, 40, C, generator=g).to(DEV).to(dtype)
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.5)
    code = {"relu": L.ACT_RELU, "leaky": L.ACT_LEAKY}[act]
    out = []
    for read_output in (False, True):
        AG._BN_READ_OUTPUT[0] = read_output
        try:
            xd = x.clone().requires_grad_()
            bn.zero_grad()
            y = AG.bn_act(xd, bn, code)
            y.backward(r)
            out.append((xd.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), y.detach().clone()))
        finally:
            AG._BN_READ_OUTPUT[0] = False
    (dx0, dg0, db0, y0), (dx1, dg1, db1, y1) = out
    tol = 1e-2 if dt != "fp32" else 2e-5                               # two runs differ by the summation order of the statistics' atomics
    assert _rel(y0.float(), y1.float()) < tol
    assert _rel(dg0, dg1) < 1e-4 and _rel(db0, db1) < 1e-4
    assert _rel(dx0.float(), dx1.float()) < tol


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
def test_sparse_regression_heads_function_vs_torch(dt):
    """SparseRegHeadsFn (csrc/head_sparse.hip) against torch: train-mode BN over the dense trunk map + leaky(0.01) + 1x1 heads,
    gathered at the object centres (duplicate centres, empty slots), output rows and every gradient (trunk map, ABN weight /
    bias, 1x1 weights / biases) for a random upstream gradient."""
    from monoflex_amd import autograd as AG
    from monoflex_amd.model.head.detector_predictor import InPlaceABN
    dtype = DT[dt]
    g = torch.Generator().manual_seed(23)
    B, H, W, C, N = 2, 12, 20, 256, 16
    ks, offs = (4, 20, 3), (0, 6, 26)
    rows = torch.zeros(N, 72)
    rows[:, 0] = (torch.rand(N, generator=g) > 0.25).float()
    rows[:, 57] = torch.randint(0, B, (N,), generator=g).float()
    rows[:, 2] = torch.randint(0, W, (N,), generator=g).float()
    rows[:, 3] = torch.randint(0, H, (N,), generator=g).float()
    rows[3] = rows[5]; rows[3, 0] = rows[5, 0] = 1.0                      # two objects on one pixel
    dout = torch.randn(N, 50, generator=g)
    ys = [(torch.randn(B, H, W, C, generator=g) * 1.3 + 0.2).to(dtype) for _ in ks]
    abns_r, w2s, b2s = [], [], []
    for k in ks:
        m = torch.nn.BatchNorm2d(C)
        with torch.no_grad():
            m.weight.copy_(torch.rand(C, generator=g) + 0.5); m.bias.copy_(torch.randn(C, generator=g) * 0.3)
        abns_r.append(m)
        w2s.append(torch.randn(k, C, 1, 1, generator=g) * 0.1)
        b2s.append(torch.randn(k, generator=g) * 0.1)
    # torch reference (fp32 on the rounded inputs)
    ref_in = [y.float().clone().requires_grad_() for y in ys]
    ref_w = [w.clone().requires_grad_() for w in w2s]
    ref_b = [b.clone().requires_grad_() for b in b2s]
    bi, cy, cx, valid = rows[:, 57].long(), rows[:, 3].long(), rows[:, 2].long(), rows[:, 0]
    tot = 0
    outs_ref = []
    for i, k in enumerate(ks):
        a = F.leaky_relu(abns_r[i](ref_in[i].permute(0, 3, 1, 2)), 0.01)
        o = F.conv2d(a, ref_w[i], ref_b[i]).permute(0, 2, 3, 1)[bi, cy, cx] * valid[:, None]
        outs_ref.append(o)
        tot = tot + (o * dout[:, offs[i]:offs[i] + k]).sum()
    tot.backward()
    # device
    abns_d = []
    for m in abns_r:
        h = InPlaceABN(C)
        h.load_state_dict({k: v for k, v in torch.nn.BatchNorm2d(C).state_dict().items()})
        with torch.no_grad():
            h.weight.copy_(m.weight); h.bias.copy_(m.bias)
        abns_d.append(h.to(DEV))
    yd = [y.to(DEV).requires_grad_() for y in ys]
    wd = [w.to(DEV).requires_grad_() for w in w2s]
    bd = [b.to(DEV).requires_grad_() for b in b2s]
    out = AG.SparseRegHeadsFn.apply(rows.to(DEV), tuple(abns_d), offs, 50, (False,) * len(ks), *yd, *[h.weight for h in abns_d], *[h.bias for h in abns_d], *wd, *bd)
    (out * dout.to(DEV)).sum().backward()
    tol = 3e-2 if dt != "fp32" else 2e-4
    for i, k in enumerate(ks):
        assert _rel(out[:, offs[i]:offs[i] + k].cpu(), outs_ref[i].detach()) < tol, i
        assert _rel(yd[i].grad.float().cpu(), ref_in[i].grad) < tol, (i, _rel(yd[i].grad.float().cpu(), ref_in[i].grad))
        assert _rel(abns_d[i].weight.grad.cpu(), abns_r[i].weight.grad) < tol and _rel(abns_d[i].bias.grad.cpu(), abns_r[i].bias.grad) < tol, i
        assert _rel(wd[i].grad.cpu(), ref_w[i].grad) < tol and _rel(bd[i].grad.cpu(), ref_b[i].grad) < tol, i
        assert _rel(abns_d[i].running_var.cpu(), abns_r[i].running_var) < 1e-3 and int(abns_d[i].num_batches_tracked) == 1
    unused = [c for c in range(50) if not any(o <= c < o + k for o, k in zip(offs, ks))]
    assert float(out[:, unused].abs().max()) == 0.0 and float(out[rows[:, 0] == 0].abs().max()) == 0.0


@pytest.mark.parametrize("offset", [False, True])
@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
def test_gram_regression_heads_vs_torch(dt, offset):
    """GramRegHeadsFn (monoflex_amd/gram_heads.py): conv3x3(64 -> 256) -> train-mode BN -> leaky(0.01) -> 1x1 heads at the object centres, with the
    BN statistics and their gradients taken from the input's patch Gram matrix instead of dense trunk maps -- against torch running the dense
    layers on the same (rounded) inputs: output rows, running statistics and EVERY gradient (feature map, trunk weights, ABN weight / bias, 1x1
    weights / biases).  Objects on the image border and corners, two objects on one pixel, neighbouring objects, empty slots."""
    from monoflex_amd.gram_heads import gram_reg_heads
    from monoflex_amd.model.head.detector_predictor import InPlaceABN
    dtype = DT[dt]
    g = torch.Generator().manual_seed(29)
    B, H, W, Cin, C, N = 2, 12, 20, 64, 256, 20
    ks, offs = (4, 20, 3), (0, 6, 26)
    rows = torch.zeros(N, 72)
    rows[:, 0] = (torch.rand(N, generator=g) > 0.2).float()
    rows[:, 57] = torch.randint(0, B, (N,), generator=g).float()
    rows[:, 2] = torch.randint(0, W, (N,), generator=g).float()
    rows[:, 3] = torch.randint(0, H, (N,), generator=g).float()
    rows[0, 2:4] = torch.tensor([0.0, 0.0]); rows[1, 2:4] = torch.tensor([W - 1.0, H - 1.0]); rows[2, 2:4] = torch.tensor([0.0, 5.0])
    rows[3] = rows[5]; rows[6] = rows[5]; rows[6, 2] = rows[5, 2] + (1.0 if rows[5, 2] < W - 1 else -1.0)       # same pixel; a neighbour
    rows[[0, 1, 2, 3, 5, 6], 0] = 1.0
    dout = torch.randn(N, 50, generator=g)
    x = (torch.randn(B, H, W, Cin, generator=g) * 0.8 + 0.1).to(dtype)
    wt = [torch.randn(C, Cin, 3, 3, generator=g) / 24.0 for _ in ks]
    if offset:
        # |mean| >> std in the trunk pre-activation (mean ~ 10, std ~ 0.8: mean^2 / var ~ 150): the Gram path's variance is E[y^2] - E[y]^2 from fp32 sums
        # finalised in double, i.e. its relative error is eps * (1 + mean^2 / var) -- this case would show a cancellation problem
        x = (x.float() + 1.5).to(dtype)
        wt = [w + 0.01 for w in wt]
    abns_r, w2s, b2s = [], [], []
    for k in ks:
        m = torch.nn.BatchNorm2d(C)
        with torch.no_grad():
            m.weight.copy_(torch.rand(C, generator=g) + 0.5); m.bias.copy_(torch.randn(C, generator=g) * 0.3)
        abns_r.append(m)
        w2s.append(torch.randn(k, C, 1, 1, generator=g) * 0.1)
        b2s.append(torch.randn(k, generator=g) * 0.1)
    rnd = (lambda t: t.to(dtype).float()) if dt != "fp32" else (lambda t: t)
    # torch reference: the dense layers in fp64 on the inputs the device path sees
    xr = x.double().permute(0, 3, 1, 2).clone().requires_grad_()
    ref_wt = [rnd(w).double().clone().requires_grad_() for w in wt]
    ref_w = [w.double().clone().requires_grad_() for w in w2s]
    ref_b = [b.double().clone().requires_grad_() for b in b2s]
    bi, cy, cx, valid = rows[:, 57].long(), rows[:, 3].long(), rows[:, 2].long(), rows[:, 0].double()
    tot, outs_ref = 0, []
    for i, k in enumerate(ks):
        bn = abns_r[i].double()
        a = F.leaky_relu(bn(F.conv2d(xr, ref_wt[i], None, 1, 1)), 0.01)
        o = F.conv2d(a, ref_w[i], ref_b[i]).permute(0, 2, 3, 1)[bi, cy, cx] * valid[:, None]
        outs_ref.append(o)
        tot = tot + (o * dout[:, offs[i]:offs[i] + k].double()).sum()
    tot.backward()
    # device
    abns_d = []
    for m in abns_r:
        h = InPlaceABN(C)
        h.load_state_dict({k_: v for k_, v in torch.nn.BatchNorm2d(C).state_dict().items()})
        with torch.no_grad():
            h.weight.copy_(m.weight.float()); h.bias.copy_(m.bias.float())
        abns_d.append(h.to(DEV))
    xd = x.to(DEV).requires_grad_()
    wtd = [w.to(DEV).requires_grad_() for w in wt]
    wd = [w.to(DEV).requires_grad_() for w in w2s]
    bd = [b.to(DEV).requires_grad_() for b in b2s]
    out, _ = gram_reg_heads(xd, rows.to(DEV), abns_d, offs, 50, wtd, [h.weight for h in abns_d], [h.bias for h in abns_d], wd, bd, sync=False)
    (out * dout.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    tol = 2e-2 if dt != "fp32" else 2e-3
    dx_ref = xr.grad.permute(0, 2, 3, 1).float()
    assert _rel(xd.grad.float().cpu(), dx_ref) < tol, ("dx", _rel(xd.grad.float().cpu(), dx_ref))
    for i, k in enumerate(ks):
        assert _rel(out[:, offs[i]:offs[i] + k].cpu(), outs_ref[i].detach().float()) < tol, ("out", i)
        assert _rel(wtd[i].grad.cpu(), ref_wt[i].grad.float()) < tol, ("dW", i, _rel(wtd[i].grad.cpu(), ref_wt[i].grad.float()))
        assert _rel(abns_d[i].weight.grad.cpu(), abns_r[i].weight.grad.float()) < tol and _rel(abns_d[i].bias.grad.cpu(), abns_r[i].bias.grad.float()) < tol, i
        assert _rel(wd[i].grad.cpu(), ref_w[i].grad.float()) < tol and _rel(bd[i].grad.cpu(), ref_b[i].grad.float()) < tol, i
        assert _rel(abns_d[i].running_var.cpu(), abns_r[i].running_var.float()) < 1e-3 and _rel(abns_d[i].running_mean.cpu(), abns_r[i].running_mean.float()) < 1e-3
        assert int(abns_d[i].num_batches_tracked) == 1
    unused = [c for c in range(50) if not any(o <= c < o + k for o, k in zip(offs, ks))]
    assert float(out[:, unused].abs().max()) == 0.0 and float(out[rows[:, 0] == 0].abs().max()) == 0.0


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("cin,cout,H,W,k,stride", [(64, 64, 24, 40, 3, 1), (64, 256, 40, 72, 3, 1), (256, 64, 13, 37, 3, 1), (128, 128, 9, 20, 3, 1),
                                                   (16, 16, 32, 64, 3, 1), (32, 32, 31, 45, 3, 1), (512, 512, 6, 10, 3, 1), (64, 128, 24, 40, 3, 2),
                                                   (64, 128, 24, 40, 1, 1)])
def test_conv_epilogue_accumulates_the_bn_statistics(cin, cout, H, W, k, stride, dt):
    """The LDS-halo 3x3 kernels add the sums / sums of squares of their (rounded) output to the following BN's scratch
    (mfx_conv_desc.stats), so the BN skips its statistics pass: same mean / rstd / output / gradients as the separate pass, for every
    halo variant the table picks (plain, wide, K-split, ragged tiles); convs other kernels run (stride 2, 1x1) report
    stats_done = 0 and the BN does its own pass."""
    from monoflex_amd import autograd as AG, ops
    from monoflex_amd import lib as L
    dtype = DT[dt]
    g = torch.Generator().manual_seed(cin + cout + H)
    B = 3
    x = torch.randn(B, H, W, cin, generator=g).to(DEV).to(dtype)
    w = (torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)).to(DEV)
    bn_a, bn_b = torch.nn.BatchNorm2d(cout).to(DEV), torch.nn.BatchNorm2d(cout).to(DEV)
    with torch.no_grad():
        bn_a.weight.copy_(torch.rand(cout, generator=g) + 0.5); bn_a.bias.copy_(torch.randn(cout, generator=g) * 0.2)
    bn_b.load_state_dict(bn_a.state_dict())
    out = []
    AG._CONV_STATS_MAX_COUT[0] = 512                           # (the model fuses up to 128 output channels; the kernels support all)
    for bn, off in ((bn_a, False), (bn_b, True)):
        AG._CONV_STATS_OFF[0] = off
        try:
            xd, wd = x.clone().requires_grad_(), w.clone().requires_grad_()
            y, done = AG.conv2d_bn_stats(xd, wd, None, stride, k // 2, bn)
            assert done == ((not off) and k == 3), done               # (the LDS-halo kernel takes stride 2 since r04)
            z = AG.bn_act(y, bn, L.ACT_RELU, stats_done=done)
            z.float().square().sum().backward()
            out.append((z.detach().float(), xd.grad.float(), wd.grad, bn.weight.grad.clone(), bn.running_mean.clone(), bn.running_var.clone()))
        finally:
            AG._CONV_STATS_OFF[0] = False
    AG._CONV_STATS_MAX_COUT[0] = 128
    tol = 2e-2 if dt != "fp32" else 1e-4
    for a_, b_ in zip(*out):
        assert _rel(a_, b_) < tol, _rel(a_, b_)
    assert float(AG._bn_scratch(bn_a.weight).abs().max()) == 0.0


def test_packed_conv_operands_follow_the_parameter():
    """The training path keeps packed copies of every conv parameter (forward and data-gradient operands) keyed by the
    parameter and its version: an in-place update (what an optimizer does) is seen by the next conv, with or without the
    step's batched re-pack, and a parameter that moved (.to / load) gets fresh buffers."""
    from monoflex_amd import autograd as AG
    g = torch.Generator().manual_seed(2)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(DEV)
    x = torch.randn(2, 16, 24, 64, generator=g).to(DEV)

    def ref():
        return F.conv2d(x.permute(0, 3, 1, 2), conv.weight, padding=1).permute(0, 2, 3, 1)
    for step in range(3):
        xd = x.clone().requires_grad_()
        y = AG.conv2d(xd, conv.weight, None, 1, 1)
        assert _rel(y, ref().detach()) < 1e-5, step
        y.square().sum().backward()
        gx = torch.autograd.grad(F.conv2d(xd.permute(0, 3, 1, 2), conv.weight.detach(), padding=1).square().sum(), xd)[0]
        assert _rel(xd.grad, gx) < 1e-4, step                     # data-gradient operand (mode 1) is current too
        with torch.no_grad():
            conv.weight.mul_(0.5).add_(0.01 * (step + 1))          # optimizer-style in-place update: version bump
        if step == 1:
            AG.pack_all_weights()                                  # the step's batched re-pack sees the new values as well
    w2 = torch.nn.Parameter(conv.weight.detach().clone() * 3)      # a different parameter object with the same shape
    assert _rel(AG.conv2d(x, w2, None, 1, 1), F.conv2d(x.permute(0, 3, 1, 2), w2, padding=1).permute(0, 2, 3, 1).detach()) < 1e-5
    tmp = conv.weight.detach() * 2                                 # a temporary (not a Parameter): packed on demand, nothing cached
    n_before = len(AG._PACKS.entries)
    assert _rel(AG.conv2d(x, tmp, None, 1, 1), F.conv2d(x.permute(0, 3, 1, 2), tmp, padding=1).permute(0, 2, 3, 1)) < 1e-5
    assert len(AG._PACKS.entries) == n_before
    import gc
    del conv, w2, y, xd                                            # the packed buffers go with their parameters
    gc.collect()
    assert len(AG._PACKS.entries) <= n_before - 3                  # conv.weight: forward + data-gradient operands; w2: forward


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
def test_batched_operand_packing_equals_the_single_operand_kernel(dt):
    """mfx_pack_conv_weights_batched (one launch for every conv operand of a step; chunks staged through LDS) against
    mfx_pack_conv_weight (one element per lane) on the operand shapes of the network, forward and data-gradient forms: the packed
    matrix and the fragment-major copy BITWISE, padding rows / columns included; the descriptors are laid out as the training path
    lays them out (monoflex_amd.autograd._PackRegistry)."""
    import numpy as np
    from monoflex_amd import lib as L, ops
    lib_ = L.load()
    dtype = DT[dt]
    E = 4 if dt == "fp32" else 8
    g = torch.Generator().manual_seed(5)
    chunk = lib_.mfx_pack_chunk_elems()
    shapes = [(64, 64, 3, 0), (64, 64, 3, 1), (27, 64, 3, 0), (64, 27, 3, 1), (256, 64, 3, 0), (256, 64, 3, 1), (512, 512, 3, 0), (512, 512, 3, 1),
              (128, 64, 3, 0), (128, 64, 3, 1), (64, 128, 1, 0), (64, 128, 1, 1), (3, 256, 1, 0), (3, 256, 1, 1), (20, 256, 1, 0), (32, 16, 3, 0),
              (32, 16, 3, 1), (1024, 1024, 3, 0), (16, 16, 1, 0)]
    items, descs, prefix, total = [], [], [0], 0
    for cout, cin, k, mode in shapes:
        w = torch.randn(cout, cin, k, k, generator=g).to(DEV)
        rows, ck_src = (cout, cin) if mode == 0 else (cin, cout)
        ck = max(E, (ck_src + E - 1) // E * E)
        if k > 1:
            ck = 1 << (ck - 1).bit_length()                      # channels per tap: a power of two for k > 1 (lookup's rule)
        K_pad = (k * k * ck + 8 * E - 1) // (8 * E) * (8 * E)
        cp = ops.cout_pad((rows + 15) // 16 * 16)
        bufs = [torch.full((cp, K_pad), float("nan"), dtype=dtype, device=DEV) for _ in range(4)]       # batched packed / frag, single packed / frag
        d = L.PackDesc()
        d.w, d.packed, d.frag = w.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr()
        d.Cout, d.Cin, d.kh, d.kw, d.mode, d.rows_pad, d.K_pad, d.ck = cout, cin, k, k, mode, cp, K_pad, ck
        descs.append(bytes(d))
        total += (cp * K_pad + chunk - 1) // chunk
        prefix.append(total)
        items.append((w, bufs, (cout, cin, k, mode, cp, K_pad, ck)))
    dt_t = torch.from_numpy(np.frombuffer(b"".join(descs), dtype=np.uint8).copy()).to(DEV)
    pt = torch.tensor(prefix, dtype=torch.int64).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib_.mfx_pack_conv_weights_batched(dt_t.data_ptr(), pt.data_ptr(), len(items), total, L.MFX_F32 if dt == "fp32" else (L.MFX_BF16 if dt == "bf16" else L.MFX_F16), st),
            "mfx_pack_conv_weights_batched")
    for w, bufs, (cout, cin, k, mode, cp, K_pad, ck) in items:
        L.check(lib_.mfx_pack_conv_weight(w.data_ptr(), cout, cin, k, k, mode, bufs[2].data_ptr(), bufs[3].data_ptr(), cp, K_pad, ck,
                                          L.MFX_F32 if dt == "fp32" else (L.MFX_BF16 if dt == "bf16" else L.MFX_F16), st), "mfx_pack_conv_weight")
    torch.cuda.synchronize()
    for w, bufs, meta in items:
        view = torch.int32 if dt == "fp32" else torch.int16
        assert torch.equal(bufs[0].view(view), bufs[2].view(view)), ("packed", meta)
        assert torch.equal(bufs[1].view(view), bufs[3].view(view)), ("frag", meta)
        assert not torch.isnan(bufs[0].float()).any() and not torch.isnan(bufs[1].float()).any(), meta       # every element was written
        assert float(bufs[0].float().abs().max()) > 0


def test_maxpool_and_upsample_grads():
    from monoflex_amd import autograd as AG
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 16, 24, generator=g)
    xr = x.clone().requires_grad_()
    yr = F.max_pool2d(xr, 2, 2)
    r = torch.randn(yr.shape, generator=g)
    (yr * r).sum().backward()
    xd = _nhwc(x).to(DEV).requires_grad_()
    yd = AG.MaxPool2x2Fn.apply(xd)
    (yd * _nhwc(r).to(DEV)).sum().backward()
    assert _rel(yd.permute(0, 3, 1, 2), yr) == 0 and _rel(xd.grad.permute(0, 3, 1, 2), xr.grad) == 0
    for f in (2, 4, 8):
        C = 64
        x = torch.randn(2, C, 6, 10, generator=g)
        w = torch.rand(C, 1, 2 * f, 2 * f, generator=g)
        sk = torch.randn(2, C, 6 * f, 10 * f, generator=g)
        xr, wr, sr = x.clone().requires_grad_(), w.clone().requires_grad_(), sk.clone().requires_grad_()
        yr = F.conv_transpose2d(xr, wr, stride=f, padding=f // 2, groups=C) + sr
        r = torch.randn(yr.shape, generator=g)
        (yr * r).sum().backward()
        xd, wd, sd = _nhwc(x).to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), _nhwc(sk).to(DEV).requires_grad_()
        yd = AG.UpsampleAddFn.apply(xd, wd, sd, f)
        (yd * _nhwc(r).to(DEV)).sum().backward()
        assert _rel(yd.permute(0, 3, 1, 2), yr) < 1e-5
        assert _rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < 1e-5 and _rel(wd.grad, wr.grad) < 1e-5
        assert _rel(sd.grad.permute(0, 3, 1, 2), sr.grad) == 0


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 64), (512, 256)])
def test_dcn_train_grads_vs_oracle(cin, cout):
    """DCN module, training form: offset conv -> DCNv2, gradients to input and all four parameters, against the C
    restatement of the reference's CPU backward (oracle/dcn_v2_ref.c)."""
    from oracle import monoflex_ref as R
    from monoflex_amd.model.backbone.DCNv2.dcn_v2 import DCN
    g = torch.Generator().manual_seed(6)
    ref = R.DCN(cin, cout)
    dev = DCN(cin, cout, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    with torch.no_grad():
        ref.weight.copy_(torch.randn(ref.weight.shape, generator=g) * 0.05)
        ref.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        ref.conv_offset_mask.weight.copy_(torch.randn(ref.conv_offset_mask.weight.shape, generator=g) * 0.02)
        ref.conv_offset_mask.bias.copy_(torch.randn(27, generator=g) * 0.5)
    dev.load_state_dict(ref.state_dict())
    dev = dev.to(DEV).train()
    x = torch.randn(2, cin, 12, 20, generator=g)
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    r = torch.randn(yr.shape, generator=g)
    (yr * r).sum().backward()
    xd = _nhwc(x).to(DEV).requires_grad_()
    yd = dev.forward_nhwc_train(xd)
    (yd * _nhwc(r).to(DEV)).sum().backward()
    assert _rel(yd.permute(0, 3, 1, 2), yr) < 1e-4
    assert _rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < 2e-4
    for n, p in dev.named_parameters():
        pr = dict(ref.named_parameters())[n]
        assert _rel(p.grad, pr.grad) < 2e-4, n


def _models(out_w, out_h):
    from monoflex_amd import synthetic as S
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.detector import KeypointDetector
    from oracle import monoflex_ref as R
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.MODEL.PRETRAIN = False
    cfg.MODEL.COMPUTE_DTYPE = "fp32"
    cfg.INPUT.WIDTH_TRAIN, cfg.INPUT.HEIGHT_TRAIN = out_w * 4, out_h * 4
    m = KeypointDetector(cfg)
    sd = S.synthetic_state_dict(m.state_dict(), seed=3, cls_bias=-1.0)
    m.load_state_dict(sd)
    ref = R.KeypointDetectorRef()
    ref.load_state_dict(sd)
    return m.to(DEV).train(), ref.train()


def _oracle_grads(ref, imgs, eidx, elen, rc, rr):
    taps = {}
    om = ref.forward_maps(imgs, eidx.long(), elen.long(), taps)
    ((taps['cls_logits'] * rc).sum() + (om['reg'] * rr).sum()).backward()      # logits tap: clone taken before the sigmoid
    return taps['cls_logits'].detach(), om['reg'].detach(), {n: p.grad for n, p in ref.named_parameters()}


def test_network_gradients_vs_oracle():
    """Whole network in training mode (batch-statistics BN everywhere, DCN offsets learned) under a linear surrogate loss.

    With ~100 ReLU/BN layers an fp32 gradient is only reproducible up to rounding-induced ReLU sign flips, so the bar is
    set by measurement rather than by a constant: an fp64 evaluation of the oracle (autograd DCN form) is the ground
    truth, the fp32 CPU oracle (C restatement of the reference backward) shows what fp32 rounding costs, and the HIP path
    must be as close to the truth as that (within 3x, per-tensor error relative to the tensor's largest entry), with every
    tensor's direction matching (cosine > 0.98)."""
    import copy
    from monoflex_amd import synthetic as S
    from oracle import monoflex_ref as R
    out_w, out_h = 64, 32
    m, ref = _models(out_w, out_h)
    ref64 = copy.deepcopy(ref).double()
    for mod in ref64.modules():
        if isinstance(mod, R.DCN):
            mod.torch_form = True
    B = 2
    imgs = S.synthetic_images(B, out_h * 4, out_w * 4, seed=11)
    eidx, elen = _edges(B, out_w, out_h)
    g = torch.Generator().manual_seed(12)
    rc = torch.randn(B, 3, out_h, out_w, generator=g)
    rr = torch.randn(B, 50, out_h, out_w, generator=g)
    c64, r64, g64 = _oracle_grads(ref64, imgs.double(), eidx, elen, rc.double(), rr.double())
    c32, r32, g32 = _oracle_grads(ref, imgs, eidx, elen, rc, rr)
    cls, reg = m.forward_train_maps(imgs.to(DEV), eidx.to(DEV), elen.to(DEV))
    ((cls * _nhwc(rc).to(DEV)).sum() + (reg * _nhwc(rr).to(DEV)).sum()).backward()
    f_h = max(_rel(cls.permute(0, 3, 1, 2), c64), _rel(reg.permute(0, 3, 1, 2), r64))
    f_32 = max(_rel(c32, c64), _rel(r32, r64))
    e_h, e_32, cos, dead = [], [], [], []
    for n, p in m.named_parameters():
        t = g64[n]
        if t is None:
            dead.append(n)
            assert p.grad is None and g32[n] is None, n
            continue
        assert p.grad is not None, n
        t = t.float()
        scale = float(t.abs().max())
        if scale < 1e-2:            # conv bias feeding a batch-statistics BN: true gradient is exactly zero, both hold noise
            assert float(p.grad.abs().max()) < 1e-1, n
            continue
        gh = p.grad.detach().cpu().float()
        e_h.append(float((gh - t).abs().max()) / scale)
        e_32.append(float((g32[n] - t).abs().max()) / scale)
        cos.append((float(F.cosine_similarity(gh.flatten(), t.flatten(), dim=0)), n))
    e_h, e_32 = np.array(e_h), np.array(e_32)
    print("forward err vs fp64: hip %.2e cpu-fp32 %.2e | grad err vs fp64: hip max %.2e med %.2e, cpu-fp32 max %.2e med %.2e, "
          "min cos %.5f" % (f_h, f_32, e_h.max(), np.median(e_h), e_32.max(), np.median(e_32), min(cos)[0]))
    assert len(dead) == 6, dead                      # outer level3/level4 project conv+BN
    assert f_h <= max(3 * f_32, 1e-4), (f_h, f_32)
    assert e_h.max() <= 3 * e_32.max() and np.median(e_h) <= 3 * np.median(e_32), (e_h.max(), e_32.max())
    assert min(cos)[0] > 0.98, min(cos)
    refb = dict(ref64.named_buffers())
    for n, b in m.named_buffers():
        if n.endswith("running_mean") or n.endswith("running_var"):
            if any(n.startswith(d.rsplit(".", 1)[0]) for d in dead):
                continue
            assert _rel(b, refb[n]) < 1e-3, n


def _edges(B, out_w, out_h):
    from monoflex_amd import synthetic as S
    ts = [S.synthetic_target(out_w, out_h, orig_size=(out_w * 4 - 8 * (i + 1), out_h * 4 - 4 * (i + 1))) for i in range(B)]
    ei = torch.stack([torch.as_tensor(np.asarray(t["edge_indices"])) for t in ts]).to(torch.int32)
    el = torch.as_tensor([int(t["edge_len"]) for t in ts], dtype=torch.int32)
    return ei, el


def test_loss_on_gpu_matches_reference_golden():
    from test_loss_golden import check_case
    check_case("b3_empty_middle_mixed_calib", DEV)


def _train_batch(B, out_w, out_h, seed0=20):
    from monoflex_amd import synthetic as S
    from monoflex_amd.structures.params_3d import make_train_target
    tg = [S.synthetic_train_target(seed0 + i, out_w=out_w, out_h=out_h, n_obj=3 + i) for i in range(B)]
    imgs = S.synthetic_images(B, out_h * 4, out_w * 4, seed=seed0)
    return imgs, tg, [make_train_target(t) for t in tg]


def test_training_forward_loss_vs_oracle():
    """KeypointDetector.forward in training mode -> (loss_dict, log_loss_dict): the 11 losses against the CPU oracle
    network (train-mode BN) evaluated with the same loss module."""
    out_w, out_h = 96, 32
    m, ref = _models(out_w, out_h)
    imgs, tg, targets = _train_batch(2, out_w, out_h)
    loss_dict, logs = m(imgs.to(DEV), [t.to(DEV) for t in targets])
    ei = torch.stack([torch.as_tensor(t["edge_indices"]) for t in tg])
    el = torch.as_tensor([int(t["edge_len"]) for t in tg])
    with torch.no_grad():
        om = ref.forward_maps(imgs, ei, el)
        want, _ = m.heads.loss_evaluator(om, targets)
    assert set(loss_dict) == set(want) and len(want) == 11
    for k in want:
        a, b = float(loss_dict[k]), float(want[k])
        assert abs(a - b) <= 3e-3 * max(1.0, abs(b)), (k, a, b)
    assert all(isinstance(v, float) for v in logs.values())


FULL_SIZE_TENSORS = (
    "backbone.base.base_layer.0.weight", "backbone.base.level0.0.weight", "backbone.base.level1.0.weight",
    "backbone.base.level2.tree1.conv1.weight", "backbone.base.level2.root.conv.weight", "backbone.base.level3.tree1.tree1.conv2.weight",
    "backbone.base.level3.tree2.root.conv.weight", "backbone.base.level4.tree2.tree1.conv1.weight", "backbone.base.level5.tree2.conv2.weight",
    "backbone.base.level5.root.conv.weight", "backbone.dla_up.ida_0.proj_1.conv.weight", "backbone.dla_up.ida_0.node_1.conv.conv_offset_mask.weight",
    "backbone.dla_up.ida_1.node_2.conv.weight", "backbone.dla_up.ida_1.up_1.weight", "backbone.dla_up.ida_2.proj_3.conv.weight",
    "backbone.dla_up.ida_2.node_1.conv.weight", "backbone.dla_up.ida_2.node_3.conv.conv_offset_mask.weight", "backbone.dla_up.ida_2.node_3.conv.conv_offset_mask.bias",
    "backbone.dla_up.ida_2.node_3.actf.0.weight", "backbone.ida_up.proj_2.conv.weight", "backbone.ida_up.node_1.conv.weight",
    "backbone.ida_up.node_2.conv.weight", "backbone.ida_up.node_2.conv.conv_offset_mask.weight", "backbone.ida_up.up_2.weight",
    "heads.predictor.class_head.0.weight", "heads.predictor.class_head.2.weight", "heads.predictor.reg_features.2.0.weight",
    "heads.predictor.reg_heads.2.0.weight", "heads.predictor.reg_features.1.0.weight", "heads.predictor.trunc_offset_conv.0.weight",
)


def _full_size_rows(gdev, gref):
    rows = []
    for n in FULL_SIZE_TENSORS:
        if gref[n].grad is None or float(gref[n].grad.abs().max()) == 0.0:      # (no truncated object in the batch: that branch has no gradient)
            continue
        a, b = gdev[n].grad.detach().double().cpu().flatten(), gref[n].grad.double().flatten()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp(min=1e-30))
        rows.append((n, cos, float((a - b).norm() / b.norm().clamp(min=1e-30))))
    return rows


@pytest.mark.parametrize("det", [False, True])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_full_size_training_step_vs_oracle(dtype, det):
    """`model(images, targets)` + backward at the benchmarked size (B = 2, 1280 x 384) against the oracle network in fp32 (CPU,
    train-mode BN, C restatement of the reference DCN backward) with the same loss module on the same inputs.

    fp32: the 11 losses and 29 gradient tensors spread over the stem, every DLA level, DCN main / offset-mask convs, the
    depthwise up-samplers and the heads (measured on MI355X: losses 1e-4, every cosine >= 0.9988, relative l2 <= 5.1e-2 --
    rounding amplified through ~100 batch-statistics BN layers; bounds ~2x).
    bf16: ONLY the 11 losses are bounded end to end (measured worst 13.6 %).  A randomly initialised DLA-34 under batch-statistics BN
    is chaotic: the bf16 feature map leaves the fp32 one by 86 % in l2 (13 % already at level5, doubling per level;
    tools/probes/train_bf16_diag.py) and whole-network gradients are uncorrelated with the oracle's although every layer is
    right -- which is what tests/test_gpu_train_fullsize.py pins, layer by layer on the same step, to bf16 rounding."""
    from monoflex_amd import lib as L, synthetic as S
    from monoflex_amd.structures.params_3d import make_train_target
    out_w, out_h, B = 320, 96, 2
    m, ref = _models(out_w, out_h)
    m.set_compute_dtype(dtype)
    tg = [S.synthetic_train_target(1000 + i) for i in range(B)]
    targets = [make_train_target(t) for t in tg]
    imgs = S.synthetic_images(B, seed=1000)
    ei = torch.stack([torch.as_tensor(t["edge_indices"]) for t in tg])
    el = torch.as_tensor([int(t["edge_len"]) for t in tg])
    lib_ = L.load()
    fused0 = lib_.mfx_get_counter(b"dcn_bt_fused")
    # det: the library's fixed-order reductions (lib.set_deterministic) -- the run-to-run spread of the atomics is gone, so the ORIGINAL bounds of
    # this test (0.28 / 0.997, loosened in r04 to 0.45 / 0.995 for that spread) hold again and are asserted in this mode
    L.set_deterministic(det)
    try:
        loss_dict, _ = m(imgs.to(DEV), [t.to(DEV) for t in targets])
        sum(loss_dict.values()).backward()
        torch.cuda.synchronize()
    finally:
        L.set_deterministic(False)
    if not det:
        assert lib_.mfx_get_counter(b"dcn_bt_fused") - fused0 == (5 if dtype == "bf16" else 0)
    om = ref.forward_maps(imgs, ei, el)
    want, _ = m.heads.loss_evaluator(om, targets)
    assert set(loss_dict) == set(want) and len(want) == 11
    worst_loss = max(abs(float(loss_dict[k]) - float(want[k])) / max(1.0, abs(float(want[k]))) for k in want)
    print("full-size %s step vs oracle (deterministic %s): worst relative loss deviation %.3e" % (dtype, det, worst_loss),
          {k: (round(float(loss_dict[k]), 5), round(float(want[k]), 5)) for k in want})
    # bf16: a randomly initialised DLA-34 under batch-statistics BN amplifies rounding differences from level to level, and the
    # BN statistics are summed with atomics, so the figure moves from run to run: 0.14 ... 0.33 observed over repeated runs on MI355X (r03 / r04;
    # one loss term sits on a kink); the layer-by-layer test (test_gpu_train_fullsize.py) is where the 16-bit arithmetic is pinned
    assert worst_loss < (3e-4 if dtype == "fp32" else (0.28 if det else 0.45)), worst_loss
    if dtype == "bf16":
        assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
        return
    sum(want.values()).backward()
    rows = _full_size_rows(dict(m.named_parameters()), dict(ref.named_parameters()))
    for r in rows:
        print("   %-70s cos %.5f  rel %.3e" % r)
    assert len(rows) >= 29
    assert min(r[1] for r in rows) > (0.997 if det else 0.995), min(rows, key=lambda r: r[1])      # (0.9962 ... 0.9988 over repeated runs without `det`: atomics' summation order)
    assert max(r[2] for r in rows) < 0.1, max(rows, key=lambda r: r[2])


def test_train_steps_update_parameters():
    """Four optimisation steps (engine.trainer.train_step, AdamW groups of solver.build_optimizer): finite losses,
    every live parameter moves, the six dead ones never get a gradient, BN running statistics move."""
    from monoflex_amd.config import get_cfg
    from monoflex_amd.engine.trainer import dead_parameter_names, train_step
    from monoflex_amd.solver import build_optimizer
    out_w, out_h = 96, 32
    m, _ = _models(out_w, out_h)
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    opt = build_optimizer(m, cfg)
    assert sum(len(g["params"]) for g in opt.param_groups) == 280
    imgs, _, targets = _train_batch(2, out_w, out_h)
    imgs, targets = imgs.to(DEV), [t.to(DEV) for t in targets]
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    rm0 = m.backbone.base.base_layer[1].running_mean.clone()
    losses = []
    for _ in range(4):
        total, loss_dict, logs = train_step(m, opt, imgs, targets)
        losses.append(float(total))
        assert all(torch.isfinite(v) for v in loss_dict.values())
    dead = set(dead_parameter_names(m))
    for n, p in m.named_parameters():
        if n in dead:
            assert p.grad is None and torch.equal(p, before[n]), n
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
            # (the edge-fusion offset branch only sees a gradient when a truncated object sits on the border)
            assert not torch.equal(p, before[n]) or float(p.grad.abs().max()) == 0, n
    assert not torch.equal(rm0, m.backbone.base.base_layer[1].running_mean)
    # same batch four times: the loss must go down.  The trajectory is not reproducible run to run (fp32 atomics in the BN
    # statistics / DCN backward reorder sums; after the first AdamW step the differences are macroscopic: step-2 losses of
    # 204 .. 259 were observed from the same start, 246.8), so the check is on the best later step, not on the last one.
    assert min(losses[1:]) < losses[0], losses


# ---- bf16 training mode (activations bf16, fp32 master weights / statistics / gradients of parameters) ------------------
@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout,k,stride,bias,f32out", [(16, 32, 3, 1, False, False), (32, 64, 3, 2, False, False),
                                                           (64, 27, 3, 1, True, True), (256, 3, 1, 1, True, True),
                                                           (64, 256, 3, 1, False, False), (128, 128, 3, 2, False, False),
                                                           (256, 512, 1, 1, False, False)])
def test_conv_grads_bf16(cin, cout, k, stride, bias, f32out, half):
    from monoflex_amd import autograd as AG
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, cin, 20, 36, generator=g).to(DT[half]).float()
    w = torch.randn(cout, cin, k, k, generator=g) * 0.1
    b = torch.randn(cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=k // 2)
    r = torch.randn(yr.shape, generator=g).to(DT[half]).float()
    (yr * r).sum().backward()
    xd = _nhwc(x).to(DEV).to(DT[half]).requires_grad_()
    wd = w.to(DEV).requires_grad_()
    bd = b.to(DEV).requires_grad_() if bias else None
    yd = AG.conv2d(xd, wd, bd, stride, k // 2, out_dtype=torch.float32 if f32out else None)
    assert yd.dtype == (torch.float32 if f32out else DT[half])
    assert _rel(yd.permute(0, 3, 1, 2), yr) < 2e-2
    (yd.float() * _nhwc(r).to(DEV)).sum().backward()
    assert _rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < 2e-2
    assert _rel(wd.grad, wr.grad) < 2e-2 and wd.grad.dtype == torch.float32
    if bias:
        assert _rel(bd.grad, br.grad) < 2e-2


@pytest.mark.parametrize("half", ["bf16", "fp16"])
def test_train_steps_bf16_mode(half):
    """16-bit training modes end to end: four AdamW steps on one batch; finite, decreasing loss; losses close to fp32 mode's.
    fp16 runs under the dynamic loss scaler (engine.trainer.LossScaler): the scale must have survived (no skipped step at 2^12)."""
    from monoflex_amd.config import get_cfg
    from monoflex_amd.engine.trainer import LossScaler, train_step
    from monoflex_amd.solver import build_optimizer
    out_w, out_h = 96, 32
    m32, _ = _models(out_w, out_h)
    m16, _ = _models(out_w, out_h)
    m16.set_compute_dtype(half)
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    imgs, _, targets = _train_batch(2, out_w, out_h)
    imgs, targets = imgs.to(DEV), [t.to(DEV) for t in targets]
    l32, _ = m32(imgs, targets)
    l16, _ = m16(imgs, targets)
    # (random weights + batch-statistics BN over a 3x1-pixel level5 map: the exp()-decoded terms swing with single logits, so
    # only the dense heat-map term is compared; per-operator bf16 accuracy is covered above)
    assert all(torch.isfinite(v) for v in l16.values())
    # (BN statistics are accumulated with atomics: run-to-run the bf16 roundings downstream differ, so the bound is loose)
    assert abs(float(l16["hm_loss"]) - float(l32["hm_loss"])) <= 0.15 * float(l32["hm_loss"])
    opt = build_optimizer(m16, cfg)
    scaler = LossScaler.for_model(m16)
    assert (scaler is not None) == (half == "fp16")
    if scaler is not None:
        scaler.attach(opt)
    before = [p.detach().clone() for p in m16.parameters()]
    n = 4 if scaler is None else 8
    losses = [float(train_step(m16, opt, imgs, targets, scaler=scaler)[0]) for _ in range(n)]
    assert all(np.isfinite(losses)) and min(losses[1:]) < losses[0], losses
    assert sum(int(not torch.equal(a, b)) for a, b in zip(before, m16.parameters())) > 100       # steps were applied
    if scaler is not None:
        # this 384 x 128 toy configuration has the steepest gradient growth (643x at scale 1): the scaler halves 2^8 until the step fits,
        # skipping those steps, and then stays; the applied steps are counted by the growth tracker and by AdamW's own step counters
        sc, halvings = float(scaler.scale), int(round(np.log2(256.0 / float(scaler.scale))))
        assert sc in [2.0 ** k for k in range(1, 9)], sc                          # (how far it backs off differs run to run: atomics order)
        assert 0 <= int(scaler.growth_tracker) <= n - halvings                 # (clean steps since the last overflow; 0: the last step itself overflowed)
        assert all(int(st["step"]) == n - halvings for st in opt.state.values())


def test_loss_scaler_skips_an_overflowing_step_exactly():
    """fp16 mode with a loss scale that must overflow (2^24): the step's gradients are non-finite, so the fused AdamW leaves the
    parameters, both moments and its step counters BITWISE untouched, the scale is halved and the growth tracker reset; the next
    step at a fitting scale is applied.  Same through the captured step (GraphedTrainStep)."""
    from monoflex_amd.config import get_cfg
    from monoflex_amd.engine.trainer import GraphedTrainStep, LossScaler, prepare_targets, train_step
    from monoflex_amd.solver import build_optimizer
    out_w, out_h = 96, 32
    m, _ = _models(out_w, out_h)
    m.set_compute_dtype("fp16")
    m.heads.loss_evaluator.log_as_float = False
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    imgs, _, targets = _train_batch(2, out_w, out_h)
    imgs, targets = imgs.to(DEV), [t.to(DEV) for t in targets]
    opt = build_optimizer(m, cfg)
    scaler = LossScaler(torch.device(DEV), init_scale=2.0 ** 4).attach(opt)
    train_step(m, opt, imgs, targets, scaler=scaler)                       # a clean step first: the optimizer state exists
    assert float(scaler.found_inf) == 0.0 and int(scaler.growth_tracker) == 1
    snap = [p.detach().clone() for p in m.parameters()]
    state = [(st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["step"].clone()) for st in opt.state.values()]
    scaler.scale.fill_(2.0 ** 24)
    train_step(m, opt, imgs, targets, scaler=scaler)
    assert float(scaler.found_inf) == 1.0 and float(scaler.scale) == 2.0 ** 23 and int(scaler.growth_tracker) == 0
    assert all(torch.equal(a, b) for a, b in zip(snap, m.parameters()))
    for (ea, es, stp), st in zip(state, opt.state.values()):
        assert torch.equal(ea, st["exp_avg"]) and torch.equal(es, st["exp_avg_sq"]) and torch.equal(stp, st["step"])
    # the captured step carries the scaler: one replay at 2^24 is skipped, the following one at 2^4 is applied
    step = GraphedTrainStep(m, opt, imgs.clone(), prepare_targets(m, targets, torch.device(DEV)), scaler=scaler, warmup=2)
    scaler.scale.fill_(2.0 ** 24)
    snap = [p.detach().clone() for p in m.parameters()]
    step()
    torch.cuda.synchronize()
    assert float(scaler.found_inf) == 1.0 and float(scaler.scale) == 2.0 ** 23
    assert all(torch.equal(a, b) for a, b in zip(snap, m.parameters()))
    scaler.scale.fill_(2.0 ** 4)
    loss = step()
    torch.cuda.synchronize()
    assert float(scaler.found_inf) == 0.0 and torch.isfinite(loss)
    assert sum(int(not torch.equal(a, b)) for a, b in zip(snap, m.parameters())) > 100


@pytest.mark.parametrize("half", ["bf16", "fp16"])
def test_stem_conv_wgrad_bf16(half):
    """bf16 stem weight gradient: matrix-core kernel over 8-element super-taps with dilation 2, mapped back to (16,3,7,7)."""
    from monoflex_amd import autograd as AG
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 32, 64, generator=g).to(DT[half]).float()
    w = (torch.randn(16, 3, 7, 7, generator=g) * 0.1)
    wr = w.clone().requires_grad_()
    yr = F.conv2d(x, wr, padding=3)
    r = torch.randn(yr.shape, generator=g).to(DT[half]).float()
    (yr * r).sum().backward()
    wd = w.to(DEV).requires_grad_()
    yd = AG.StemConvFn.apply(x.to(DEV), wd, DT[half])
    (yd.float() * _nhwc(r).to(DEV)).sum().backward()
    assert _rel(yd.permute(0, 3, 1, 2), yr) < 2e-2 and _rel(wd.grad, wr.grad) < 2e-2


@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 64)])
def test_dcn_train_grads_bf16_vs_oracle(cin, cout, half):
    """bf16 DCN backward (bf16 x / dy / d(columns), fp32 accumulation of every gradient) against the C oracle's fp32
    gradients on bf16-representable inputs; tolerance of a bf16 pipeline."""
    from oracle import monoflex_ref as R
    from monoflex_amd.model.backbone.DCNv2.dcn_v2 import DCN
    g = torch.Generator().manual_seed(8)
    ref = R.DCN(cin, cout)
    dev = DCN(cin, cout, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    with torch.no_grad():
        ref.weight.copy_((torch.randn(ref.weight.shape, generator=g) * 0.05).to(DT[half]).float())
        ref.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        ref.conv_offset_mask.weight.copy_((torch.randn(ref.conv_offset_mask.weight.shape, generator=g) * 0.02).to(DT[half]).float())
        ref.conv_offset_mask.bias.copy_(torch.randn(27, generator=g) * 0.5)
    dev.load_state_dict(ref.state_dict())
    dev = dev.to(DEV).train()
    x = torch.randn(2, cin, 12, 20, generator=g).to(DT[half]).float()
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    r = torch.randn(yr.shape, generator=g).to(DT[half]).float()
    (yr * r).sum().backward()
    xd = _nhwc(x).to(DEV).to(DT[half]).requires_grad_()
    yd = dev.forward_nhwc_train(xd)
    (yd.float() * _nhwc(r).to(DEV)).sum().backward()
    assert _rel(yd.permute(0, 3, 1, 2), yr) < 2e-2
    assert _rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < 3e-2
    refp = dict(ref.named_parameters())
    for n, p in dev.named_parameters():
        assert _rel(p.grad, refp[n].grad) < 4e-2, n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,H,W,off_std", [(64, 64, 13, 37, 0.5), (64, 128, 24, 40, 3.0), (128, 64, 9, 50, 6.0), (256, 64, 12, 20, 12.0),
                                                  (512, 256, 12, 40, 2.0)])
def test_dcn_tile_owned_backward_vs_oracle(cin, cout, H, W, off_std, dtype):
    """Second-generation DCN backward (dcn_bwd_tile.hip: grad_input accumulated per 8x16 tile in LDS, ring samples through a
    work list, far corners through the fp32 side buffer, grad_weight as an MFMA GEMM over the stored columns) against the C
    restatement of the reference backward: ragged tiles, 1/2/4 channel slices, offsets from sub-pixel to far beyond the
    8-pixel ring (std 12: most corners take the far path, many leave the image), and against the first-generation kernels."""
    from oracle import monoflex_ref as R
    from monoflex_amd import autograd as AG
    from monoflex_amd.model.backbone.DCNv2.dcn_v2 import DCN
    g = torch.Generator().manual_seed(18)
    bf = dtype != torch.float32                                  # 16-bit activations (bf16 / fp16)
    rnd = (lambda t: t.to(dtype).float()) if bf else (lambda t: t)
    ref = R.DCN(cin, cout)
    dev = DCN(cin, cout, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    with torch.no_grad():
        ref.weight.copy_(rnd(torch.randn(ref.weight.shape, generator=g) * 0.05))
        ref.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        ref.conv_offset_mask.weight.copy_(rnd(torch.randn(ref.conv_offset_mask.weight.shape, generator=g) * (0.3 / (9 * cin) ** 0.5)))
        b = torch.randn(27, generator=g) * off_std
        b[18:] = torch.randn(9, generator=g)
        ref.conv_offset_mask.bias.copy_(b)
    dev.load_state_dict(ref.state_dict())
    dev = dev.to(DEV).train()
    x = rnd(torch.randn(2, cin, H, W, generator=g))
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    r = rnd(torch.randn(yr.shape, generator=g))
    (yr * r).sum().backward()
    got = {}
    for gen in ("v1", "v2"):
        AG._DCN_BWD_V1[0] = gen == "v1"
        try:
            dev.zero_grad(set_to_none=True)
            xd = _nhwc(x).to(DEV).to(dtype).requires_grad_()
            yd = dev.forward_nhwc_train(xd)
            (yd.float() * _nhwc(r).to(DEV)).sum().backward()
            torch.cuda.synchronize()
            got[gen] = [xd.grad.float().permute(0, 3, 1, 2).cpu()] + [p.grad.float().cpu() for _, p in dev.named_parameters()]
        finally:
            AG._DCN_BWD_V1[0] = False
    names = ["input"] + [n for n, _ in dev.named_parameters()]
    want = [xr.grad] + [dict(ref.named_parameters())[n].grad for n in names[1:]]
    tol = 4e-2 if bf else 3e-4
    for n, a, w_ in zip(names, got["v2"], want):
        assert _rel(a, w_) < tol, (n, _rel(a, w_))
    if not bf:                                                   # fp32: both generations are exact up to summation order
        for n, a, b_ in zip(names, got["v2"], got["v1"]):
            assert _rel(a, b_) < 3e-4, (n, _rel(a, b_))


@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("C,Cout,H,W,fly", [(64, 64, 32, 64, 1), (64, 64, 32, 64, 0), (128, 64, 12, 40, 0)])
def test_dcn_backward_raw_gradient_in_the_activation_type(C, Cout, H, W, fly, half):
    """mfx_dcn_backward_v2_rt with `raw_in_act_dtype` (r06): the offset / mask gradient rows written in the 16-bit activation type are the fp32 rows
    of mfx_dcn_backward_v2 rounded once -- bit for bit what the cast in between produced -- for the three kernels that write them (gcol-free fused
    sample + weight gradient, the fused kernel behind a d(columns) GEMM, the plain sample kernel of the wide layers); the other outputs are the same call."""
    from monoflex_amd import lib as L
    lib_ = L.load()
    dt = DT[half]
    g = torch.Generator().manual_seed(5)
    B = 2
    x = torch.randn(B, H, W, C, generator=g).to(DEV).to(dt)
    om = torch.zeros(B, H, W, 32)
    om[..., :18] = torch.randn(B, H, W, 18, generator=g) * 2.5
    om[..., 18:27] = torch.sigmoid(torch.randn(B, H, W, 9, generator=g))
    om = om.to(DEV)
    w = (torch.randn(Cout, C, 3, 3, generator=g) * 0.05).to(DEV)
    dy = torch.randn(B, H, W, Cout, generator=g).to(DEV).to(dt)
    code = L.MFX_BF16 if half == "bf16" else L.MFX_F16
    nws = lib_.mfx_dcn_backward_v2_workspace_bytes(B, C, H, W, Cout, code)
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())                                      # noqa: E731
    outs = {}
    try:
        L.check(lib_.mfx_set_option(b"dcn_bt_fuse_min_chunks", 1), "opt")
        L.check(lib_.mfx_set_option(b"dcn_bt_fly", fly), "opt")
        for raw16 in (0, 1):
            dx = torch.empty_like(x)
            draw = torch.full((B, H, W, 32), float("nan"), dtype=dt if raw16 else torch.float32, device=DEV)
            dw, db = torch.empty(Cout, C, 3, 3, device=DEV), torch.empty(Cout, device=DEV)
            L.check(lib_.mfx_dcn_backward_v2_rt(ptr(x), ptr(om), ptr(w), ptr(dy), ptr(dx), ptr(draw), raw16, ptr(dw), ptr(db), B, C, H, W, Cout, code,
                                                ptr(ws), nws, None), "mfx_dcn_backward_v2_rt")
            torch.cuda.synchronize()
            outs[raw16] = (draw, dw, db)
    finally:
        L.check(lib_.mfx_set_option(b"dcn_bt_fuse_min_chunks", 1024), "opt")
        L.check(lib_.mfx_set_option(b"dcn_bt_fly", 1), "opt")
    assert outs[0][0].dtype == torch.float32 and outs[1][0].dtype == dt
    assert torch.equal(outs[0][0].to(dt), outs[1][0])
    assert bool((outs[1][0][..., 27:] == 0).all()) and float(outs[1][0].float().abs().max()) > 0
    assert _rel(outs[1][1], outs[0][1]) < 1e-5 and _rel(outs[1][2], outs[0][2]) < 1e-5


@pytest.mark.parametrize("form", ["fly", "fused", "unfused"])
def test_dcn_backward_is_repeatable(form):
    """The tile-owned DCN backward launched five times on the same inputs (64 -> 64 @ 2 x 96 x 320, bf16: the shape of the training step's five
    full-resolution layers): the offset / mask gradient rows and the weight gradient carry no atomics, so every launch must produce the same BITS.
    r06 found them differing run to run in ~0.03 % of the samples (lanes 48..63 of a wave; up to 20 % of an entry) with the SLP-vectorised
    packed-fp32 code of that translation unit -- build.py compiles it with -fno-slp-vectorize since (profiles/r06_dcnbwd_repeatability.md).
    grad_input is exempt: its far corners are added with packed 16-bit atomics (arrival-order rounding)."""
    from monoflex_amd import lib as L
    lib_ = L.load()
    dt, code = torch.bfloat16, L.MFX_BF16
    g = torch.Generator().manual_seed(5)
    B, C, Cout, H, W = 2, 64, 64, 96, 320
    x = torch.randn(B, H, W, C, generator=g).to(DEV).to(dt)
    om = torch.zeros(B, H, W, 32)
    om[..., :18] = torch.randn(B, H, W, 18, generator=g) * 1.5
    om[..., 18:27] = torch.sigmoid(torch.randn(B, H, W, 9, generator=g))
    om = om.to(DEV)
    w = (torch.randn(Cout, C, 3, 3, generator=g) * 0.05).to(DEV)
    dy = torch.randn(B, H, W, Cout, generator=g).to(DEV).to(dt)
    nws = lib_.mfx_dcn_backward_v2_workspace_bytes(B, C, H, W, Cout, code)
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())                                      # noqa: E731
    runs = []
    try:
        L.check(lib_.mfx_set_option(b"dcn_bt_fuse_wgrad", 0 if form == "unfused" else 1), "opt")
        L.check(lib_.mfx_set_option(b"dcn_bt_fly", 1 if form == "fly" else 0), "opt")
        for _ in range(5):
            dx = torch.empty_like(x)
            draw = torch.zeros((B, H, W, 32), dtype=dt, device=DEV)
            dw, db = torch.empty(Cout, C, 3, 3, device=DEV), torch.empty(Cout, device=DEV)
            L.check(lib_.mfx_dcn_backward_v2_rt(ptr(x), ptr(om), ptr(w), ptr(dy), ptr(dx), ptr(draw), 1, ptr(dw), ptr(db), B, C, H, W, Cout, code,
                                                ptr(ws), nws, None), "mfx_dcn_backward_v2_rt")
            torch.cuda.synchronize()
            runs.append((draw, dw))
    finally:
        L.check(lib_.mfx_set_option(b"dcn_bt_fuse_wgrad", 1), "opt")
        L.check(lib_.mfx_set_option(b"dcn_bt_fly", 1), "opt")
    for draw, dw in runs[1:]:
        assert torch.equal(draw, runs[0][0]), "offset / mask gradient rows differ between launches: %d entries" % int((draw != runs[0][0]).sum())
        if form != "unfused":                                    # (the unfused form's weight-gradient GEMM may add its slabs with atomics)
            assert torch.equal(dw, runs[0][1]), "weight gradient differs between launches: %d entries" % int((dw != runs[0][1]).sum())


@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("H,W,off_std,min_chunks", [(96, 320, 1.5, None), (96, 320, 5.0, None), (32, 64, 2.5, 1), (9, 32, 6.0, 1)])
def test_dcn_fused_sample_wgrad_kernel_vs_oracle(H, W, off_std, min_chunks, half):
    """`dcn_bwd_sample_wgrad_kernel` (dcn_bwd_tile.hip: grad_offset / grad_mask + grad_weight on the matrix cores from an LDS
    column tile, bf16, C = Cout = 64, W % 32 == 0) against the C restatement of the reference backward
    (dcn_v2_cuda.cu:206-335, dcn_v2_im2col_cuda.cu:197-327): all five gradients, at the shape the benchmarked training step
    dispatches it on (B=2, 64->64 @ 96x320: 1920 chunks >= the default threshold) and on small maps forced onto it through the
    option `dcn_bt_fuse_min_chunks`.  The launch counter proves the fused kernel ran; the unfused kernels (option
    `dcn_bt_fuse_wgrad` = 0) on the same inputs are the A/B side: same bf16 operands, so they agree far below the bf16 bar."""
    from oracle import monoflex_ref as R
    from monoflex_amd import lib as L
    from monoflex_amd.model.backbone.DCNv2.dcn_v2 import DCN
    g = torch.Generator().manual_seed(77)
    rnd = lambda t: t.to(DT[half]).float()                         # noqa: E731
    ref = R.DCN(64, 64)
    dev = DCN(64, 64, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    with torch.no_grad():
        ref.weight.copy_(rnd(torch.randn(ref.weight.shape, generator=g) * 0.05))
        ref.bias.copy_(torch.randn(64, generator=g) * 0.1)
        ref.conv_offset_mask.weight.copy_(rnd(torch.randn(ref.conv_offset_mask.weight.shape, generator=g) * (0.3 / 576 ** 0.5)))
        b = torch.randn(27, generator=g) * off_std
        b[18:] = torch.randn(9, generator=g)
        ref.conv_offset_mask.bias.copy_(b)
    dev.load_state_dict(ref.state_dict())
    dev = dev.to(DEV).train()
    x = rnd(torch.randn(2, 64, H, W, generator=g))
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    r = rnd(torch.randn(yr.shape, generator=g))
    (yr * r).sum().backward()
    lib_ = L.load()
    names = ["input"] + [n for n, _ in dev.named_parameters()]
    want = [xr.grad] + [dict(ref.named_parameters())[n].grad for n in names[1:]]
    got = {}
    try:
        if min_chunks is not None:
            L.check(lib_.mfx_set_option(b"dcn_bt_fuse_min_chunks", min_chunks), "opt")
        # four forms of the same backward: "fly2" (option dcn_bt_fly = 2) = the fused sample + weight-gradient kernel rebuilds d(columns) from dy on
        # the matrix cores and writes it for the tile kernel (no d(columns) GEMM, no re-read), "fly" (= 1, THE DEFAULT) = nothing materialised at all
        # (the tile kernel splats dy and multiplies by W per tap), 1 = the fused kernel reading a d(columns) GEMM's output, 0 = five launches
        for form in ("fly2", "fly", 1, 0):
            L.check(lib_.mfx_set_option(b"dcn_bt_fuse_wgrad", 0 if form == 0 else 1), "opt")
            L.check(lib_.mfx_set_option(b"dcn_bt_fly", {"fly2": 2, "fly": 1}.get(form, 0)), "opt")
            before, before_fly = lib_.mfx_get_counter(b"dcn_bt_fused"), lib_.mfx_get_counter(b"dcn_bt_fly")
            dev.zero_grad(set_to_none=True)
            xd = _nhwc(x).to(DEV).to(DT[half]).requires_grad_()
            yd = dev.forward_nhwc_train(xd)
            (yd.float() * _nhwc(r).to(DEV)).sum().backward()
            torch.cuda.synchronize()
            assert lib_.mfx_get_counter(b"dcn_bt_fused") - before == (0 if form == 0 else 1), "the fused kernel %s" % ("ran" if form == 0 else "did not run")
            assert lib_.mfx_get_counter(b"dcn_bt_fly") - before_fly == (1 if form in ("fly", "fly2") else 0), form
            got[form] = [xd.grad.float().permute(0, 3, 1, 2).cpu()] + [p.grad.float().cpu() for _, p in dev.named_parameters()]
    finally:
        L.check(lib_.mfx_set_option(b"dcn_bt_fuse_wgrad", 1), "opt")
        L.check(lib_.mfx_set_option(b"dcn_bt_fly", 1), "opt")
        L.check(lib_.mfx_set_option(b"dcn_bt_fuse_min_chunks", 1024), "opt")
    assert _rel(yd.permute(0, 3, 1, 2), yr) < 2e-2
    for form in ("fly2", "fly", 1):
        for n, a, w_ in zip(names, got[form], want):
            # bars of test_dcn_train_grads_bf16_vs_oracle for what the fused kernels produce (grad_weight, grad_offset / grad_mask through the
            # offset conv's parameters).  grad_input's largest entry-wise error over the 3.9 M entries of the full-size map moves with the
            # arrival order of the packed-16-bit far-corner atomics (observed 2.2e-2 ... 3.2e-2 over repeated runs): a gross bound here
            assert _rel(a, w_) < (6e-2 if n == "input" else 4e-2), (form, n, _rel(a, w_))
        for n, a, b_ in zip(names, got[form], got[0]):
            # (far corners are added with packed-16-bit atomics, whose rounding depends on arrival order, so two runs of the SAME kernels
            # differ by up to ~2 % of the largest entry at these offsets; the gcol-free form rounds the splatted dy where the others round
            # d(columns): the same size of error, elsewhere)
            assert _rel(a, b_) < (6e-2 if n == "input" else 1e-2), ("%s vs unfused" % form, n, _rel(a, b_))


@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout,k,stride,H,W", [(64, 256, 3, 1, 48, 96), (64, 64, 3, 1, 40, 72), (128, 128, 3, 2, 64, 80),
                                                   (256, 256, 3, 1, 24, 80), (512, 512, 3, 1, 12, 40), (64, 192, 3, 1, 33, 47),
                                                   (128, 27, 3, 1, 48, 96), (64, 27, 3, 1, 37, 75), (64, 64, 3, 1, 96, 40),
                                                   (16, 16, 3, 1, 64, 96), (32, 32, 3, 1, 48, 80), (16, 32, 3, 1, 37, 75), (32, 16, 3, 1, 41, 70)])
def test_conv_wgrad_transposed_read_kernel_vs_torch(cin, cout, k, stride, H, W, half):
    """Second-generation weight-gradient kernel (wgrad_tr.hip: natural-layout LDS tiles + ds_read_b64_tr_b16, 64x64 wave
    blocks, BK = 192) against torch autograd of F.conv2d on bf16-representable operands, and against the first-generation
    kernel; ragged pixel slabs, stride 2, Cout not a multiple of 128."""
    from monoflex_amd import autograd as AG, lib as L
    g = torch.Generator().manual_seed(31)
    B = 3
    x = torch.randn(B, cin, H, W, generator=g).to(DT[half]).float()
    w = torch.randn(cout, cin, k, k, generator=g) * 0.05
    wr = w.clone().requires_grad_()
    yr = F.conv2d(x, wr, None, stride=stride, padding=k // 2)
    r = torch.randn(yr.shape, generator=g).to(DT[half]).float()
    (yr * r).sum().backward()
    lib_ = L.load()
    got = {}
    try:
        for tr in (1, 2, 0):                                     # 1: LDS-patch form where it applies, 2: plain transposed-read form, 0: first generation
            L.check(lib_.mfx_set_option(b"wgrad_tr", 1 if tr else 0), "opt")
            L.check(lib_.mfx_set_option(b"wgrad_patch", 1 if tr == 1 else 0), "opt")
            xd = _nhwc(x).to(DEV).to(DT[half])
            wd = w.to(DEV).requires_grad_()
            yd = AG.conv2d(xd, wd, None, stride, k // 2)
            (yd.float() * _nhwc(r).to(DEV)).sum().backward()
            got[tr] = wd.grad.detach().cpu()
    finally:
        L.check(lib_.mfx_set_option(b"wgrad_tr", 1), "opt")
        L.check(lib_.mfx_set_option(b"wgrad_patch", 1), "opt")
    # dy is rounded to bf16 by the conv's output dtype on both sides; fp32 accumulation: only summation order differs
    for v in (1, 2):
        assert _rel(got[v], got[0]) < 2e-3, (v, _rel(got[v], got[0]))
        assert _rel(got[v], wr.grad) < 1.5e-2, (v, _rel(got[v], wr.grad))


@pytest.mark.parametrize("name", ["b2", "b3_empty_middle_mixed_calib", "b1_many"])
def test_fused_focal_loss_kernel_vs_reference_golden(name):
    """csrc/loss_kernels.hip (sigmoid + clamp + penalty-reduced focal loss + gradient, one pass) against the fixture recorded
    from the reference's Loss_Computation (tests/golden/loss.npz): the heat-map loss value, and the gradient with respect to
    the class map -- the golden holds dL/dp for p = sigmoid_hm(logits); the kernel returns dL/dz = dL/dp * p(1-p)."""
    from test_loss_golden import case_inputs, evaluator
    from monoflex_amd.structures.params_3d import make_train_target
    g = np.load(os.path.join(ROOT, "tests", "golden", "loss.npz"), allow_pickle=True)
    tg, cls, reg = case_inputs(name)
    p = cls.double()
    logits = torch.log(p / (1 - p)).float()                                    # sigmoid_hm(logits) == cls up to fp32 rounding
    z = logits.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_()
    ev = evaluator()
    preds = {"cls": cls.to(DEV), "reg": reg.to(DEV).requires_grad_(), "cls_logits_nhwc": z}
    loss_dict, _ = ev(preds, [make_train_target(t).to(DEV) for t in tg])
    ref = float(g["%s/loss/hm_loss" % name])
    assert abs(float(loss_dict["hm_loss"]) - ref) <= 3e-5 * max(1.0, abs(ref)), (float(loss_dict["hm_loss"]), ref)
    for k in loss_dict:                                                        # the other ten terms are untouched by the fused path
        r = float(g["%s/loss/%s" % (name, k)])
        assert abs(float(loss_dict[k]) - r) <= 3e-5 * max(1.0, abs(r)), k
    sum(loss_dict.values()).backward()
    dz = z.grad.detach().cpu().permute(0, 3, 1, 2).double().flatten()          # NCHW order like the golden's flat index
    idx = torch.as_tensor(g["%s/grad_cls_idx" % name])
    pp = p.flatten()[idx]
    want = torch.as_tensor(g["%s/grad_cls_samples" % name]).double() * pp * (1 - pp)
    inside = (pp > 1.0001e-4) & (pp < 1 - 1.0001e-4)                           # on the clamp the reference's autograd passes 0 too
    assert float((dz[idx][inside] - want[inside]).abs().max()) <= 2e-5 * max(1e-3, float(want.abs().max()))
    # and against the unfused torch path on the device (same inputs): total gradient mass
    cls_d = cls.to(DEV).requires_grad_()
    ld2, _ = ev({"cls": cls_d, "reg": reg.to(DEV)}, [make_train_target(t).to(DEV) for t in tg])
    ld2["hm_loss"].backward()
    full = (cls_d.grad.double() * p.to(DEV) * (1 - p.to(DEV))).cpu().flatten()
    assert float((dz - full).abs().max()) <= 2e-5 * max(1e-3, float(full.abs().max()))


@pytest.mark.parametrize("name", ["b2", "b3_empty_middle_mixed_calib", "b1_many"])
def test_fused_object_loss_kernel_vs_reference_golden(name):
    """mfx_object_loss (one wavefront per object, forward-mode gradient rows) against the fixture recorded from the reference's
    Loss_Computation: all eleven loss values, the logged means, and the gradient of the summed loss at the object centres."""
    from test_loss_golden import check_case, evaluator
    assert evaluator().fused_object_loss
    check_case(name, DEV)


def test_fused_object_loss_inside_a_wider_map_and_weighted_terms():
    """The 50 channels at an offset inside a 64-channel pixel (ld 64, ch_off 8), and backward with a different incoming gradient
    per term: against autograd of the tensor-op form on the same device."""
    from test_loss_golden import case_inputs, evaluator, TERM_NAMES
    from monoflex_amd import autograd as AG
    from monoflex_amd.structures.params_3d import make_train_target
    tg, cls, reg = case_inputs("b3_empty_middle_mixed_calib")
    ev = evaluator()
    targets = [make_train_target(t).to(DEV) for t in tg]
    heat, tv = ev.prepare_targets(targets, DEV)
    B, C, H, W = reg.shape
    wide = torch.randn(B, H, W, 64, device=DEV)
    wide[..., 8:58] = reg.to(DEV).permute(0, 2, 3, 1)
    wide.requires_grad_()
    terms, logged = AG.ObjectLossFn.apply(wide, tv["object_rows"], ev.object_loss_cfg(), 8)
    gout = torch.linspace(0.5, 1.5, len(TERM_NAMES), device=DEV)
    (terms * gout).sum().backward()
    assert float(wide.grad[..., :8].abs().max()) == 0 and float(wide.grad[..., 58:].abs().max()) == 0
    ev.fused_object_loss = False
    reg_d = reg.to(DEV).requires_grad_()
    loss_dict, logs = ev({"cls": cls.to(DEV), "reg": reg_d}, (heat, tv))
    sum(loss_dict[k] * gout[i] for i, k in enumerate(TERM_NAMES)).backward()
    for i, k in enumerate(TERM_NAMES):
        assert abs(float(terms[i]) - float(loss_dict[k])) <= 2e-5 * max(1.0, abs(float(loss_dict[k]))), k
    want = reg_d.grad.permute(0, 2, 3, 1)
    assert float((wide.grad[..., 8:58] - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    for k, v in zip(('depth_MAE', 'center_MAE', '02_MAE', '13_MAE', 'lower_MAE', 'hard_MAE', 'soft_MAE', 'mean_MAE'), logged[3:11].tolist()):
        assert abs(v - logs[k]) <= 1e-4 * max(1.0, abs(logs[k])), k


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("with_edge", [False, True])
def test_gram_heads_hip_node_equals_the_torch_node(dt, with_edge):
    """csrc/gram_heads.hip + GramRegHeadsHipFn (hand-written forward and backward: no torch op, no vendor GEMM inside the node) against
    GramRegHeadsFn (the torch-differentiated form, itself pinned to fp64 dense layers above and on the CPU): table, edge activation rows,
    running statistics and every gradient -- feature map, trunk weights, ABN weight / bias, 1x1 weights / biases -- on border / coincident /
    empty object rows and, `with_edge`, the 3d_offset branch's activation at edge-sequence pixels (duplicates included).
    Reference: model/head/detector_predictor.py:125-165."""
    from monoflex_amd import gram_heads as GH
    from monoflex_amd.model.head.detector_predictor import InPlaceABN
    dtype = DT[dt]
    g = torch.Generator().manual_seed(31)
    B, H, W, Cin, C, N = 2, 14, 22, 64, 256, 24
    ks, offs = (4, 2, 20, 3), (0, 4, 6, 26)
    rows = torch.zeros(N, 72)
    rows[:, 0] = (torch.rand(N, generator=g) > 0.2).float()
    rows[:, 57] = torch.randint(0, B, (N,), generator=g).float()
    rows[:, 2] = torch.randint(0, W, (N,), generator=g).float()
    rows[:, 3] = torch.randint(0, H, (N,), generator=g).float()
    rows[0, 2:4] = torch.tensor([0.0, 0.0]); rows[1, 2:4] = torch.tensor([W - 1.0, H - 1.0]); rows[3] = rows[5]
    rows[[0, 1, 3, 5], 0] = 1.0
    x = (torch.randn(B, H, W, Cin, generator=g) * 0.8 + 0.1).to(dtype)
    wt = [torch.randn(C, Cin, 3, 3, generator=g) / 24.0 for _ in ks]
    w2 = [torch.randn(k, C, 1, 1, generator=g) * 0.1 for k in ks]
    b2 = [torch.randn(k, generator=g) * 0.1 for k in ks]
    gam = [torch.rand(C, generator=g) + 0.5 for _ in ks]
    bet = [torch.randn(C, generator=g) * 0.3 for _ in ks]
    dout = torch.randn(N, 50, generator=g)
    extra = None
    if with_edge:                                              # border pixels of both images, with repeats (replicate padding of the edge sequence)
        pix = [b * H * W + y * W + 0 for b in range(B) for y in range(H)] + [b * H * W + 0 * W + xx for b in range(B) for xx in range(W)]
        extra = torch.tensor(pix + pix[:5] + [pix[-1]] * 3, dtype=torch.long)
        dact = torch.randn(extra.numel(), C, generator=g)
    res = {}
    for hip in (False, True):
        GH.HIP_NODE[0] = hip
        try:
            abns = []
            for i in range(len(ks)):
                h = InPlaceABN(C).to(DEV)
                with torch.no_grad():
                    h.weight.copy_(gam[i]); h.bias.copy_(bet[i])
                abns.append(h)
            xd = x.to(DEV).requires_grad_()
            wtd = [w.to(DEV).requires_grad_() for w in wt]
            wd = [w.to(DEV).requires_grad_() for w in w2]
            bd = [b.to(DEV).requires_grad_() for b in b2]
            out, ae = GH.gram_reg_heads(xd, rows.to(DEV), abns, offs, 50, wtd, [h.weight for h in abns], [h.bias for h in abns], wd, bd, sync=False,
                                        extra_branch=1 if with_edge else -1, extra_rows=extra.to(DEV) if with_edge else None)
            loss = (out * dout.to(DEV)).sum()
            if with_edge:
                loss = loss + (ae.float() * dact.to(DEV)).sum()
            loss.backward()
            torch.cuda.synchronize()
            res[hip] = dict(out=out.detach().float().cpu(), ae=None if ae is None else ae.detach().float().cpu(), dx=xd.grad.float().cpu(),
                            dw=[w.grad.cpu() for w in wtd], dg=[h.weight.grad.cpu() for h in abns], db=[h.bias.grad.cpu() for h in abns],
                            dw2=[w.grad.cpu() for w in wd], db2=[b.grad.cpu() for b in bd],
                            rm=[h.running_mean.cpu() for h in abns], rv=[h.running_var.cpu() for h in abns], nbt=[int(h.num_batches_tracked) for h in abns])
        finally:
            GH.HIP_NODE[0] = True
    a, b_ = res[True], res[False]
    assert _rel(a["out"], b_["out"]) < 2e-3, ("out", _rel(a["out"], b_["out"]))
    if with_edge:
        assert _rel(a["ae"], b_["ae"]) < 1.5e-2, ("act_e", _rel(a["ae"], b_["ae"]))          # (both rounded to the 16-bit type)
    assert _rel(a["dx"], b_["dx"]) < 2e-2, ("dx", _rel(a["dx"], b_["dx"]))
    for i in range(len(ks)):
        for key, tol in (("dw", 1e-2), ("dg", 5e-3), ("db", 5e-3), ("dw2", 2e-3), ("db2", 1e-4), ("rm", 1e-4), ("rv", 1e-3)):
            assert _rel(a[key][i], b_[key][i]) < tol, (key, i, _rel(a[key][i], b_[key][i]))
    assert a["nbt"] == b_["nbt"] == [1] * len(ks)
