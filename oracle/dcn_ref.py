"""oracle/dcn_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/dcn_v2_ref.c (the plain-C restatement of the
reference DCNv2 CPU path) exposed with the call signatures of the reference's
`_ext` module (model/backbone/DCNv2/src/vision.cpp:3-8, src/dcn_v2.h:9-23,48-59)
and of `dcn_v2.py`'s autograd Function (model/backbone/DCNv2/dcn_v2.py:16-54).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libdcn_v2_ref.so")
_lib = None


def build(force=False):
    """Compile oracle/dcn_v2_ref.c with gcc (see oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "dcn_v2_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.c_void_p
        i = ctypes.c_int
        _lib.dcn_ref_forward.argtypes = [fp] * 6 + [i] * 14
        _lib.dcn_ref_forward.restype = i
        _lib.dcn_ref_backward.argtypes = [fp] * 11 + [i] * 14
        _lib.dcn_ref_backward.restype = i
        _lib.dcn_ref_im2col_image.argtypes = [fp] * 4 + [i] * 12
        _lib.dcn_ref_im2col_image.restype = i
    return _lib


def _f32c(t):
    assert t.device.type == "cpu"
    return t.detach().to(torch.float32).contiguous()


def _out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw):
    return ((H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1,
            (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1)


def dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w,
                   pad_h, pad_w, dilation_h, dilation_w, deformable_group, use_torch_gemm=True):
    """Same contract as reference `_ext.dcn_v2_forward` (src/dcn_v2.h:9-46), CPU fp32.

    use_torch_gemm=True mirrors dcn_v2_cpu.cpp literally: C im2col per image, BLAS GEMM
    (here torch.addmm) -- fast enough for full-size layers.  False runs the all-C path.
    """
    x, w, b, off, msk = map(_f32c, (input, weight, bias, offset, mask))
    B, C, H, W = x.shape
    Cout = w.shape[0]
    assert w.shape[1] == C and w.shape[2] == kernel_h and w.shape[3] == kernel_w
    Ho, Wo = _out_hw(H, W, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w)
    out = torch.empty(B, Cout, Ho, Wo, dtype=torch.float32)
    L = lib()
    if not use_torch_gemm:
        rc = L.dcn_ref_forward(x.data_ptr(), w.data_ptr(), b.data_ptr(), off.data_ptr(), msk.data_ptr(),
                               out.data_ptr(), B, C, H, W, Cout, kernel_h, kernel_w, stride_h, stride_w,
                               pad_h, pad_w, dilation_h, dilation_w, deformable_group)
        assert rc == 0
        return out
    K = C * kernel_h * kernel_w
    cols = torch.empty(K, Ho * Wo, dtype=torch.float32)
    w2 = w.view(Cout, K)
    for n in range(B):
        L.dcn_ref_im2col_image(x[n].data_ptr(), off[n].data_ptr(), msk[n].data_ptr(), cols.data_ptr(),
                               C, H, W, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                               dilation_h, dilation_w, deformable_group)
        # dcn_v2_cpu.cpp:82-85 bias broadcast, :101-104 out += W * columns
        out[n] = torch.addmm(b.view(Cout, 1).expand(Cout, Ho * Wo), w2, cols).view(Cout, Ho, Wo)
    return out


def dcn_v2_backward(input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h, stride_w,
                    pad_h, pad_w, dilation_h, dilation_w, deformable_group):
    """Same contract as reference `_ext.dcn_v2_backward` (src/dcn_v2.h:48-92)."""
    x, w, b, off, msk, go = map(_f32c, (input, weight, bias, offset, mask, grad_output))
    B, C, H, W = x.shape
    Cout = w.shape[0]
    gi, gw, gb = torch.zeros_like(x), torch.zeros_like(w), torch.zeros_like(b)
    goff, gm = torch.zeros_like(off), torch.zeros_like(msk)
    rc = lib().dcn_ref_backward(x.data_ptr(), w.data_ptr(), b.data_ptr(), off.data_ptr(), msk.data_ptr(),
                                go.data_ptr(), gi.data_ptr(), goff.data_ptr(), gm.data_ptr(),
                                gw.data_ptr(), gb.data_ptr(), B, C, H, W, Cout, kernel_h, kernel_w,
                                stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, deformable_group)
    assert rc == 0
    return [gi, goff, gm, gw, gb]


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class _DCNv2Ref(torch.autograd.Function):
    """Mirror of reference `_DCNv2` (dcn_v2.py:16-51) on the C oracle."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
        ctx.cfg = (_pair(weight.shape[2:4]), _pair(stride), _pair(padding), _pair(dilation), deformable_groups)
        (kh, kw), (sh, sw), (ph, pw), (dh, dw), dg = ctx.cfg
        out = dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg)
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        (kh, kw), (sh, sw), (ph, pw), (dh, dw), dg = ctx.cfg
        gi, goff, gm, gw, gb = dcn_v2_backward(input, weight, bias, offset, mask, grad_output,
                                               kh, kw, sh, sw, ph, pw, dh, dw, dg)
        return gi, goff, gm, gw, gb, None, None, None, None


dcn_v2_conv = _DCNv2Ref.apply


def dcn_v2_torch(input, offset, mask, weight, bias, stride=1, padding=1, dilation=1):
    """Independent pure-torch formulation (explicit integer-corner gather), deformable_groups=1.

    Used to cross-check the C restatement (two independently written forms of
    dcn_v2_im2col_cpu.cpp:127-196 must agree) and differentiable through autograd.
    """
    B, C, H, W = input.shape
    Cout, _, kh, kw = weight.shape
    sh, sw = _pair(stride); ph, pw = _pair(padding); dh, dw = _pair(dilation)
    Ho, Wo = _out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
    ys = (torch.arange(Ho, dtype=input.dtype) * sh - ph).view(1, Ho, 1)
    xs = (torch.arange(Wo, dtype=input.dtype) * sw - pw).view(1, 1, Wo)
    flat = input.reshape(B, C, H * W)
    cols = []
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            h = ys + i * dh + offset[:, 2 * k]
            w = xs + j * dw + offset[:, 2 * k + 1]
            inside = (h > -1) & (w > -1) & (h < H) & (w < W)
            h0 = torch.floor(h); w0 = torch.floor(w)
            lh = h - h0; lw = w - w0
            val = 0
            for (hc, wc, wt) in ((h0, w0, (1 - lh) * (1 - lw)), (h0, w0 + 1, (1 - lh) * lw),
                                 (h0 + 1, w0, lh * (1 - lw)), (h0 + 1, w0 + 1, lh * lw)):
                ok = inside & (hc >= 0) & (hc <= H - 1) & (wc >= 0) & (wc <= W - 1)
                idx = (hc.clamp(0, H - 1) * W + wc.clamp(0, W - 1)).long().view(B, 1, Ho * Wo).expand(B, C, Ho * Wo)
                g = flat.gather(2, idx).view(B, C, Ho, Wo)
                val = val + g * (wt * ok.to(input.dtype)).unsqueeze(1)
            cols.append(val * mask[:, k].unsqueeze(1))
    col = torch.stack(cols, dim=2)                      # B, C, kh*kw, Ho, Wo
    out = torch.einsum("ock,bckhw->bohw", weight.view(Cout, C, kh * kw), col)
    return out + bias.view(1, Cout, 1, 1)


def dcn_v2_grid_sample(x, off, msk, w, b, pad=1):
    """Modulated deformable 3x3 / stride-1 convolution with torch.nn.functional.grid_sample as the sampler: the bilinear rule (zero beyond the map, corner
    by corner) and its input / coordinate derivatives are the LIBRARY's, not a restatement written for this repository.  Channel 2k of `off` moves tap k
    along y, 2k + 1 along x (dcn_v2_im2col_cpu.cpp:150-161)."""
    B, C, H, W = x.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=x.dtype), torch.arange(W, dtype=x.dtype), indexing="ij")
    out = b.view(1, -1, 1, 1).expand(B, w.shape[0], H, W).clone()
    for k in range(9):
        i, j = divmod(k, 3)
        py = ys + (i - pad) + off[:, 2 * k]
        px = xs + (j - pad) + off[:, 2 * k + 1]
        grid = torch.stack((2 * px / (W - 1) - 1, 2 * py / (H - 1) - 1), dim=-1)
        smp = torch.nn.functional.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True) * msk[:, k:k + 1]
        out = out + torch.einsum("oc,bchw->bohw", w[:, :, i, j], smp)
    return out
