"""Tensor-level wrappers over the C ABI (include/monoflex_hip.h).

Activations are torch CUDA tensors in physical NHWC layout, shape (B, H, W, C), dtype float32
(parity mode) or bfloat16 (perf mode).  torch is used for device memory and the current HIP stream
only; every arithmetic op below runs in libmonoflex_hip.so.  Weight packing (done once per
`prepare`) uses torch indexing ops -- it is not on the hot path.
"""
import ctypes
import os
import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import lib as L


def _stream():
    """The current HIP stream of the CURRENT device: every entry point below runs under `on_tensor_device`, which makes the
    operands' device current first, so a model on cuda:N launches on cuda:N's stream whatever the caller's current device is."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _first_cuda_device(args):
    for a in args:
        if torch.is_tensor(a):
            if a.is_cuda:
                return a.device
        elif isinstance(a, (list, tuple)):
            d = _first_cuda_device(a)
            if d is not None:
                return d
    return None


def on_tensor_device(fn):
    """Run `fn` with the device of its first CUDA tensor argument current (kernels, streams and scratch buffers are all
    looked up through the current device); other CUDA operands must live on the same device."""
    import functools

    @functools.wraps(fn)
    def run(*args, **kwargs):
        dev = _first_cuda_device(args) or _first_cuda_device(tuple(kwargs.values()))
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return run


def _dt(dtype):
    if dtype == torch.float32:
        return L.MFX_F32
    if dtype == torch.bfloat16:
        return L.MFX_BF16
    if dtype == torch.float16:
        return L.MFX_F16
    raise TypeError("MonoFlex HIP kernels take float32, bfloat16 or float16, got %s" % dtype)


# Compute tag of the split-precision mode (include/monoflex_hip.h MFX_F16X2): activations are ordinary float32 tensors, the GEMM
# kernels (conv2d / cat_conv1x1 / dcn / heads_fused) multiply fp16 (hi, lo) operand pairs.  The tag travels with the PACKED WEIGHTS
# (`pack_*(…, dtype=F16X2)` -> `.split`), which is what selects the kernel; modules learn it from `compute_tag`.
F16X2 = "f16x2"


def compute_tag(module, dtype):
    """The pack / kernel tag a module uses for activations of `dtype`: F16X2 when the module was switched to the split-precision
    mode (KeypointDetector.set_compute_dtype("fp16x2") marks every sub-module) and the activations are fp32."""
    if dtype == torch.float32 and module.__dict__.get("_mfx_split", False):
        return F16X2
    return dtype


def storage_dtype(dtype):
    return torch.float32 if dtype == F16X2 else dtype


def _elems(dtype):
    return 4 if (dtype == torch.float32 or dtype == F16X2) else 8


def split_chunks(w):
    """fp32 tensor (element count a multiple of 4, chunks of 4 consecutive values) -> the same shape, float32-TYPED, every 16-byte
    chunk holding [4 hi halves | 4 lo halves] of its 4 values: hi = fp16(x), lo = fp16(x - hi) (csrc/common.h f32s_t)."""
    w = w.detach().float().contiguous()
    c = w.view(-1, 4)
    hi = c.half()
    lo = (c - hi.float()).half()
    return torch.cat((hi, lo), 1).contiguous().view(torch.float32).view(w.shape)


def cast_operand(w, dtype):
    """Weights as the kernels of compute tag `dtype` read them."""
    return split_chunks(w) if dtype == F16X2 else w.to(dtype)


def pair_steps(x, dim):
    """Split-precision fragment-major weights for the kernels that walk K in step PAIRS (csrc/heads.hip): `x` is float32-typed with
    16-byte chunks [hi hi | lo lo] (dwords) in its last axis and the K step on axis `dim`; the result replaces that axis by
    [pair][hi | lo] and every chunk by [its dwords of step 2p | of step 2p+1]: one 8-element fp16 MFMA operand of hi (lo) halves."""
    sh = list(x.shape)
    dim = dim % len(sh)
    assert sh[-1] == 4 and sh[dim] % 2 == 0
    lead, mid = sh[:dim], sh[dim + 1:-1]
    nl, nm = len(lead), len(mid)
    v = x.reshape(*lead, sh[dim] // 2, 2, *mid, 2, 2)                  # [.., pair, step in pair, mid.., hi/lo, dword]
    perm = list(range(nl)) + [nl, nl + 2 + nm] + [nl + 2 + i for i in range(nm)] + [nl + 1, nl + 3 + nm]
    return v.permute(*perm).contiguous().view(*lead, sh[dim] // 2, 2, *mid, 4)


def split_weight_scale(w):
    """Power of two s (python float) that brings max |w| * s into [2^11, 2^12): the lo halves of the scaled weights are then normal
    fp16 numbers down to |w| = 2^-14 of the largest one (unscaled, every lo half of a |w| < 0.25 weight is an fp16 SUBNORMAL, i.e.
    carries a 3e-8 absolute error -- ~1e-6 relative on DLA-34's weights, the largest error term of the split mode).  The kernels'
    epilogues undo it exactly: pack_* fold 1/s into the per-channel `scale`."""
    m = float(w.detach().abs().max())
    if not (m > 0.0) or not math.isfinite(m):
        return 1.0
    return 2.0 ** (11 - math.floor(math.log2(m)))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _need_cuda(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("MonoFlex HIP operator called with a CPU tensor: the product path has no CPU "
                               "fallback (the CPU oracle lives in oracle/ and is test-only)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("MonoFlex HIP operator: operands on different devices (%s and %s)" % (dev, t.device))


def _pow2(v):
    return v > 0 and (v & (v - 1)) == 0


def _round_up(v, m):
    return (v + m - 1) // m * m


def cout_pad(c):
    return 16 if c <= 16 else 32 if c <= 32 else _round_up(c, 64)


# --------------------------------------------------------------------------------------------
# packed parameter containers
# --------------------------------------------------------------------------------------------
@dataclass
class PackedConv:
    w: torch.Tensor                 # [Cout_pad][K_pad]
    scale: Optional[torch.Tensor]   # fp32 [Cout_pad]
    shift: Optional[torch.Tensor]
    kh: int
    kw: int
    stride: int
    pad_h: int
    pad_w: int
    dil_w: int
    Ck: int
    Cout: int
    Cout_pad: int
    K_pad: int
    act: int
    w_frag: Optional[torch.Tensor] = None   # fragment-major copy for the LDS-halo kernel (3x3 / stride 1)
    w_frag_f16: Optional[torch.Tensor] = None   # same, IEEE fp16 (DCN LDS-patch kernel, bf16 mode)
    w_frag_pair: Optional[torch.Tensor] = None  # split precision: w_frag with its K steps paired (pair_steps; mfx_conv_desc.w_frag_pair)
    w_pair_f16: Optional[torch.Tensor] = None   # IEEE fp16, tap-pair K order of the fourth-generation DCN kernel (dcn_pair_fragments; mfx_dcn_desc.w_pair_f16)
    ps: Optional["PackedConv"] = None           # DCN as project-then-sample: the same weights as ONE 1x1 conv C -> 9*Cout (rows (tap, n)); built on first use (dcn_ps_pack)
    split: bool = False                     # split-precision operands (F16X2): fp32 activations, MFX_F16X2 kernels


def fragment_major(w2d, dtype):
    """[Cout_pad][K_pad] -> [Cout_pad/16][K_pad/(4E)][4 kq][16 n][E]: one MFMA weight fragment (16 rows x 64 bytes of K)
    per contiguous KiB, lane (kq*16 + n) owning 16 bytes."""
    E = _elems(dtype)
    N, K = w2d.shape
    assert N % 16 == 0 and K % (4 * E) == 0
    return w2d.view(N // 16, 16, K // (4 * E), 4, E).permute(0, 2, 3, 1, 4).contiguous()


def fold_bn(bn, conv_bias=None, cout_padded=None):
    """Eval-mode BatchNorm as y = x*scale + shift (a preceding conv bias folded in)."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    if conv_bias is not None:
        shift = shift + conv_bias.detach().float() * scale
    if cout_padded is not None and cout_padded > scale.numel():
        padn = cout_padded - scale.numel()
        scale = torch.cat((scale, scale.new_ones(padn)))
        shift = torch.cat((shift, shift.new_zeros(padn)))
    return scale.contiguous(), shift.contiguous()


def _pad_rows_cols(w2d, rows, cols):
    out = w2d.new_zeros(rows, cols)
    out[:w2d.shape[0], :w2d.shape[1]] = w2d
    return out


def pack_conv(weight, dtype, scale=None, shift=None, stride=1, pad=0, act=L.ACT_NONE, cout=None):
    """weight (Cout,Cin,kh,kw) -> K-contiguous [Cout_pad][K_pad], K = (tap, channel)."""
    Cout, Cin, kh, kw = weight.shape
    if not _pow2(Cin) or Cin < _elems(dtype):
        raise ValueError("pack_conv: Cin must be a power of two >= %d (got %d)" % (_elems(dtype), Cin))
    K = kh * kw * Cin
    # 128 bytes of K lets the kernel run 64- or 128-byte k-iterations; a K of 64 bytes (the 32 -> 64 1x1 "project" conv of DLA level 2 in 16-bit modes) stays at
    # 64: padded to 128 the generic kernel loaded every row twice as wide as it is (19.3 -> 13.4 us at 8 x 96 x 320, tools/pointwise_bench.py)
    bk = 8 * _elems(dtype) if K >= 8 * _elems(dtype) else 4 * _elems(dtype)
    K_pad = _round_up(K, bk)
    cout = Cout if cout is None else cout
    cp = cout_pad(cout)
    w2 = weight.detach().float().permute(0, 2, 3, 1).reshape(Cout, K)
    if dtype == F16X2:
        ws = split_weight_scale(w2)
        w2 = w2 * ws
        scale = (scale.detach().float() if scale is not None else torch.ones(Cout, device=weight.device)) / ws
    w2 = cast_operand(_pad_rows_cols(w2, cp, K_pad), dtype).contiguous()

    def padv(v, fill):
        if v is None:
            return None
        v = v.detach().float()
        if v.numel() < cp:
            v = torch.cat((v, v.new_full((cp - v.numel(),), fill)))
        return v.contiguous()
    wf = fragment_major(w2, dtype) if (kh == 3 and kw == 3 and stride in (1, 2) and pad == 1) else None
    pk = PackedConv(w2, padv(scale, 1.0), padv(shift, 0.0), kh, kw, stride, pad, pad, 1, Cin, cout, cp, K_pad, act, wf, split=dtype == F16X2)
    if wf is not None and dtype == F16X2 and Cin >= 32:
        pk.w_frag_pair = pair_steps(wf, 1)
    return pk


# stem geometry: zero-padded NHWC4 image, 3 columns left / 5 right, 3 rows top/bottom
STEM_PAD_H, STEM_PAD_WL, STEM_PAD_WR = 3, 3, 5


def pack_stem(weight, dtype, scale, shift, act=L.ACT_RELU):
    """7x7/s1/p3 conv on 3 channels (dla_dcn.py:268-272) over the padded NHWC4 image.
    bf16: a 16-byte chunk is 2 adjacent pixels x 4 ch -> 7 x 4 'super taps' of 8 elements, dil_w = 2.
    f32 : a chunk is 1 pixel x 4 ch -> 7 x 7 taps of 4 elements."""
    Cout = weight.shape[0]
    w = weight.detach().float()
    w4 = torch.cat((w, w.new_zeros(Cout, 1, 7, 7)), dim=1)              # (Cout,4,7,7)
    if dtype == F16X2 and Cout == 16:
        # split precision, dedicated kernel (csrc/stem.hip stem_conv7x7_split_kernel): the fp16 super-tap matrix twice -- hi halves, lo halves
        w8 = torch.cat((w4, w4.new_zeros(Cout, 4, 7, 1)), dim=3)
        wp = w8.permute(0, 2, 3, 1).reshape(Cout, 7, 4, 2, 4).reshape(Cout, 7 * 4 * 8)
        ws = split_weight_scale(wp)
        wp = wp * ws
        hi = wp.half()
        lo = (wp - hi.float()).half()
        return PackedConv(torch.cat((hi, lo), 0).contiguous(), (scale.detach().float() / ws).contiguous(), shift.contiguous(), 7, 4, 1, 0, 0, 2, 8, Cout,
                          cout_pad(Cout), wp.shape[1], act, split=True)
    if dtype in (torch.bfloat16, torch.float16):
        w8 = torch.cat((w4, w4.new_zeros(Cout, 4, 7, 1)), dim=3)        # kw 7 -> 8
        # [n][th][j][u][c] with kw = 2j+u
        wp = w8.permute(0, 2, 3, 1).reshape(Cout, 7, 4, 2, 4).reshape(Cout, 7 * 4 * 8)
        kh, kw, Ck, dil = 7, 4, 8, 2
    else:
        wp = w4.permute(0, 2, 3, 1).reshape(Cout, 7 * 7 * 4)
        kh, kw, Ck, dil = 7, 7, 4, 1
    bk = 8 * _elems(dtype)
    K_pad = _round_up(wp.shape[1], bk)
    cp = cout_pad(Cout)
    if dtype == F16X2:
        ws = split_weight_scale(wp)
        wp, scale = wp * ws, scale.detach().float() / ws
    wp = cast_operand(_pad_rows_cols(wp, cp, K_pad), dtype).contiguous()
    return PackedConv(wp, scale.contiguous(), shift.contiguous(), kh, kw, 1, 0, 0, dil, Ck, Cout, cp, K_pad, act, split=dtype == F16X2)


# --------------------------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------------------------
@on_tensor_device
def conv2d(x, p: PackedConv, res=None, out_dtype=None, rowmap=None, x_channels=None, x_ch_off=0,
           out_hw=None, in_hw=None, stats=None):
    """y = act(conv(x)*scale + shift (+res)).  x: (B,H,W,Cx) NHWC.  With `rowmap` (int32 [M], pixel
    indices into the (B,Ho,Wo) grid, -1 = zero row) the output is the dense (M, Cout) row list."""
    _need_cuda(x, res, rowmap)
    B, H, W, Cx = x.shape
    if in_hw is not None:
        H, W = in_hw
    out_dtype = out_dtype or x.dtype
    if out_hw is None:
        Ho = (H + 2 * p.pad_h - ((p.kh - 1) + 1)) // p.stride + 1
        Wo = (W + 2 * p.pad_w - (p.dil_w * (p.kw - 1) + 1)) // p.stride + 1
    else:
        Ho, Wo = out_hw
    M = B * Ho * Wo if rowmap is None else rowmap.numel()
    y = torch.empty((B, Ho, Wo, p.Cout) if rowmap is None else (M, p.Cout), dtype=out_dtype, device=x.device)
    d = L.ConvDesc()
    d.x = x.data_ptr() + x_ch_off * x.element_size()
    d.w, d.scale, d.shift = p.w.data_ptr(), (p.scale.data_ptr() if p.scale is not None else None), \
        (p.shift.data_ptr() if p.shift is not None else None)
    d.w_frag = p.w_frag.data_ptr() if p.w_frag is not None else None
    d.w_frag_pair = p.w_frag_pair.data_ptr() if p.w_frag_pair is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.y = y.data_ptr()
    d.rowmap = rowmap.data_ptr() if rowmap is not None else None
    d.B, d.H, d.W, d.x_pixstride, d.Ck = B, H, W, Cx, p.Ck
    d.kh, d.kw, d.stride, d.pad_h, d.pad_w, d.dil_w = p.kh, p.kw, p.stride, p.pad_h, p.pad_w, p.dil_w
    d.Ho, d.Wo, d.M, d.Cout, d.Cout_pad, d.K_pad = Ho, Wo, M, p.Cout, p.Cout_pad, p.K_pad
    d.ldy, d.ldres = p.Cout, (res.shape[-1] if res is not None else 0)
    d.act, d.dtype, d.out_dtype = p.act, (L.MFX_F16X2 if p.split else _dt(x.dtype)), _dt(out_dtype)
    if rowmap is None and M * p.Cout_pad <= SPLITK_MAX_ELEMS and p.K_pad * x.element_size() >= 2048:
        ws = _splitk_workspace(x.device)                      # small-M / long-K layers: lets the library split K
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if stats is not None:
        # train-mode BN statistics of y accumulated by the conv's epilogue into the BN layer's scratch (where the kernel that
        # runs supports it: conv2d.last_stats_done tells the caller whether a statistics pass is still needed)
        done = ctypes.c_int(0)
        d.stats, d.stats_ncopy, d.stats_done = stats.data_ptr(), L.load().mfx_bn_ncopy(p.Cout), ctypes.pointer(done)
        L.check(L.load().mfx_conv2d_nhwc(ctypes.byref(d), _stream()), "mfx_conv2d_nhwc")
        conv2d.last_stats_done = bool(done.value)
        return y
    L.check(L.load().mfx_conv2d_nhwc(ctypes.byref(d), _stream()), "mfx_conv2d_nhwc")
    return y


conv2d.last_stats_done = False


@dataclass
class PackedCat:
    w: torch.Tensor
    scale: torch.Tensor
    shift: torch.Tensor
    Cseg: int
    Cout: int
    Cout_pad: int
    K_pad: int
    act: int
    split: bool = False


def pack_cat(weight, dtype, scale, shift, src_channels, act=L.ACT_RELU):
    Cout, Ctot = weight.shape[:2]
    assert sum(src_channels) == Ctot
    Cseg = min(src_channels)
    assert all(c % Cseg == 0 for c in src_channels) and _pow2(Cseg)
    cp = cout_pad(Cout)
    w2 = weight.detach().float().reshape(Cout, Ctot)
    if dtype == F16X2:
        ws = split_weight_scale(w2)
        w2, scale = w2 * ws, scale.detach().float() / ws
    w2 = cast_operand(_pad_rows_cols(w2, cp, Ctot), dtype).contiguous()
    return PackedCat(w2, scale.contiguous(), shift.contiguous(), Cseg, Cout, cp, Ctot, act, split=dtype == F16X2)


@on_tensor_device
def cat_conv1x1(srcs, p: PackedCat):
    """Root: 1x1 conv over the virtual concat of `srcs` (list of (B,H,W,Ci) tensors)."""
    _need_cuda(*srcs)
    B, H, W, _ = srcs[0].shape
    y = torch.empty((B, H, W, p.Cout), dtype=srcs[0].dtype, device=srcs[0].device)
    d = L.CatDesc()
    n = 0
    for s in srcs:
        C = s.shape[3]
        for part in range(C // p.Cseg):
            d.src[n], d.stride[n], d.off[n] = s.data_ptr(), C, part * p.Cseg
            n += 1
    d.nseg, d.Cseg = n, p.Cseg
    d.w, d.res, d.y = p.w.data_ptr(), None, y.data_ptr()
    d.scale = p.scale.data_ptr() if p.scale is not None else None
    d.shift = p.shift.data_ptr() if p.shift is not None else None
    d.M, d.Cout, d.Cout_pad, d.K_pad, d.ldy, d.ldres, d.act = B * H * W, p.Cout, p.Cout_pad, p.K_pad, p.Cout, 0, p.act
    d.dtype = L.MFX_F16X2 if p.split else _dt(y.dtype)
    L.check(L.load().mfx_cat_conv1x1_nhwc(ctypes.byref(d), _stream()), "mfx_cat_conv1x1_nhwc")
    return y


def dcn_pair_fragments(weight, cout_pad):
    """(Cout, Cin, 3, 3) -> the fourth-generation DCN kernel's fp16 weights (csrc/dcn_lds.hip, mfx_dcn_desc.w_pair_f16): the input channels in slices
    of 16, five MFMA k-steps (K = 32) per slice, k-step j = taps (2j, 2j + 1) x the slice's 16 channels (the tenth tap is zeros); fragment-major
    [cout_pad / 16][Cin / 16 * 5][4 kq][16 n][8]: lane (kq, n) holds tap 2j + (kq >> 1), channels 16 s + 8 (kq & 1) .. + 7 of output channel 16 nf + n."""
    Cout, Cin, kh, kw = weight.shape
    assert kh == 3 and kw == 3 and Cin % 16 == 0 and cout_pad % 16 == 0
    w = weight.detach().float().permute(0, 2, 3, 1).reshape(Cout, 9, Cin)
    wp = w.new_zeros(cout_pad, 10, Cin)
    wp[:Cout, :9] = w
    wp = wp.view(cout_pad, 5, 2, Cin // 16, 2, 8).permute(0, 3, 1, 2, 4, 5)      # (n, slice, step, tap parity, channel half, 8)
    w2d = wp.reshape(cout_pad, (Cin // 16) * 5 * 32).to(torch.float16).contiguous()
    return fragment_major(w2d, torch.float16)


def add_f16_fragments(p: PackedConv, weight):
    """Attach the fp16 fragment-major weights the DCN LDS-patch kernels multiply with (bf16 / fp16 mode, 3x3/s1/p1; split precision: the (hi, lo) halves
    of the scaled weights as two consecutive arrays each -- csrc/dcn_lds.hip dcn_lds_split_kernel)."""
    if p.split and p.kh == 3 and p.kw == 3 and p.Cout_pad == 64 and weight.shape[1] % 32 == 0 and p.K_pad == 9 * weight.shape[1]:
        Cout, Cin, kh, kw = weight.shape
        w = weight.detach().float().to(p.w.device)
        ws = split_weight_scale(w.permute(0, 2, 3, 1).reshape(Cout, -1))        # the scale pack_conv folded into p.scale (same matrix, same maximum)
        w = w * ws
        hi = w.half()
        lo = (w - hi.float()).half()
        p.w_pair_f16 = torch.cat((dcn_pair_fragments(hi.float(), p.Cout_pad).reshape(-1), dcn_pair_fragments(lo.float(), p.Cout_pad).reshape(-1))).contiguous()

        def frag(h):
            return fragment_major(_pad_rows_cols(h.float().permute(0, 2, 3, 1).reshape(Cout, -1), p.Cout_pad, p.K_pad).half().contiguous(), torch.float16).reshape(-1)
        p.w_frag_f16 = torch.cat((frag(hi), frag(lo))).contiguous()
        return p
    if p.w_frag is not None and p.w.dtype in (torch.float16, torch.bfloat16) and p.kh == 3 and p.kw == 3 and p.Cout_pad == 64 \
            and weight.shape[1] % 16 == 0 and not p.split:
        p.w_pair_f16 = dcn_pair_fragments(weight.to(p.w.device), p.Cout_pad)
    if p.w_frag is not None and p.w.dtype == torch.float16:
        p.w_frag_f16 = p.w_frag                                  # fp16 mode: the fragments already are IEEE fp16
    elif p.w_frag is not None and p.w.dtype == torch.bfloat16:
        Cout, Cin, kh, kw = weight.shape
        w2 = _pad_rows_cols(weight.detach().float().permute(0, 2, 3, 1).reshape(Cout, kh * kw * Cin), p.Cout_pad, p.K_pad)
        p.w_frag_f16 = fragment_major(w2.to(device=p.w.device, dtype=torch.float16).contiguous(), torch.float16)
    return p


def _dcn_desc(x, offmask, p: PackedConv, y, off: Optional[PackedConv] = None, offmask_out=None):
    B, H, W, C = x.shape
    Ho, Wo = y.shape[1], y.shape[2]
    d = L.DcnDesc()
    d.x, d.w, d.y = x.data_ptr(), p.w.data_ptr(), y.data_ptr()
    d.offmask = offmask.data_ptr() if offmask is not None else None
    d.w_frag = p.w_frag.data_ptr() if p.w_frag is not None else None
    d.w_frag_f16 = p.w_frag_f16.data_ptr() if p.w_frag_f16 is not None else None
    d.w_pair_f16 = p.w_pair_f16.data_ptr() if p.w_pair_f16 is not None else None
    d.scale = p.scale.data_ptr() if p.scale is not None else None
    d.shift = p.shift.data_ptr() if p.shift is not None else None
    d.B, d.H, d.W, d.C = B, H, W, C
    d.kh, d.kw, d.stride, d.pad, d.dil = p.kh, p.kw, p.stride, p.pad_h, p.dil_w
    d.Ho, d.Wo, d.Cout, d.Cout_pad, d.K_pad, d.ldy, d.act = Ho, Wo, p.Cout, p.Cout_pad, p.K_pad, p.Cout, p.act
    d.dtype = L.MFX_F16X2 if p.split else _dt(x.dtype)
    if off is not None and off.w_frag_f16 is not None and off.shift is not None and off.Cout_pad == 32 and off.K_pad == 9 * C:
        d.off_w_frag_f16, d.off_shift = off.w_frag_f16.data_ptr(), off.shift.data_ptr()
        d.offmask_out = offmask_out.data_ptr() if offmask_out is not None else None
    return d


@on_tensor_device
def dcn(x, offmask, p: PackedConv):
    """Fused DCNv2 + scale/shift + act.  x (B,H,W,C) NHWC, offmask fp32 (B,Ho,Wo,32)."""
    _need_cuda(x, offmask)
    B = x.shape[0]
    Ho, Wo = offmask.shape[1], offmask.shape[2]
    y = torch.empty((B, Ho, Wo, p.Cout), dtype=x.dtype, device=x.device)
    d = _dcn_desc(x, offmask, p, y)
    if B * Ho * Wo * p.Cout_pad <= SPLITK_MAX_ELEMS:           # small maps: lets the library split K over workgroups
        ws = _splitk_workspace(x.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    L.check(L.load().mfx_dcn_nhwc(ctypes.byref(d), _stream()), "mfx_dcn_nhwc")
    return y


# DCN as "project, then sample" (csrc/dcn_ps.hip): which layers take it.  The projected map holds 9 * Cout values per pixel and is written and read back;
# measured at B = 8 (profiles/r06_dcn_ps.md): it beats the fused gather kernels where that map is <= 18 MB (512 -> 256 @ 12 x 40: 49 -> 39 us,
# 256 -> 64 @ 24 x 80: 30 -> 24 us) and loses from 35 MB up (the projection GEMM is bound by writing the map: 70.8 MB take 33-47 us with this library's
# 1x1 kernel AND with the vendor's GEMM), so the byte limit below keeps it to the two small-map channel-reducing modules.
DCN_PS = [os.environ.get("MFX_DCN_PS", "1") != "0"]      # MFX_DCN_PS=0: every layer on the fused gather kernels (A/B)
DCN_PS_MAX_BYTES = [int(os.environ.get("MFX_DCN_PS_MAX_MB", "24")) << 20]
PROJECT_AS = [os.environ.get("MFX_PROJECT_AS", "1") != "0"]      # the projection on csrc/gemm_as.hip (0: mfx_conv2d_nhwc's 1x1 kernel)


def dcn_ps_pack(p: PackedConv):
    """The DCN weights [Cout][(tap, c)] as a 1x1 conv C -> 9 * Cout whose output row is [(tap, n)] (no scale / shift / activation: those follow the sampling)."""
    if p.ps is None:
        C = p.K_pad // 9
        w = p.w[:p.Cout].view(p.Cout, 9, C).permute(1, 0, 2).reshape(9 * p.Cout, C, 1, 1)
        p.ps = pack_conv(w, p.w.dtype, None, None, stride=1, pad=0, act=L.ACT_NONE)
    return p.ps


def dcn_ps_applies(x, p: PackedConv):
    B, H, W, C = x.shape
    if getattr(p, "transient", False):                         # training: the projection operand would be re-packed (six launches) every step
        return False
    return (DCN_PS[0] and x.dtype in (torch.bfloat16, torch.float16) and not p.split and p.kh == 3 and p.kw == 3 and p.stride == 1 and p.pad_h == 1
            and p.dil_w == 1 and p.K_pad == 9 * C and C >= 128 and p.Cout == p.Cout_pad and p.Cout in (64, 128, 256)
            and B * H * W * 9 * p.Cout * 2 <= DCN_PS_MAX_BYTES[0])


@on_tensor_device
def dcn_ps(x, offmask, p: PackedConv):
    """DCNv2 + scale/shift + act as two launches: the 1x1 projection of the whole map (dense GEMM), then the bilinear sampling of the projected map."""
    _need_cuda(x, offmask)
    B, H, W, C = x.shape
    pp = dcn_ps_pack(p)
    # measured (profiles/r06_dcn_ps.md, B = 8): the activation-stationary kernel wins where the map is store-bound or wide -- K = 128 (35.9 -> 23.0 us,
    # 63.3 -> 35.1), K = 512 (29.7 -> 26.5), K = 256 with 2304 outputs (47.4 -> 35.0) -- and loses on 256 -> 576 / 1152 (16.6 -> 20.4, 27.4 -> 25.8: a tie)
    if PROJECT_AS[0] and pp.K_pad == C and (C in (128, 512) or (C == 256 and 9 * p.Cout >= 2304)):
        proj = torch.empty((B, H, W, 9 * p.Cout), dtype=x.dtype, device=x.device)      # rows [(tap, n)]
        L.check(L.load().mfx_project_nhwc(_ptr(x), _ptr(pp.w), _ptr(proj), B * H * W, C, 9 * p.Cout, C, 9 * p.Cout, _dt(x.dtype), _stream()), "mfx_project_nhwc")
    else:
        proj = conv2d(x, pp)                                    # the tiled implicit-GEMM kernel
    y = torch.empty((B, H, W, p.Cout), dtype=x.dtype, device=x.device)
    L.check(L.load().mfx_dcn_sample_nhwc(_ptr(proj), _ptr(offmask), _ptr(p.scale), _ptr(p.shift), _ptr(y), B, H, W, p.Cout, p.Cout, p.act,
                                         _dt(x.dtype), _stream()), "mfx_dcn_sample_nhwc")
    return y


@on_tensor_device
def dcn_module(x, p_off: PackedConv, p: PackedConv, need_offmask=False):
    """The DCN module of the reference (dcn_v2.py:118-128): offset/mask conv (27 -> 32 channels, fp32 out, sigmoid on the mask channels)
    followed by the fused DCNv2.  Where the library's LDS-patch kernel takes the layer (mfx_dcn_fuses_offset_conv: 64 -> 64 on large
    16-bit maps) the offset conv runs INSIDE it -- one launch, the (B,H,W,32) offset map exists only if `need_offmask` (training: the
    backward pass reads it).  -> (y, offmask or None)"""
    _need_cuda(x)
    B, H, W, C = x.shape
    if p.stride == 1 and p.pad_h == 1 and p.kh == 3 and p.kw == 3:
        y = torch.empty((B, H, W, p.Cout), dtype=x.dtype, device=x.device)
        om = torch.empty((B, H, W, 32), dtype=torch.float32, device=x.device) if need_offmask else None
        d = _dcn_desc(x, None, p, y, off=p_off, offmask_out=om)
        if d.off_w_frag_f16 and L.load().mfx_dcn_fuses_offset_conv(ctypes.byref(d)):
            L.check(L.load().mfx_dcn_nhwc(ctypes.byref(d), _stream()), "mfx_dcn_nhwc")
            return y, om
    om = conv2d(x, p_off, out_dtype=torch.float32)
    if dcn_ps_applies(x, p):
        return dcn_ps(x, om, p), om
    return dcn(x, om, p), om


@on_tensor_device
def maxpool2x2(x):
    _need_cuda(x)
    B, H, W, C = x.shape
    y = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    L.check(L.load().mfx_maxpool2x2_nhwc(_ptr(x), _ptr(y), B, H, W, C, _dt(x.dtype), _stream()), "mfx_maxpool2x2_nhwc")
    return y


def pack_upsample(weight):
    """(C,1,k,k) depthwise deconv weight -> fp32 [k*k][C]."""
    C, _, k, _ = weight.shape
    return weight.detach().float().reshape(C, k * k).t().contiguous()


@on_tensor_device
def upsample_add(x, w_taps, f, skip=None):
    _need_cuda(x, w_taps, skip)
    B, H, W, C = x.shape
    y = torch.empty((B, H * f, W * f, C), dtype=x.dtype, device=x.device)
    L.check(L.load().mfx_upsample_add_nhwc(_ptr(x), _ptr(w_taps), _ptr(skip), _ptr(y), B, H, W, C, f, _dt(x.dtype), _stream()),
            "mfx_upsample_add_nhwc")
    return y


@on_tensor_device
def nchw_to_nhwc(x, dtype, channels=None):
    """fp32 NCHW -> NHWC of `dtype`, channel axis zero-padded to `channels`."""
    _need_cuda(x)
    x = x.float().contiguous()
    B, C, H, W = x.shape
    ld = channels or C
    y = torch.empty((B, H, W, ld), dtype=dtype, device=x.device)
    L.check(L.load().mfx_nchw_to_nhwc(_ptr(x), _ptr(y), B, C, H, W, ld, _dt(dtype), _stream()), "mfx_nchw_to_nhwc")
    return y


@on_tensor_device
def nhwc_to_nchw(x, channels=None):
    _need_cuda(x)
    B, H, W, ld = x.shape
    C = channels or ld
    y = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    L.check(L.load().mfx_nhwc_to_nchw(_ptr(x), _ptr(y), B, C, H, W, ld, _dt(x.dtype), _stream()), "mfx_nhwc_to_nchw")
    return y


@on_tensor_device
def pack_image(images, dtype):
    """(B,3,H,W) fp32 NCHW -> zero-padded NHWC4 (B, H+6, W+8, 4) for the stem conv."""
    _need_cuda(images)
    images = images.float().contiguous()
    dtype = storage_dtype(dtype)
    B, C, H, W = images.shape
    assert C == 3
    y = torch.empty((B, H + 2 * STEM_PAD_H, W + STEM_PAD_WL + STEM_PAD_WR, 4), dtype=dtype, device=images.device)
    L.check(L.load().mfx_pack_image_nhwc4(_ptr(images), _ptr(y), B, H, W, STEM_PAD_H, STEM_PAD_WL, STEM_PAD_WR,
                                          _dt(dtype), _stream()), "mfx_pack_image_nhwc4")
    return y


@on_tensor_device
def stem_conv(images, p: PackedConv):
    """bf16 stem: (B,3,H,W) fp32 NCHW -> (B,H,W,16) bf16 NHWC, conv7x7 + scale/shift + act in one kernel."""
    _need_cuda(images)
    images = images.float().contiguous()
    B, C, H, W = images.shape
    assert C == 3 and p.w.dtype in (torch.bfloat16, torch.float16) and p.Cout == 16
    y = torch.empty((B, H, W, 16), dtype=torch.float32 if p.split else p.w.dtype, device=images.device)
    L.check(L.load().mfx_stem_conv7x7_nchw(_ptr(images), _ptr(p.w), _ptr(p.scale), _ptr(p.shift), _ptr(y), B, H, W, 16, p.K_pad, p.act,
                                           L.MFX_F16X2 if p.split else _dt(p.w.dtype), _stream()), "mfx_stem_conv7x7_nchw")
    return y


@on_tensor_device
def f1_fused(images, p_stem: PackedConv, p_l0: PackedConv, p_l1: PackedConv):
    """(B,3,H,W) fp32 NCHW -> level1 map (B,H/2,W/2,32): stem + level0 + level1 (each conv + folded BN + ReLU) in one kernel (csrc/f1_fused.hip);
    the packs are the ones the three separate launches use (pack_stem, pack_conv)."""
    _need_cuda(images)
    images = images.float().contiguous()
    B, C, H, W = images.shape
    split = p_stem.split                                            # split precision: fp32 map out, (hi, lo) fp16 operand pairs
    dt = torch.float32 if split else p_stem.w.dtype
    assert C == 3 and (split or dt in (torch.bfloat16, torch.float16)) and p_stem.Cout == 16 and p_l0.Cout == 16 and p_l1.Cout == 32 and p_l1.stride == 2
    assert p_l0.split == split and p_l1.split == split
    key = "_f1_w160"
    for p in (p_l0, p_l1):                                          # [Cout][160] slice of the K-padded (tap, channel) matrix, cached on the pack
        if not hasattr(p, key):
            assert p.K_pad >= 160 and p.Ck == 16
            setattr(p, key, p.w[:, :160].contiguous())
    y = torch.empty((B, H // 2, W // 2, 32), dtype=dt, device=images.device)
    L.check(L.load().mfx_f1_fused(_ptr(images), _ptr(p_stem.w), _ptr(p_stem.scale), _ptr(p_stem.shift),
                                  _ptr(getattr(p_l0, key)), _ptr(p_l0.scale), _ptr(p_l0.shift),
                                  _ptr(getattr(p_l1, key)), _ptr(p_l1.scale), _ptr(p_l1.shift),
                                  _ptr(y), B, H, W, p_stem.K_pad, L.MFX_F16X2 if split else _dt(dt), _stream()), "mfx_f1_fused")
    return y


@dataclass
class PackedHeads:
    w1: torch.Tensor
    scale1: torch.Tensor
    shift1: torch.Tensor
    w2: torch.Tensor
    bias2: torch.Tensor
    K_pad: int
    ch_off: list
    c_out: list
    ld_out: int
    split: bool = False
    w2_scale: Optional[list] = None          # per branch, multiplies the 1x1 sums before the bias (split precision: 1 / the weights' packing scale)
    w1_32: Optional[torch.Tensor] = None     # packs of the v_mfma_f32_32x32x16 form of the kernel (16-bit modes; mfx_heads_desc.w1_32 / w2_32)
    w2_32: Optional[torch.Tensor] = None


@on_tensor_device
def heads_fused(x, p: PackedHeads, planar_classes=0):
    """-> (head map fp32 (B,H,W,ld_out), class-planar logits (B,planar_classes,H*W) or None)."""
    _need_cuda(x)
    B, H, W, C = x.shape
    assert C == 64
    out = torch.empty((B, H, W, p.ld_out), dtype=torch.float32, device=x.device)
    planar = torch.empty((B, planar_classes, H * W), dtype=torch.float32, device=x.device) if planar_classes else None
    d = L.HeadsDesc()
    d.planar, d.planar_c = (planar.data_ptr() if planar is not None else None), planar_classes
    d.x, d.w1, d.scale1, d.shift1 = x.data_ptr(), p.w1.data_ptr(), p.scale1.data_ptr(), p.shift1.data_ptr()
    d.w2, d.bias2, d.out = p.w2.data_ptr(), p.bias2.data_ptr(), out.data_ptr()
    d.w1_32 = p.w1_32.data_ptr() if p.w1_32 is not None else None
    d.w2_32 = p.w2_32.data_ptr() if p.w2_32 is not None else None
    d.B, d.H, d.W, d.nbranch, d.K_pad, d.ld_out = B, H, W, len(p.c_out), p.K_pad, p.ld_out
    d.dtype = L.MFX_F16X2 if p.split else _dt(x.dtype)
    for i, (o, c) in enumerate(zip(p.ch_off, p.c_out)):
        d.ch_off[i], d.c_out[i] = o, c
        d.w2_scale[i] = p.w2_scale[i] if p.w2_scale is not None else 1.0
    L.check(L.load().mfx_heads_fused(ctypes.byref(d), _stream()), "mfx_heads_fused")
    return out, planar


@on_tensor_device
def edge_scatter_add(out, ch_off, C, v, edge_xy, edge_len, planar=None):
    _need_cuda(out, v, edge_xy, edge_len, planar)
    B, H, W, ld = out.shape
    Lmax = edge_xy.shape[1]
    L.check(L.load().mfx_edge_scatter_add(_ptr(out), ld, ch_off, C, _ptr(v), v.shape[-1], _ptr(edge_xy), _ptr(edge_len),
                                          B, Lmax, H, W, _ptr(planar), _stream()), "mfx_edge_scatter_add")


@on_tensor_device
def decode_topk(hmap, ch_off, ncls, K, planar=None):
    """Per-(image,class) NMS + top-K.  Reads the class-planar logits when given (coalesced), else the NHWC map."""
    _need_cuda(hmap, planar)
    B, H, W, ld = hmap.shape
    scores = torch.empty((B, ncls, K), dtype=torch.float32, device=hmap.device)
    index = torch.empty((B, ncls, K), dtype=torch.int32, device=hmap.device)
    if planar is not None:
        src, bs, cs, ps = planar.data_ptr(), ncls * H * W, H * W, 1
    else:
        src, bs, cs, ps = hmap.data_ptr() + 4 * ch_off, H * W * ld, 1, ld
    lib = L.load()
    ws = torch.empty(int(lib.mfx_decode_topk_workspace_bytes(ncls, B, K)), dtype=torch.uint8, device=hmap.device)
    L.check(lib.mfx_decode_topk(ctypes.c_void_p(src), bs, cs, ps, ncls, B, H, W, K, _ptr(scores), _ptr(index), _ptr(ws), ws.numel(),
                                _stream()), "mfx_decode_topk")
    return scores, index


@on_tensor_device
def decode_boxes(hmap, reg_off, scores, index, calib, pad, img_size, threshold, depth_mode="soft"):
    """`depth_mode`: the reference's `output_depth` name (lib.DEPTH_MODES; 'oracle' needs ground truth and is not a decode mode)."""
    _need_cuda(hmap, scores, index, calib, pad, img_size)
    if depth_mode not in L.DEPTH_MODES:
        raise ValueError("decode_boxes: output_depth %r is not one of %s" % (depth_mode, sorted(L.DEPTH_MODES)))
    B, H, W, ld = hmap.shape
    ncls, K = scores.shape[1], scores.shape[2]
    det = torch.empty((B, K, 14), dtype=torch.float32, device=hmap.device)
    topk = torch.empty((B, K, 5), dtype=torch.float32, device=hmap.device)
    valid = torch.empty((B, K), dtype=torch.int32, device=hmap.device)
    L.check(L.load().mfx_decode_boxes_mode(_ptr(hmap), ld, reg_off, _ptr(scores), _ptr(index), ncls, B, H, W, K, _ptr(calib),
                                           _ptr(pad), _ptr(img_size), ctypes.c_float(threshold), L.DEPTH_MODES[depth_mode], _ptr(det), _ptr(topk),
                                           _ptr(valid), _stream()), "mfx_decode_boxes_mode")
    return det, topk, valid


# ---- reference `_ext` boundary (NCHW fp32) ---------------------------------------------------------
_ws_cache = {}


SPLITK_MAX_ELEMS = 4 * 1024 * 1024          # output elements (M * Cout_pad) up to which split-K is offered
_splitk_ws = {}


def _splitk_workspace(device):
    """One persistent fp32 scratch per device (9 splits x SPLITK_MAX_ELEMS): stable address, so captured graphs stay valid;
    launches on one stream are ordered, so consecutive layers can share it."""
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture the scratch must come from THAT graph's memory pool and live exactly as long as the ops
        # that use it (an entry cached from an earlier capture would belong to another graph's pool): allocate per call, the
        # pool reuses the block for the next layer in stream order, which replays faithfully
        return torch.empty(9 * SPLITK_MAX_ELEMS, dtype=torch.float32, device=device)
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)      # concurrent streams must not share it
    if key not in _splitk_ws:
        _splitk_ws[key] = torch.empty(9 * SPLITK_MAX_ELEMS, dtype=torch.float32, device=device)
    return _splitk_ws[key]


def _workspace(nbytes, device):
    if torch.cuda.is_current_stream_capturing():                # see _splitk_workspace
        return torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def _ext_check(x, w, off, msk, kh, kw, dg):
    """The reference's argument checks (src/cuda/dcn_v2_cuda.cu:60-84, dcn_v2.py:84-87) with its messages, as RuntimeError."""
    C = x.shape[1]
    if w.shape[2] != kh or w.shape[3] != kw:
        raise RuntimeError("Input shape and kernel shape wont match: (%d x %d vs %d x %d)." % (kh, kw, w.shape[2], w.shape[3]))
    if w.shape[1] != C:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (C, w.shape[1]))
    if dg < 1 or C % dg != 0:
        raise RuntimeError("dcn_v2: %d input channels cannot be split into %d deformable groups" % (C, dg))
    if off.shape[1] != 2 * dg * kh * kw or msk.shape[1] != dg * kh * kw:
        raise RuntimeError("dcn_v2: offset / mask must have 2*dg*kh*kw = %d / dg*kh*kw = %d channels (got %d / %d)"
                           % (2 * dg * kh * kw, dg * kh * kw, off.shape[1], msk.shape[1]))


@on_tensor_device
def ext_dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg):
    """`_ext.dcn_v2_forward` (src/dcn_v2.h:9-23): NCHW fp32 in and out; ONE call of the C entry, which is as general as the reference's:
    deformable groups (its own example uses 2, testcuda.py:169-180) are looped inside `mfx_dcn_v2_forward` (group g = channel slice g with its
    own 2*kh*kw offset and kh*kw mask channels, dcn_v2_im2col_cuda.cu:147-156), stride / padding / dilation are per axis."""
    _need_cuda(input, weight, bias, offset, mask)
    ts = [t.float().contiguous() for t in (input, weight, bias, offset, mask)]
    x, w, b, off, msk = ts
    _ext_check(x, w, off, msk, kh, kw, dg)
    B, C, H, W = x.shape
    Cout = w.shape[0]
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    lib_ = L.load()
    nbytes = lib_.mfx_dcn_v2_workspace_bytes_g(B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, 0)
    ws = _workspace(nbytes, x.device)
    out = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    L.check(lib_.mfx_dcn_v2_forward(_ptr(x), _ptr(w), _ptr(b), _ptr(off), _ptr(msk), _ptr(out), B, C, H, W, Cout, kh, kw,
                                    sh, sw, ph, pw, dh, dw, dg, _ptr(ws), ws.numel(), _stream()), "mfx_dcn_v2_forward")
    return out


@on_tensor_device
def ext_dcn_v2_backward(input, weight, bias, offset, mask, grad_output, kh, kw, sh, sw, ph, pw, dh, dw, dg):
    """`_ext.dcn_v2_backward` (src/dcn_v2.h:48-59) -> [grad_input, grad_offset, grad_mask, grad_weight, grad_bias]; deformable
    groups and per-axis geometry inside the C entry, as in the forward."""
    _need_cuda(input, weight, bias, offset, mask, grad_output)
    x, w, b, off, msk, go = [t.float().contiguous() for t in (input, weight, bias, offset, mask, grad_output)]
    _ext_check(x, w, off, msk, kh, kw, dg)
    B, C, H, W = x.shape
    Cout = w.shape[0]
    lib_ = L.load()
    nbytes = lib_.mfx_dcn_v2_workspace_bytes_g(B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, 1)
    ws = _workspace(nbytes, x.device)
    gi, gw, gb = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
    goff, gm = torch.empty_like(off), torch.empty_like(msk)
    L.check(lib_.mfx_dcn_v2_backward(_ptr(x), _ptr(w), _ptr(b), _ptr(off), _ptr(msk), _ptr(go), _ptr(gi), _ptr(goff), _ptr(gm),
                                     _ptr(gw), _ptr(gb), B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg,
                                     _ptr(ws), ws.numel(), _stream()), "mfx_dcn_v2_backward")
    return [gi, goff, gm, gw, gb]
