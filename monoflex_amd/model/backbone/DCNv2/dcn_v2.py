"""DCNv2 operator surface of the reference (model/backbone/DCNv2/dcn_v2.py:16-128):
`_DCNv2` autograd Function, `dcn_v2_conv`, `DCNv2`, `DCN` -- same names, arguments and error
behaviour, backed by the gfx950 kernels through `_ext`."""
import math

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import _ext as _backend
from .... import lib as L
from .... import ops


class _DCNv2(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
        ctx.stride = _pair(stride)
        ctx.padding = _pair(padding)
        ctx.dilation = _pair(dilation)
        ctx.kernel_size = _pair(weight.shape[2:4])
        ctx.deformable_groups = deformable_groups
        output = _backend.dcn_v2_forward(input, weight, bias, offset, mask,
                                         ctx.kernel_size[0], ctx.kernel_size[1], ctx.stride[0], ctx.stride[1],
                                         ctx.padding[0], ctx.padding[1], ctx.dilation[0], ctx.dilation[1],
                                         ctx.deformable_groups)
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        grad_input, grad_offset, grad_mask, grad_weight, grad_bias = _backend.dcn_v2_backward(
            input, weight, bias, offset, mask, grad_output,
            ctx.kernel_size[0], ctx.kernel_size[1], ctx.stride[0], ctx.stride[1],
            ctx.padding[0], ctx.padding[1], ctx.dilation[0], ctx.dilation[1], ctx.deformable_groups)
        return grad_input, grad_offset, grad_mask, grad_weight, grad_bias, None, None, None, None


dcn_v2_conv = _DCNv2.apply


class DCNv2(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.Tensor(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        self.bias.data.zero_()

    def forward(self, input, offset, mask):
        assert 2 * self.deformable_groups * self.kernel_size[0] * self.kernel_size[1] == offset.shape[1]
        assert self.deformable_groups * self.kernel_size[0] * self.kernel_size[1] == mask.shape[1]
        return dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding,
                           self.dilation, self.deformable_groups)


OFFSET_CONV_FP32 = [False]


class DCN(DCNv2):
    """DCNv2 + its own 27-channel offset/mask conv (zero-initialised), reference dcn_v2.py:97-128."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        channels_ = self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = nn.Conv2d(self.in_channels, channels_, kernel_size=self.kernel_size,
                                          stride=self.stride, padding=self.padding, bias=True)
        self.init_offset()
        self._packs = {}

    def init_offset(self):
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    # ---- NHWC fast path ------------------------------------------------------------------------
    def packed_offset(self, dtype):
        key = ("off", dtype)
        if key not in self._packs:
            c = self.conv_offset_mask
            p = ops.pack_conv(c.weight, dtype, None, c.bias, stride=self.stride[0], pad=self.padding[0],
                              act=L.ACT_DCN_OFFMASK, cout=32)
            self._packs[key] = ops.add_f16_fragments(p, c.weight)            # fp16 fragments: the LDS-patch DCN kernel can run this conv itself
        return self._packs[key]

    def packed_main(self, dtype, bn=None, act=L.ACT_NONE):
        key = ("main", dtype, id(bn), act)
        if key not in self._packs:
            if bn is not None:
                scale, shift = ops.fold_bn(bn, self.bias)
            else:
                scale, shift = None, self.bias
            p = ops.pack_conv(self.weight, dtype, scale, shift, stride=self.stride[0], pad=self.padding[0], act=act)
            p.dil_w = self.dilation[0]
            ops.add_f16_fragments(p, self.weight)
            self._packs[key] = p
        return self._packs[key]

    def forward_nhwc(self, x, bn=None, act=L.ACT_NONE):
        """x (B,H,W,C) NHWC; offset/mask conv (fp32 out, sigmoid on the 9 mask channels fused) then the
        fused gather+MFMA kernel with bias (+BN, +act) folded into its epilogue."""
        if OFFSET_CONV_FP32[0] and x.dtype != torch.float32:    # ablation switch (tools/bf16_ablation.py): offsets from fp32 operands
            offmask = ops.conv2d(x.float(), self.packed_offset(torch.float32), out_dtype=torch.float32)
        else:
            tag = ops.compute_tag(self, x.dtype)
            return ops.dcn_module(x, self.packed_offset(tag), self.packed_main(tag, bn, act))[0]
        return ops.dcn(x, offmask, self.packed_main(ops.compute_tag(self, x.dtype), bn, act))

    def forward_nhwc_train(self, x):
        """Differentiable NHWC form: offset/mask conv -> DCNv2 (gradients to x, offsets, mask logits, weight, bias)."""
        from .... import autograd as AG
        c = self.conv_offset_mask
        # (B,H,W,32), 27 used; the sigmoid of the 9 mask channels runs in the conv epilogue, and DCNFn returns the gradient
        # of the pre-activation (both generations of the backward fold the sigmoid derivative in)
        return AG.dcn_module(x, c.weight, c.bias, self.weight, self.bias, self.stride[0], self.padding[0], self.dilation[0])

    def forward(self, input):
        """(B,C,H,W) NCHW -> (B,Cout,Ho,Wo) NCHW, the reference's own call form (dcn_v2.py:118-128), differentiable with respect to
        the input and all four parameters whenever autograd is recording (testcuda.py:169-180 `example_dconv`: forward, then
        `error.backward()`).  One deformable group and square geometry run the NHWC operators (`AG.dcn_module`: offset/mask conv with
        the sigmoid fused, DCNv2, both HIP backward kernels); deformable_groups > 1 goes through `dcn_v2_conv` like the reference does,
        with the offset/mask conv on the HIP conv kernels.  The layout changes at the boundary are data movement only."""
        from .... import autograd as AG
        kh, kw = self.kernel_size
        square = self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1] and self.dilation[0] == self.dilation[1]
        recording = torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not square:
            raise RuntimeError("DCN: stride / padding / dilation must be square on this build (the MonoFlex model only builds square "
                               "3x3 DCNs, dla_dcn.py:391; `_ext.dcn_v2_forward` reports the same)")
        if self.deformable_groups == 1 and not recording:
            dtype = torch.float32 if input.dtype == torch.float32 else torch.bfloat16
            x = ops.nchw_to_nhwc(input, dtype)
            return ops.nhwc_to_nchw(self.forward_nhwc(x))                    # fused eval kernels (no graph recorded)
        x = input.float().permute(0, 2, 3, 1).contiguous()                   # NHWC; autograd sees the permute
        c = self.conv_offset_mask
        if self.deformable_groups == 1:
            y = AG.dcn_module(x, c.weight, c.bias, self.weight, self.bias, self.stride[0], self.padding[0], self.dilation[0])
            return y.permute(0, 3, 1, 2).contiguous()
        # general group count: out = conv_offset_mask(input); o1, o2, mask = chunk(out, 3); offset = cat(o1, o2); mask = sigmoid(mask)
        out = AG.conv2d(x, c.weight, c.bias, self.stride[0], self.padding[0], out_dtype=torch.float32).permute(0, 3, 1, 2)
        n = self.deformable_groups * kh * kw
        offset, mask = out[:, :2 * n].contiguous(), torch.sigmoid(out[:, 2 * n:3 * n]).contiguous()
        return dcn_v2_conv(input.float(), offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation, self.deformable_groups)
