"""Stand-in for the reference's pybind module `_ext` (model/backbone/DCNv2/src/vision.cpp:3-8):
same four names, same positional arguments, served by libmonoflex_hip.so on gfx950.

    dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg) -> output
    dcn_v2_backward(input, weight, bias, offset, mask, grad_output, kh, ...) -> [gi, goff, gmask, gw, gb]
"""
from .... import ops


def dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w,
                   pad_h, pad_w, dilation_h, dilation_w, deformable_group):
    return ops.ext_dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w,
                                  pad_h, pad_w, dilation_h, dilation_w, deformable_group)


def dcn_v2_backward(input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h, stride_w,
                    pad_h, pad_w, dilation_h, dilation_w, deformable_group):
    # the reference reads grad_output's raw pointer without .contiguous();
    # ops.ext_dcn_v2_backward makes it contiguous defensively
    return ops.ext_dcn_v2_backward(input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w,
                                   stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, deformable_group)


def dcn_v2_psroi_pooling_forward(*args, **kwargs):
    raise RuntimeError("dcn_v2_psroi_pooling_forward: not supported on this build (never called by MonoFlex)")


def dcn_v2_psroi_pooling_backward(*args, **kwargs):
    raise RuntimeError("dcn_v2_psroi_pooling_backward: not supported on this build (never called by MonoFlex)")
