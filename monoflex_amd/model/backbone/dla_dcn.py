"""DLA-34 + DLAUp/IDAUp backbone on the gfx950 kernels.

Module tree, attribute names and state_dict keys follow the reference
(/root/reference/model/backbone/dla_dcn.py:20-452) so its checkpoints load unchanged; the
nn.Conv2d / nn.BatchNorm2d / nn.ConvTranspose2d children are parameter holders only -- every
forward below runs NHWC through libmonoflex_hip.so:

  conv + BN(eval, folded) + ReLU (+ residual)  -> one implicit-GEMM launch      (ops.conv2d)
  Root: cat -> 1x1 conv -> BN -> ReLU          -> one launch, no concat tensor   (ops.cat_conv1x1)
  DeformConv: DCN + BN + ReLU                  -> offset conv + fused DCN kernel (DCN.forward_nhwc)
  up(proj(x)) + skip                           -> one depthwise-deconv launch    (ops.upsample_add)

Eval mode folds BN (running statistics) into the conv epilogues.  Training mode (module.training) runs the same
graph unfused through the differentiable operators of monoflex_amd.autograd: conv -> train-mode BN(+act,+res),
fp32 activations, gradients by the HIP backward kernels.
"""
import math
import os

import numpy as np
import torch
from torch import nn

from ... import autograd as AG
from ... import lib as L
from ... import ops
from .DCNv2.dcn_v2 import DCN

BN_MOMENTUM = 0.1


def build_backbone(cfg):
    return DLASeg(base_name=cfg.MODEL.BACKBONE.CONV_BODY, pretrained=cfg.MODEL.PRETRAIN,
                  down_ratio=cfg.MODEL.BACKBONE.DOWN_RATIO, last_level=5)


def invalidate_packs(module):
    """Drop every cached packed-weight tensor under `module` (call after changing parameters)."""
    for m in module.modules():
        if hasattr(m, "_packs"):
            m._packs.clear()


def _eval_only(m):
    if m.training:
        raise NotImplementedError("%s: this entry point is the fused eval-mode path (BatchNorm folded into the conv epilogue from the "
                                  "running statistics); call .eval(), or use the module's training forward (forward_nhwc_train / "
                                  "model(images, targets) in train mode)" % type(m).__name__)


def _train_conv_bn(x, conv, bn, act, res=None):
    """Training form of conv -> BN(batch statistics) -> act (+res before the act)."""
    y, done = AG.conv2d_bn_stats(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], bn)
    return AG.bn_act(y, bn, act, res, stats_done=done)


def _conv_bn(owner, key, conv, bn, dtype, act):
    packs = owner.__dict__.setdefault("_packs", {})
    dtype = ops.compute_tag(owner, dtype)
    k = (key, dtype)
    if k not in packs:
        scale, shift = ops.fold_bn(bn)
        packs[k] = ops.pack_conv(conv.weight, dtype, scale, shift, stride=conv.stride[0], pad=conv.padding[0], act=act)
    return packs[k]


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=dilation, bias=False, dilation=dilation)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=dilation, bias=False, dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.stride = stride

    def forward(self, x, residual=None):                      # x NHWC
        if residual is None:
            residual = x
        if self.training:
            if residual is x and self.stride == 1 and AG.RESIDUAL_ALIAS[0]:
                # identity residual: x feeds conv1 AND the add behind bn2.  conv1's node hands x on as a second output, so the residual's gradient
                # arrives at that node and rides the data-gradient conv's epilogue (no autograd add over the map)
                y, done, xr = AG.conv2d_bn_stats(x, self.conv1.weight, self.conv1.bias, 1, self.conv1.padding[0], self.bn1, alias=True)
                out = AG.bn_act(y, self.bn1, L.ACT_RELU, None, stats_done=done)
                return _train_conv_bn(out, self.conv2, self.bn2, L.ACT_RELU, res=xr)
            out = _train_conv_bn(x, self.conv1, self.bn1, L.ACT_RELU)
            return _train_conv_bn(out, self.conv2, self.bn2, L.ACT_RELU, res=residual)
        out = ops.conv2d(x, _conv_bn(self, "c1", self.conv1, self.bn1, x.dtype, L.ACT_RELU))
        return ops.conv2d(out, _conv_bn(self, "c2", self.conv2, self.bn2, x.dtype, L.ACT_RELU), res=residual)


class Root(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, residual):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 1, stride=1, bias=False, padding=(kernel_size - 1) // 2)
        self.bn = nn.BatchNorm2d(out_channels, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.residual = residual
        assert not residual, "dla34 uses residual_root=False (dla_dcn.py:264)"

    def forward(self, *xs):
        if self.training:
            return AG.bn_act(AG.CatConv1x1Fn.apply(self.conv.weight, *xs), self.bn, L.ACT_RELU)
        packs = self.__dict__.setdefault("_packs", {})
        chans = tuple(t.shape[3] for t in xs)
        tag = ops.compute_tag(self, xs[0].dtype)
        k = (chans, tag)
        if k not in packs:
            scale, shift = ops.fold_bn(self.bn)
            packs[k] = ops.pack_cat(self.conv.weight, tag, scale, shift, chans, act=L.ACT_RELU)
        return ops.cat_conv1x1(list(xs), packs[k])


class Tree(nn.Module):
    def __init__(self, levels, block, in_channels, out_channels, stride=1, level_root=False, root_dim=0,
                 root_kernel_size=1, dilation=1, root_residual=False):
        super().__init__()
        if root_dim == 0:
            root_dim = 2 * out_channels
        if level_root:
            root_dim += in_channels
        if levels == 1:
            self.tree1 = block(in_channels, out_channels, stride, dilation=dilation)
            self.tree2 = block(out_channels, out_channels, 1, dilation=dilation)
        else:
            self.tree1 = Tree(levels - 1, block, in_channels, out_channels, stride, root_dim=0,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels, root_dim=root_dim + out_channels,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
        if levels == 1:
            self.root = Root(root_dim, out_channels, root_kernel_size, root_residual)
        self.level_root, self.root_dim, self.levels = level_root, root_dim, levels
        self.downsample = nn.MaxPool2d(stride, stride=stride) if stride > 1 else None
        self.project = None
        if in_channels != out_channels:
            self.project = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, bias=False),
                                         nn.BatchNorm2d(out_channels, momentum=BN_MOMENTUM))

    def forward(self, x, residual=None, children=None, bottom=None):       # dla_dcn.py:246-259
        children = [] if children is None else children
        # `bottom`: a levels > 1 Tree and its nested tree1 both max-pool the SAME x (dla_dcn.py:248 runs twice on level3 / level4); the outer
        # one hands its result down instead (r05: two of the six pooling launches of a pass, and of their backward, gone)
        if bottom is None:
            if self.training:
                bottom = AG.MaxPool2x2Fn.apply(x) if self.downsample else x
            else:
                bottom = ops.maxpool2x2(x) if self.downsample else x
        if self.levels == 1:
            # a levels>1 Tree hands `residual` to a nested Tree that recomputes its own:
            # the outer level3/level4 `project` output is dead in the reference (its parameters never get a
            # gradient), so it is not computed here
            if not self.project:
                residual = bottom
            elif self.training:
                residual = _train_conv_bn(bottom, self.project[0], self.project[1], L.ACT_NONE)
            else:
                residual = ops.conv2d(bottom, _conv_bn(self, "proj", self.project[0], self.project[1], x.dtype, L.ACT_NONE))
        if self.level_root:
            children.append(bottom)
        if self.levels > 1 and isinstance(self.tree1, Tree) and self.tree1.downsample is not None and self.downsample is not None:
            x1 = self.tree1(x, residual, bottom=bottom)
        else:
            x1 = self.tree1(x, residual)
        if self.levels == 1:
            x2 = self.tree2(x1)
            return self.root(x2, x1, *children)
        children.append(x1)
        return self.tree2(x1, children=children)


class DLA(nn.Module):
    def __init__(self, levels, channels, num_classes=1000, block=BasicBlock, residual_root=False, linear_root=False):
        super().__init__()
        self.channels, self.num_classes = channels, num_classes
        self.base_layer = nn.Sequential(nn.Conv2d(3, channels[0], kernel_size=7, stride=1, padding=3, bias=False),
                                        nn.BatchNorm2d(channels[0], momentum=BN_MOMENTUM), nn.ReLU(inplace=True))
        self.level0 = self._make_conv_level(channels[0], channels[0], levels[0])
        self.level1 = self._make_conv_level(channels[0], channels[1], levels[1], stride=2)
        self.level2 = Tree(levels[2], block, channels[1], channels[2], 2, level_root=False, root_residual=residual_root)
        self.level3 = Tree(levels[3], block, channels[2], channels[3], 2, level_root=True, root_residual=residual_root)
        self.level4 = Tree(levels[4], block, channels[3], channels[4], 2, level_root=True, root_residual=residual_root)
        self.level5 = Tree(levels[5], block, channels[4], channels[5], 2, level_root=True, root_residual=residual_root)

    @staticmethod
    def _make_conv_level(inplanes, planes, convs, stride=1, dilation=1):
        modules = []
        for i in range(convs):
            modules.extend([nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride if i == 0 else 1,
                                      padding=dilation, bias=False, dilation=dilation),
                            nn.BatchNorm2d(planes, momentum=BN_MOMENTUM), nn.ReLU(inplace=True)])
            inplanes = planes
        return nn.Sequential(*modules)

    def forward(self, images, dtype, cut=None):
        """images: (B,3,H,W) fp32 NCHW -> list of 6 NHWC feature maps (strides 1..32).  `cut` (training): see DLASeg.forward_nhwc."""
        if self.training:
            return self._forward_train(images, dtype, cut)
        packs = self.__dict__.setdefault("_packs", {})
        tag = ops.compute_tag(self, dtype)
        if ("stem", tag) not in packs:
            scale, shift = ops.fold_bn(self.base_layer[1])
            packs[("stem", tag)] = ops.pack_stem(self.base_layer[0].weight, tag, scale, shift)
        B, _, H, W = images.shape
        if FUSE_F1[0] and ((dtype in (torch.bfloat16, torch.float16) and tag == dtype) or tag == ops.F16X2) and packs[("stem", tag)].Cout == 16 \
                and len(self.level0) == 3 and len(self.level1) == 3 and H % 2 == 0 and W % 2 == 0 and self.channels[:2] == [16, 32]:
            # stem -> level0 -> level1 in one kernel: the two full-resolution 16-channel maps never reach memory.  Nothing downstream reads
            # them (DLAUp starts at level 2, first_level = 2), so y[0] is None on this path (FUSE_F1[0] = False brings it back: tests of the
            # per-stage goldens)
            x = ops.f1_fused(images, packs[("stem", tag)], _conv_bn(self.level0, "c0", self.level0[0], self.level0[1], dtype, L.ACT_RELU),
                             _conv_bn(self.level1, "c0", self.level1[0], self.level1[1], dtype, L.ACT_RELU))
            y = [None, x]
            for i in range(2, 6):
                x = getattr(self, "level{}".format(i))(x)
                y.append(x)
            return y
        if (dtype in (torch.bfloat16, torch.float16) or tag == ops.F16X2) and packs[("stem", tag)].Cout == 16:
            x = ops.stem_conv(images, packs[("stem", tag)])                 # reads the NCHW planes directly
        else:
            x = ops.conv2d(ops.pack_image(images, dtype), packs[("stem", tag)], out_hw=(H, W))
        y = []
        for i in range(6):
            lvl = getattr(self, "level{}".format(i))
            if i < 2:
                for j in range(0, len(lvl), 3):
                    x = ops.conv2d(x, _conv_bn(lvl, "c%d" % j, lvl[j], lvl[j + 1], dtype, L.ACT_RELU))
            else:
                x = lvl(x)
            y.append(x)
        return y


def _dla_forward_train(self, images, dtype, cut=None):
    x = AG.bn_act(AG.StemConvFn.apply(images, self.base_layer[0].weight, dtype), self.base_layer[1], L.ACT_RELU)
    y = []
    for i in range(6):
        lvl = getattr(self, "level{}".format(i))
        if cut is not None and i == 4:
            x = cut("level3", [x])[0]                           # level4 reads level3's map through a gradient cut
        if i < 2:
            for j in range(0, len(lvl), 3):
                x = _train_conv_bn(x, lvl[j], lvl[j + 1], L.ACT_RELU)
        else:
            x = lvl(x)
        y.append(x)
    return y


DLA._forward_train = _dla_forward_train


def dla34(pretrained=True, **kwargs):
    """DLA-34 trunk. `pretrained`: False = random init; a path = ImageNet weights from that file; True = the file named
    by $MONOFLEX_DLA34_WEIGHTS (the reference downloads dla34-ba72cf86.pth, dla_dcn.py:333-344 -- there is no network here)."""
    model = DLA([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], block=BasicBlock, **kwargs)
    if pretrained:
        path = pretrained if isinstance(pretrained, str) else os.environ.get("MONOFLEX_DLA34_WEIGHTS", "")
        if not path or not os.path.exists(path):
            raise RuntimeError("dla34(pretrained=True) needs the ImageNet weights http://dl.yf.io/dla/models/imagenet/"
                               "dla34-ba72cf86.pth as a local file: set MONOFLEX_DLA34_WEIGHTS=<path> or MODEL.PRETRAIN "
                               "to the path (no network in this build), or set MODEL.PRETRAIN False and load a checkpoint")
        from ...utils.model_serialization import load_dla_imagenet
        load_dla_imagenet(model, path)
    return model


def fill_up_weights(up):                                       # dla_dcn.py:372-381
    w = up.weight.data
    f = math.ceil(w.size(2) / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    for i in range(w.size(2)):
        for j in range(w.size(3)):
            w[0, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
    for ch in range(1, w.size(0)):
        w[ch, 0, :, :] = w[0, 0, :, :]


class DeformConv(nn.Module):
    def __init__(self, chi, cho):
        super().__init__()
        self.actf = nn.Sequential(nn.BatchNorm2d(cho, momentum=BN_MOMENTUM), nn.ReLU(inplace=True))
        self.conv = DCN(chi, cho, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)

    def forward(self, x):                                      # NHWC; DCN + BN + ReLU in one kernel epilogue
        if self.training:
            return AG.bn_act(self.conv.forward_nhwc_train(x), self.actf[0], L.ACT_RELU)
        return self.conv.forward_nhwc(x, bn=self.actf[0], act=L.ACT_RELU)


class IDAUp(nn.Module):
    def __init__(self, o, channels, up_f):
        super().__init__()
        for i in range(1, len(channels)):
            c, f = channels[i], int(up_f[i])
            up = nn.ConvTranspose2d(o, o, f * 2, stride=f, padding=f // 2, output_padding=0, groups=o, bias=False)
            fill_up_weights(up)
            setattr(self, "proj_" + str(i), DeformConv(c, o))
            setattr(self, "up_" + str(i), up)
            setattr(self, "node_" + str(i), DeformConv(o, o))

    def forward(self, layers, startp, endp):                   # dla_dcn.py:419-425, in-place list semantics kept
        packs = self.__dict__.setdefault("_packs", {})
        if not self.training and PARALLEL_PROJ[0] and endp - startp > 2 and layers[startp].is_cuda:
            return self._forward_parallel_proj(layers, startp, endp, packs)
        for i in range(startp + 1, endp):
            k = i - startp
            up = getattr(self, "up_" + str(k))
            if k not in packs:
                packs[k] = ops.pack_upsample(up.weight)
            t = getattr(self, "proj_" + str(k))(layers[i])
            if self.training:
                t = AG.UpsampleAddFn.apply(t, up.weight, layers[i - 1], up.stride[0])
                layers[i] = getattr(self, "node_" + str(k))(t)
                continue
            t = ops.upsample_add(t, packs[k], up.stride[0], skip=layers[i - 1])      # up(proj(x_i)) + x_{i-1}
            layers[i] = getattr(self, "node_" + str(k))(t)


def _idaup_forward_parallel_proj(self, layers, startp, endp, packs):
    """Inference: proj_k(layers[startp + k]) reads only its own level's map, so the proj DCN modules of one IDAUp are independent
    of each other and of the node chain before them.  They are issued on forked streams (inside a hipGraph capture these become
    parallel branches): the small-map DCN launches (120-480 workgroups on 256 CUs) overlap instead of running one after another."""
    cur = torch.cuda.current_stream()
    n = endp - startp - 1
    side = self.__dict__.setdefault("_side_streams", [])
    while len(side) < n - 1:
        side.append(torch.cuda.Stream())
    proj = [None] * n
    for k in range(2, n + 1):                                  # proj_2 .. proj_n on side streams
        st = side[k - 2]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            proj[k - 1] = getattr(self, "proj_" + str(k))(layers[startp + k])
    proj[0] = self.proj_1(layers[startp + 1])
    for k in range(1, n + 1):
        i = startp + k
        up = getattr(self, "up_" + str(k))
        if k not in packs:
            packs[k] = ops.pack_upsample(up.weight)
        if k >= 2:
            cur.wait_stream(side[k - 2])
        t = ops.upsample_add(proj[k - 1], packs[k], up.stride[0], skip=layers[i - 1])
        layers[i] = getattr(self, "node_" + str(k))(t)


FUSE_F1 = [os.environ.get("MFX_FUSE_F1", "1") == "1"]         # inference, 16-bit maps: stem + level0 + level1 as one kernel (csrc/f1_fused.hip)

IDAUp._forward_parallel_proj = _idaup_forward_parallel_proj
PARALLEL_PROJ = [__import__("os").environ.get("MFX_PARALLEL_PROJ", "0") == "1"]     # measured neutral at B = 8 (3.18-3.21 vs 3.20 ms/step): off


class DLAUp(nn.Module):
    def __init__(self, startp, channels, scales, in_channels=None):
        super().__init__()
        self.startp = startp
        if in_channels is None:
            in_channels = channels
        self.channels = channels
        channels = list(channels)
        in_channels = list(in_channels)
        scales = np.array(scales, dtype=int)
        for i in range(len(channels) - 1):
            j = -i - 2
            setattr(self, "ida_{}".format(i), IDAUp(channels[j], in_channels[j:], scales[j:] // scales[j]))
            scales[j + 1:] = scales[j]
            in_channels[j + 1:] = [channels[j] for _ in channels[j + 1:]]

    def forward(self, layers):
        out = [layers[-1]]
        for i in range(len(layers) - self.startp - 1):
            getattr(self, "ida_{}".format(i))(layers, len(layers) - i - 2, len(layers))
            out.insert(0, layers[-1])
        return out


class DLASeg(nn.Module):
    def __init__(self, base_name, pretrained, down_ratio, last_level):
        super().__init__()
        assert down_ratio in [2, 4, 8, 16]
        self.first_level = int(np.log2(down_ratio))
        self.last_level = last_level
        self.base = globals()[base_name](pretrained=pretrained)
        channels = self.base.channels
        scales = [2 ** i for i in range(len(channels[self.first_level:]))]
        self.dla_up = DLAUp(self.first_level, channels[self.first_level:], scales)
        self.out_channels = channels[self.first_level]
        self.ida_up = IDAUp(self.out_channels, channels[self.first_level:self.last_level],
                            [2 ** i for i in range(self.last_level - self.first_level)])
        self.compute_dtype = torch.float32

    def forward_nhwc(self, images, cut=None):
        """`cut(name, tensors) -> tensors` (training, optional): called at three points of the forward pass -- level3 -> level4
        ("level3"), DLA base -> DLAUp ("base", the six level maps) and DLAUp -> IDAUp ("dla_up") -- so that a trainer can replace
        the maps by detached leaves and run the backward pass in four pieces (heads + IDAUp, DLAUp, level5/4, level3..stem), each
        followed by the all-reduce of its own gradients (engine/trainer.GraphedTrainStep; KeypointDetector.backward_thunks)."""
        cut = cut if (cut is not None and self.training) else None
        x = self.base(images, self.compute_dtype, cut) if cut is not None else self.base(images, self.compute_dtype)
        if cut is not None:
            x = cut("base", list(x))
        x = self.dla_up(list(x))
        # the reference clones x[0..2] because IDAUp mutates its list argument (dla_dcn.py:53-56); here the
        # list itself is fresh and tensors are never written in place, so no copy is needed
        y = [x[i] for i in range(self.last_level - self.first_level)]
        if cut is not None:
            y = cut("dla_up", y)
        self.ida_up(y, 0, len(y))
        return y[-1]

    def forward(self, images, cut=None):
        """(B,3,H,W) -> (B,64,H/4,W/4): logical NCHW view of the NHWC result (channels_last strides)."""
        return self.forward_nhwc(images, cut).permute(0, 3, 1, 2)
