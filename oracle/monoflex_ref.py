"""oracle/monoflex_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU fp32 restatement (plain torch ops + the C DCN oracle) of the reference's hot path:

    image batch -> DLA-34 -> DLAUp/IDAUp (16 DCNv2) -> 9 head branches + edge fusion
                -> sigmoid/clamp -> 3x3 NMS -> top-K -> POI gather -> 3D-box decode

Module attribute names reproduce the reference's state_dict keys so the same
weights load into the reference (golden generation, oracle/gen_golden.py), into
this oracle and into the HIP model.  Citations are relative to /root/reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
Pinning: tests/test_oracle_golden.py checks this file against fixtures captured
from the reference's own Python (tests/golden/*.npz, made by oracle/gen_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import dcn_ref

BN_MOMENTUM = 0.1                                   # model/backbone/dla_dcn.py:18
PI = math.pi

# Static parameters of the path: runs/monoflex.yaml over config/defaults.py
REG_HEADS = [['2d_dim'], ['3d_offset'], ['corner_offset'], ['corner_uncertainty'], ['3d_dim'],
             ['ori_cls', 'ori_offset'], ['depth'], ['depth_uncertainty']]
REG_CHANNELS = [[4], [2], [20], [3], [3], [8, 8], [1], [1]]
DIM_MEAN = ((3.8840, 1.5261, 1.6286), (0.8423, 1.7607, 0.6602), (1.7635, 1.7372, 0.5968))
DEPTH_RANGE = (0.1, 100.0)
DOWN_RATIO = 4
EPS_KPT = 1e-3                                      # model/anno_encoder.py:14


def key2channel(key):
    """model/layers/utils.py:22-37 -- channel slice of a regression key in the 50-ch map."""
    keys = [k for g in REG_HEADS for k in g]
    chans = [c for g in REG_CHANNELS for c in g]
    i = keys.index(key)
    s = sum(chans[:i])
    return slice(s, s + chans[i])


# --------------------------------------------------------------------------------------
# DCN module on the C oracle (model/backbone/DCNv2/dcn_v2.py:57-128)
# --------------------------------------------------------------------------------------
class DCN(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.zeros(cout))
        stdv = 1.0 / math.sqrt(cin * 9)                         # dcn_v2.py:75-81
        self.weight.data.uniform_(-stdv, stdv)
        self.conv_offset_mask = nn.Conv2d(cin, 27, 3, 1, 1, bias=True)
        self.conv_offset_mask.weight.data.zero_()               # dcn_v2.py:114-116
        self.conv_offset_mask.bias.data.zero_()

    def forward(self, x):
        out = self.conv_offset_mask(x)                          # dcn_v2.py:118-128
        offset = out[:, :18]                                    # chunk(3)+cat(o1,o2) == first 18 channels
        mask = torch.sigmoid(out[:, 18:27])
        if getattr(self, "torch_form", False):                  # dtype-generic autograd form (fp64 ground truth in tests)
            return dcn_ref.dcn_v2_torch(x, offset, mask, self.weight, self.bias, 1, 1, 1)
        return dcn_ref.dcn_v2_conv(x, offset, mask, self.weight, self.bias, 1, 1, 1, 1)


class DeformConv(nn.Module):                                    # dla_dcn.py:384-396
    def __init__(self, chi, cho):
        super().__init__()
        self.actf = nn.Sequential(nn.BatchNorm2d(cho, momentum=BN_MOMENTUM), nn.ReLU(inplace=True))
        self.conv = DCN(chi, cho)

    def forward(self, x):
        return self.actf(self.conv(x))


# --------------------------------------------------------------------------------------
# DLA-34 base (model/backbone/dla_dcn.py:70-98, 185-331)
# --------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout, momentum=BN_MOMENTUM)

    def forward(self, x, residual=None):
        if residual is None:
            residual = x
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + residual)


class Root(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, 1, 0, bias=False)
        self.bn = nn.BatchNorm2d(cout, momentum=BN_MOMENTUM)

    def forward(self, *xs):                                     # residual_root=False for dla34
        return F.relu(self.bn(self.conv(torch.cat(xs, 1))))


class Tree(nn.Module):
    def __init__(self, levels, cin, cout, stride=1, level_root=False, root_dim=0):
        super().__init__()
        if root_dim == 0:
            root_dim = 2 * cout
        if level_root:
            root_dim += cin
        if levels == 1:
            self.tree1 = BasicBlock(cin, cout, stride)
            self.tree2 = BasicBlock(cout, cout, 1)
            self.root = Root(root_dim, cout)
        else:
            self.tree1 = Tree(levels - 1, cin, cout, stride, root_dim=0)
            self.tree2 = Tree(levels - 1, cout, cout, root_dim=root_dim + cout)
        self.level_root, self.levels = level_root, levels
        self.downsample = nn.MaxPool2d(stride, stride=stride) if stride > 1 else None
        self.project = None
        if cin != cout:
            self.project = nn.Sequential(nn.Conv2d(cin, cout, 1, 1, bias=False),
                                         nn.BatchNorm2d(cout, momentum=BN_MOMENTUM))

    def forward(self, x, residual=None, children=None):         # dla_dcn.py:246-259
        children = [] if children is None else children
        bottom = self.downsample(x) if self.downsample else x
        residual = self.project(bottom) if self.project else bottom
        if self.level_root:
            children.append(bottom)
        x1 = self.tree1(x, residual)          # a nested Tree ignores `residual`
        if self.levels == 1:
            x2 = self.tree2(x1)
            return self.root(x2, x1, *children)
        children.append(x1)
        return self.tree2(x1, children=children)


def _conv_level(cin, cout, stride):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, bias=False),
                         nn.BatchNorm2d(cout, momentum=BN_MOMENTUM), nn.ReLU(inplace=True))


class DLA34(nn.Module):
    channels = [16, 32, 64, 128, 256, 512]

    def __init__(self):
        super().__init__()
        c = self.channels
        self.base_layer = nn.Sequential(nn.Conv2d(3, c[0], 7, 1, 3, bias=False),
                                        nn.BatchNorm2d(c[0], momentum=BN_MOMENTUM), nn.ReLU(inplace=True))
        self.level0 = _conv_level(c[0], c[0], 1)
        self.level1 = _conv_level(c[0], c[1], 2)
        self.level2 = Tree(1, c[1], c[2], 2, level_root=False)
        self.level3 = Tree(2, c[2], c[3], 2, level_root=True)
        self.level4 = Tree(2, c[3], c[4], 2, level_root=True)
        self.level5 = Tree(1, c[4], c[5], 2, level_root=True)

    def forward(self, x):
        y = []
        x = self.base_layer(x)
        for i in range(6):
            x = getattr(self, 'level%d' % i)(x)
            y.append(x)
        return y


def bilinear_up_weight(w):
    """dla_dcn.py:372-381 fixed bilinear initialisation of the depthwise deconv."""
    k = w.shape[2]
    f = math.ceil(k / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    for i in range(k):
        for j in range(k):
            w[0, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
    w[1:, 0] = w[0, 0]


class IDAUp(nn.Module):                                          # dla_dcn.py:399-425
    def __init__(self, o, channels, up_f):
        super().__init__()
        for i in range(1, len(channels)):
            f = int(up_f[i])
            setattr(self, 'proj_%d' % i, DeformConv(channels[i], o))
            up = nn.ConvTranspose2d(o, o, f * 2, stride=f, padding=f // 2, groups=o, bias=False)
            bilinear_up_weight(up.weight.data)
            setattr(self, 'up_%d' % i, up)
            setattr(self, 'node_%d' % i, DeformConv(o, o))

    def forward(self, layers, startp, endp):
        for i in range(startp + 1, endp):
            k = i - startp
            layers[i] = getattr(self, 'up_%d' % k)(getattr(self, 'proj_%d' % k)(layers[i]))
            layers[i] = getattr(self, 'node_%d' % k)(layers[i] + layers[i - 1])


class DLAUp(nn.Module):                                          # dla_dcn.py:429-452
    def __init__(self, startp, channels, scales):
        super().__init__()
        self.startp = startp
        channels = list(channels)
        in_channels = list(channels)
        scales = np.array(scales, dtype=int)
        for i in range(len(channels) - 1):
            j = -i - 2
            setattr(self, 'ida_%d' % i, IDAUp(channels[j], in_channels[j:], scales[j:] // scales[j]))
            scales[j + 1:] = scales[j]
            in_channels[j + 1:] = [channels[j] for _ in channels[j + 1:]]

    def forward(self, layers):
        out = [layers[-1]]
        for i in range(len(layers) - self.startp - 1):
            getattr(self, 'ida_%d' % i)(layers, len(layers) - i - 2, len(layers))
            out.insert(0, layers[-1])
        return out


class DLASeg(nn.Module):                                         # dla_dcn.py:30-58
    def __init__(self):
        super().__init__()
        self.first_level, self.last_level = 2, 5
        self.base = DLA34()
        ch = self.base.channels
        self.dla_up = DLAUp(self.first_level, ch[self.first_level:], [2 ** i for i in range(4)])
        self.out_channels = ch[self.first_level]
        self.ida_up = IDAUp(self.out_channels, ch[self.first_level:self.last_level], [2 ** i for i in range(3)])

    def forward(self, x, taps=None):
        x = self.base(x)
        if taps is not None:
            taps['base'] = [t for t in x]
        x = self.dla_up(list(x))
        y = [x[i].clone() for i in range(self.last_level - self.first_level)]
        self.ida_up(y, 0, len(y))
        return y[-1]


# --------------------------------------------------------------------------------------
# Heads (model/head/detector_predictor.py:21-165).  InPlaceABN (third-party, not vendored,
# requirements.txt:14) is restated as BatchNorm2d(eps=1e-5) + leaky_relu(0.01): parity
# unpinned at that boundary.
# --------------------------------------------------------------------------------------
class ABN(nn.BatchNorm2d):
    def forward(self, x):
        return F.leaky_relu(super().forward(x), 0.01)


class Predictor(nn.Module):
    def __init__(self, cin=64, head_conv=256, classes=3, init_p=0.01):
        super().__init__()
        self.head_conv = head_conv
        self.class_head = nn.Sequential(nn.Conv2d(cin, head_conv, 3, padding=1, bias=False),
                                        ABN(head_conv, momentum=0.1),
                                        nn.Conv2d(head_conv, classes, 1, bias=True))
        self.class_head[-1].bias.data.fill_(-np.log(1 / init_p - 1))       # :60
        self.reg_features, self.reg_heads = nn.ModuleList(), nn.ModuleList()
        for idx, keys in enumerate(REG_HEADS):
            self.reg_features.append(nn.Sequential(nn.Conv2d(cin, head_conv, 3, padding=1, bias=False),
                                                   ABN(head_conv, momentum=0.1)))
            heads = nn.ModuleList()
            for ki, key in enumerate(keys):
                h = nn.Conv2d(head_conv, REG_CHANNELS[idx][ki], 1, bias=True)
                if 'uncertainty' in key:
                    nn.init.xavier_normal_(h.weight, gain=0.01)             # :87-88
                if key == '3d_offset':
                    self.offset_index = [idx, ki]
                nn.init.constant_(h.bias, 0)                                # :93
                heads.append(h)
            self.reg_heads.append(heads)

        def trunc(cout):
            return nn.Sequential(nn.Conv1d(head_conv, head_conv, 3, padding=1, padding_mode='replicate'),
                                 nn.BatchNorm1d(head_conv, momentum=0.1), nn.Identity(),
                                 nn.Conv1d(head_conv, cout, 1))
        self.trunc_heatmap_conv = trunc(classes)                            # :111-119
        self.trunc_offset_conv = trunc(2)

    def forward(self, features, edge_indices, edge_lens, taps=None):
        """edge_indices (B,832,2) int64 (x,y); edge_lens (B,) -- the per-image `targets` fields."""
        b, c, h, w = features.shape
        feature_cls = self.class_head[:-1](features)
        output_cls = self.class_head[-1](feature_cls)
        output_regs = []
        for i, feat_head in enumerate(self.reg_features):
            reg_feature = feat_head(features)
            for j, out_head in enumerate(self.reg_heads[i]):
                output_reg = out_head(reg_feature)
                if [i, j] == self.offset_index:                              # :136-158
                    grid = edge_indices.view(b, -1, 1, 2).to(features.dtype)
                    grid = torch.stack((grid[..., 0] / (w - 1) * 2 - 1, grid[..., 1] / (h - 1) * 2 - 1), -1)
                    fused = torch.cat((feature_cls, reg_feature), dim=1)
                    edge_feat = F.grid_sample(fused, grid, align_corners=True).squeeze(-1)
                    edge_cls = self.trunc_heatmap_conv(edge_feat[:, :self.head_conv])
                    edge_off = self.trunc_offset_conv(edge_feat[:, self.head_conv:])
                    if taps is not None:
                        taps['edge_cls'], taps['edge_off'] = edge_cls, edge_off
                    for k in range(b):
                        n = int(edge_lens[k])
                        idx = edge_indices[k, :n]
                        output_cls[k, :, idx[:, 1], idx[:, 0]] += edge_cls[k, :, :n]
                        output_reg[k, :, idx[:, 1], idx[:, 0]] += edge_off[k, :, :n]
                output_regs.append(output_reg)
        if taps is not None:
            taps['cls_logits'] = output_cls.clone()
        cls = torch.sigmoid(output_cls).clamp(min=1e-4, max=1 - 1e-4)        # layers/utils.py:39-43
        return {'cls': cls, 'reg': torch.cat(output_regs, dim=1)}


# --------------------------------------------------------------------------------------
# Decode (model/layers/utils.py:45-145, model/head/detector_infer.py:77-237,
# model/anno_encoder.py:69-86,124-155,187-295).  One image at a time: the reference decode is
# batch-1 only.
# --------------------------------------------------------------------------------------
def nms_hm(hm):
    hmax = F.max_pool2d(hm, 3, stride=1, padding=1)
    return hm * (hmax == hm).float()


def select_topk(hm, K=50):
    """layers/utils.py:61-100 with the torch-1.4 integer floor division made explicit."""
    b, c, h, w = hm.shape
    flat = hm.view(b, c, -1)
    sc_all, ind_all = torch.topk(flat, K)
    ys_all = torch.div(ind_all, w, rounding_mode='floor').float()
    xs_all = (ind_all % w).float()
    sc, ind = torch.topk(sc_all.view(b, -1), K)
    cls = torch.div(ind, K, rounding_mode='floor').float()
    ind_all = ind_all.view(b, -1).gather(1, ind)
    ys = ys_all.view(b, -1).gather(1, ind)
    xs = xs_all.view(b, -1).gather(1, ind)
    return sc, ind_all, cls, ys, xs


class Calib:
    """The 6 scalars decode needs, derived as data/datasets/kitti_utils.py:213-218."""
    def __init__(self, P):
        P = np.asarray(P, dtype=np.float64).reshape(3, 4)
        self.P = P
        self.c_u, self.c_v, self.f_u, self.f_v = P[0, 2], P[1, 2], P[0, 0], P[1, 1]
        self.b_x, self.b_y = P[0, 3] / (-self.f_u), P[1, 3] / (-self.f_v)

    def as_f32(self):
        return np.array([self.f_u, self.f_v, self.c_u, self.c_v, self.b_x, self.b_y], dtype=np.float32)


def box_iou(a, b):
    """engine/visualize_infer.py:23-27 (axis-aligned boxes, no +1)."""
    inter = max(min(a[2], b[2]) - max(a[0], b[0]), 0) * max(min(a[3], b[3]) - max(a[1], b[1]), 0)
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def oracle_depth_choice(boxes, clses, d_all, u_all, gt_boxes, gt_clses, gt_depths, iou_thresh=0.5):
    """detector_infer.py:238-277 `get_oracle_depths`: per detection, the nearest ground-truth box of its class (centre distance);
    if their 2D IoU reaches 0.5 the estimate closest to the true depth, otherwise the mean of the estimates.
    -> (depth, sigma, choice) with choice in {-1: mean, 0..3: column of d_all}."""
    depth, sigma = d_all.mean(dim=1), u_all.mean(dim=1)
    choice = torch.full((d_all.shape[0],), -1, dtype=torch.long)
    if gt_boxes.shape[0] == 0:
        return depth, sigma, choice
    gt_c = (gt_boxes[:, :2] + gt_boxes[:, 2:]) / 2
    for i in range(boxes.shape[0]):
        c = (boxes[i, :2] + boxes[i, 2:]) / 2
        dis = torch.sum((c.reshape(1, 2) - gt_c) ** 2, dim=1)
        dis[gt_clses != clses[i]] = 9999
        near = torch.argmin(dis)
        if box_iou(boxes[i].numpy(), gt_boxes[near].numpy()) < iou_thresh:
            continue
        k = torch.argmin(torch.abs(d_all[i] - gt_depths[near]))
        depth[i], sigma[i], choice[i] = d_all[i, k], u_all[i, k], k
    return depth, sigma, choice


def decode_image(cls_hm, reg, calib, pad_size, img_size, threshold=0.2, K=50, output_depth='soft', gt=None):
    """cls_hm (1,3,H,W) post sigmoid/clamp; reg (1,50,H,W); returns dict with top-K and (N,14) rows.
    `output_depth`: detector_infer.py:149-198; 'oracle' takes gt = dict(boxes (G,4), clses (G,), depths (G,))."""
    assert cls_hm.shape[0] == 1
    heat = nms_hm(cls_hm)
    scores, indexs, clses, ys, xs = select_topk(heat, K)
    out = dict(scores=scores[0].clone(), indexs=indexs[0].clone(), clses=clses[0].clone(),
               ys=ys[0].clone(), xs=xs[0].clone())
    pts = torch.stack((xs.view(-1), ys.view(-1)), dim=1)
    w = reg.shape[3]
    pois = reg[0].permute(1, 2, 0).reshape(-1, reg.shape[1])[indexs.view(-1)]     # utils.py:120-145
    out['pois'] = pois.clone()
    scores = scores.view(-1)
    valid = scores >= threshold
    out['valid'] = valid.clone()
    if valid.sum() == 0:                                                         # infer.py:106-113
        out['result'] = scores.new_zeros(0, 14)
        return out
    scores, clses, pts, pois = scores[valid], clses.view(-1)[valid], pts[valid], pois[valid]
    pad = torch.as_tensor(pad_size, dtype=torch.float32).view(1, 2)

    reg2d = F.relu(pois[:, key2channel('2d_dim')])
    off3d = pois[:, key2channel('3d_offset')]
    dim_off = pois[:, key2channel('3d_dim')]
    ori = torch.cat((pois[:, key2channel('ori_cls')], pois[:, key2channel('ori_offset')]), dim=1)

    # anno_encoder.py:69-86
    box = torch.cat((pts - reg2d[:, :2], pts + reg2d[:, 2:]), dim=1) * DOWN_RATIO - pad.repeat(1, 2)
    box[:, 0::2] = box[:, 0::2].clamp(min=0, max=img_size[0] - 1)
    box[:, 1::2] = box[:, 1::2].clamp(min=0, max=img_size[1] - 1)
    # anno_encoder.py:221-243  (exp, mean only)
    dims = dim_off.exp() * torch.tensor(DIM_MEAN, dtype=torch.float32)[clses.long()]
    # anno_encoder.py:124-140  inv_sigmoid
    d_direct = (1 / torch.sigmoid(pois[:, key2channel('depth')].squeeze(-1)) - 1).clamp(*DEPTH_RANGE)
    u_direct = pois[:, key2channel('depth_uncertainty')].exp()
    # anno_encoder.py:187-219
    kp = pois[:, key2channel('corner_offset')].view(-1, 10, 2)
    h3d = dims[:, 1]
    f_u = float(calib.f_u)
    dc = kp[:, 8, 1] - kp[:, 9, 1]
    d02 = kp[:, [0, 2], 1] - kp[:, [4, 6], 1]
    d13 = kp[:, [1, 3], 1] - kp[:, [5, 7], 1]
    zc = f_u * h3d / (F.relu(dc) * DOWN_RATIO + EPS_KPT)
    z02 = (f_u * h3d.unsqueeze(-1) / (F.relu(d02) * DOWN_RATIO + EPS_KPT)).mean(dim=1)
    z13 = (f_u * h3d.unsqueeze(-1) / (F.relu(d13) * DOWN_RATIO + EPS_KPT)).mean(dim=1)
    d_kpt = torch.stack([t.clamp(*DEPTH_RANGE) for t in (zc, z02, z13)], dim=1)
    u_kpt = pois[:, key2channel('corner_uncertainty')].exp()
    # infer.py:149-198: which estimate becomes the depth, and the uncertainty that goes with it
    d_all = torch.cat((d_direct.unsqueeze(1), d_kpt), dim=1)
    u_all = torch.cat((u_direct, u_kpt), dim=1)
    wts = 1 / u_all
    single = {'direct': 0, 'keypoints_center': 1, 'keypoints_02': 2, 'keypoints_13': 3}
    if output_depth == 'soft':
        wts = wts / wts.sum(dim=1, keepdim=True)
        depth = torch.sum(d_all * wts, dim=1)
        sigma = torch.sum(wts * u_all, dim=1)
    elif output_depth == 'hard':
        depth = d_all[torch.arange(d_all.shape[0]), wts.argmax(dim=1)]
        sigma = u_all.min(dim=1).values
    elif output_depth == 'mean':
        depth, sigma = d_all.mean(dim=1), u_all.mean(dim=1)
    elif output_depth == 'keypoints_avg':
        depth, sigma = d_kpt.mean(dim=1), u_kpt.mean(dim=1)
    elif output_depth in single:
        depth, sigma = d_all[:, single[output_depth]], u_all[:, single[output_depth]]
    elif output_depth == 'oracle':
        depth, sigma, out['oracle_choice'] = oracle_depth_choice(box, clses, d_all, u_all, gt['boxes'], gt['clses'], gt['depths'])
    else:
        raise ValueError(output_depth)
    # anno_encoder.py:142-155 + kitti_utils.py:350-369
    uv = (pts + off3d) * DOWN_RATIO - pad
    x = ((uv[:, 0] - float(calib.c_u)) * depth) / float(calib.f_u) + float(calib.b_x)
    y = ((uv[:, 1] - float(calib.c_v)) * depth) / float(calib.f_v) + float(calib.b_y)
    loc = torch.stack((x, y, depth), dim=1)
    # anno_encoder.py:245-295 multi-bin
    bins = torch.softmax(ori[:, :8].view(-1, 4, 2), dim=2)[..., 1]
    best = bins.argmax(dim=1)
    centers = torch.tensor([0, PI / 2, PI, -PI / 2], dtype=torch.float32)
    alphas = ori.new_zeros(ori.shape[0])
    for i in range(4):
        m = best == i
        alphas[m] = torch.atan2(ori[m, 8 + 2 * i], ori[m, 9 + 2 * i]) + centers[i]
    rotys = alphas + torch.atan2(loc[:, 0], loc[:, 2])
    rotys = torch.where(rotys > PI, rotys - 2 * PI, rotys)
    rotys = torch.where(rotys < -PI, rotys + 2 * PI, rotys)
    alphas = torch.where(alphas > PI, alphas - 2 * PI, alphas)
    alphas = torch.where(alphas < -PI, alphas + 2 * PI, alphas)
    loc[:, 1] += dims[:, 1] / 2                                                 # infer.py:215
    dims = dims.roll(shifts=-1, dims=1)                                         # (l,h,w)->(h,w,l)
    final = scores * (1 - torch.clamp(sigma, min=0.01, max=1))                  # infer.py:225-227
    out['result'] = torch.cat([clses.view(-1, 1), alphas.view(-1, 1), box, dims, loc,
                               rotys.view(-1, 1), final.view(-1, 1)], dim=1)
    out['vis_scores'] = scores
    return out


class KeypointDetectorRef(nn.Module):
    """model/detector.py:11-37 (eval path), returning the intermediate maps as well."""

    def __init__(self):
        super().__init__()
        self.backbone = DLASeg()
        self.heads = nn.Module()
        self.heads.predictor = Predictor(self.backbone.out_channels)

    def forward_maps(self, images, edge_indices, edge_lens, taps=None):
        feat = self.backbone(images, taps)
        if taps is not None:
            taps['feature'] = feat
        return self.heads.predictor(feat, edge_indices, edge_lens, taps)

    @torch.no_grad()
    def detect(self, images, targets):
        """targets: list of dicts(calib=Calib, pad_size=(px,py), size=(W,H), edge_indices, edge_len)."""
        ei = torch.stack([torch.as_tensor(t['edge_indices']) for t in targets])
        el = torch.as_tensor([int(t['edge_len']) for t in targets])
        maps = self.forward_maps(images, ei, el)
        return [decode_image(maps['cls'][i:i + 1], maps['reg'][i:i + 1], t['calib'], t['pad_size'], t['size'])
                for i, t in enumerate(targets)], maps
