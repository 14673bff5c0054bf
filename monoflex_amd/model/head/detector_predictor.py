"""Detection heads (reference model/head/detector_predictor.py:19-169) on the gfx950 kernels.

Parameter names follow the reference (`class_head.{0,1,2}`, `reg_features.{i}.{0,1}`,
`reg_heads.{i}.{j}`, `trunc_heatmap_conv.{0,1,3}`, `trunc_offset_conv.{0,1,3}`).  InPlaceABN
(third-party, not vendored) is held as a plain module with BatchNorm2d's parameter/buffer names;
its semantics here are BN(eps=1e-5) -> leaky_relu(0.01) (`abn_abs_weight`
selects upstream's |gamma|+eps variant).

Forward: one fused launch for the nine 3x3->ABN->1x1 branches, then the edge-fusion tail
(re-evaluates the two needed trunks at the <=832 border points only, Conv1d k3 (replicate pad) +
BN1d + Conv1d 1x1, scatter-add into the class / 3d_offset channels).
"""
import numpy as np
import torch
from torch import nn

from ... import autograd as AG
from ... import lib as L
from ... import ops

HM_LD = 64          # fp32 head map row: [0:3] class logits, [8:58] regression channels
REG_OFF = 8


# training: the sparse regression branches through the patch Gram matrix (gram_heads.py) instead of seven dense trunk maps; MFX_GRAM_HEADS=0 or
# GRAM_HEADS[0] = False brings the dense trunks back (tests compare the two)
GRAM_HEADS = [__import__("os").environ.get("MFX_GRAM_HEADS", "1") != "0"]
GRAM_OFFSET = [__import__("os").environ.get("MFX_GRAM_OFFSET", "1") != "0"]       # the 3d_offset branch in that sparse set too (its edge fusion reads gathered rows)


class InPlaceABN(nn.Module):
    """Parameter holder for the head's fused BN + leaky_relu(0.01): the parameter / buffer names of a BatchNorm2d
    (weight, bias, running_mean, running_var, num_batches_tracked -- the stand-in the golden fixtures were recorded
    with), but -- like upstream's inplace_abn.InPlaceABN -- NOT a torch `_BatchNorm` subclass, so
    `torch.nn.SyncBatchNorm.convert_sync_batchnorm` leaves the nine head ABNs on rank-local statistics exactly as in the
    reference (tools/plain_train_net.py:131-132)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu", activation_param=0.01):
        super().__init__()
        if not affine:
            raise NotImplementedError("InPlaceABN holder: affine=True only")
        self.num_features, self.eps, self.momentum, self.affine, self.track_running_stats = num_features, eps, momentum, True, True
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.activation, self.activation_param = activation, activation_param

    def forward(self, x):
        raise RuntimeError("InPlaceABN is a parameter holder: the fused HIP heads kernel applies it")


class _predictor(nn.Module):
    fused_edge_nodes = True         # training: head conv + edge-row gather as one autograd node, border scatter as one kernel (False: torch indexing)

    def __init__(self, cfg, in_channels):
        super().__init__()
        classes = len(cfg.DATASETS.DETECT_CLASSES)
        self.regression_head_cfg = cfg.MODEL.HEAD.REGRESSION_HEADS
        self.regression_channel_cfg = cfg.MODEL.HEAD.REGRESSION_CHANNELS
        self.output_width = cfg.INPUT.WIDTH_TRAIN // cfg.MODEL.BACKBONE.DOWN_RATIO
        self.output_height = cfg.INPUT.HEIGHT_TRAIN // cfg.MODEL.BACKBONE.DOWN_RATIO
        self.head_conv = cfg.MODEL.HEAD.NUM_CHANNEL
        self.use_inplace_abn = cfg.MODEL.INPLACE_ABN
        self.bn_momentum = cfg.MODEL.HEAD.BN_MOMENTUM
        self.abn_abs_weight = False
        if not self.use_inplace_abn or cfg.MODEL.HEAD.USE_NORMALIZATION != "BN":
            raise NotImplementedError("the HIP heads implement the runs/monoflex.yaml configuration "
                                      "(INPLACE_ABN True, BN): conv3x3 -> BN -> leaky_relu(0.01) -> conv1x1")
        if self.head_conv != 256 or in_channels != 64:
            raise NotImplementedError("fused heads kernel is built for 64 -> 256 trunks")

        def trunk():
            return [nn.Conv2d(in_channels, self.head_conv, kernel_size=3, padding=1, bias=False),
                    InPlaceABN(self.head_conv, momentum=self.bn_momentum, activation="leaky_relu")]
        self.class_head = nn.Sequential(*trunk(), nn.Conv2d(self.head_conv, classes, kernel_size=1, padding=0, bias=True))
        self.class_head[-1].bias.data.fill_(-np.log(1 / cfg.MODEL.HEAD.INIT_P - 1))
        self.reg_features, self.reg_heads = nn.ModuleList(), nn.ModuleList()
        for idx, keys in enumerate(self.regression_head_cfg):
            self.reg_features.append(nn.Sequential(*trunk()))
            head_list = nn.ModuleList()
            for key_index, key in enumerate(keys):
                out_head = nn.Conv2d(self.head_conv, self.regression_channel_cfg[idx][key_index], kernel_size=1, bias=True)
                if key.find('uncertainty') >= 0 and cfg.MODEL.HEAD.UNCERTAINTY_INIT:
                    torch.nn.init.xavier_normal_(out_head.weight, gain=0.01)
                if key == '3d_offset':
                    self.offset_index = [idx, key_index]
                nn.init.constant_(out_head.bias, 0)
                head_list.append(out_head)
            self.reg_heads.append(head_list)

        self.enable_edge_fusion = cfg.MODEL.HEAD.ENABLE_EDGE_FUSION
        self.edge_fusion_kernel_size = cfg.MODEL.HEAD.EDGE_FUSION_KERNEL_SIZE
        self.edge_fusion_relu = cfg.MODEL.HEAD.EDGE_FUSION_RELU
        if self.enable_edge_fusion:
            if cfg.MODEL.HEAD.EDGE_FUSION_NORM != 'BN' or self.edge_fusion_kernel_size != 3:
                raise NotImplementedError("edge fusion: k=3 + BN1d configuration only")
            act = nn.ReLU(inplace=True) if self.edge_fusion_relu else nn.Identity()

            def trunc(cout):
                k = self.edge_fusion_kernel_size
                return nn.Sequential(nn.Conv1d(self.head_conv, self.head_conv, kernel_size=k, padding=k // 2, padding_mode='replicate'),
                                     nn.BatchNorm1d(self.head_conv, momentum=self.bn_momentum), act,
                                     nn.Conv1d(self.head_conv, cout, kernel_size=1))
            self.trunc_heatmap_conv = trunc(classes)
            self.trunc_offset_conv = trunc(2)
        self.num_classes = classes
        self._packs = {}

    # ---- packing ---------------------------------------------------------------------------------
    def _abn_fold(self, abn):
        if self.abn_abs_weight:                                   # upstream inplace_abn: gamma_eff = |gamma| + eps
            g = abn.weight.detach().float().abs() + abn.eps
            scale = g / torch.sqrt(abn.running_var.detach().float() + abn.eps)
            return scale, abn.bias.detach().float() - abn.running_mean.detach().float() * scale
        return ops.fold_bn(abn)

    def _pack(self, dtype):
        key = ("heads", dtype)
        if key in self._packs:
            return self._packs[key]
        trunks = [self.class_head] + list(self.reg_features)
        w1, sc, sh = [], [], []
        for t in trunks:
            w1.append(t[0].weight.detach().float().permute(0, 2, 3, 1).reshape(self.head_conv, -1))
            s, b = self._abn_fold(t[1])
            sc.append(s); sh.append(b)
        K = w1[0].shape[1]
        assert K == 576 and self.head_conv == 256
        E = 4 if dtype in (torch.float32, ops.F16X2) else 8
        steps = K // (4 * E)
        nb = len(trunks)
        dev = w1[0].device
        w2_scale = None
        if dtype == ops.F16X2:                                     # weights times a power of two per branch (ops.split_weight_scale), undone by scale1 / w2_scale
            for i in range(len(w1)):
                ws = ops.split_weight_scale(w1[i])
                w1[i], sc[i] = w1[i] * ws, sc[i] / ws
        # 3x3 weights, fragment-major: [branch][wave wn 4][step][frag j 4][k-group kq 4][row nl 16][E]  (lane = kq*16+nl)
        W1 = ops.cast_operand(torch.stack(w1, 0).view(nb, 4, 4, 16, steps, 4, E).permute(0, 1, 4, 2, 5, 3, 6).contiguous(), dtype)
        if dtype == ops.F16X2:
            W1 = ops.pair_steps(W1, 2)                              # [branch][wn][step pair][hi | lo][j][kq][nl][4]: heads.hip walks K in step pairs
        w2 = torch.zeros(nb, 32, self.head_conv, device=dev)
        b2 = torch.zeros(nb, 32, device=dev)
        ch_off, c_out = [0], [self.num_classes]
        w2[0, :self.num_classes] = self.class_head[2].weight.detach().float().reshape(self.num_classes, -1)
        b2[0, :self.num_classes] = self.class_head[2].bias.detach().float()
        off = REG_OFF
        for i, heads in enumerate(self.reg_heads):
            r = 0
            for h in heads:
                c = h.weight.shape[0]
                w2[i + 1, r:r + c] = h.weight.detach().float().reshape(c, -1)
                b2[i + 1, r:r + c] = h.bias.detach().float()
                r += c
            ch_off.append(off); c_out.append(r)
            off += r
        assert off <= HM_LD
        if dtype == ops.F16X2:
            w2_scale = []
            for i in range(nb):
                ws = ops.split_weight_scale(w2[i])
                w2[i] *= ws
                w2_scale.append(1.0 / ws)
        # 1x1 weights, fragment-major with the K order the kernel's accumulators arrive in (heads.hip TrunkPack):
        #   bf16: [branch][wn][kb 2][of 2][g 4][o_l 16][half 2][q 4], trunk channel n = 64wn + 32kb + 16half + 4g + q
        #   f32 : [branch][wn][kb 4][of 2][g 4][o_l 16][e 4],          n = 64wn + 16kb + 4g + e
        if dtype in (torch.float32, ops.F16X2):
            W2 = ops.cast_operand(w2.view(nb, 2, 16, 4, 4, 4, 4).permute(0, 3, 4, 1, 5, 2, 6).contiguous(), dtype)
            if dtype == ops.F16X2:
                W2 = ops.pair_steps(W2, 2)                          # [branch][wn][kb pair][hi | lo][of][g][o_l][4]
        else:
            W2 = w2.view(nb, 2, 16, 4, 2, 2, 4, 4).permute(0, 3, 4, 1, 6, 2, 5, 7).contiguous().to(dtype)
        p = ops.PackedHeads(W1, torch.cat(sc).contiguous(), torch.cat(sh).contiguous(),
                            W2, b2.contiguous(), K, ch_off, c_out, HM_LD, split=dtype == ops.F16X2, w2_scale=w2_scale)
        if dtype in (torch.bfloat16, torch.float16):
            # the same weights for the v_mfma_f32_32x32x16 form (csrc/heads.hip heads_fused32_kernel; option "heads_mfma32"):
            #   3x3: [branch][wn 4][K-step 36][rb 2][h 2][row 32][8], channel 64 wn + 32 rb + row, k = 16 s + 8 h + e
            #   1x1: [branch][wn 4][rb 2][t 2][h 2][o 32][a 2][q 4], trunk channel n = 64 wn + 32 rb + 16 t + 8 a + 4 h + q (k-slot e = 4 a + q)
            w1s = torch.stack(w1, 0)
            p.w1_32 = w1s.view(nb, 4, 2, 32, 36, 2, 8).permute(0, 1, 4, 2, 5, 3, 6).contiguous().to(dtype)
            p.w2_32 = w2.view(nb, 32, 4, 2, 2, 2, 2, 4).permute(0, 2, 3, 4, 6, 1, 5, 7).contiguous().to(dtype)
        # edge fusion: trunks of the class branch and of the 3d_offset branch at the border points
        if self.enable_edge_fusion:
            oi = self.offset_index[0]
            w_e = torch.cat((self.class_head[0].weight, self.reg_features[oi][0].weight), 0)
            s0, b0 = self._abn_fold(self.class_head[1])
            s1, b1 = self._abn_fold(self.reg_features[oi][1])
            p.edge_trunk = ops.pack_conv(w_e, dtype, torch.cat((s0, s1)), torch.cat((b0, b1)), stride=1, pad=1, act=L.ACT_LEAKY)
            p.edge_branches = []
            for seq, cout, choff in ((self.trunc_heatmap_conv, self.num_classes, 0),
                                     (self.trunc_offset_conv, 2, REG_OFF + sum(sum(c) for c in self.regression_channel_cfg[:oi])
                                      + sum(self.regression_channel_cfg[oi][:self.offset_index[1]]))):
                c1, bn, c3 = seq[0], seq[1], seq[3]
                scale, shift = ops.fold_bn(bn, c1.bias)
                # Conv1d weight (256,256,3) -> conv over a 1 x (L+2) "image", taps along W
                pk1 = ops.pack_conv(c1.weight.detach().unsqueeze(2), dtype, scale, shift, stride=1, pad=0,
                                    act=L.ACT_RELU if self.edge_fusion_relu else L.ACT_NONE)
                pk2 = ops.pack_conv(c3.weight.detach().unsqueeze(2), dtype, None, c3.bias, stride=1, pad=0, act=L.ACT_NONE, cout=4)
                p.edge_branches.append((pk1, pk2, cout, choff))
        self._packs[key] = p
        return p

    # ---- forward ---------------------------------------------------------------------------------
    def forward_nhwc(self, features, edge_indices=None, edge_lens=None, edge_rowmap=None):
        """features (B,H,W,64) NHWC -> fp32 head map (B,H,W,64): [0:3] class logits (pre-sigmoid, after
        edge fusion), [8:58] the 50 regression channels.  edge_indices int32 (B,L,2) (x,y), edge_lens int32 (B,)."""
        if self.training:
            raise RuntimeError("forward_nhwc is the fused eval path; training goes through forward_train")
        p = self._pack(ops.compute_tag(self, features.dtype))
        hm, planar = ops.heads_fused(features, p, planar_classes=self.num_classes)
        self.last_cls_planar = planar                               # (B,3,H*W) class logits for the top-K kernel
        if self.enable_edge_fusion:
            if edge_indices is None:
                raise ValueError("edge fusion is enabled: targets must carry edge_indices / edge_len")
            B, H, W, _ = features.shape
            Lmax = edge_indices.shape[1]
            rowmap = edge_rowmap if edge_rowmap is not None else make_edge_rowmap(edge_indices, H, W)
            trunk = ops.conv2d(features, p.edge_trunk, rowmap=rowmap)   # (B*(L+2), 512)
            trunk = trunk.view(B, 1, Lmax + 2, 2 * self.head_conv)
            for bi, (pk1, pk2, cout, choff) in enumerate(p.edge_branches):
                f1 = ops.conv2d(trunk, pk1, x_ch_off=bi * self.head_conv)                  # (B,1,L,256)
                o = ops.conv2d(f1, pk2, out_dtype=torch.float32)                           # (B,1,L,4) fp32
                ops.edge_scatter_add(hm, choff, cout, o, edge_indices, edge_lens, planar=planar if choff == 0 else None)
        return hm

    def forward_train(self, features, edge_indices=None, edge_lens=None, object_rows=None, plan=None):
        """Training form (detector_predictor.py:125-169), unfused and differentiable: per branch
        conv3x3 -> ABN(batch statistics, leaky 0.01) -> one 1x1 conv over the branch's stacked heads; edge fusion
        gathers the two trunks at the border points.  Returns (class logits (B,H,W,ncls), regression (B,H,W,50)).

        With `object_rows` (the loss's packed object table, fp32 [N,72]) the regression branches whose activation nothing else
        reads are evaluated at the object centres only (csrc/head_sparse.hip) and the second result is the GATHERED table
        (N,50) -- the rows select_point_of_interest would pick (layers/utils.py:120-145); the class head (dense focal loss) and
        the 3d_offset head (edge fusion reads its trunk) keep the dense path.

        `plan` (`edge_plan`): the index tensors of the edge fusion that are functions of the targets alone, prepared with the batch
        (engine/trainer.prepare_targets) instead of being rebuilt by ~45 tiny launches inside every step."""
        B, H, W, _ = features.shape
        trunks = [self.class_head] + list(self.reg_features)
        lasts = [[self.class_head[2]]] + [list(h) for h in self.reg_heads]
        oi = self.offset_index[0]
        sparse = object_rows is not None
        feats, outs, sp = [], [], []
        # the sparse regression branches need no dense trunk map at all when their statistics come from the input's patch Gram matrix
        # (monoflex_amd/gram_heads.py); the class head (dense focal loss) and the 3d_offset head (edge fusion) keep theirs
        gram = sparse and GRAM_HEADS[0]
        fuse_nodes = self.enable_edge_fusion and self.fused_edge_nodes
        # ... and the 3d_offset branch joins them where the edge fusion reads its trunk through the gathered-rows node: the fusion needs the
        # trunk ACTIVATION at the <= L + 2 edge-sequence pixels of every image, the loss needs the head at the object centres -- both sparse
        gram_off = gram and GRAM_OFFSET[0] and (fuse_nodes or not self.enable_edge_fusion)
        is_sparse = lambda bi: sparse and bi != 0 and (bi - 1 != oi or gram_off)                      # noqa: E731
        dense_ids = [bi for bi in range(len(trunks)) if not (gram and is_sparse(bi))]
        ys = dict(zip(dense_ids, AG.fanout_conv(features, [trunks[bi][0].weight for bi in dense_ids], 1)))   # one summed gradient for `features`
        edge_rows = {}
        if self.enable_edge_fusion:
            if edge_indices is None:
                raise ValueError("edge fusion is enabled: targets must carry edge_indices / edge_len")
            if fuse_nodes:                                   # functions of the targets only, shared by both fusions
                if plan is None:
                    plan = self.edge_plan(edge_indices, edge_lens, None, H, W)
                rowmap, rows_center, valid_l = plan["rowmap"], plan["rows_center"], plan["valid_l"]
        for bi, (t, heads) in enumerate(zip(trunks, lasts)):
            w = heads[0].weight if len(heads) == 1 else torch.cat([h.weight for h in heads], 0)
            b = heads[0].bias if len(heads) == 1 else torch.cat([h.bias for h in heads], 0)
            y, done = ys.get(bi), False
            if is_sparse(bi):
                sp.append((bi - 1, y, t[1], w, b, done, t[0].weight))
                feats.append(None); outs.append(None)
                continue
            f = AG.bn_act(y, t[1], L.ACT_LEAKY, stats_done=done)
            feats.append(f)
            if fuse_nodes and (bi == 0 or bi - 1 == oi):
                # head conv + gather of the trunk's rows at the edge pixels in one node (one data gradient for f)
                yo, er = AG.HeadConvGatherFn.apply(f, w, b, rowmap)
                outs.append(yo if yo.shape[-1] == w.shape[0] else yo[..., :w.shape[0]])
                edge_rows[bi] = er
            else:
                outs.append(AG.conv2d(f, w, b, 1, 0, out_dtype=torch.float32))
        cls, regs = outs[0], outs[1:]
        if self.enable_edge_fusion:
            oi, oj = self.offset_index
            Lmax = edge_indices.shape[1]
            if not fuse_nodes:
                pos = torch.arange(-1, Lmax + 1, device=features.device).clamp(0, Lmax - 1)      # replicate padding, k=3
                xy = edge_indices[:, pos].long()
                bidx = torch.arange(B, device=features.device).view(B, 1).expand(B, Lmax + 2)
            new = []
            gram_tab = None
            if gram_off:
                # the sparse set as ONE node: the (N, 50) table of all eight regression branches at the object centres + the 3d_offset trunk's
                # activation at the edge-sequence pixels (its edge-fusion input)
                gram_tab, edge_rows[1 + oi] = self._gram_table(features, object_rows, sp, rowmap if fuse_nodes else None, oi)
            for bi_f, f, seq, base in ((0, feats[0], self.trunc_heatmap_conv, cls), (1 + oi, feats[1 + oi], self.trunc_offset_conv, regs[oi])):
                if fuse_nodes:
                    e = edge_rows[bi_f].view(B, 1, Lmax + 2, self.head_conv)
                else:
                    e = f[bidx, xy[..., 1], xy[..., 0]].view(B, 1, Lmax + 2, self.head_conv)  # grid_sample at integer points
                c1, bn, c3 = seq[0], seq[1], seq[3]
                h1 = AG.bn_act(AG.conv2d(e, c1.weight.unsqueeze(2), c1.bias, 1, 0), bn,
                               L.ACT_RELU if self.edge_fusion_relu else L.ACT_NONE)
                o = AG.conv2d(h1, c3.weight.unsqueeze(2), c3.bias, 1, 0, out_dtype=torch.float32).view(B, Lmax, -1)
                lo = 0 if base is cls else sum(self.regression_channel_cfg[oi][:oj])
                co = o.shape[-1]
                if fuse_nodes and base is None:                 # (3d_offset in the sparse set: its fused edge rows are merged into the table below)
                    o_off, lo_off = o, sum(sum(c) for c in self.regression_channel_cfg[:oi]) + lo
                    new.append(None)
                    continue
                if fuse_nodes:
                    new.append(AG.EdgeScatterAddFn.apply(base, o, edge_indices, edge_lens, lo, rows_center, valid_l))
                    continue
                # static-shape scatter (no nonzero()/host sync, graph-capturable): positions >= edge_len contribute zeros;
                # the valid border pixels are unique, so accumulate == the reference's '+='
                valid = (torch.arange(Lmax, device=features.device).view(1, Lmax) < edge_lens.view(B, 1).long()).to(o.dtype)
                ys, xs = edge_indices[..., 1].long(), edge_indices[..., 0].long()
                bl = torch.arange(B, device=features.device).view(B, 1).expand(B, Lmax)
                add = torch.zeros(B, H, W, co, dtype=o.dtype, device=o.device).index_put((bl, ys, xs), o * valid.unsqueeze(-1), accumulate=True)
                if co != base.shape[-1]:
                    add = torch.nn.functional.pad(add, (lo, base.shape[-1] - lo - co))
                new.append(base + add)
            cls, regs[oi] = new[0], new[1]
        if not sparse:
            return cls, torch.cat(regs, dim=3)
        # gathered regression table in the reference's channel order
        starts = [sum(sum(c) for c in self.regression_channel_cfg[:i]) for i in range(len(self.regression_channel_cfg))]
        rows = object_rows
        if gram_off:
            if gram_tab is None:                                 # (no edge fusion: the table alone)
                gram_tab, _ = self._gram_table(features, rows, sp, None, oi)
            elif fuse_nodes:
                # an object centre that is a border pixel receives that pixel's fused edge output (valid border pixels are unique per image):
                # slot map pixel -> edge position, -1 elsewhere; invalid positions write to a dummy slot
                Lmax = edge_indices.shape[1]
                if "e_idx" not in plan:
                    plan = dict(plan, **self.edge_plan_rows(rows, rows_center, valid_l, B, H, W))
                co = o_off.shape[-1]
                # (index_select / index_add: their gradients are one launch each; advanced indexing sorts its indices in the backward pass)
                add = o_off.reshape(B * Lmax, co).float().index_select(0, plan["e_idx"]) * plan["hit"]
                gram_tab = gram_tab.index_add(1, plan["off_cols"], add)                       # columns lo_off .. lo_off + co
            return cls, gram_tab
        if gram:
            from monoflex_amd.gram_heads import gram_reg_heads
            tab, _ = gram_reg_heads(features, rows, [e[2] for e in sp], [starts[e[0]] for e in sp], 50, [e[6] for e in sp],
                                    [e[2].weight for e in sp], [e[2].bias for e in sp], [e[3] for e in sp], [e[4] for e in sp],
                                    sync=any(bool(getattr(e[2], "sync_bn", False)) for e in sp))
        else:
            tab = AG.SparseRegHeadsFn.apply(rows, tuple(e[2] for e in sp), tuple(starts[e[0]] for e in sp), 50, tuple(e[5] for e in sp),
                                            *[e[1] for e in sp], *[e[2].weight for e in sp], *[e[2].bias for e in sp],
                                            *[e[3] for e in sp], *[e[4] for e in sp])
        bidx, cx, cy = rows[:, 57].long().clamp(0, B - 1), rows[:, 2].long().clamp(0, W - 1), rows[:, 3].long().clamp(0, H - 1)
        lo, n_off = starts[oi], sum(self.regression_channel_cfg[oi])
        # the dense 3d_offset head at the centres (index_select: its gradient is one index_add_, no sort as behind advanced indexing)
        off_rows = regs[oi].reshape(-1, regs[oi].shape[-1]).index_select(0, (bidx * H + cy) * W + cx)[:, :n_off]
        return cls, torch.cat((tab[:, :lo], off_rows, tab[:, lo + n_off:]), dim=1)

    def edge_plan(self, edge_indices, edge_lens, object_rows=None, H=None, W=None):
        """The edge fusion's index tensors, functions of the targets only (static shapes, no host sync): `rowmap` long [B (L + 2)] pixel rows of
        the sequence positions -1 .. L (replicate padding, k = 3), `rows_center` [B L] the positions 0 .. L - 1, `valid_l` float [B, L, 1] =
        (l < edge_len[b]); with the loss's object table also `e_idx` / `hit` (`edge_plan_rows`)."""
        H, W = H or self.output_height, W or self.output_width
        B, Lm = edge_indices.shape[0], edge_indices.shape[1]
        rm = make_edge_rowmap(edge_indices, H, W).long()
        plan = {"rowmap": rm, "rows_center": rm.view(B, Lm + 2)[:, 1:-1].reshape(-1).contiguous(),
                "valid_l": (torch.arange(Lm, device=rm.device).view(1, Lm) < edge_lens.view(B, 1)).float().unsqueeze(-1)}
        if object_rows is not None:
            plan.update(self.edge_plan_rows(object_rows, plan["rows_center"], plan["valid_l"], B, H, W))
        return plan

    def edge_plan_rows(self, rows, rows_center, valid_l, B, H, W):
        """`e_idx` long [N]: the flat edge position whose border pixel is object n's centre pixel (0 where none), `hit` float [N, 1]: whether there
        is one, `off_cols` long [2]: the columns of the regression table the fused 3d_offset output is added to."""
        e_idx, hit = edge_position_of_rows(rows, rows_center, valid_l, B, H, W)
        oi, oj = self.offset_index
        lo = sum(sum(c) for c in self.regression_channel_cfg[:oi]) + sum(self.regression_channel_cfg[oi][:oj])
        return {"e_idx": e_idx.clamp_min(0), "hit": hit.unsqueeze(1).to(torch.float32),
                "off_cols": torch.arange(lo, lo + self.trunc_offset_conv[3].out_channels, device=rows.device)}

    def _gram_table(self, features, rows, sp, edge_rowmap, oi):
        """All sparse branches in `sp` through monoflex_amd/gram_heads.py: (table (N, 50), the 3d_offset trunk's activation rows at `edge_rowmap`)."""
        from monoflex_amd.gram_heads import gram_reg_heads
        starts = [sum(sum(c) for c in self.regression_channel_cfg[:i]) for i in range(len(self.regression_channel_cfg))]
        jb = [k for k, e in enumerate(sp) if e[0] == oi]
        return gram_reg_heads(features, rows, [e[2] for e in sp], [starts[e[0]] for e in sp], 50, [e[6] for e in sp],
                              [e[2].weight for e in sp], [e[2].bias for e in sp], [e[3] for e in sp], [e[4] for e in sp],
                              sync=any(bool(getattr(e[2], "sync_bn", False)) for e in sp),
                              extra_branch=jb[0] if (edge_rowmap is not None and jb) else -1, extra_rows=edge_rowmap if jb else None)

    def forward(self, features, targets, object_rows=None):
        """Reference surface: features (B,64,H,W) (any strides) + targets -> {'cls','reg'} NCHW views.  Training with
        `object_rows` (see forward_train): {'cls', 'reg_rows' (N,50), 'cls_logits_nhwc'} -- what the loss reads, no dense 'reg'."""
        x = features.permute(0, 2, 3, 1).contiguous()
        ei, el = getattr(targets, "edge", None) or stack_edge_fields(targets, x.device)
        if self.training and object_rows is not None:
            logits, reg_rows = self.forward_train(x, ei, el, object_rows, plan=getattr(targets, "edge_plan", None))
            # 'cls' (see below) is built when somebody reads it: the fused loss takes the raw logits
            return _LazyMaps(lambda: torch.sigmoid(logits).clamp(min=1e-4, max=1 - 1e-4).permute(0, 3, 1, 2),
                             {'reg': None, 'reg_rows': reg_rows, 'cls_logits_nhwc': logits})
        if self.training:
            logits, reg = self.forward_train(x, ei, el)
            # 'cls' keeps the reference's contract (sigmoid_hm of the logits, NCHW, differentiable); this build's loss uses the raw
            # NHWC logits and does sigmoid + clamp + focal + its gradient in one kernel, so the backward of this view never runs
            # there -- but any other consumer of maps['cls'] (or the loss's tensor-op fallback) still trains the class head
            cls = torch.sigmoid(logits).clamp(min=1e-4, max=1 - 1e-4)
            return {'cls': cls.permute(0, 3, 1, 2), 'reg': reg.permute(0, 3, 1, 2), 'cls_logits_nhwc': logits}
        hm = self.forward_nhwc(x, ei, el)
        cls = torch.sigmoid(hm[..., :self.num_classes]).clamp(min=1e-4, max=1 - 1e-4).permute(0, 3, 1, 2)
        return {'cls': cls, 'reg': hm[..., REG_OFF:REG_OFF + 50].permute(0, 3, 1, 2), 'hm_nhwc': hm, 'cls_planar': self.last_cls_planar}


class _LazyMaps(dict):
    """The predictor's output dict whose 'cls' entry (sigmoid + clamp of the logits, NCHW view: two launches and an autograd branch) is
    computed on first access."""

    def __init__(self, make_cls, items):
        super().__init__(items)
        self._make_cls = make_cls

    def __missing__(self, key):
        if key != 'cls':
            raise KeyError(key)
        self['cls'] = v = self._make_cls()
        return v

    def get(self, key, default=None):
        return self[key] if key == 'cls' else super().get(key, default)

    def __contains__(self, key):
        return key == 'cls' or super().__contains__(key)


def edge_position_of_rows(rows, rows_center, valid_l, B, H, W):
    """For every row of the object table (fp32 [N, 72]: column 0 valid flag, 57 image, 2 / 3 centre x / y): the flat edge-sequence position
    b * L + l whose border pixel IS the object's centre pixel, and whether there is one.  `rows_center` = flat pixel of every (b, l),
    `valid_l` = (l < edge_len[b]); the valid border pixels of an image are unique, invalid positions are written to a dummy slot.
    Static shapes, no host sync (used inside the captured training step)."""
    npx = B * H * W
    ok = valid_l.reshape(-1) > 0
    slot = torch.full((npx + 1,), -1, dtype=torch.long, device=rows.device)
    slot.scatter_(0, torch.where(ok, rows_center, torch.full_like(rows_center, npx)), torch.arange(rows_center.numel(), device=rows.device))
    bidx, cx, cy = rows[:, 57].long().clamp(0, B - 1), rows[:, 2].long().clamp(0, W - 1), rows[:, 3].long().clamp(0, H - 1)
    e_idx = slot[(bidx * H + cy) * W + cx]
    return e_idx, (e_idx >= 0) & (rows[:, 0] > 0)


def make_edge_rowmap(edge_indices, H, W):
    """int32 [B*(L+2)] pixel rows of the edge sequence positions -1..L (replicate padding of the k=3 Conv1d,
    detector_predictor.py:111-119).  Depends on the targets only: computed once per batch, not per forward."""
    B, Lmax = edge_indices.shape[0], edge_indices.shape[1]
    pos = torch.arange(-1, Lmax + 1, device=edge_indices.device).clamp(0, Lmax - 1)
    xy = edge_indices[:, pos].long()
    rm = torch.arange(B, device=edge_indices.device).view(B, 1) * (H * W) + xy[..., 1] * W + xy[..., 0]
    return rm.to(torch.int32).reshape(-1).contiguous()


def stack_edge_fields(targets, device):
    ei = torch.stack([torch.as_tensor(t.get_field("edge_indices")) for t in targets]).to(device=device, dtype=torch.int32)
    el = torch.stack([torch.as_tensor(t.get_field("edge_len")) for t in targets]).to(device=device, dtype=torch.int32)
    return ei.contiguous(), el.contiguous()


def make_predictor(cfg, in_channels):
    if cfg.MODEL.HEAD.PREDICTOR != "Base_Predictor":
        raise KeyError(cfg.MODEL.HEAD.PREDICTOR)
    return _predictor(cfg, in_channels)
