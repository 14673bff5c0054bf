"""Training loss of the detection head (reference model/head/detector_loss.py:21-517, with the helpers it calls:
model/anno_encoder.py:88-295, model/layers/focal_loss.py:29-55, model/layers/iou_loss.py:7-49,
model/layers/utils.py:120-145, data/datasets/kitti_utils.py:350-369).

Same constructor, `prepare_targets`, `__call__(predictions, targets) -> (loss_dict, log_loss_dict)`, the same 11
loss names, weights and values.  The formulation differs: the reference compacts the valid objects with boolean
indexing (dynamic shapes, one host sync per index, ~30 `.item()` calls); here every per-object quantity keeps its
static (B*MAX_OBJECTS) shape and selections are 0/1 weights, so

    mean over selected rows  ==  sum(w * v) / sum(w)

is evaluated without leaving the device -- no sync inside the loss, one batched transfer for the log dict -- and the
whole step can be stream-ordered behind the backward kernels.  Rows that are not selected are fed neutral inputs
before any division / log so that no NaN can leak into a gradient through a zero weight.

Objects-free batches give zero regression losses here; the reference raises (UnboundLocalError) on them.
"""
import math

import torch
import torch.nn.functional as F

from ..layers.utils import Converter_key2channel

PI = math.pi


def make_loss_evaluator(cfg):
    return Loss_Computation(cfg)


class _LossDict(dict):
    """The loss terms by name; `.total` (optional) = their sum, already on the autograd graph (engine/trainer.total_loss)."""
    total = None


def _wmean(v, w, floor=1.0):
    """sum(w*v)/max(sum(w), floor): mean of v over the rows selected by the 0/1 weights w."""
    return (v * w).sum() / torch.clamp(w.sum(), min=floor)


class Loss_Computation:
    def __init__(self, cfg):
        H = cfg.MODEL.HEAD
        self.key2channel = Converter_key2channel(keys=H.REGRESSION_HEADS, channels=H.REGRESSION_CHANNELS)
        self.max_objs = cfg.DATASETS.MAX_OBJECTS
        self.loss_keys = list(H.LOSS_NAMES)
        self.loss_weights = dict(zip(H.LOSS_NAMES, H.INIT_LOSS_WEIGHT))
        self.dim_weight = torch.as_tensor(H.DIMENSION_WEIGHT, dtype=torch.float32).view(1, 3)
        self.uncertainty_range = H.UNCERTAINTY_RANGE
        self.focal_alpha, self.focal_beta = H.LOSS_PENALTY_ALPHA, H.LOSS_BETA
        lt = list(H.LOSS_TYPE)
        if H.HEATMAP_TYPE != 'centernet' or lt[1] != 'L1' or lt[3] != 'L1' or lt[2] not in ('giou', 'iou', 'linear_iou'):
            raise NotImplementedError("loss types %s / heatmap %s: runs/monoflex.yaml uses centernet focal + L1 + giou + L1"
                                      % (lt, H.HEATMAP_TYPE))
        self.iou_type = lt[2]
        if cfg.INPUT.ORIENTATION != 'multi-bin':
            raise NotImplementedError("orientation loss: the multi-bin encoding of runs/monoflex.yaml")
        self.orien_bin_size = cfg.INPUT.ORIENTATION_BIN_SIZE
        self.trunc_offset_loss_type = H.TRUNCATION_OFFSET_LOSS
        self.separate_trunc_offset = 'trunc_offset_loss' in self.loss_keys
        self.modify_invalid_keypoint_depths = H.MODIFY_INVALID_KEYPOINT_DEPTH
        self.corner_loss_depth = H.CORNER_LOSS_DEPTH
        keys = self.key2channel.keys
        need = ('2d_dim', '3d_offset', 'corner_offset', 'corner_uncertainty', '3d_dim', 'ori_cls', 'ori_offset', 'depth',
                'depth_uncertainty')
        names = ('hm_loss', 'bbox_loss', 'depth_loss', 'offset_loss', 'orien_loss', 'dims_loss', 'corner_loss', 'keypoint_loss',
                 'keypoint_depth_loss', 'weighted_avg_depth_loss')
        if any(k not in keys for k in need) or any(n not in self.loss_keys for n in names) \
                or self.corner_loss_depth not in ('soft_combine', 'hard_combine', 'direct', 'keypoint_mean'):
            raise NotImplementedError("Loss_Computation is built for the head/loss set of runs/monoflex.yaml")
        # Anno_Encoder constants (anno_encoder.py:11-48)
        self.depth_mode, self.depth_range = H.DEPTH_MODE, H.DEPTH_RANGE
        self.depth_ref = tuple(H.DEPTH_REFERENCE)
        self.dim_mean = torch.as_tensor(H.DIMENSION_MEAN, dtype=torch.float32)
        self.dim_std = torch.as_tensor(H.DIMENSION_STD, dtype=torch.float32)
        self.dim_modes = list(H.DIMENSION_REG)
        self.down_ratio = cfg.MODEL.BACKBONE.DOWN_RATIO
        self.EPS = 1e-3
        self.log_as_float = True          # reference returns python floats in log_loss_dict; False keeps 0-d tensors
        self.fused_object_loss = True     # CUDA maps: the regression terms run as ONE kernel (csrc/object_loss_math.h); False = tensor ops
        self._obj_cfg = None
        self._consts = {}

    def _const(self, name, values, device, dtype=torch.float32):
        """Small constant tables live on the device once (no host-to-device copy per step: those cannot be graph-captured)."""
        key = (name, str(device), dtype)
        if key not in self._consts:
            self._consts[key] = torch.as_tensor(values, dtype=dtype).to(device)
        return self._consts[key]

    # ---------------------------------------------------------------------------------------------
    def prepare_targets(self, targets, device=None):
        """Stack the per-image fields (detector_loss.py:85-114) and the six calibration scalars of every image."""
        def st(name):
            t = torch.stack([torch.as_tensor(x.get_field(name)) for x in targets])
            return t.to(device) if device is not None else t
        names = ("cls_ids", "target_centers", "2d_bboxes", "keypoints", "keypoints_depth_mask", "dimensions", "locations", "rotys",
                 "alphas", "orientations", "pad_size", "reg_mask", "reg_weight", "offset_3D", "trunc_mask")
        d = {n: st(n) for n in names}
        d["bboxes"] = d.pop("2d_bboxes")
        calibs = [x.get_field("calib") for x in targets]
        d["calib"] = calibs
        cal = torch.tensor([[c.f_u, c.f_v, c.c_u, c.c_v, c.b_x, c.b_y] for c in calibs], dtype=torch.float32)
        d["calib_f32"] = cal.to(d["reg_mask"].device)
        if all(x.has_field("ori_img") for x in targets):
            d["ori_imgs"] = torch.stack([torch.as_tensor(x.get_field("ori_img")) for x in targets])
        d["object_rows"] = self.pack_objects(d)
        return st("hm"), d

    def pack_objects(self, d):
        """The stacked target fields as ONE fp32 table, a row per (image, object slot) (csrc/object_loss_math.h R_*): what the
        per-object loss kernel reads.  Pure function of the targets -- built once per batch with them, outside the step."""
        from ... import lib as L
        rm = d["reg_mask"]
        B, M = rm.shape[0], rm.shape[1]
        N, dev = B * M, rm.device
        if d["keypoints"].shape[-2] != 10 or d["orientations"].shape[-1] != 8:
            return None                                    # the kernel is built for 10 keypoints / 4 orientation bins
        f = lambda t, k: t.reshape(N, k).to(device=dev, dtype=torch.float32)
        bidx = torch.arange(B, device=dev).view(B, 1).expand(B, M).reshape(N)
        cal = d["calib_f32"].to(dev)
        # the reference indexes the calibration list by the RANK of the image among those that own an object (anno_encoder.py:198-199)
        present = rm.reshape(B, -1).bool().any(dim=1)
        rank = (torch.cumsum(present.long(), 0) - 1).clamp(min=0)
        cols = [f(rm, 1), f(d["cls_ids"], 1), f(d["target_centers"], 2), f(d["bboxes"], 4), f(d["keypoints"], 30),
                f(d["keypoints_depth_mask"], 3), f(d["dimensions"], 3), f(d["locations"], 3)[:, 2:3], f(d["rotys"], 1),
                f(d["orientations"], 8), f(d["offset_3D"], 2), f(d["trunc_mask"], 1), bidx.float().view(N, 1), cal[bidx],
                d["pad_size"].to(device=dev, dtype=torch.float32).reshape(B, 2)[bidx], cal[:, 0][rank][bidx].view(N, 1)]
        rows = torch.cat(cols, dim=1)
        return torch.nn.functional.pad(rows, (0, L.OBJ_ROW - rows.shape[1])).contiguous()

    def object_loss_cfg(self):
        """mfx_object_loss_cfg of this evaluator (include/monoflex_hip.h); None when a setting is outside what the kernel covers."""
        from ... import lib as L
        if getattr(self, "_obj_cfg", None) is not None:
            return self._obj_cfg
        W, k = self.loss_weights, self.key2channel
        if self.orien_bin_size != 4 or self.depth_mode not in ('exp', 'linear', 'inv_sigmoid') or self.depth_range is None \
                or len(self.dim_mean) != 3:
            return None
        c = L.ObjectLossCfg()
        names = ('bbox_loss', 'depth_loss', 'offset_loss', 'trunc_offset_loss', 'orien_loss', 'dims_loss', 'corner_loss', 'keypoint_loss',
                 'keypoint_depth_loss', 'weighted_avg_depth_loss')
        for i, n in enumerate(names):
            c.w[i] = float(W.get(n, 0.0))
        for i, v in enumerate(self.dim_mean.flatten().tolist()):
            c.dim_mean[i] = v
        for i, v in enumerate(self.dim_std.flatten().tolist()):
            c.dim_std[i] = v
        for i, v in enumerate(self.dim_weight.flatten().tolist()):
            c.dim_weight[i] = v
        c.depth_ref[0], c.depth_ref[1] = float(self.depth_ref[0]), float(self.depth_ref[1])
        c.depth_range[0], c.depth_range[1] = float(self.depth_range[0]), float(self.depth_range[1])
        lo, hi = self.uncertainty_range if self.uncertainty_range is not None else (-float('inf'), float('inf'))
        c.unc_lo, c.unc_hi, c.down_ratio, c.eps = float(lo), float(hi), float(self.down_ratio), float(self.EPS)
        c.depth_mode = ('exp', 'linear', 'inv_sigmoid').index(self.depth_mode)
        c.has_depth_range, c.dim_exp, c.dim_use_std = 1, int(self.dim_modes[0] == 'exp'), int(bool(self.dim_modes[2]))
        c.iou_type = ('giou', 'iou', 'linear_iou').index(self.iou_type)
        c.corner_depth_mode = ('direct', 'keypoint_mean', 'soft_combine', 'hard_combine').index(self.corner_loss_depth)
        c.separate_trunc, c.trunc_log = int(self.separate_trunc_offset), int(self.trunc_offset_loss_type != 'L1')
        c.modify_invalid = int(bool(self.modify_invalid_keypoint_depths))
        for i, key in enumerate(('2d_dim', '3d_offset', 'corner_offset', 'corner_uncertainty', '3d_dim', 'ori_cls', 'ori_offset', 'depth',
                                 'depth_uncertainty')):
            c.ch[i] = k(key).start
        self._obj_cfg = c
        return c

    # ---- Anno_Encoder pieces, batched over all B*MAX_OBJECTS rows --------------------------------------
    def _decode_depth(self, off):                                                 # anno_encoder.py:124-140
        if self.depth_mode == 'exp':
            d = off.exp()
        elif self.depth_mode == 'linear':
            d = off * self.depth_ref[1] + self.depth_ref[0]
        elif self.depth_mode == 'inv_sigmoid':
            d = 1 / torch.sigmoid(off) - 1
        else:
            raise ValueError(self.depth_mode)
        if self.depth_range is not None:
            d = torch.clamp(d, min=self.depth_range[0], max=self.depth_range[1])
        return d

    def _decode_dimension(self, cls_id, off):                                     # anno_encoder.py:217-239
        mean = self._const("dim_mean", self.dim_mean, off.device)[cls_id]
        if self.dim_modes[0] == 'exp':
            off = off.exp()
        if self.dim_modes[2]:
            return off * self._const("dim_std", self.dim_std, off.device)[cls_id] + mean
        return off * mean

    def _decode_location(self, points, offsets, depths, cal, pad):                # anno_encoder.py:142-156 + kitti_utils.py:350-369
        uv = (points + offsets) * self.down_ratio - pad
        x = (uv[:, 0] - cal[:, 2]) * depths / cal[:, 0] + cal[:, 4]
        y = (uv[:, 1] - cal[:, 3]) * depths / cal[:, 1] + cal[:, 5]
        return torch.stack((x, y, depths), dim=1)

    def _keypoint_depths(self, kpts, dims, f_u):                                  # anno_encoder.py:185-215
        h3d = dims[:, 1]
        center_h = kpts[:, -2, 1] - kpts[:, -1, 1]
        c02 = kpts[:, 0:3:2, 1] - kpts[:, 4:7:2, 1]                               # corners (0,2) - (4,6); slices, not index lists
        c13 = kpts[:, 1:4:2, 1] - kpts[:, 5:8:2, 1]
        dc = f_u * h3d / (F.relu(center_h) * self.down_ratio + self.EPS)
        d02 = (f_u.unsqueeze(-1) * h3d.unsqueeze(-1) / (F.relu(c02) * self.down_ratio + self.EPS)).mean(dim=1)
        d13 = (f_u.unsqueeze(-1) * h3d.unsqueeze(-1) / (F.relu(c13) * self.down_ratio + self.EPS)).mean(dim=1)
        return torch.stack([torch.clamp(t, min=self.depth_range[0], max=self.depth_range[1]) for t in (dc, d02, d13)], dim=1)

    def _decode_roty(self, vec, locs):                                            # anno_encoder.py:241-295 (multi-bin)
        nb = self.orien_bin_size
        conf = torch.softmax(vec[:, :nb * 2].reshape(-1, nb, 2), dim=2)[..., 1]
        best = conf.argmax(dim=1, keepdim=True)
        off = vec[:, nb * 2:].reshape(-1, nb, 2)
        centers = self._const("alpha_centers", [0, PI / 2, PI, -PI / 2], vec.device, vec.dtype)[:nb]
        alpha_all = torch.atan2(off[..., 0], off[..., 1]) + centers
        alphas = alpha_all.gather(1, best).squeeze(1)
        rotys = alphas + torch.atan2(locs[:, 0], locs[:, 2])
        rotys = torch.where(rotys > PI, rotys - 2 * PI, rotys)
        return torch.where(rotys < -PI, rotys + 2 * PI, rotys)

    def encode_box3d(self, rotys, dims, locs):                                    # anno_encoder.py:88-122
        c, s = rotys.cos(), rotys.sin()
        l, h, w = dims[:, 0:1] * 0.5, dims[:, 1:2] * 0.5, dims[:, 2:3] * 0.5
        sx = self._const("corner_sx", [-1, -1, 1, 1, -1, -1, 1, 1], dims.device, dims.dtype)        # -l/2 .. l/2 per corner
        sy = self._const("corner_sy", [1, 1, 1, 1, -1, -1, -1, -1], dims.device, dims.dtype)
        sz = self._const("corner_sz", [-1, 1, 1, -1, -1, 1, 1, -1], dims.device, dims.dtype)
        x, y, z = l * sx, h * sy, w * sz
        X = c[:, None] * x + s[:, None] * z + locs[:, 0:1]
        Y = y + locs[:, 1:2]
        Z = -s[:, None] * x + c[:, None] * z + locs[:, 2:3]
        return torch.stack((X, Y, Z), dim=2)                                       # (N, 8, 3)

    # ---------------------------------------------------------------------------------------------
    def prepare_predictions(self, tv, predictions):
        """Targets and decoded predictions for all N = B*MAX_OBJECTS rows, plus the 0/1 selection weights."""
        reg = predictions['reg']                                                  # (B,C,H,W), any strides
        B, C, H, W = reg.shape
        N = B * self.max_objs
        dev = reg.device
        valid = tv["reg_mask"].reshape(N).bool()
        v = valid.float()
        bidx = torch.arange(B, device=dev).view(B, 1).expand(B, self.max_objs).reshape(N)
        cal, pad = tv["calib_f32"][bidx], tv["pad_size"].to(reg.dtype)[bidx]
        pts_i = tv["target_centers"].reshape(N, 2).long()
        pts = pts_i.to(reg.dtype)
        box = tv["bboxes"].reshape(N, 4).to(reg.dtype)
        t_h, t_w = box[:, 3] - box[:, 1], box[:, 2] - box[:, 0]
        m2d = valid & (t_h > 0) & (t_w > 0)
        t_reg2d = torch.cat((pts - box[:, :2], box[:, 2:] - pts), dim=1)
        t_cls = tv["cls_ids"].reshape(N).long().clamp(min=0)
        t_depth = tv["locations"][..., -1].reshape(N).to(reg.dtype)
        t_roty = tv["rotys"].reshape(N).to(reg.dtype)
        t_off = tv["offset_3D"].reshape(N, 2).to(reg.dtype)
        t_dims = tv["dimensions"].reshape(N, 3).to(reg.dtype)
        t_ori = tv["orientations"].reshape(N, -1).to(reg.dtype)
        t_loc = self._decode_location(pts, t_off, t_depth, cal, pad)
        targets = {'reg_2D': t_reg2d, 'offset_3D': t_off, 'depth_3D': t_depth, 'orien_3D': t_ori, 'dims_3D': t_dims,
                   'corners_3D': self.encode_box3d(t_roty, t_dims, t_loc), 'width_2D': t_w, 'height_2D': t_h, 'rotys_3D': t_roty,
                   'cat_3D': torch.cat((t_loc, t_dims, t_roty[:, None]), dim=1),
                   'trunc_mask_3D': tv["trunc_mask"].reshape(N).bool() & valid}
        # POI gather straight from the NHWC map (layers/utils.py:120-145)
        flat = reg.permute(0, 2, 3, 1).reshape(B, H * W, C)
        idx = (pts_i[:, 1] * W + pts_i[:, 0]).view(B, self.max_objs, 1).expand(B, self.max_objs, C)
        poi = flat.gather(1, idx).reshape(N, C)
        k = self.key2channel
        preds = {'reg_2D': F.relu(poi[:, k('2d_dim')]), 'offset_3D': poi[:, k('3d_offset')],
                 'orien_3D': torch.cat((poi[:, k('ori_cls')], poi[:, k('ori_offset')]), dim=1)}
        p_dims = self._decode_dimension(t_cls, poi[:, k('3d_dim')])
        preds['dims_3D'] = p_dims
        p_depth = self._decode_depth(poi[:, k('depth')].squeeze(-1))
        preds['depth_3D'] = p_depth
        lo, hi = (self.uncertainty_range if self.uncertainty_range is not None else (-float('inf'), float('inf')))
        preds['depth_uncertainty'] = torch.clamp(poi[:, k('depth_uncertainty')].squeeze(-1), min=lo, max=hi)
        kp = tv["keypoints"].reshape(N, -1, 3).to(reg.dtype)
        targets['keypoints'], targets['keypoints_mask'] = kp[..., :2], kp[..., 2] * v[:, None]
        kdm = tv["keypoints_depth_mask"].reshape(N, 3).bool()
        targets['keypoints_depth_mask'] = kdm
        p_kp = poi[:, k('corner_offset')].reshape(N, -1, 2)
        # the reference indexes the calibration list by the RANK of the image among those that own an object
        # (anno_encoder.py:198-199, `calibs[idx]` not `calibs[gt_idx]`) -- kept
        present = tv["reg_mask"].reshape(B, -1).bool().any(dim=1)
        rank = (torch.cumsum(present.long(), 0) - 1).clamp(min=0)
        f_u = tv["calib_f32"][:, 0][rank][bidx]
        preds['keypoints'] = p_kp
        preds['keypoints_depths'] = self._keypoint_depths(p_kp, p_dims, f_u)
        preds['corner_offset_uncertainty'] = torch.clamp(poi[:, k('corner_uncertainty')], min=lo, max=hi)
        if self.corner_loss_depth == 'direct':
            corner_depth = p_depth
        elif self.corner_loss_depth == 'keypoint_mean':
            corner_depth = preds['keypoints_depths'].mean(dim=1)
        else:
            unc = torch.cat((preds['depth_uncertainty'].unsqueeze(-1), preds['corner_offset_uncertainty']), dim=1).exp()
            depths = torch.cat((p_depth.unsqueeze(-1), preds['keypoints_depths']), dim=1)
            if self.corner_loss_depth == 'soft_combine':
                wts = 1 / unc
                wts = wts / wts.sum(dim=1, keepdim=True)
                corner_depth = torch.sum(depths * wts, dim=1)
                preds['weighted_depths'] = corner_depth
            else:
                corner_depth = depths.gather(1, unc.argmin(dim=1, keepdim=True)).squeeze(1)
        p_loc = self._decode_location(pts, preds['offset_3D'], corner_depth, cal, pad)
        p_roty = self._decode_roty(preds['orien_3D'], p_loc)
        preds.update({'corners_3D': self.encode_box3d(p_roty, p_dims, p_loc), 'rotys_3D': p_roty,
                      'cat_3D': torch.cat((p_loc, p_dims, p_roty[:, None]), dim=1)})
        sel = {'valid': valid, 'reg_2D': m2d}
        return targets, preds, sel, {'object_weights': tv["reg_weight"].reshape(N)}

    # ---------------------------------------------------------------------------------------------
    def _focal(self, pred, target):                                                # focal_loss.py:29-55
        pos = target.eq(1).float()
        neg = (target.lt(1) & target.ge(0)).float()
        neg_w = torch.pow(1 - target, self.focal_beta)
        pos_loss = torch.log(pred) * torch.pow(1 - pred, self.focal_alpha) * pos
        neg_loss = torch.log(1 - pred) * torch.pow(pred, self.focal_alpha) * neg_w * neg
        return -neg_loss.sum() - pos_loss.sum(), pos.sum()

    def _iou(self, pred, target):                                                  # iou_loss.py:12-49
        pl, pt, pr, pb = pred.unbind(1)
        tl, tt, tr, tb = target.unbind(1)
        t_area, p_area = (tl + tr) * (tt + tb), (pl + pr) * (pt + pb)
        w_i = torch.min(pl, tl) + torch.min(pr, tr)
        h_i = torch.min(pb, tb) + torch.min(pt, tt)
        g_w = torch.max(pl, tl) + torch.max(pr, tr)
        g_h = torch.max(pb, tb) + torch.max(pt, tt)
        ac = g_w * g_h + 1e-7
        inter = w_i * h_i
        union = t_area + p_area - inter
        ious = (inter + 1.0) / (union + 1.0)
        if self.iou_type == 'iou':
            return -torch.log(ious), ious
        if self.iou_type == 'linear_iou':
            return 1 - ious, ious
        return 1 - (ious - (ac - union) / ac), ious

    def _multibin(self, vec, gt, v):                                               # detector_loss.py:495-517
        nb = self.orien_bin_size
        cls_losses, reg_losses, reg_cnt = 0, 0, 0
        for i in range(nb):
            ce = F.cross_entropy(vec[:, 2 * i:2 * i + 2], gt[:, i].long(), reduction='none')
            cls_losses = cls_losses + _wmean(ce, v)
            wi = (gt[:, i] == 1).float() * v
            s = nb * 2 + i * 2
            off = F.normalize(vec[:, s:s + 2])
            l1 = (off[:, 0] - torch.sin(gt[:, nb + i])).abs() + (off[:, 1] - torch.cos(gt[:, nb + i])).abs()
            reg_losses = reg_losses + (l1 * wi).sum()
            reg_cnt = reg_cnt + wi.sum()
        return cls_losses / nb + reg_losses / torch.clamp(reg_cnt, min=1)

    def _heat_term(self, predictions, heat, dev):
        if predictions.get('cls_logits_nhwc') is not None and predictions['cls_logits_nhwc'].is_cuda:
            # fused heat-map term: the predictor hands over its fp32 NHWC logits; sigmoid + clamp + focal + gradient in one pass
            from ... import autograd as AG
            hm_loss, num_pos = AG.FocalLossFn.apply(predictions['cls_logits_nhwc'], heat.to(dev), float(self.focal_alpha), float(self.focal_beta))
        else:
            hm_loss, num_pos = self._focal(predictions['cls'], heat.to(dev))
        return self.loss_weights['hm_loss'] * hm_loss / torch.clamp(num_pos, 1)

    def _fused(self, predictions, heat, tv, dev):
        """Device form of __call__: heat-map term + ONE launch for the nine regression terms and their gradient rows."""
        from ... import autograd as AG
        from ... import lib as L
        cfg, rows = self.object_loss_cfg(), tv.get("object_rows")
        if rows is None:
            rows = self.pack_objects(tv)
        if cfg is None or rows is None:
            raise NotImplementedError("fused object loss: 4 orientation bins, 10 keypoints, a depth range")
        if predictions.get('reg') is None:
            terms, logged = AG.ObjectLossFn.apply(predictions['reg_rows'].float(), rows.to(dev), cfg, 0)      # (N,50): row n = object row n
        else:
            reg = predictions['reg'].permute(0, 2, 3, 1)                          # the predictor's NHWC map: a view
            terms, logged = AG.ObjectLossFn.apply(reg.float(), rows.to(dev), cfg, 0)
        t = terms.unbind(0)
        loss_dict = _LossDict({'hm_loss': self._heat_term(predictions, heat, dev), 'bbox_loss': t[0], 'dims_loss': t[5], 'orien_loss': t[4],
                               'offset_loss': t[2]})
        if self.separate_trunc_offset and terms.shape[0] == 10:
            loss_dict.total = terms.sum() + loss_dict['hm_loss']           # every entry of `terms` is one of the dict's terms
        if self.separate_trunc_offset:
            loss_dict['trunc_offset_loss'] = t[3]
        loss_dict.update({'corner_loss': t[6], 'depth_loss': t[1], 'keypoint_loss': t[7], 'keypoint_depth_loss': t[8],
                          'weighted_avg_depth_loss': t[9]})
        with torch.no_grad():
            lg = logged.unbind(0)
            logs = {'2D_IoU': lg[0], '3D_IoU': lg[0] * 0, 'depth_loss': lg[1], 'keypoint_depth_loss': lg[2]}
            for key, val in loss_dict.items():
                if key not in logs:
                    logs[key] = val.detach()
            logs.update(dict(zip(('depth_MAE', 'center_MAE', '02_MAE', '13_MAE', 'lower_MAE', 'hard_MAE', 'soft_MAE', 'mean_MAE'), lg[3:11])))
            if self.log_as_float:
                vals = torch.stack([x.float() for x in logs.values()]).tolist()     # the step's single host sync
                logs = dict(zip(logs.keys(), vals))
        return loss_dict, logs

    def __call__(self, predictions, targets):
        dev = (predictions['reg'] if predictions.get('reg') is not None else predictions['reg_rows']).device
        prepared = targets if isinstance(targets, tuple) else getattr(targets, "loss", None)
        heat, tv = prepared if prepared is not None else self.prepare_targets(targets, dev)
        W = self.loss_weights
        if predictions.get('reg') is None:                         # the predictor handed over the gathered regression table
            return self._fused(predictions, heat, tv, dev)
        if predictions['reg'].is_cuda and self.fused_object_loss and self.object_loss_cfg() is not None \
                and (tv.get("object_rows") is not None or tv["keypoints"].shape[-2] == 10):
            return self._fused(predictions, heat, tv, dev)          # else: the tensor-op form below (same device, ~900 launches)
        T, P, sel, _ = self.prepare_predictions(tv, predictions)
        valid, v = sel['valid'], sel['valid'].float()
        v2 = sel['reg_2D'].float()
        one = torch.ones((), device=dev)

        hm_loss = self._heat_term(predictions, heat, dev)

        # 2D box (GIoU); unselected rows get a unit box on both sides
        m2 = sel['reg_2D'][:, None]
        l2d, iou2d = self._iou(torch.where(m2, P['reg_2D'], one), torch.where(m2, T['reg_2D'], one))
        reg_2D_loss = W['bbox_loss'] * _wmean(l2d, v2)
        t_depth_safe = torch.where(valid, T['depth_3D'], one)
        depth_MAE = (P['depth_3D'] - T['depth_3D']).abs() / t_depth_safe

        # direct depth with aleatoric uncertainty
        d_l1 = W['depth_loss'] * (P['depth_3D'] - T['depth_3D']).abs()
        real_depth_loss = _wmean(d_l1.detach(), v)
        depth_loss = _wmean(d_l1 * torch.exp(-P['depth_uncertainty']) + P['depth_uncertainty'] * W['depth_loss'], v)

        # projected-centre offset: L1 inside, log(1+L1) for truncated objects
        off_l1 = (P['offset_3D'] - T['offset_3D']).abs().sum(dim=1)
        trunc = T['trunc_mask_3D'].float()
        if self.separate_trunc_offset:
            tl = off_l1 if self.trunc_offset_loss_type == 'L1' else torch.log(1 + off_l1)
            trunc_offset_loss = W['trunc_offset_loss'] * (tl * trunc).sum() / torch.clamp(trunc.sum(), min=1)
            offset_loss = W['offset_loss'] * _wmean(off_l1, v * (1 - trunc))
        else:
            offset_loss = W['offset_loss'] * _wmean(off_l1, v)

        orien_loss = W['orien_loss'] * self._multibin(P['orien_3D'], T['orien_3D'], v)
        dims_l1 = (P['dims_3D'] - T['dims_3D']).abs() * self._const('dim_weight', self.dim_weight, dev)
        dims_loss = W['dims_loss'] * _wmean(dims_l1.sum(dim=1), v)
        # (N,8) per-corner L1 sums averaged over all N*8 entries (detector_loss.py:338-339: `.sum(dim=2).mean()`)
        corner_loss = W['corner_loss'] * _wmean((P['corners_3D'] - T['corners_3D']).abs().sum(dim=2).mean(dim=1), v)
        kmask = T['keypoints_mask']
        kp_l1 = (P['keypoints'] - T['keypoints']).abs().sum(dim=2)
        keypoint_loss = (W['keypoint_loss'] * kp_l1 * kmask).sum() / torch.clamp(kmask.sum(), min=1)

        # depths solved from the three keypoint groups
        kd = P['keypoints_depths']
        km = (T['keypoints_depth_mask'] & valid[:, None]).float()
        kinv = ((~T['keypoints_depth_mask']) & valid[:, None]).float()
        t_kd = T['depth_3D'].unsqueeze(-1)
        unc = P['corner_offset_uncertainty']
        kd_valid = W['keypoint_depth_loss'] * (kd - t_kd).abs()
        kd_invalid = W['keypoint_depth_loss'] * (kd.detach() - t_kd).abs()
        log_valid_kd = _wmean(kd_valid.detach(), km)
        kd_valid = kd_valid * torch.exp(-unc) + W['keypoint_depth_loss'] * unc
        kd_invalid = kd_invalid * torch.exp(-unc)
        keypoint_depth_loss = _wmean(kd_valid, km)
        if self.modify_invalid_keypoint_depths:
            keypoint_depth_loss = keypoint_depth_loss + _wmean(kd_invalid, kinv)

        comb_depth = torch.cat((P['depth_3D'].unsqueeze(1), kd), dim=1)
        comb_unc = torch.cat((P['depth_uncertainty'].unsqueeze(1), unc), dim=1).exp()
        cw = 1 / comb_unc
        cw = cw / cw.sum(dim=1, keepdim=True)
        soft_depths = torch.sum(comb_depth * cw, dim=1)
        soft_depth_loss = W['weighted_avg_depth_loss'] * _wmean((soft_depths - T['depth_3D']).abs(), v)

        loss_dict = {'hm_loss': hm_loss, 'bbox_loss': reg_2D_loss, 'dims_loss': dims_loss, 'orien_loss': orien_loss,
                     'offset_loss': offset_loss}
        if self.separate_trunc_offset:
            loss_dict['trunc_offset_loss'] = trunc_offset_loss
        loss_dict.update({'corner_loss': corner_loss, 'depth_loss': depth_loss, 'keypoint_loss': keypoint_loss,
                          'keypoint_depth_loss': keypoint_depth_loss, 'weighted_avg_depth_loss': soft_depth_loss})

        with torch.no_grad():                                                      # log-only quantities (detector_loss.py:396-482)
            kMAE = (kd - t_kd).abs() / t_depth_safe.unsqueeze(-1)
            cMAE = torch.cat((depth_MAE.unsqueeze(1), kMAE), dim=1)
            hard = cMAE.gather(1, comb_unc.argmin(dim=1, keepdim=True)).squeeze(1)
            logs = {'2D_IoU': _wmean(iou2d, v2), '3D_IoU': torch.zeros((), device=dev),
                    'depth_loss': real_depth_loss, 'keypoint_depth_loss': log_valid_kd}
            for key, val in loss_dict.items():
                if key not in logs:
                    logs[key] = val.detach()
            logs.update({'depth_MAE': _wmean(depth_MAE, v), 'center_MAE': _wmean(kMAE[:, 0], v),
                         '02_MAE': _wmean(kMAE[:, 1], v), '13_MAE': _wmean(kMAE[:, 2], v),
                         'lower_MAE': _wmean(cMAE.min(dim=1)[0], v), 'hard_MAE': _wmean(hard, v),
                         'soft_MAE': _wmean((soft_depths - T['depth_3D']).abs() / t_depth_safe, v),
                         'mean_MAE': _wmean((comb_depth.mean(dim=1) - T['depth_3D']).abs() / t_depth_safe, v)})
            if self.log_as_float:
                vals = torch.stack([x.float() for x in logs.values()]).tolist()     # the step's single host sync
                logs = dict(zip(logs.keys(), vals))
        return loss_dict, logs
