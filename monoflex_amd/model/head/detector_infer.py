"""PostProcessor (reference model/head/detector_infer.py:20-237) on the gfx950 decode kernels.

The reference decode is batch-1 only; here every image of the batch is decoded by
its own workgroup with the batch-1 semantics (its own calib / pad_size; the 2D-box clamp uses image
0's padded size exactly like `out_size = out_size[0]`, anno_encoder.py:79).  The device always
produces K=50 rows + a validity mask (static shapes, hipGraph-capturable); the `score >= threshold`
compaction of detector_infer.py:102-119 happens when the result is handed to the host.
"""
import numpy as np
import torch
from torch import nn

from ... import lib as L
from ... import ops
from ..layers.utils import Converter_key2channel
from .detector_predictor import HM_LD, REG_OFF


def make_post_processor(cfg):
    key2channel = Converter_key2channel(keys=cfg.MODEL.HEAD.REGRESSION_HEADS, channels=cfg.MODEL.HEAD.REGRESSION_CHANNELS)
    return PostProcessor(cfg=cfg, key2channel=key2channel)


class PostProcessor(nn.Module):
    def __init__(self, cfg, key2channel, anno_encoder=None):
        super().__init__()
        self.key2channel = key2channel
        self.det_threshold = cfg.TEST.DETECTIONS_THRESHOLD
        self.max_detection = cfg.TEST.DETECTIONS_PER_IMG
        self.output_width = cfg.INPUT.WIDTH_TRAIN // cfg.MODEL.BACKBONE.DOWN_RATIO
        self.output_height = cfg.INPUT.HEIGHT_TRAIN // cfg.MODEL.BACKBONE.DOWN_RATIO
        self.output_depth = cfg.MODEL.HEAD.OUTPUT_DEPTH
        self.uncertainty_as_conf = cfg.TEST.UNCERTAINTY_AS_CONFIDENCE
        self.num_classes = len(cfg.DATASETS.DETECT_CLASSES)
        keys = key2channel.keys
        want = ['2d_dim', '3d_offset', 'corner_offset', 'corner_uncertainty', '3d_dim', 'ori_cls', 'ori_offset',
                'depth', 'depth_uncertainty']
        ok = (keys == want and key2channel.channels == [4, 2, 20, 3, 3, 8, 8, 1, 1] and (self.output_depth in L.DEPTH_MODES or self.output_depth == 'oracle')
              and self.uncertainty_as_conf and cfg.MODEL.HEAD.DEPTH_MODE == 'inv_sigmoid'
              and list(cfg.MODEL.HEAD.DIMENSION_REG) == ['exp', True, False] and cfg.INPUT.ORIENTATION == 'multi-bin'
              and cfg.MODEL.BACKBONE.DOWN_RATIO == 4 and self.num_classes == 3)
        # constants compiled into mfx_decode_boxes (decode.hip DecodeConst; reference config/defaults.py:175,206-208): a config
        # that changes them would train with its own values and decode with the built-in ones, so it is refused
        built_in_mean = ((3.8840, 1.5261, 1.6286), (0.8423, 1.7607, 0.6602), (1.7635, 1.7372, 0.5968))
        mean = [tuple(float(v) for v in row) for row in cfg.MODEL.HEAD.DIMENSION_MEAN]
        ok = ok and len(mean) == 3 and all(abs(a - b) < 1e-6 for ra, rb in zip(mean, built_in_mean) for a, b in zip(ra, rb))
        ok = ok and [float(v) for v in cfg.MODEL.HEAD.DEPTH_RANGE] == [0.1, 100.0]
        if not ok:
            raise NotImplementedError("the HIP decode kernel implements the runs/monoflex.yaml decode "
                                      "(output_depth soft / hard / mean / direct / keypoints_* / oracle, inv_sigmoid depth in [0.1, 100], exp dims with the KITTI "
                                      "DIMENSION_MEAN, multi-bin orientation)")

    @staticmethod
    def prepare_targets(targets, device):
        """pad_size (B,2) int32, calib (B,6) fp32 [f_u,f_v,c_u,c_v,b_x,b_y], size (2,) int32 of image 0."""
        pad = torch.stack([torch.as_tensor(t.get_field("pad_size")) for t in targets]).to(device=device, dtype=torch.int32)
        calib = torch.from_numpy(np.stack([t.get_field("calib").as_f32() if hasattr(t.get_field("calib"), "as_f32")
                                           else _calib_f32(t.get_field("calib")) for t in targets])).to(device)
        size = torch.tensor(list(targets[0].size), dtype=torch.int32, device=device)
        return pad.contiguous(), calib.contiguous(), size

    def decode_device(self, hm, pad, calib, size, cls_planar=None):
        """hm fp32 (B,H,W,64) -> det (B,K,14), topk (B,K,5) [score, flat index, cls, y, x], valid (B,K) int32."""
        if self.output_depth == 'oracle':
            raise ValueError("output_depth = 'oracle' reads the ground truth of each image: call the module (forward) with the dataset's targets")
        scores, index = ops.decode_topk(hm, 0, self.num_classes, self.max_detection, planar=cls_planar)
        # (`output_depth` is read at every call: engine/inference.py:166 re-assigns it between the passes of `eval_all_depths`)
        return ops.decode_boxes(hm, REG_OFF, scores, index, calib, pad, size, float(self.det_threshold), depth_mode=self.output_depth)

    def forward(self, predictions, targets, features=None, test=False, refine_module=None):
        hm = predictions['hm_nhwc']
        pad, calib, size = self.prepare_targets(targets, hm.device)
        if self.output_depth == 'oracle':
            det, topk, valid = self.decode_oracle(hm, pad, calib, size, predictions.get('cls_planar'), targets)
        else:
            det, topk, valid = self.decode_device(hm, pad, calib, size, predictions.get('cls_planar'))
        keep = valid.bool()
        results = [det[b][keep[b]] for b in range(det.shape[0])]          # host sync, as detector_infer.py:106
        vis_scores = [topk[b][keep[b], 0] for b in range(det.shape[0])]
        eval_utils = {'dis_ious': None, 'depth_errors': None, 'vis_scores': vis_scores[0] if len(results) == 1 else vis_scores,
                      'topk': topk, 'valid': valid, 'det_all': det}
        visualize_preds = {'heat_map': predictions['cls']}
        result = results[0] if len(results) == 1 else results
        return result, eval_utils, visualize_preds

    # the single-estimate decodes whose rows `decode_oracle` chooses among, in the column order of the reference's
    # pred_combined_depths (detector_infer.py:173: direct, then the three keypoint groups)
    ORACLE_COLUMNS = ('direct', 'keypoints_center', 'keypoints_02', 'keypoints_13')

    def decode_oracle(self, hm, pad, calib, size, cls_planar, targets):
        """`output_depth = 'oracle'` (detector_infer.py:199-202, get_oracle_depths :238-277; the first method engine/inference.py:154 evaluates):
        every detection takes, of its four depth estimates, the one closest to the depth of the ground-truth object it overlaps (nearest box centre
        of its class, 2D IoU >= 0.5), and the mean of the four when it overlaps none.  The depth only enters a row through location, rotation and
        the uncertainty-scaled score, so the row of a detection under 'oracle' IS its row from the decode of the chosen single estimate (or of
        'mean'): five launches of the box kernel on one top-K, and a per-detection choice of row on the host, where the ground truth is.
        (The reference reads targets[0] only -- it evaluates at batch 1; here image b reads targets[b].)"""
        scores, index = ops.decode_topk(hm, 0, self.num_classes, self.max_detection, planar=cls_planar)
        dec = {m: ops.decode_boxes(hm, REG_OFF, scores, index, calib, pad, size, float(self.det_threshold), depth_mode=m)
               for m in ('mean',) + self.ORACLE_COLUMNS}
        det, topk, valid = dec['mean']
        rows = {m: d[0].cpu() for m, d in dec.items()}
        valid_h = valid.cpu().bool()
        out = rows['mean'].clone()
        for b, t in enumerate(targets):
            mask = torch.as_tensor(t.get_field('reg_mask')).bool().cpu()
            gt_cls = torch.as_tensor(t.get_field('cls_ids')).cpu()[mask]
            gt_box = torch.as_tensor(t.get_field('gt_bboxes')).cpu()[mask].float()
            gt_depth = torch.as_tensor(t.get_field('locations')).cpu()[mask][:, -1].float()
            if gt_box.shape[0] == 0:
                continue
            gt_centre = (gt_box[:, :2] + gt_box[:, 2:]) / 2
            for i in torch.nonzero(valid_h[b]).flatten().tolist():
                box = rows['mean'][b, i, 2:6]
                dis = torch.sum((((box[:2] + box[2:]) / 2).reshape(1, 2) - gt_centre) ** 2, dim=1)
                dis[gt_cls != int(rows['mean'][b, i, 0])] = 9999
                near = int(torch.argmin(dis))
                if _box_iou(box.numpy(), gt_box[near].numpy()) < 0.5:            # (a 0 / 0 overlap is not "< 0.5": such a pair counts as met, :268-270)
                    continue
                est = torch.stack([rows[m][b, i, 11] for m in self.ORACLE_COLUMNS])     # row[11] = location z = the depth that decode used
                out[b, i] = rows[self.ORACLE_COLUMNS[int(torch.argmin(torch.abs(est - gt_depth[near])))]][b, i]
        return out.to(det.device), topk, valid


def _box_iou(a, b):
    """engine/visualize_infer.py:23-27, in the float32 scalars the reference computes it in."""
    with np.errstate(invalid='ignore', divide='ignore'):
        inter = max(min(a[2], b[2]) - max(a[0], b[0]), 0) * max(min(a[3], b[3]) - max(a[1], b[1]), 0)
        return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def _calib_f32(c):
    return np.array([c.f_u, c.f_v, c.c_u, c.c_v, c.b_x, c.b_y], dtype=np.float32)
