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
        ok = (keys == want and key2channel.channels == [4, 2, 20, 3, 3, 8, 8, 1, 1] and self.output_depth == 'soft'
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
                                      "(soft depth fusion, inv_sigmoid depth in [0.1, 100], exp dims with the KITTI "
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
        scores, index = ops.decode_topk(hm, 0, self.num_classes, self.max_detection, planar=cls_planar)
        return ops.decode_boxes(hm, REG_OFF, scores, index, calib, pad, size, float(self.det_threshold))

    def forward(self, predictions, targets, features=None, test=False, refine_module=None):
        hm = predictions['hm_nhwc']
        pad, calib, size = self.prepare_targets(targets, hm.device)
        det, topk, valid = self.decode_device(hm, pad, calib, size, predictions.get('cls_planar'))
        keep = valid.bool()
        results = [det[b][keep[b]] for b in range(det.shape[0])]          # host sync, as detector_infer.py:106
        vis_scores = [topk[b][keep[b], 0] for b in range(det.shape[0])]
        eval_utils = {'dis_ious': None, 'depth_errors': None, 'vis_scores': vis_scores[0] if len(results) == 1 else vis_scores,
                      'topk': topk, 'valid': valid, 'det_all': det}
        visualize_preds = {'heat_map': predictions['cls']}
        result = results[0] if len(results) == 1 else results
        return result, eval_utils, visualize_preds


def _calib_f32(c):
    return np.array([c.f_u, c.f_v, c.c_u, c.c_v, c.b_x, c.b_y], dtype=np.float32)
