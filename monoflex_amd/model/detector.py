"""KeypointDetector (reference model/detector.py:11-37): backbone + heads, same constructor,
attributes (`backbone`, `backbone.out_channels`, `heads.{predictor,post_processor}`), state_dict keys
and eval-mode return triple.  `detect_device` is the static-shape device pipeline used by bench.py
and hipGraph capture."""
import torch
from torch import nn

from ..structures.image_list import to_image_list
from .backbone import build_backbone
from .backbone.dla_dcn import invalidate_packs
from .head.detector_head import bulid_head
from .head.detector_predictor import make_edge_rowmap, stack_edge_fields

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}


class KeypointDetector(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        self.heads = bulid_head(cfg, self.backbone.out_channels)
        self.test = cfg.DATASETS.TEST_SPLIT == 'test'
        self.set_compute_dtype(cfg.MODEL.get("COMPUTE_DTYPE", "fp32"))

    def set_compute_dtype(self, name):
        """'fp32' = parity mode (f32 MFMA, <=1e-3 on logits), 'bf16' = perf mode (bf16 MFMA, fp32 accumulate)."""
        self.compute_dtype = _DTYPES[name] if isinstance(name, str) else name
        self.backbone.compute_dtype = self.compute_dtype
        return self

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        invalidate_packs(self)
        return r

    def train(self, mode=True):
        invalidate_packs(self)
        return super().train(mode)

    # ---- device pipeline: no host sync, static shapes ------------------------------------------------
    def detect_device(self, images, edge_indices, edge_lens, pad, calib, size, edge_rowmap=None):
        feat = self.backbone.forward_nhwc(images)
        hm = self.heads.predictor.forward_nhwc(feat, edge_indices, edge_lens, edge_rowmap)
        det, topk, valid = self.heads.post_processor.decode_device(hm, pad, calib, size, self.heads.predictor.last_cls_planar)
        return det, topk, valid, hm

    def forward_train_maps(self, images, edge_indices, edge_lens):
        """Training-mode network: (B,3,H,W) images -> (class logits (B,h,w,ncls), regression (B,h,w,50)), NHWC,
        differentiable (HIP forward + backward kernels; BN on batch statistics)."""
        feat = self.backbone.forward_nhwc(images)
        return self.heads.predictor.forward_train(feat, edge_indices, edge_lens)

    def device_targets(self, targets, device):
        """Device-side view of the per-image targets: (edge_indices, edge_lens, pad, calib, size, edge_rowmap)."""
        ei, el = stack_edge_fields(targets, device)
        pad, calib, size = self.heads.post_processor.prepare_targets(targets, device)
        pr = self.heads.predictor
        return ei, el, pad, calib, size, make_edge_rowmap(ei, pr.output_height, pr.output_width)

    def forward(self, images, targets=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        images = to_image_list(images)
        if not images.tensors.is_cuda:
            raise RuntimeError("KeypointDetector (HIP build) needs CUDA/HIP tensors: there is no CPU fallback")
        if self.training:                                           # detector.py:30-33 -> (loss_dict, log_loss_dict)
            return self.heads(self.backbone(images.tensors), targets)
        with torch.no_grad():
            features = self.backbone(images.tensors)
            return self.heads(features, targets, test=self.test)
