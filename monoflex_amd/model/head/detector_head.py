"""Detection head container (reference model/head/detector_head.py:9-29).

Holds the three pieces under the attribute names the reference's scripts and checkpoints use -- `predictor` (nine fused
branches + edge fusion), `loss_evaluator` (training), `post_processor` (decode) -- and routes a call by mode:
training -> `(loss_dict, log_loss_dict)`, evaluation -> `(result, eval_utils, visualize_preds)`."""
from torch import nn

from . import detector_infer, detector_loss, detector_predictor


class Detect_Head(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        parts = (("predictor", detector_predictor.make_predictor(cfg, in_channels)),
                 ("loss_evaluator", detector_loss.make_loss_evaluator(cfg)),       # a plain callable (no parameters), like the reference's
                 ("post_processor", detector_infer.make_post_processor(cfg)))
        for name, part in parts:                        # attribute order = the reference's state_dict order
            setattr(self, name, part)

    def losses(self, maps, targets):
        """Head maps of a training batch -> the 11 weighted loss terms and their logged values."""
        return self.loss_evaluator(maps, targets)

    def detections(self, maps, targets, test=False, features=None):
        """Head maps of an evaluation batch -> decoded (N,14) rows per image plus the decode by-products."""
        return self.post_processor(maps, targets, test=test, features=features)

    sparse_regression = True       # training on the device: regression branches evaluated at the object centres only

    def forward(self, features, targets=None, test=False):
        if self.training and self.sparse_regression and features.is_cuda and self.loss_evaluator.fused_object_loss \
                and self.loss_evaluator.object_loss_cfg() is not None:
            prepared = targets if isinstance(targets, tuple) else getattr(targets, "loss", None)
            if prepared is None:
                prepared = self.loss_evaluator.prepare_targets(targets, features.device)
            rows = prepared[1].get("object_rows")
            if rows is not None:
                maps = self.predictor(features, targets, object_rows=rows.to(features.device))
                return self.loss_evaluator(maps, prepared)
        maps = self.predictor(features, targets)
        return self.losses(maps, targets) if self.training else self.detections(maps, targets, test, features)


def bulid_head(cfg, in_channels):                       # (sic) the reference's spelling, imported by model/detector.py
    return Detect_Head(cfg, in_channels)
