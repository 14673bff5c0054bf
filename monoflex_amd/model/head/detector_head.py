"""Detect_Head (reference model/head/detector_head.py:9-29): predictor + loss evaluator (train) / post-processor (eval)."""
from torch import nn

from .detector_infer import make_post_processor
from .detector_loss import make_loss_evaluator
from .detector_predictor import make_predictor


class Detect_Head(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        self.predictor = make_predictor(cfg, in_channels)
        self.loss_evaluator = make_loss_evaluator(cfg)
        self.post_processor = make_post_processor(cfg)

    def forward(self, features, targets=None, test=False):
        x = self.predictor(features, targets)
        if self.training:
            return self.loss_evaluator(x, targets)
        return self.post_processor(x, targets, test=test, features=features)


def bulid_head(cfg, in_channels):
    return Detect_Head(cfg, in_channels)
