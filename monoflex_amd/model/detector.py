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

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
           "fp16": torch.float16, "float16": torch.float16,      # fp16: IEEE half activations, same MFMA rate as bf16 (training: under a loss scaler)
           "fp16x2": torch.float32}      # split precision (inference): fp32 activations, fp16 (hi, lo) MFMA operand pairs (ops.F16X2)


class KeypointDetector(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        self.heads = bulid_head(cfg, self.backbone.out_channels)
        self.test = cfg.DATASETS.TEST_SPLIT == 'test'
        self.set_compute_dtype(cfg.MODEL.get("COMPUTE_DTYPE", "fp32"))

    def set_compute_dtype(self, name):
        """'fp32' = parity mode (f32 MFMA, <=1e-3 on logits), 'bf16' = perf mode (bf16 MFMA, fp32 accumulate; inference and training),
        'fp16' = perf mode with IEEE-half activations (same MFMA rate, three more mantissa bits: ~8x closer to the reference in inference;
        training in fp16 runs under engine.trainer.LossScaler, which do_train / GraphedTrainStep create by themselves),
        'fp16x2' = parity-grade perf mode for inference: fp32 activations in memory, every MFMA operand split into an fp16 (hi, lo)
        pair, four fp16 products per element pair accumulated in fp32 -- fp32-grade results at 4x the fp32 matrix rate (training
        in this mode runs the fp32 kernels)."""
        self.compute_dtype = _DTYPES[name] if isinstance(name, str) else name
        self.backbone.compute_dtype = self.compute_dtype
        split = name == "fp16x2"
        for m in self.modules():                                   # ops.compute_tag: which packed weights / kernels a module's eval path uses
            m.__dict__["_mfx_split"] = split
        self.compute_mode = name if isinstance(name, str) else {v: k for k, v in _DTYPES.items() if len(k) == 4}[name]
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

    # ---- backward pass in pieces (data-parallel overlap of the gradient exchange; engine/trainer.GraphedTrainStep) -----------------
    BACKWARD_SEGMENTS = 4

    def backward_segment_of(self, name):
        """Which of the four backward pieces produces the gradient of parameter `name`: 0 = heads + IDAUp (first to finish),
        1 = DLAUp, 2 = DLA level5 / level4, 3 = level3 .. stem (last)."""
        if name.startswith("heads.") or name.startswith("backbone.ida_up."):
            return 0
        if name.startswith("backbone.dla_up."):
            return 1
        if name.startswith("backbone.base.level4.") or name.startswith("backbone.base.level5."):
            return 2
        return 3

    def set_backward_cuts(self, cut):
        """The next training forward passes call `cut(name, tensors)` at the segment boundaries (None: one autograd graph again)."""
        self._cut = cut

    @staticmethod
    def backward_thunks(losses, cuts):
        """The four pieces of `losses.backward()` over the cut forward pass, in execution order.  `cuts[name] = (maps the forward
        produced, the detached leaves the next stage consumed)`; a piece back-propagates the leaves' gradients into the maps.
        level3's map has two consumers (level4 through "level3", DLAUp through "base"): their gradients are added first."""
        # the pieces below hard-wire DLA-34 at DOWN_RATIO 4 (first_level 2): levels 0 / 1 feed nothing but level 2, so their cut
        # leaves carry no gradient of their own.  Another first level would hand DLAUp those maps and their gradients would be
        # dropped silently -- refuse instead.
        base_leaves = cuts["base"][1]
        if len(base_leaves) != 6:
            raise RuntimeError("backward_thunks: expected the six DLA level maps at the 'base' cut, got %d" % len(base_leaves))

        def check_unused_levels():
            for i in (0, 1):
                if base_leaves[i].grad is not None:
                    raise RuntimeError("backward_thunks: level%d's map received a gradient through the 'base' cut -- this cut layout "
                                       "assumes first_level == 2 (DOWN_RATIO 4), where only levels 2..5 feed DLAUp" % i)

        def run(tensors, grads):
            pairs = [(t, g) for t, g in zip(tensors, grads) if g is not None and t.requires_grad]
            torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])

        def dla_up():
            o, l = cuts["dla_up"]
            run(o, [x.grad for x in l])

        def level54():
            o, l = cuts["base"]
            run(o[4:6], [l[4].grad, l[5].grad])

        def level3_to_stem():
            check_unused_levels()
            o, l = cuts["base"]
            g3, gx = l[3].grad, cuts["level3"][1][0].grad
            run([o[2], o[3]], [l[2].grad, gx if g3 is None else (g3 if gx is None else g3 + gx)])
        return [lambda: losses.backward(), dla_up, level54, level3_to_stem]

    def forward(self, images, targets=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        images = to_image_list(images)
        if not images.tensors.is_cuda:
            raise RuntimeError("KeypointDetector (HIP build) needs CUDA/HIP tensors: there is no CPU fallback")
        if self.training:                                           # detector.py:30-33 -> (loss_dict, log_loss_dict)
            return self.heads(self.backbone(images.tensors, getattr(self, "_cut", None)), targets)
        with torch.no_grad():
            features = self.backbone(images.tensors)
            return self.heads(features, targets, test=self.test)
