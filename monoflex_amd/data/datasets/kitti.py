"""KITTI dataset front (reference data/datasets/kitti.py:28-525).

Same constructor, directory layout (image_2 / label_2 / calib / ImageSets/<split>.txt), `__len__` and
`__getitem__ -> (image, target, original_idx)` contract as the reference, with the work split differently: the host reads
and parses files (PNG decode, label and calib text) and tosses the flip coin; padding, flipping, normalisation, all
geometry, the Gaussian heat maps and the border walk run on the GPU (mfx_kitti_encode_targets /
mfx_kitti_preprocess_u8), a whole batch per launch.  `load_raw` + `encode_batch` is the batch path the collator uses;
`__getitem__` is the same thing for a batch of one.

Differences from the reference, by design: tensors in the target live on the device; `ori_img` (a uint8 copy of the padded
frame kept for visualisation) is not produced; USE_RIGHT_IMAGE is not implemented (runs/monoflex.yaml: False)."""
import os
import random

import numpy as np
import torch

from ...structures.params_3d import ParamsList
from ..encode import TARGET_FIELDS, EncodeParams, encode_targets, preprocess_images
from .kitti_utils import Calibration, read_label_records, RECORD_WIDTH

TARGET_ORDER = ("cls_ids", "target_centers", "keypoints", "keypoints_depth_mask", "dimensions", "locations", "reg_mask",
                "reg_weight", "offset_3D", "2d_bboxes", "pad_size", "rotys", "trunc_mask", "alphas", "orientations", "hm",
                "gt_bboxes", "occlusions", "truncations", "edge_len", "edge_indices")        # kitti.py:496-523


class RawSample:
    """What the host produces per sample: decoded frame, label records, camera matrix, flip decision."""
    __slots__ = ("frame", "records", "calib", "flip", "original_idx")

    def __init__(self, frame, records, calib, flip, original_idx):
        self.frame, self.records, self.calib, self.flip, self.original_idx = frame, records, calib, flip, original_idx


class KITTIDataset(torch.utils.data.Dataset):
    def __init__(self, cfg, root, is_train=True, transforms=None, augment=True, device="cuda"):
        super().__init__()
        self.root = root
        self.image_dir, self.label_dir, self.calib_dir = (os.path.join(root, d) for d in ("image_2", "label_2", "calib"))
        self.split = cfg.DATASETS.TRAIN_SPLIT if is_train else cfg.DATASETS.TEST_SPLIT
        self.is_train = is_train
        self.transforms = transforms                 # kept for signature parity; normalisation runs in the frame kernel
        imageset = self.imageset_txt = os.path.join(root, "ImageSets", "{}.txt".format(self.split))
        if not os.path.exists(imageset):
            raise FileNotFoundError("ImageSets file not exist, dir = {}".format(imageset))
        with open(imageset) as f:
            self.image_files = [line.replace("\n", "") + ".png" for line in f if line.strip()]
        self.label_files = [i.replace(".png", ".txt") for i in self.image_files]
        self.classes = tuple(cfg.DATASETS.DETECT_CLASSES)
        self.num_classes, self.num_samples = len(self.classes), len(self.image_files)
        if cfg.DATASETS.USE_RIGHT_IMAGE and is_train:
            raise NotImplementedError("DATASETS.USE_RIGHT_IMAGE is not implemented in this build")
        self.params = EncodeParams.from_cfg(cfg)
        self.flip_p = float(cfg.INPUT.AUG_PARAMS[0][0]) if (is_train and augment) else 0.0      # augmentations/__init__.py:15-23
        self.pixel_mean, self.pixel_std = tuple(cfg.INPUT.PIXEL_MEAN), tuple(cfg.INPUT.PIXEL_STD)
        if cfg.INPUT.TO_BGR:
            raise NotImplementedError("INPUT.TO_BGR is not implemented in this build")
        self.input_width, self.input_height = self.params.in_w, self.params.in_h
        self.down_ratio = self.params.down
        self.output_width, self.output_height = self.input_width // self.down_ratio, self.input_height // self.down_ratio
        self.max_edge_length = (self.output_width + self.output_height) * 2
        self.max_objs = self.params.max_objs
        self.enable_edge_fusion = cfg.MODEL.HEAD.ENABLE_EDGE_FUSION
        self.device = torch.device(device)

    def __len__(self):
        return self.num_samples

    # ---- host side: files -> raw sample -------------------------------------------------------------------
    def get_image(self, idx):
        from PIL import Image
        return np.asarray(Image.open(os.path.join(self.image_dir, self.image_files[idx])).convert("RGB"), dtype=np.uint8)

    def get_calibration(self, idx, use_right_cam=False):
        return Calibration(os.path.join(self.calib_dir, self.label_files[idx]), use_right_cam=use_right_cam)

    def get_label_objects(self, idx):
        """Records of the objects in DETECT_CLASSES (read_label + filtrate_objects); empty for the test split."""
        if self.split == "test":
            return np.zeros((0, RECORD_WIDTH), dtype=np.float64)
        return read_label_records(os.path.join(self.label_dir, self.label_files[idx]), self.classes)

    def get_edge_utils(self, image_size, pad_size, down_ratio=4):
        """Border walk of the valid image area on the output grid as an (n, 2) int64 tensor of (x, y) (kitti.py:126-179);
        the device encoder emits the same sequence zero-padded to max_edge_length."""
        img_w, img_h = image_size
        x0, y0 = -(-int(pad_size[0]) // down_ratio), -(-int(pad_size[1]) // down_ratio)
        x1, y1 = (int(pad_size[0]) + img_w - 1) // down_ratio, (int(pad_size[1]) + img_h - 1) // down_ratio
        pts = [(x0, y) for y in range(y0, y1)] + [(x, y1) for x in range(x0, x1)] + \
              [(x1, y) for y in range(y1, y0, -1)] + [(x, y0) for x in range(x1, x0 - 1, -1)]
        return torch.tensor(pts, dtype=torch.int64).reshape(-1, 2)

    def load_raw(self, idx):
        if idx >= self.num_samples:
            raise IndexError(idx)
        flip = self.flip_p > 0 and random.random() < self.flip_p            # augmentations.py:38
        return RawSample(self.get_image(idx), self.get_label_objects(idx), self.get_calibration(idx), bool(flip),
                         self.image_files[idx][:6])

    # ---- device side: raw samples -> network input + targets ----------------------------------------------
    def encode_batch(self, samples, check=True):
        """[RawSample] -> ((B,3,H,W) float32 images, [ParamsList], [original_idx], stacked field dict)."""
        frames = [s.frame for s in samples]
        flips = [s.flip for s in samples]
        sizes = [(f.shape[1], f.shape[0]) for f in frames]
        images = preprocess_images(frames, flips, self.params, self.device, self.pixel_mean, self.pixel_std)
        fields = encode_targets([s.records for s in samples], [s.calib.P for s in samples], sizes, flips, self.params,
                                self.device, check=check)
        test_split = self.split == "test"
        targets = []
        for b, s in enumerate(samples):
            t = ParamsList(image_size=(self.input_width, self.input_height), is_train=self.is_train)
            calib = s.calib.flipped(sizes[b][0]) if s.flip else s.calib
            if test_split:                                                   # kitti.py:287-299
                t.add_field("pad_size", fields["pad_size"][b])
                t.add_field("calib", calib)
                if self.enable_edge_fusion:
                    t.add_field("edge_len", fields["edge_len"][b])
                    t.add_field("edge_indices", fields["edge_indices"][b])
            else:
                for k in TARGET_ORDER:
                    if k in ("edge_len", "edge_indices") and not self.enable_edge_fusion:
                        continue
                    t.add_field(k, fields[k][b])
                    if k == "locations":
                        t.add_field("calib", calib)                          # same position as in the reference's field list
            targets.append(t)
        return images, targets, [s.original_idx for s in samples], fields

    def __getitem__(self, idx):
        images, targets, ids, _ = self.encode_batch([self.load_raw(idx)])
        return images[0], targets[0], ids[0]
