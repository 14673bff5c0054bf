"""Per-image target container (reference structures/params_3d.py:5-56) and the camera calibration
scalars the decode needs (reference data/datasets/kitti_utils.py:160-218, 350-369)."""
import numpy as np
import torch


class Calibration:
    """Projection matrix P (3,4) of the rectified camera; f/c/b scalars derived as the reference does."""

    def __init__(self, P):
        self.P = np.asarray(P, dtype=np.float64).reshape(3, 4)
        self.c_u, self.c_v = self.P[0, 2], self.P[1, 2]
        self.f_u, self.f_v = self.P[0, 0], self.P[1, 1]
        self.b_x = self.P[0, 3] / (-self.f_u)
        self.b_y = self.P[1, 3] / (-self.f_v)

    def as_f32(self):
        return np.array([self.f_u, self.f_v, self.c_u, self.c_v, self.b_x, self.b_y], dtype=np.float32)

    def project_image_to_rect(self, uv_depth):
        x = ((uv_depth[:, 0] - self.c_u) * uv_depth[:, 2]) / self.f_u + self.b_x
        y = ((uv_depth[:, 1] - self.c_v) * uv_depth[:, 2]) / self.f_v + self.b_y
        out = np.zeros_like(uv_depth) if isinstance(uv_depth, np.ndarray) else uv_depth.new_zeros(uv_depth.shape)
        out[:, 0], out[:, 1], out[:, 2] = x, y, uv_depth[:, 2]
        return out


class ParamsList:
    def __init__(self, image_size, is_train=True):
        self.size = image_size            # (W, H) of the padded frame
        self.is_train = is_train
        self.extra_fields = {}

    def add_field(self, field, field_data):
        if not isinstance(field_data, (Calibration, torch.Tensor)) and not hasattr(field_data, "f_u"):
            field_data = torch.as_tensor(field_data)
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def to(self, device):
        target = ParamsList(self.size, self.is_train)
        for k, v in self.extra_fields.items():
            target.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return target

    def __len__(self):
        return int(torch.count_nonzero(self.extra_fields["reg_mask"])) if self.is_train else 0

    def __repr__(self):
        return "ParamsList(regress_number={}, image_width={}, image_height={})".format(len(self), self.size[0], self.size[1])


def make_test_target(tgt):
    """synthetic.synthetic_target() dict -> ParamsList with the test-split fields (kitti.py:287-299)."""
    t = ParamsList(image_size=tgt["size"], is_train=False)
    t.add_field("pad_size", tgt["pad_size"])
    t.add_field("calib", Calibration(tgt["P"]))
    t.add_field("edge_len", tgt["edge_len"])
    t.add_field("edge_indices", tgt["edge_indices"])
    return t


TRAIN_FIELDS = ("hm", "cls_ids", "target_centers", "reg_mask", "trunc_mask", "reg_weight", "offset_3D", "keypoints",
                "keypoints_depth_mask", "dimensions", "locations", "rotys", "alphas", "orientations", "2d_bboxes", "gt_bboxes")


def make_train_target(tgt):
    """synthetic.synthetic_train_target() dict -> ParamsList with the training-split fields (kitti.py:302-333)."""
    t = ParamsList(image_size=tgt["size"], is_train=True)
    t.add_field("pad_size", tgt["pad_size"])
    t.add_field("calib", Calibration(tgt["P"]))
    t.add_field("edge_len", tgt["edge_len"])
    t.add_field("edge_indices", tgt["edge_indices"])
    for k in TRAIN_FIELDS:
        t.add_field(k, tgt[k])
    return t
