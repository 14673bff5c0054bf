"""Device-side sample encoding: the ctypes front of mfx_kitti_encode_targets / mfx_kitti_preprocess_u8.

Replaces the numeric half of the reference's KITTIDataset.__getitem__ (data/datasets/kitti.py:231-525), its flip
augmentation (data/augmentations/augmentations.py:33-78) and ToTensor+Normalize (data/transforms/transforms.py:15-31):
one launch pair encodes the targets of a whole batch, one launch converts its frames.  Inputs travel as a single pinned
host buffer per call; nothing is synchronised unless the caller asks for the status check."""
import ctypes

import numpy as np
import torch

from .. import lib as L
from .datasets.kitti_utils import RECORD_WIDTH

# target field -> (descriptor member, trailing shape (M = max_objs, C/H/W = heat map, E = edge slots), dtype)
TARGET_FIELDS = {
    "hm": ("hm", ("C", "H", "W"), torch.float32),
    "cls_ids": ("cls_ids", ("M",), torch.int32),
    "target_centers": ("target_centers", ("M", 2), torch.int32),
    "keypoints": ("keypoints", ("M", 10, 3), torch.float32),
    "keypoints_depth_mask": ("keypoints_depth_mask", ("M", 3), torch.float32),
    "dimensions": ("dimensions", ("M", 3), torch.float32),
    "locations": ("locations", ("M", 3), torch.float32),
    "reg_mask": ("reg_mask", ("M",), torch.uint8),
    "reg_weight": ("reg_weight", ("M",), torch.float32),
    "offset_3D": ("offset_3D", ("M", 2), torch.float32),
    "2d_bboxes": ("bboxes", ("M", 4), torch.float32),
    "gt_bboxes": ("gt_bboxes", ("M", 4), torch.float32),
    "rotys": ("rotys", ("M",), torch.float32),
    "trunc_mask": ("trunc_mask", ("M",), torch.uint8),
    "alphas": ("alphas", ("M",), torch.float32),
    "orientations": ("orientations", ("M", 8), torch.float32),
    "occlusions": ("occlusions", ("M",), torch.float64),
    "truncations": ("truncations", ("M",), torch.float64),
    "pad_size": ("pad_size", (2,), torch.int64),
    "edge_indices": ("edge_indices", ("E", 2), torch.int64),
    "edge_len": ("edge_len", (), torch.int64),
    "P": ("P_out", (3, 4), torch.float64),
    "heat_radius": ("heat_radius", ("M", 4), torch.int32),
    "status": ("status", (), torch.int32),
}
STATUS_MESSAGES = {1: "more objects of the detect classes than DATASETS.MAX_OBJECTS",
                   2: "truncated object whose 2D box centre lies outside the image (the reference fails on it)",
                   4: "truncated object: the centre line meets no image border",
                   8: "boundary heat map with both radii > 0"}


class EncodeParams:
    """The dataset settings the encoder depends on (reference KITTIDataset.__init__, kitti.py:60-98)."""

    def __init__(self, in_w=1280, in_h=384, down=4, max_objs=40, num_classes=3, filter_annos=(0.9, 20.0), filter_enable=True,
                 edge_ratio=0.5):
        self.in_w, self.in_h, self.down, self.max_objs, self.num_classes = in_w, in_h, down, max_objs, num_classes
        self.filter_trunc = float(filter_annos[0]) if filter_enable else -1.0
        self.filter_size = float(filter_annos[1])
        self.edge_ratio = float(edge_ratio)

    @classmethod
    def from_cfg(cls, cfg):
        unsupported = []
        if cfg.INPUT.HEATMAP_CENTER != "3D": unsupported.append("INPUT.HEATMAP_CENTER != '3D'")
        if not cfg.DATASETS.CONSIDER_OUTSIDE_OBJS: unsupported.append("DATASETS.CONSIDER_OUTSIDE_OBJS False")
        if cfg.INPUT.APPROX_3D_CENTER != "intersect": unsupported.append("INPUT.APPROX_3D_CENTER != 'intersect'")
        if not cfg.INPUT.ADJUST_BOUNDARY_HEATMAP: unsupported.append("INPUT.ADJUST_BOUNDARY_HEATMAP False")
        if not cfg.INPUT.KEYPOINT_VISIBLE_MODIFY: unsupported.append("INPUT.KEYPOINT_VISIBLE_MODIFY False")
        if cfg.INPUT.ORIENTATION != "multi-bin" or cfg.INPUT.ORIENTATION_BIN_SIZE != 4:
            unsupported.append("INPUT.ORIENTATION other than 4-bin 'multi-bin'")
        if unsupported:
            raise NotImplementedError("the device target encoder implements the runs/monoflex.yaml settings only: " + "; ".join(unsupported))
        return cls(cfg.INPUT.WIDTH_TRAIN, cfg.INPUT.HEIGHT_TRAIN, cfg.MODEL.BACKBONE.DOWN_RATIO, cfg.DATASETS.MAX_OBJECTS,
                   len(cfg.DATASETS.DETECT_CLASSES), tuple(cfg.DATASETS.FILTER_ANNOS), bool(cfg.DATASETS.FILTER_ANNO_ENABLE),
                   cfg.INPUT.HEATMAP_RATIO)

    def dims(self):
        out_w, out_h = self.in_w // self.down, self.in_h // self.down
        return {"M": self.max_objs, "C": self.num_classes, "H": out_h, "W": out_w, "E": 2 * (out_w + out_h)}


def pack_inputs(records, Ps, sizes, flips, params):
    """Host-side packing of one batch: list of (n_i,14) record arrays, (3,4) matrices, (w,h) sizes, flip flags ->
    dict of numpy arrays in the descriptor's input layout. Raises when a sample has more objects than MAX_OBJECTS
    (the reference overruns its fixed-size arrays there)."""
    B, M = len(records), params.max_objs
    rec = np.zeros((B, M, RECORD_WIDTH), dtype=np.float64)
    n_obj = np.zeros(B, dtype=np.int32)
    for b, r in enumerate(records):
        r = np.asarray(r, dtype=np.float64).reshape(-1, RECORD_WIDTH)
        if r.shape[0] > M:
            raise IndexError("sample %d has %d objects of the detect classes, DATASETS.MAX_OBJECTS is %d" % (b, r.shape[0], M))
        rec[b, :r.shape[0]] = r
        n_obj[b] = r.shape[0]
    return dict(records=rec, n_obj=n_obj, P=np.asarray(Ps, dtype=np.float64).reshape(B, 3, 4).copy(),
                img_wh=np.asarray(sizes, dtype=np.int32).reshape(B, 2).copy(), flip=np.asarray(flips, dtype=np.int32).reshape(B).copy())


def _align16(n):
    return (n + 15) // 16 * 16


def _upload(arrays, device):
    """Several small numpy arrays -> one pinned buffer -> one async copy; returns device views in the original dtypes."""
    total = sum(_align16(a.nbytes) for a in arrays.values())
    host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
    hn, views, off = host.numpy(), {}, 0
    for k, a in arrays.items():
        hn[off:off + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        views[k] = (off, a.shape, a.dtype)
        off += _align16(a.nbytes)
    dev = host.to(device, non_blocking=True)
    return {k: _view(dev, o, shape, getattr(torch, np.dtype(dt).name)) for k, (o, shape, dt) in views.items()}


def _view(buf, offset, shape, dtype):
    n = int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size()
    return buf[offset:offset + n].view(dtype).view(*shape)


def _alloc_outputs(B, dims, device):
    """All output fields as views of one device allocation (one allocator call per batch instead of 24)."""
    specs, total = {}, 0
    for name, (_, shape, dt) in TARGET_FIELDS.items():
        full = (B,) + tuple(dims.get(s, s) for s in shape)
        specs[name] = (total, full, dt)
        total += _align16(int(np.prod(full, dtype=np.int64)) * torch.empty((), dtype=dt).element_size())
    buf = torch.empty(total, dtype=torch.uint8, device=device)
    return {name: _view(buf, o, full, dt) for name, (o, full, dt) in specs.items()}


def encode_targets(records, Ps, sizes, flips, params, device, check=True):
    """Batch of raw labels -> {field: (B, ...) device tensor} with every field of the reference's training target
    (kitti.py:496-523; "calib" is left to the caller, "P" is the camera matrix after the flip).
    check=True synchronises once to turn a non-zero status into the exception the reference would have raised."""
    if torch.device(device).type != "cuda":
        raise RuntimeError("encode_targets runs on the GPU (HIP kernels); there is no CPU fallback")
    lib = L.load()
    inp = _upload(pack_inputs(records, Ps, sizes, flips, params), device)
    B, dims = len(records), params.dims()
    out = _alloc_outputs(B, dims, device)
    d = L.KittiDesc()
    for k, t in inp.items():
        setattr(d, k, t.data_ptr())
    for name, (member, _, _) in TARGET_FIELDS.items():
        setattr(d, member, out[name].data_ptr())
    d.B, d.max_objs, d.in_w, d.in_h, d.down, d.num_classes = B, params.max_objs, params.in_w, params.in_h, params.down, params.num_classes
    d.filter_trunc, d.filter_size, d.edge_ratio = params.filter_trunc, params.filter_size, params.edge_ratio
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream().cuda_stream
        L.check(lib.mfx_kitti_encode_targets(ctypes.byref(d), ctypes.c_void_p(stream)), "mfx_kitti_encode_targets")
    if check:
        check_status(out["status"])
    return out


def check_status(status):
    bad = torch.nonzero(status).flatten().tolist()
    if bad:
        code = int(status[bad[0]])
        msgs = [m for bit, m in STATUS_MESSAGES.items() if code & bit]
        raise ValueError("target encoding failed for sample %d of the batch: %s" % (bad[0], "; ".join(msgs)))


def preprocess_images(frames, flips, params, device, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """List of (h_i, w_i, 3) uint8 RGB frames (numpy) -> (B, 3, in_h, in_w) float32 device tensor: flip, centre zero padding,
    /255, (x - mean) / std (kitti.py:218-228, transforms.py:15-31; INPUT.TO_BGR False)."""
    if torch.device(device).type != "cuda":
        raise RuntimeError("preprocess_images runs on the GPU (HIP kernel); there is no CPU fallback")
    lib = L.load()
    B = len(frames)
    sizes = np.zeros((B, 2), dtype=np.int32)
    offsets = np.zeros(B, dtype=np.int64)
    total = 0
    for b, f in enumerate(frames):
        if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
            raise ValueError("frame %d: expected (h, w, 3) uint8, got %s %s" % (b, f.shape, f.dtype))
        if f.shape[0] > params.in_h or f.shape[1] > params.in_w:
            raise ValueError("frame %d (%dx%d) is larger than the network input %dx%d"
                             % (b, f.shape[1], f.shape[0], params.in_w, params.in_h))
        sizes[b] = (f.shape[1], f.shape[0])
        offsets[b] = total
        total += _align16(f.size)
    host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
    hn = host.numpy()
    for b, f in enumerate(frames):
        hn[offsets[b]:offsets[b] + f.size] = f.reshape(-1)
    pixels = host.to(device, non_blocking=True)
    meta = _upload(dict(offsets=offsets, img_wh=sizes, flip=np.asarray(flips, dtype=np.int32).reshape(B).copy()), device)
    out = torch.empty((B, 3, params.in_h, params.in_w), dtype=torch.float32, device=device)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream().cuda_stream
        L.check(lib.mfx_kitti_preprocess_u8(pixels.data_ptr(), meta["offsets"].data_ptr(), meta["img_wh"].data_ptr(),
                                            meta["flip"].data_ptr(), out.data_ptr(), B, params.in_w, params.in_h, m3, s3,
                                            ctypes.c_void_p(stream)), "mfx_kitti_preprocess_u8")
    return out
