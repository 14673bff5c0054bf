"""Deterministic synthetic weights, images and per-image targets for the hot path.

There is no network for datasets or checkpoints, so every test, golden fixture and
benchmark runs on KITTI-shaped synthetic data.  Everything here is a pure
function of integer seeds through torch's CPU generator, so this container and the GPU
box regenerate bit-identical tensors (same torch build in both).

Weights are "trained-like" rather than the reference's init distributions: the init leaves
every DCN offset conv at zero (reference dcn_v2.py:114-116) and BN at identity, which would
exercise neither the deformable sampling nor the BN folding.  Here activations stay O(1)
through all ~55 layers and DCN offsets have a std of ~1.5 pixels, including samples that
leave the feature map.
"""
import math

import numpy as np
import torch

# typical KITTI P2
KITTI_P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728],
                     [0.0, 721.5377, 172.854, 0.2163791],
                     [0.0, 0.0, 1.0, 0.002745884]], dtype=np.float64)


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def synthetic_images(batch, height=384, width=1280, seed=1000):
    """(B,3,H,W) fp32 ~ N(0,1); image i uses seed+i (ImageNet-normalised KITTI pixels are ~N(0,1))."""
    return torch.stack([torch.randn(3, height, width, generator=_gen(seed + i)) for i in range(batch)])


def edge_indices(image_size, pad_size, down_ratio=4):
    """Border of the un-padded image on the stride-4 grid, in the reference's traversal order
    (data/datasets/kitti.py:126-179): left top->bottom, bottom left->right, right bottom->top,
    top right->left.  Returns (n,2) int64 (x,y)."""
    img_w, img_h = image_size
    x_min, y_min = int(math.ceil(pad_size[0] / down_ratio)), int(math.ceil(pad_size[1] / down_ratio))
    x_max, y_max = (pad_size[0] + img_w - 1) // down_ratio, (pad_size[1] + img_h - 1) // down_ratio
    pts = [(x_min, y) for y in range(y_min, y_max)]
    pts += [(x, y_max) for x in range(x_min, x_max)]
    pts += [(x_max, y) for y in range(y_max, y_min, -1)]
    pts += [(x, y_min) for x in range(x_max, x_min - 1, -1)]
    return torch.tensor(pts, dtype=torch.int64).view(-1, 2)


def synthetic_target(out_w=320, out_h=96, down_ratio=4, orig_size=None, P=KITTI_P2):
    """Test-split `targets` fields of one image (data/datasets/kitti.py:287-299) as a plain dict:
    pad_size (2,), size (W,H) of the padded frame, calib P (3,4), edge_indices (max_len,2), edge_len."""
    W, H = out_w * down_ratio, out_h * down_ratio
    if orig_size is None:                        # 1242x375 inside 1280x384; scaled for small grids
        orig_size = (W - 38, H - 9) if (W, H) == (1280, 384) else (W - 2 * (W // 64), H - 2 * (H // 64))
    pad = ((W - orig_size[0]) // 2, (H - orig_size[1]) // 2)          # kitti.py:218-228 centre pad
    ei = edge_indices(orig_size, pad, down_ratio)
    max_len = (out_w + out_h) * 2                                      # kitti.py:69
    padded = torch.zeros(max_len, 2, dtype=torch.int64)
    padded[:ei.shape[0]] = ei
    return dict(pad_size=torch.tensor(pad, dtype=torch.int64), size=(W, H), P=np.array(P, dtype=np.float64),
                edge_indices=padded, edge_len=int(ei.shape[0]) - 1)    # kitti.py:282-285 count-1


def synthetic_state_dict(template_state, seed=0, cls_bias=-1.0, offset_std=1.5):
    """Fill a state_dict with the reference's 478 keys (taken from `template_state`: name -> tensor
    of the right shape) deterministically.  Rules by key suffix / shape:
      conv weight            N(0, 2/fan_in) * gain   (He; keeps activations O(1) through ReLU)
      BN weight/bias/mean/var U(0.8,1.2) / N(0,0.1) / N(0,0.1) / U(0.8,1.2)
      conv_offset_mask       weight scaled so offsets ~ N(0, offset_std), bias N(0,0.2)
      depthwise up_k.weight  bilinear kernel * U(0.9,1.1)
      head 1x1 convs         N(0, 1/fan_in); class bias = cls_bias; other biases N(0,0.1)
    """
    g = _gen(seed)
    out = {}
    for name in sorted(template_state.keys()):
        t = template_state[name]
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            v = torch.zeros(shape, dtype=t.dtype)
        elif name.endswith("running_mean"):
            v = torch.randn(shape, generator=g) * 0.1
        elif name.endswith("running_var"):
            v = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif ".up_" in name and name.endswith("weight"):
            k = shape[2]
            f = math.ceil(k / 2)
            c = (2 * f - 1 - f % 2) / (2.0 * f)
            ker = torch.tensor([[(1 - abs(i / f - c)) * (1 - abs(j / f - c)) for j in range(k)] for i in range(k)])
            v = ker.view(1, 1, k, k) * (torch.rand(shape[0], 1, 1, 1, generator=g) * 0.2 + 0.9)
        elif "conv_offset_mask" in name:
            if name.endswith("weight"):
                fan_in = shape[1] * shape[2] * shape[3]
                # inputs to a DCN are post-ReLU features with E[x^2] ~ 1 -> out std ~ sqrt(fan_in)*w_std
                v = torch.randn(shape, generator=g) * (offset_std / math.sqrt(fan_in))
            else:
                v = torch.randn(shape, generator=g) * 0.2
        elif len(shape) >= 3 and name.endswith("weight"):              # conv2d / conv1d / DCN weight
            fan_in = int(np.prod(shape[1:]))
            is_head_out = ("class_head.2" in name or "reg_heads" in name or name.endswith("conv.3.weight")
                           or ".3.weight" in name)
            std = math.sqrt((1.0 if is_head_out else 2.0) / fan_in)
            v = torch.randn(shape, generator=g) * std
        elif name.endswith("weight"):                                   # BN gamma
            v = torch.rand(shape, generator=g) * 0.4 + 0.8
            if ".bn2." in name:                                         # residual branch: keep sums O(1)
                v = v * 0.4
        elif name.endswith("bias"):
            if "class_head.2" in name:
                v = torch.full(shape, float(cls_bias))
            elif "bn" in name or "actf" in name or ".1.bias" in name or "project.1" in name:
                v = torch.randn(shape, generator=g) * 0.1
            else:
                v = torch.randn(shape, generator=g) * 0.1
        else:
            raise KeyError("synthetic_state_dict: unhandled key %s %s" % (name, shape))
        out[name] = v.to(t.dtype).reshape(shape).contiguous()
    return out


# ------------------------------------------------------------------------------------------------
# Training-split targets (reference data/datasets/kitti.py:302-333).  The dataset pipeline is out
# of scope; this draws seeded 3D boxes in front of the KITTI camera and derives every field the loss consumes from
# their projection, so the fields are mutually consistent (centres, offsets, keypoints, depths, multi-bin angles).
# ------------------------------------------------------------------------------------------------
_DIM_MEAN = np.array(((3.8840, 1.5261, 1.6286), (0.8423, 1.7607, 0.6602), (1.7635, 1.7372, 0.5968)))
_DIM_STD = np.array(((0.4259, 0.1367, 0.1022), (0.2349, 0.1133, 0.1427), (0.1766, 0.0948, 0.1242)))


def _multibin(alpha, num_bin=4, margin=1 / 6):
    """4 bin flags + 4 residuals (kitti.py:181-200): bin centres 0, pi/2, pi, -pi/2, overlap margin pi/12."""
    centers = np.array([0, np.pi / 2, np.pi, -np.pi / 2])
    rng = np.pi / num_bin + 2 * np.pi / num_bin * margin
    off = alpha - centers
    off = np.where(off > np.pi, off - 2 * np.pi, off)
    off = np.where(off < -np.pi, off + 2 * np.pi, off)
    out = np.zeros(2 * num_bin)
    hit = np.abs(off) < rng
    out[:num_bin][hit] = 1
    out[num_bin:][hit] = off[hit]
    return out


def synthetic_train_target(seed, out_w=320, out_h=96, down_ratio=4, n_obj=None, max_objs=40, P=KITTI_P2):
    """One image's training `targets` fields as a dict (plus the test-split fields of synthetic_target)."""
    rs = np.random.RandomState(seed)
    base = synthetic_target(out_w, out_h, down_ratio, P=P)
    W, H = out_w * down_ratio, out_h * down_ratio
    pad = np.asarray(base["pad_size"], dtype=np.float64)
    img_w, img_h = W - 2 * pad[0], H - 2 * pad[1]
    Pm = np.asarray(P, dtype=np.float64).reshape(3, 4)
    n = int(rs.randint(1, 9)) if n_obj is None else n_obj
    f = dict(hm=np.zeros((3, out_h, out_w), np.float32), cls_ids=np.zeros(max_objs, np.int32),
             target_centers=np.zeros((max_objs, 2), np.int32), reg_mask=np.zeros(max_objs, np.uint8),
             trunc_mask=np.zeros(max_objs, np.uint8), reg_weight=np.zeros(max_objs, np.float32),
             offset_3D=np.zeros((max_objs, 2), np.float32), keypoints=np.zeros((max_objs, 10, 3), np.float32),
             keypoints_depth_mask=np.zeros((max_objs, 3), np.float32), dimensions=np.zeros((max_objs, 3), np.float32),
             locations=np.zeros((max_objs, 3), np.float32), rotys=np.zeros(max_objs, np.float32),
             alphas=np.zeros(max_objs, np.float32), orientations=np.zeros((max_objs, 8), np.float32))
    f["2d_bboxes"] = np.zeros((max_objs, 4), np.float32)
    f["gt_bboxes"] = np.zeros((max_objs, 4), np.float32)
    # focal length / image size of the small test grids is not KITTI's: scale the scene so boxes land inside
    sx = W / 1280.0

    def project(p3):
        q = Pm[:, :3] @ p3.T + Pm[:, 3:4]
        uv = (q[:2] / q[2:]).T
        uv[:, 0] = (uv[:, 0] - Pm[0, 2]) * sx + img_w / 2            # keep the principal point centred on small grids
        uv[:, 1] = (uv[:, 1] - Pm[1, 2]) * sx + img_h / 2
        return uv

    ys, xs = np.mgrid[0:out_h, 0:out_w]
    for i in range(n):
        c = int(rs.randint(0, 3))
        dims = _DIM_MEAN[c] + _DIM_STD[c] * rs.randn(3) * 0.5                       # (l, h, w)
        z = rs.uniform(6, 55)
        x = rs.uniform(-1.0, 1.0) * z * 1.1                                          # |x|/z > 0.86 leaves the image: truncated objects
        yb = 1.65 + rs.randn() * 0.1                                               # bottom of the box on the road
        ry = rs.uniform(-np.pi, np.pi)
        l, h, w = dims
        R = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
        cor = np.array([[l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2], [0, 0, 0, 0, -h, -h, -h, -h],
                        [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2]])
        cor = (R @ cor).T + np.array([x, yb, z])
        pts3 = np.vstack((cor, [[x, yb, z]], [[x, yb - h, z]]))                     # 8 corners, bottom centre, top centre
        loc = np.array([x, yb - h / 2, z])
        uv = project(pts3)
        cuv = project(loc[None])[0]
        box = np.array([uv[:8, 0].min(), uv[:8, 1].min(), uv[:8, 0].max(), uv[:8, 1].max()])
        box = np.clip(box, 0, [img_w - 1, img_h - 1, img_w - 1, img_h - 1])
        if box[2] - box[0] < 2 or box[3] - box[1] < 2:
            continue
        inside = 0 <= cuv[0] <= img_w - 1 and 0 <= cuv[1] <= img_h - 1
        centre = cuv if inside else np.clip(cuv, 0, [img_w - 1, img_h - 1])        # truncated: nearest border point
        cf = (centre + pad) / down_ratio
        ci = np.floor(cf).astype(np.int32)
        ci = np.clip(ci, 0, [out_w - 1, out_h - 1])
        f["cls_ids"][i], f["target_centers"][i] = c, ci
        f["reg_mask"][i], f["reg_weight"][i], f["trunc_mask"][i] = 1, 1.0, 0 if inside else 1
        f["offset_3D"][i] = (cuv + pad) / down_ratio - ci
        f["gt_bboxes"][i] = box
        f["2d_bboxes"][i] = (box + np.tile(pad, 2)) / down_ratio
        kf = (uv + pad) / down_ratio
        vis = (uv[:, 0] >= 0) & (uv[:, 0] <= img_w - 1) & (uv[:, 1] >= 0) & (uv[:, 1] <= img_h - 1) & (pts3[:, 2] > 0)
        f["keypoints"][i, :, :2] = kf - ci
        f["keypoints"][i, :, 2] = vis
        f["keypoints_depth_mask"][i] = (vis[8] & vis[9], vis[[0, 2, 4, 6]].all(), vis[[1, 3, 5, 7]].all())
        f["dimensions"][i], f["locations"][i], f["rotys"][i] = dims, loc, ry
        alpha = ry - np.arctan2(x, z)
        alpha = alpha - 2 * np.pi if alpha > np.pi else alpha + 2 * np.pi if alpha < -np.pi else alpha
        f["alphas"][i] = alpha
        f["orientations"][i] = _multibin(alpha)
        bw, bh = (box[2] - box[0]) / down_ratio, (box[3] - box[1]) / down_ratio
        sigma = max(1.0, 0.1 * float(np.hypot(bw, bh))) / 1.5
        gauss = np.exp(-((xs - ci[0]) ** 2 + (ys - ci[1]) ** 2) / (2 * sigma * sigma)).astype(np.float32)
        gauss[ci[1], ci[0]] = 1.0
        f["hm"][c] = np.maximum(f["hm"][c], gauss)
    base.update(f)
    return base


# ------------------------------------------------------------------------------------------------
# KITTI label text for the input-pipeline tests and benchmarks (data/datasets/kitti.py consumes label_2/*.txt lines).
# ------------------------------------------------------------------------------------------------
def synthetic_kitti_labels(seed, img_w, img_h, n_obj, z_range=(4, 60), occl_max=3):
    """KITTI-format label lines (2-decimal text, like label_2/*.txt) for seeded boxes in front of (and around) the camera:
    in-image objects, objects whose 3D centre projects outside the image, objects straddling or behind the camera
    plane, over-truncated small boxes (annotation filter), and classes outside DETECT_CLASSES."""
    rs = np.random.RandomState(seed)
    P = np.asarray(KITTI_P2, dtype=np.float64).reshape(3, 4)
    dims = {"Car": (1.53, 1.63, 3.88), "Pedestrian": (1.76, 0.66, 0.84), "Cyclist": (1.74, 0.60, 1.76),
            "Van": (2.2, 1.9, 5.1), "Truck": (3.2, 2.6, 10.0), "Misc": (1.9, 1.5, 3.5)}                # (h, w, l)
    names = ["Car"] * 5 + ["Pedestrian"] * 2 + ["Cyclist"] * 2 + ["Van", "Truck", "Misc", "DontCare"]
    lines = []
    for k in range(n_obj):
        typ = names[rs.randint(len(names))]
        if typ == "DontCare":
            x1, y1 = rs.uniform(0, img_w - 60), rs.uniform(0, img_h - 40)
            lines.append("DontCare -1 -1 -10 %.2f %.2f %.2f %.2f -1 -1 -1 -1000 -1000 -1000 -10"
                         % (x1, y1, x1 + rs.uniform(5, 50), y1 + rs.uniform(5, 30)))
            continue
        h, w, l = np.array(dims[typ]) * (1 + 0.1 * rs.randn(3))
        mode = rs.randint(10)
        z = rs.uniform(*z_range)
        x = rs.uniform(-0.75, 0.75) * z
        if mode == 0:
            x = np.sign(rs.randn()) * rs.uniform(0.82, 1.0) * z          # centre projects beyond the left/right border
        elif mode == 1:
            z, x = rs.uniform(1.2, 3.0), rs.uniform(-2.5, 2.5)            # box straddles the camera plane
        elif mode == 2 and k % 2 == 0:
            z = -rs.uniform(2, 20)                                        # behind the camera
        y = 1.65 + 0.1 * rs.randn()
        ry = rs.uniform(-np.pi, np.pi)
        c, s_ = np.cos(ry), np.sin(ry)
        xs = np.array([l / 2, l / 2, -l / 2, -l / 2] * 2)
        ys = np.array([0, 0, 0, 0, -h, -h, -h, -h])
        zs = np.array([w / 2, -w / 2, -w / 2, w / 2] * 2)
        X, Y, Z = c * xs + s_ * zs + x, ys + y, -s_ * xs + c * zs + z
        Zc = np.maximum(Z, 0.1)
        u = (P[0, 0] * X + P[0, 2] * Zc + P[0, 3]) / Zc
        v = (P[1, 1] * Y + P[1, 2] * Zc + P[1, 3]) / Zc
        full = np.array([u.min(), v.min(), u.max(), v.max()])
        box = np.array([max(full[0], 0), max(full[1], 0), min(full[2], img_w - 1), min(full[3], img_h - 1)])
        if box[2] - box[0] < 1 or box[3] - box[1] < 1:                    # not visible at all: KITTI would not label it
            box = np.array([rs.uniform(0, img_w - 30), rs.uniform(0, img_h - 30), 0, 0])
            box[2:] = box[:2] + rs.uniform(3, 25, 2)
        area_full = max((full[2] - full[0]) * (full[3] - full[1]), 1e-6)
        trunc = float(np.clip(1 - (box[2] - box[0]) * (box[3] - box[1]) / area_full, 0, 1))
        if mode == 3:
            trunc = 0.95                                                  # annotation filter: dropped when the box is <= 20 px
        alpha = ry - np.arctan2(x, z)
        lines.append("%s %.2f %d %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f"
                     % (typ, trunc, rs.randint(0, occl_max + 1), alpha, box[0], box[1], box[2], box[3], h, w, l, x, y, z, ry))
    return lines


def synthetic_detections(seed, label_lines, img_w, img_h, recall=0.8, n_false=3):
    """(N,14) float32 detection rows in the detector's output format [cls, alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry,
    score] (detector_infer.py:215-218) derived from KITTI label lines: most labelled Car/Pedestrian/Cyclist objects are
    detected with perturbed boxes, a few are missed or mis-classified, and some false positives are added."""
    rs = np.random.RandomState(seed)
    ids = {"Car": 0, "Pedestrian": 1, "Cyclist": 2}
    rows = []
    for line in label_lines:
        d = line.split(" ")
        if d[0] not in ids or rs.rand() > recall:
            continue
        v = np.array([float(x) for x in d[1:15]])
        box, hwl, loc, ry = v[3:7].copy(), v[7:10].copy(), v[10:13].copy(), v[13]
        if loc[2] <= 0:
            continue
        q = rs.choice([0.3, 1.0, 3.0])                                    # detection quality: tight, typical, sloppy
        box += rs.randn(4) * 2.0 * q
        hwl *= 1 + 0.04 * q * rs.randn(3)
        loc += rs.randn(3) * np.array([0.05, 0.03, 0.02 * loc[2]]) * q
        ry = ry + 0.08 * q * rs.randn()
        cls = ids[d[0]] if rs.rand() > 0.05 else int(rs.randint(0, 3))
        alpha = ry - np.arctan2(loc[0], loc[2])
        rows.append([cls, alpha, box[0], box[1], box[2], box[3], hwl[0], hwl[1], hwl[2], loc[0], loc[1], loc[2], ry, rs.uniform(0.2, 1.0)])
    for _ in range(int(rs.randint(0, n_false + 1))):
        cls = int(rs.randint(0, 3))
        x1, y1 = rs.uniform(0, img_w - 80), rs.uniform(0, img_h - 60)
        z = rs.uniform(5, 60)
        rows.append([cls, rs.uniform(-3, 3), x1, y1, x1 + rs.uniform(10, 80), y1 + rs.uniform(10, 60), 1.5, 1.6, 3.9,
                     rs.uniform(-0.5, 0.5) * z, 1.7, z, rs.uniform(-3, 3), rs.uniform(0.2, 0.7)])
    return np.asarray(rows, dtype=np.float32).reshape(-1, 14)
