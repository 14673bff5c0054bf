"""oracle/kitti_encode_ref.py -- TEST INFRASTRUCTURE (CPU restatement; never imported by the product path).

numpy restatement of the reference's per-sample input pipeline and training-target encoding:

  label line -> object            data/datasets/kitti_utils.py:61-97 (Object3d.__init__), :31-40 (convertRot2Alpha)
  8 corners                       data/datasets/kitti_utils.py:115-133 (generate_corners3d)
  projection                      data/datasets/kitti_utils.py:316-325 (Calibration.project_rect_to_image)
  horizontal flip                 data/augmentations/augmentations.py:33-78 (RandomHorizontallyFlip)
  centre pad + edge indices       data/datasets/kitti.py:218-228 (pad_image), :126-179 (get_edge_utils), :268-285
  target encoding                 data/datasets/kitti.py:301-525
  truncated-centre intersection   data/datasets/kitti_utils.py:990-1028 (approx_proj_center)
  heat-map rasterisation          model/heatmap_coder.py:37-124 (gaussian_radius, gaussian2D, draw_umich_gaussian[_2D])
  multi-bin angle code            data/datasets/kitti.py:181-200 (encode_alpha_multibin)
  ToTensor + Normalize            data/transforms/transforms.py:15-31, data/transforms/build.py:3-17

Pinned by tests/golden/kitti_encode.npz = outputs of the reference's own KITTIDataset.__getitem__ on a generated
KITTI-format directory (oracle/gen_golden.py kitti; tests/test_kitti_encode_cpu.py).  Arithmetic is float64 like the
reference's numpy code; fields are stored in the reference's dtypes.  Settings are those of runs/monoflex.yaml (the only
configuration on the hot path): 3D heat-map centre, outside objects kept with the 'intersect' centre, boundary heat-maps,
modified keypoint visibility, multi-bin (4) orientation, annotation filter [0.9, 20].
"""
import math

import numpy as np

TYPE_ID = {"Car": 0, "Pedestrian": 1, "Cyclist": 2, "Van": -4, "Truck": -4, "Person_sitting": -2, "Tram": -99,
           "Misc": -99, "DontCare": -1}                              # config/__init__.py:3-13
PIXEL_MEAN, PIXEL_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)    # config/defaults.py:34-36


class Obj:
    """Fields of kitti_utils.Object3d that the encoder reads (kitti_utils.py:64-92)."""

    def __init__(self, line):
        d = line.split(" ")
        v = [float(x) for x in d[1:]]
        self.type = d[0]
        self.truncation, self.occlusion = v[0], int(v[1])
        self.xmin, self.ymin, self.xmax, self.ymax = v[3], v[4], v[5], v[6]
        self.box2d = np.array([self.xmin, self.ymin, self.xmax, self.ymax], dtype=np.float32)
        self.h, self.w, self.l = v[7], v[8], v[9]
        self.t = np.array((v[10], v[11], v[12]), dtype=np.float32)
        self.ry = v[13]
        self.alpha = rot2alpha(self.ry, self.t[2], self.t[0])


def rot2alpha(ry, z, x):                                             # kitti_utils.py:31-40
    a = ry - math.atan2(x, z)
    while a > math.pi:
        a -= 2 * math.pi
    while a < -math.pi:
        a += 2 * math.pi
    return a


def read_objects(lines, classes=("Car", "Pedestrian", "Cyclist")):
    """read_label + filtrate_objects (kitti_utils.py:443-447, kitti.py:202-216)."""
    objs = [Obj(l.rstrip()) for l in lines if l.strip()]
    return [o for o in objs if o.type in classes]


def corners3d(o):                                                    # kitti_utils.py:115-133
    l, h, w = o.l, o.h, o.w
    xs = [l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2]
    ys = [0, 0, 0, 0, -h, -h, -h, -h]
    zs = [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2]
    c, s = np.cos(o.ry), np.sin(o.ry)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.dot(R, np.vstack([xs, ys, zs])).T + o.t


def project(P, pts):                                                 # kitti_utils.py:316-325
    hom = np.hstack((pts, np.ones((pts.shape[0], 1))))
    q = np.dot(hom, P.T)
    q[:, 0] /= q[:, 2]
    q[:, 1] /= q[:, 2]
    return q[:, :2], q[:, 2]


def flip(objs, P, img_w):
    """RandomHorizontallyFlip with the coin already tossed (augmentations.py:38-76). Returns the flipped P."""
    for o in objs:
        w = o.xmax - o.xmin
        o.xmin = img_w - o.xmax - 1
        o.xmax = o.xmin + w
        o.box2d = np.array([o.xmin, o.ymin, o.xmax, o.ymax], dtype=np.float32)
        ry = (-math.pi - o.ry) if o.ry < 0 else (math.pi - o.ry)
        while ry > math.pi:
            ry -= 2 * math.pi
        while ry < -math.pi:
            ry += 2 * math.pi
        o.ry = ry
        t = o.t.copy()
        t[0] = -t[0]
        o.t = t
        o.alpha = rot2alpha(ry, o.t[2], o.t[0])
    P = P.copy()
    P[0, 2] = img_w - P[0, 2] - 1
    P[0, 3] = -P[0, 3]
    return P


def pad_size(img_w, img_h, in_w=1280, in_h=384):                     # kitti.py:218-228
    return np.array([(in_w - img_w) // 2, (in_h - img_h) // 2])


def edge_indices(img_w, img_h, pad, down=4, max_len=832):
    """get_edge_utils + the zero padding / count-1 of __getitem__ (kitti.py:126-179, 276-284): border pixels of the valid
    image area on the stride-4 grid, walked left (down), bottom (right), right (up), top (left)."""
    x0, y0 = int(np.ceil(pad[0] / down)), int(np.ceil(pad[1] / down))
    x1, y1 = (pad[0] + img_w - 1) // down, (pad[1] + img_h - 1) // down
    pts = [(x0, y) for y in range(y0, y1)] + [(x, y1) for x in range(x0, x1)] + \
          [(x1, y) for y in range(y1, y0, -1)] + [(x, y0) for x in range(x1, x0 - 1, -1)]
    out = np.zeros((max_len, 2), dtype=np.int64)
    out[:len(pts)] = np.array(pts, dtype=np.int64).reshape(-1, 2)
    return out, len(pts) - 1


def intersect_center(pc, c2d, img_w, img_h):
    """approx_proj_center (kitti_utils.py:990-1028): the image-border point on the line projected-centre -> 2D box
    centre that is closest to the projected centre. The reference fits the line with np.polyfit(deg 1) through the two
    points, i.e. the exact line; it is written in closed form here."""
    if not (0 <= c2d[0] <= img_w - 1 and 0 <= c2d[1] <= img_h - 1):
        return None
    with np.errstate(divide="ignore", invalid="ignore"):
        a = (c2d[1] - pc[1]) / (c2d[0] - pc[0])
        b = pc[1] - a * pc[0]
        cand = []
        if 0 <= b <= img_h - 1:
            cand.append((0.0, b))
        ry = (img_w - 1) * a + b
        if 0 <= ry <= img_h - 1:
            cand.append((img_w - 1.0, ry))
        tx = -b / a
        if 0 <= tx <= img_w - 1:
            cand.append((tx, 0.0))
        bx = (img_h - 1 - b) / a
        if 0 <= bx <= img_w - 1:
            cand.append((bx, img_h - 1.0))
    cand = np.array(cand)
    return cand[np.argmin(np.linalg.norm(cand - pc.reshape(1, 2), axis=1))]


def gaussian_radius(h, w, min_overlap=0.7):                          # heatmap_coder.py:37-57
    b1 = h + w
    c1 = w * h * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * c1)) / 2
    b2 = 2 * (h + w)
    c2 = (1 - min_overlap) * w * h
    r2 = (b2 + np.sqrt(b2 ** 2 - 16 * c2)) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (h + w)
    c3 = (min_overlap - 1) * w * h
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def _gauss_circular(r):                                              # gaussian2D (heatmap_coder.py:59-67), sigma = (2r+1)/6
    sigma = (2 * r + 1) / 6
    y, x = np.ogrid[-float(r):r + 1, -float(r):r + 1]
    g = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    g[g < np.finfo(g.dtype).eps * g.max()] = 0
    return g


def _gauss_ellipse(rx, ry):                                          # ellip_gaussian2D (heatmap_coder.py:126-135)
    sx, sy = (2 * rx + 1) / 6, (2 * ry + 1) / 6
    y, x = np.ogrid[-float(ry):ry + 1, -float(rx):rx + 1]
    g = np.exp(-(x * x) / (2 * sx * sx) - (y * y) / (2 * sy * sy))
    g[g < np.finfo(g.dtype).eps * g.max()] = 0
    return g


def draw_gaussian(hm, cx, cy, rx, ry, circular):
    """draw_umich_gaussian / draw_umich_gaussian_2D (heatmap_coder.py:85-124): element-wise max of the map with the
    Gaussian window clipped to the map (the eps*max cut-off never triggers inside a +-r window: min value e^-9)."""
    H, W = hm.shape
    g = _gauss_circular(rx) if circular else _gauss_ellipse(rx, ry)
    left, right = min(cx, rx), min(W - cx, rx + 1)
    top, bottom = min(cy, ry), min(H - cy, ry + 1)
    dst = hm[cy - top:cy + bottom, cx - left:cx + right]
    src = g[ry - top:ry + bottom, rx - left:rx + right]
    if min(src.shape) > 0 and min(dst.shape) > 0:
        np.maximum(dst, src, out=dst)


def multibin(alpha, num_bin=4, margin=1 / 6):                        # kitti.py:181-200
    centers = np.array([0, np.pi / 2, np.pi, -np.pi / 2])
    bin_size = 2 * np.pi / num_bin
    rng = bin_size / 2 + bin_size * margin
    off = alpha - centers
    off[off > np.pi] -= 2 * np.pi
    off[off < -np.pi] += 2 * np.pi
    out = np.zeros(2 * num_bin)
    for i in range(num_bin):
        if abs(off[i]) < rng:
            out[i], out[i + num_bin] = 1, off[i]
    return out


def encode_sample(lines, P, img_w, img_h, do_flip=False, in_w=1280, in_h=384, down=4, max_objs=40,
                  filter_annos=(0.9, 20), edge_ratio=0.5):
    """Everything KITTIDataset.__getitem__ puts into the training `target` (kitti.py:231-525), as a dict of arrays."""
    P = np.asarray(P, dtype=np.float64).reshape(3, 4)
    objs = read_objects(lines)
    if do_flip:
        P = flip(objs, P, img_w)
    pad = pad_size(img_w, img_h, in_w, in_h)
    out_w, out_h = in_w // down, in_h // down
    x_min, y_min = int(np.ceil(pad[0] / down)), int(np.ceil(pad[1] / down))
    x_max, y_max = (pad[0] + img_w - 1) // down, (pad[1] + img_h - 1) // down
    ei, el = edge_indices(img_w, img_h, pad, down, (out_w + out_h) * 2)
    f = dict(hm=np.zeros((3, out_h, out_w), np.float32), cls_ids=np.zeros(max_objs, np.int32),
             target_centers=np.zeros((max_objs, 2), np.int32), gt_bboxes=np.zeros((max_objs, 4), np.float32),
             keypoints=np.zeros((max_objs, 10, 3), np.float32), keypoints_depth_mask=np.zeros((max_objs, 3), np.float32),
             dimensions=np.zeros((max_objs, 3), np.float32), locations=np.zeros((max_objs, 3), np.float32),
             rotys=np.zeros(max_objs, np.float32), alphas=np.zeros(max_objs, np.float32),
             offset_3D=np.zeros((max_objs, 2), np.float32), occlusions=np.zeros(max_objs), truncations=np.zeros(max_objs),
             orientations=np.zeros((max_objs, 8), np.float32), reg_mask=np.zeros(max_objs, np.uint8),
             trunc_mask=np.zeros(max_objs, np.uint8), reg_weight=np.zeros(max_objs, np.float32))
    f["2d_bboxes"] = np.zeros((max_objs, 4), np.float32)
    if len(objs) > max_objs:
        raise IndexError("more than MAX_OBJECTS=%d objects of the detect classes" % max_objs)   # the reference overruns its arrays
    for i, o in enumerate(objs):
        cls_id = TYPE_ID[o.type]
        locs = o.t.copy()
        locs[1] = locs[1] - o.h / 2                                  # float32 arithmetic (numpy>=2 weak python scalars)
        if locs[-1] <= 0:
            continue
        c3 = corners3d(o)
        c2, _ = project(P, c3)
        pbox = np.array([c2[:, 0].min(), c2[:, 1].min(), c2[:, 0].max(), c2[:, 1].max()])
        if pbox[0] >= 0 and pbox[1] >= 0 and pbox[2] <= img_w - 1 and pbox[3] <= img_h - 1:
            box = pbox.copy()                                        # float64
        else:
            box = o.box2d.copy()                                     # float32: all later box arithmetic stays float32
        if o.truncation >= filter_annos[0] and (box[2:] - box[:2]).min() <= filter_annos[1]:
            continue
        pc, _ = project(P, locs.reshape(-1, 3))
        pc = pc[0]
        inside = (0 <= pc[0] <= img_w - 1) and (0 <= pc[1] <= img_h - 1)
        approx = not inside
        if approx:
            tpc = intersect_center(pc, (box[:2] + box[2:]) / 2, img_w, img_h)
            if tpc is None:
                raise TypeError("truncated object whose 2D box centre is outside the image")   # reference: unpack of None
        else:
            tpc = pc.copy()
        k3 = np.concatenate((c3, np.stack((c3[:4].mean(axis=0), c3[4:].mean(axis=0)))), axis=0)
        k2, _ = project(P, k3)
        vis = (k2[:, 0] >= 0) & (k2[:, 0] <= img_w - 1) & (k2[:, 1] >= 0) & (k2[:, 1] <= img_h - 1) & (k3[:, -1] > 0)
        vis = np.append(np.tile(vis[:4] | vis[4:8], 2), np.tile(vis[8] | vis[9], 2))          # KEYPOINT_VISIBLE_MODIFY
        dvalid = np.stack((vis[[8, 9]].all(), vis[[0, 2, 4, 6]].all(), vis[[1, 3, 5, 7]].all()))
        k2 = (k2 + pad.reshape(1, 2)) / down
        tpc = (tpc + pad) / down
        pc = (pc + pad) / down
        box[0::2] += pad[0]
        box[1::2] += pad[1]
        box /= down
        bdim = box[2:] - box[:2]
        tc = tpc.round().astype(int)
        tc[0] = np.clip(tc[0], x_min, x_max)
        tc[1] = np.clip(tc[1], y_min, y_max)
        pred_2d = tc[0] >= box[0] and tc[1] >= box[1] and tc[0] <= box[2] and tc[1] <= box[3]
        if (bdim > 0).all() and 0 <= tc[0] <= out_w - 1 and 0 <= tc[1] <= out_h - 1:
            if approx:
                bw = min(tc[0] - box[0], box[2] - tc[0])
                bh = min(tc[1] - box[1], box[3] - tc[1])
                rx, ry = max(0, int(bw * edge_ratio)), max(0, int(bh * edge_ratio))
                assert min(rx, ry) == 0
                draw_gaussian(f["hm"][cls_id], int(tc[0]), int(tc[1]), rx, ry, circular=False)
            else:
                r = max(0, int(gaussian_radius(bdim[1], bdim[0])))
                draw_gaussian(f["hm"][cls_id], int(tc[0]), int(tc[1]), r, r, circular=True)
            f["cls_ids"][i] = cls_id
            f["target_centers"][i] = tc
            f["offset_3D"][i] = pc - tc
            f["gt_bboxes"][i] = o.box2d
            if pred_2d:
                f["2d_bboxes"][i] = box
            f["keypoints"][i] = np.concatenate((k2 - tc.reshape(1, -1), vis[:, None].astype(np.float32)), axis=1)
            f["keypoints_depth_mask"][i] = dvalid
            f["dimensions"][i] = (o.l, o.h, o.w)
            f["locations"][i] = locs
            f["rotys"][i], f["alphas"][i] = o.ry, o.alpha
            f["orientations"][i] = multibin(o.alpha)
            f["reg_mask"][i], f["reg_weight"][i], f["trunc_mask"][i] = 1, 1, int(approx)
            f["occlusions"][i], f["truncations"][i] = float(o.occlusion), o.truncation
    f.update(pad_size=pad, edge_indices=ei, edge_len=el, P=P, size=np.array([in_w, in_h]))
    return f


def transform_image(img_u8, do_flip=False, in_w=1280, in_h=384, mean=PIXEL_MEAN, std=PIXEL_STD):
    """(h,w,3) uint8 RGB -> (3,in_h,in_w) float32: optional left-right flip (augmentations.py:40), centre zero pad
    (kitti.py:218-228), ToTensor (/255) and Normalize (transforms.py:15-31; TO_BGR False). The padding is zero BEFORE
    normalisation, i.e. -mean/std in the output."""
    img = img_u8[:, ::-1] if do_flip else img_u8
    h, w, _ = img.shape
    canvas = np.zeros((in_h, in_w, 3), dtype=np.uint8)
    py, px = (in_h - h) // 2, (in_w - w) // 2
    canvas[py:py + h, px:px + w] = img
    x = canvas.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    m = np.asarray(mean, dtype=np.float32).reshape(3, 1, 1)
    s = np.asarray(std, dtype=np.float32).reshape(3, 1, 1)
    return (x - m) / s
