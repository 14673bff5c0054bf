"""oracle/kitti_eval_ref.py -- TEST INFRASTRUCTURE (CPU restatement; never imported by the product path).

numpy restatement of the reference's KITTI result writer and AP evaluation:

  result rows -> text              data/datasets/evaluation/kitti_object_eval_python/evaluate.py:34-62
  label / result parsing           .../kitti_common.py:294-349 (get_label_anno, get_label_annos)
  2D box overlap                   .../eval.py:83-110 (image_box_overlap)
  rotated BEV intersection         .../rotate_iou.py:17-268 (corners, vertex collection, angular sort, fan area)
  3D overlap                       .../eval.py:118-153 (d3_box_overlap[_kernel])
  ignore rules per class/level     .../eval.py:27-80 (clean_data)
  greedy matching                  .../eval.py:156-286 (compute_statistics_jit)
  41 recall sample thresholds      .../eval.py:8-24 (get_thresholds)
  PR accumulation, envelopes, AP   .../eval.py:448-582 (eval_class, get_mAP, get_mAP_R40)
  report text + result dict        .../eval.py:648-741 (get_official_eval_result)

Pinned by tests/golden/kitti_eval.npz: the reference's own modules executed in this container with numba emulated in Python
(oracle/gen_golden.py eval), on 12 generated label files + synthetic detections: result files byte-identical, overlap
matrices and every AP value equal (tests/test_kitti_eval_cpu.py).  The rotated-box arithmetic is float32 like the
reference's kernels; everything else float64.
"""
import math

import numpy as np

ID_TO_NAME = {0: "Car", 1: "Pedestrian", 2: "Cyclist"}                                  # evaluate.py:36-40
EVAL_NAMES = ["car", "pedestrian", "cyclist", "van", "person_sitting", "truck"]            # eval.py:28
MIN_HEIGHT, MAX_OCCLUSION, MAX_TRUNCATION = (40, 25, 25), (0, 1, 2), (0.15, 0.3, 0.5)        # eval.py:29-31
F32 = np.float32


# ---- text formats -------------------------------------------------------------------------------------------------------
def result_text(det):
    """(N,14) detection rows -> the text the reference writes (csv writer, ' ' delimiter, values rounded to 4 decimals as
    float32 and printed through python float repr; the trailing newline stays: the reference's strip step dies on a missing
    `os` import inside a bare try/except)."""
    det = np.asarray(det)
    if len(det) == 0:
        return "\n"
    rows = []
    for p in det:
        p = p.round(4)
        rows.append(" ".join([ID_TO_NAME[int(p[0])], "0", "0"] + [repr(v) for v in p[1:].tolist()]))
    return "\n".join(rows) + "\n"


def parse_annos(text):
    """One label / result file -> dict of arrays (kitti_common.py:294-332). `dimensions` is reordered (h,w,l)->(l,h,w)."""
    lines = text.splitlines(keepends=True)
    rows = [] if (len(lines) == 0 or len(lines[0]) < 15) else [l.strip().split(" ") for l in lines]
    a = dict(name=np.array([r[0] for r in rows]), truncated=np.array([float(r[1]) for r in rows]),
             occluded=np.array([int(r[2]) for r in rows]), alpha=np.array([float(r[3]) for r in rows]),
             bbox=np.array([[float(v) for v in r[4:8]] for r in rows]).reshape(-1, 4),
             dimensions=np.array([[float(v) for v in r[8:11]] for r in rows]).reshape(-1, 3)[:, [2, 0, 1]],
             location=np.array([[float(v) for v in r[11:14]] for r in rows]).reshape(-1, 3),
             rotation_y=np.array([float(r[14]) for r in rows]).reshape(-1))
    a["score"] = np.array([float(r[15]) for r in rows]) if (rows and len(rows[0]) == 16) else np.zeros(len(rows))
    return a


# ---- overlaps -----------------------------------------------------------------------------------------------------------
def bbox_overlap(boxes, query, criterion=-1):
    """(N,4) x (K,4) axis-aligned overlap; criterion -1 IoU, 0 / area(boxes), 1 / area(query) (eval.py:83-110)."""
    N, K = len(boxes), len(query)
    out = np.zeros((N, K), dtype=np.float64)
    for k in range(K):
        q = query[k]
        qa = (q[2] - q[0]) * (q[3] - q[1])
        for n in range(N):
            b = boxes[n]
            iw = min(b[2], q[2]) - max(b[0], q[0])
            ih = min(b[3], q[3]) - max(b[1], q[1])
            if iw > 0 and ih > 0:
                ba = (b[2] - b[0]) * (b[3] - b[1])
                ua = {-1: ba + qa - iw * ih, 0: ba, 1: qa}.get(criterion, 1.0)
                out[n, k] = iw * ih / ua
    return out


def _corners(r):
    """rotate_iou.py:205-228: corners of (cx, cy, dx, dy, angle), clockwise, float32."""
    c, s = math.cos(r[4]), math.sin(r[4])
    xs = np.array([-r[2] / 2, -r[2] / 2, r[2] / 2, r[2] / 2], dtype=F32)
    ys = np.array([-r[3] / 2, r[3] / 2, r[3] / 2, -r[3] / 2], dtype=F32)
    out = np.zeros(8, dtype=F32)
    for i in range(4):
        out[2 * i] = c * xs[i] + s * ys[i] + r[0]
        out[2 * i + 1] = -s * xs[i] + c * ys[i] + r[1]
    return out


def _inside(px, py, q):
    """rotate_iou.py:166-182: projection test against edges AB and AD of quadrilateral q."""
    ab0, ab1, ad0, ad1 = q[2] - q[0], q[3] - q[1], q[6] - q[0], q[7] - q[1]
    ap0, ap1 = px - q[0], py - q[1]
    abab, abap = ab0 * ab0 + ab1 * ab1, ab0 * ap0 + ab1 * ap1
    adad, adap = ad0 * ad0 + ad1 * ad1, ad0 * ap0 + ad1 * ap1
    return abab >= abap and abap >= 0 and adad >= adap and adap >= 0


def _edge_cross(p1, p2, i, j):
    """rotate_iou.py:77-116: proper intersection of edge i of p1 with edge j of p2, or None."""
    A, B = p1[2 * i:2 * i + 2], p1[2 * ((i + 1) % 4):2 * ((i + 1) % 4) + 2]
    C, D = p2[2 * j:2 * j + 2], p2[2 * ((j + 1) % 4):2 * ((j + 1) % 4) + 2]
    BA0, BA1, DA0, CA0, DA1, CA1 = B[0] - A[0], B[1] - A[1], D[0] - A[0], C[0] - A[0], D[1] - A[1], C[1] - A[1]
    acd = DA1 * CA0 > CA1 * DA0
    bcd = (D[1] - B[1]) * (C[0] - B[0]) > (C[1] - B[1]) * (D[0] - B[0])
    if acd == bcd:
        return None
    if (CA1 * BA0 > BA1 * CA0) == (DA1 * BA0 > BA1 * DA0):
        return None
    DC0, DC1 = D[0] - C[0], D[1] - C[1]
    ABBA, CDDC = A[0] * B[1] - B[0] * A[1], C[0] * D[1] - D[0] * C[1]
    DH = BA1 * DC0 - BA0 * DC1
    return (ABBA * DC0 - BA0 * CDDC) / DH, (ABBA * DC1 - BA1 * CDDC) / DH


def rotated_intersection(r1, r2):
    """Area of the intersection of two rotated rectangles (rotate_iou.py:231-247): collect corners of each inside the other
    and the edge crossings, sort them around their centroid by the reference's monotone angle key, sum the fan triangles."""
    p1, p2 = _corners(r1), _corners(r2)
    pts = np.zeros(16, dtype=F32)
    n = 0
    for i in range(4):                                            # rotate_iou.py:185-202
        if _inside(p1[2 * i], p1[2 * i + 1], p2):
            pts[2 * n], pts[2 * n + 1] = p1[2 * i], p1[2 * i + 1]; n += 1
        if _inside(p2[2 * i], p2[2 * i + 1], p1):
            pts[2 * n], pts[2 * n + 1] = p2[2 * i], p2[2 * i + 1]; n += 1
    for i in range(4):
        for j in range(4):
            x = _edge_cross(p1, p2, i, j)
            if x is not None:
                pts[2 * n], pts[2 * n + 1] = x; n += 1
    if n > 0:                                                     # rotate_iou.py:33-69
        cen = np.zeros(2, dtype=F32)
        for i in range(n):
            cen[0] += pts[2 * i]; cen[1] += pts[2 * i + 1]
        cen[0] /= n; cen[1] /= n
        key = np.zeros(16, dtype=F32)
        v = np.zeros(2, dtype=F32)
        for i in range(n):
            v[0], v[1] = pts[2 * i] - cen[0], pts[2 * i + 1] - cen[1]
            d = math.sqrt(v[0] * v[0] + v[1] * v[1])
            v[0], v[1] = v[0] / d, v[1] / d
            if v[1] < 0:
                v[0] = -2 - v[0]
            key[i] = v[0]
        for i in range(1, n):                                     # insertion sort, ascending key
            if key[i - 1] > key[i]:
                t, tx, ty, j = key[i], pts[2 * i], pts[2 * i + 1], i
                while j > 0 and key[j - 1] > t:
                    key[j], pts[2 * j], pts[2 * j + 1] = key[j - 1], pts[2 * j - 2], pts[2 * j - 1]
                    j -= 1
                key[j], pts[2 * j], pts[2 * j + 1] = t, tx, ty
    area = 0.0                                                    # rotate_iou.py:17-30
    for i in range(n - 2):
        a, b, c = pts[:2], pts[2 * i + 2:2 * i + 4], pts[2 * i + 4:2 * i + 6]
        area += abs(((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0)
    return area


def rotated_overlap(boxes, query, criterion=-1):
    """(N,5) x (K,5) -> (N,K) float32 (rotate_iou.py:250-333). The kernel evaluates (query, box) in that order, so
    criterion 0 normalises by the QUERY box's area and 1 by the box's; 2 returns the raw intersection area."""
    boxes, query = np.asarray(boxes).astype(F32), np.asarray(query).astype(F32)
    out = np.zeros((len(boxes), len(query)), dtype=F32)
    for n in range(len(boxes)):
        for k in range(len(query)):
            r1, r2 = query[k], boxes[n]
            a1, a2 = r1[2] * r1[3], r2[2] * r2[3]
            inter = rotated_intersection(r1, r2)
            out[n, k] = inter / (a1 + a2 - inter) if criterion == -1 else inter / a1 if criterion == 0 else \
                inter / a2 if criterion == 1 else inter
    return out


def box3d_overlap(boxes, query, criterion=-1):
    """(N,7) x (K,7) [x,y,z,l,h,w,ry] camera-frame 3D IoU: BEV intersection x height overlap (eval.py:118-153)."""
    inc = rotated_overlap(boxes[:, [0, 2, 3, 5, 6]], query[:, [0, 2, 3, 5, 6]], 2)
    for i in range(len(boxes)):
        for j in range(len(query)):
            if inc[i, j] > 0:
                ih = min(boxes[i, 1], query[j, 1]) - max(boxes[i, 1] - boxes[i, 4], query[j, 1] - query[j, 4])
                if ih > 0:
                    v1, v2 = boxes[i, 3] * boxes[i, 4] * boxes[i, 5], query[j, 3] * query[j, 4] * query[j, 5]
                    vol = ih * inc[i, j]
                    ua = {-1: v1 + v2 - vol, 0: v1, 1: v2}.get(criterion, vol)
                    inc[i, j] = vol / ua
                else:
                    inc[i, j] = 0.0
    return inc


def image_overlaps(dt, gt, metric):
    """(num_dt, num_gt) overlap matrix of one image for metric 0 bbox / 1 bev / 2 3d (eval.py:329-401; the reference computes
    all pairs of a 50-image part at once and slices the diagonal blocks -- same values)."""
    if metric == 0:
        return bbox_overlap(dt["bbox"], gt["bbox"])
    if metric == 1:
        mk = lambda a: np.concatenate([a["location"][:, [0, 2]], a["dimensions"][:, [0, 2]], a["rotation_y"][:, None]], axis=1)
        return rotated_overlap(mk(dt), mk(gt)).astype(np.float64)
    mk = lambda a: np.concatenate([a["location"], a["dimensions"], a["rotation_y"][:, None]], axis=1)
    return box3d_overlap(mk(dt), mk(gt)).astype(np.float64)


# ---- matching and AP ----------------------------------------------------------------------------------------------------
def ignore_flags(gt, dt, cls, level):
    """eval.py:27-80: per ground truth 0 = counts, 1 = matched without credit, -1 = other class; per detection likewise;
    plus the DontCare boxes and the number of counting ground truths."""
    name = EVAL_NAMES[cls]
    ig, idt, dc, nvalid = [], [], [], 0
    for i in range(len(gt["name"])):
        g = gt["name"][i].lower()
        kind = 1 if g == name else 0 if (name == "pedestrian" and g == "person_sitting") or (name == "car" and g == "van") else -1
        hard = (gt["occluded"][i] > MAX_OCCLUSION[level] or gt["truncated"][i] > MAX_TRUNCATION[level]
                or gt["bbox"][i, 3] - gt["bbox"][i, 1] <= MIN_HEIGHT[level])
        if kind == 1 and not hard:
            ig.append(0); nvalid += 1
        elif kind == 0 or (hard and kind == 1):
            ig.append(1)
        else:
            ig.append(-1)
        if gt["name"][i] == "DontCare":
            dc.append(gt["bbox"][i])
    for i in range(len(dt["name"])):
        if abs(dt["bbox"][i, 3] - dt["bbox"][i, 1]) < MIN_HEIGHT[level]:
            idt.append(1)
        else:
            idt.append(0 if dt["name"][i].lower() == name else -1)
    dc = np.stack(dc, 0).astype(np.float64) if dc else np.zeros((0, 4))
    return nvalid, np.array(ig, dtype=np.int64), np.array(idt, dtype=np.int64), dc


def match(ov, gt_alpha, dt_bbox, dt_alpha, dt_score, ig, idt, dc, metric, min_ov, thresh=0.0, count_fp=False, aos=False):
    """Greedy assignment of detections to ground truths in label order (eval.py:156-286). Returns tp, fp, fn, the summed
    orientation similarity (-1 if undefined) and the scores of the true positives."""
    nd, ng = len(dt_score), len(ig)
    taken = [False] * nd
    below = [count_fp and dt_score[j] < thresh for j in range(nd)]
    NONE = -10000000
    tp = fp = fn = 0
    sim = 0
    tp_scores, deltas = [], []
    for i in range(ng):
        if ig[i] == -1:
            continue
        best, val, max_ov, from_ignored = -1, NONE, 0, False
        for j in range(nd):
            if idt[j] == -1 or taken[j] or below[j]:
                continue
            o = ov[j, i]
            if not count_fp and o > min_ov and dt_score[j] > val:
                best, val = j, dt_score[j]
            elif count_fp and o > min_ov and (o > max_ov or from_ignored) and idt[j] == 0:
                max_ov, best, val, from_ignored = o, j, 1, False
            elif count_fp and o > min_ov and val == NONE and idt[j] == 1:
                best, val, from_ignored = j, 1, True
        if val == NONE and ig[i] == 0:
            fn += 1
        elif val != NONE and (ig[i] == 1 or idt[best] == 1):
            taken[best] = True
        elif val != NONE:
            tp += 1
            tp_scores.append(dt_score[best])
            if aos:
                deltas.append(gt_alpha[i] - dt_alpha[best])
            taken[best] = True
    if count_fp:
        fp = sum(1 for j in range(nd) if not (taken[j] or idt[j] == -1 or idt[j] == 1 or below[j]))
        stuff = 0
        if metric == 0:
            odc = bbox_overlap(dt_bbox, dc, 0)
            for i in range(len(dc)):
                for j in range(nd):
                    if taken[j] or idt[j] in (-1, 1) or below[j]:
                        continue
                    if odc[j, i] > min_ov:
                        taken[j] = True; stuff += 1
        fp -= stuff
        if aos:                                                  # fp zeros, then (1 + cos(delta)) / 2 per true positive
            terms = np.zeros(fp + len(deltas))
            for i, d in enumerate(deltas):
                terms[fp + i] = (1.0 + np.cos(d)) / 2.0
            sim = np.sum(terms) if (tp > 0 or fp > 0) else -1
    return tp, fp, fn, sim, np.array(tp_scores)


def sample_thresholds(scores, num_gt, npts=41):
    """Scores at which recall first reaches k/40 (eval.py:8-24)."""
    scores = np.sort(np.asarray(scores, dtype=np.float64))[::-1]
    cur, out = 0, []
    for i, s in enumerate(scores):
        l = (i + 1) / num_gt
        r = (i + 2) / num_gt if i < len(scores) - 1 else l
        if (r - cur) < (cur - l) and i < len(scores) - 1:
            continue
        out.append(s)
        cur += 1 / (npts - 1.0)
    return out


def precision_curves(gts, dts, classes, metric, min_overlaps, aos=False, overlaps=None, tables=None):
    """eval.py:448-570 -> precision, recall, aos arrays [class, level, overlap set, 41].  `overlaps`: per-image (num_dt, num_gt)
    matrices to use instead of computing them; `tables`: dict that receives {(m, level, k): (thresholds, pr)}."""
    ovs = overlaps if overlaps is not None else [image_overlaps(d, g, metric) for g, d in zip(gts, dts)]
    nc, nk = len(classes), len(min_overlaps)
    prec, rec, ori = (np.zeros((nc, 3, nk, 41)) for _ in range(3))
    for m, cls in enumerate(classes):
        for level in range(3):
            flags = [ignore_flags(g, d, cls, level) for g, d in zip(gts, dts)]
            total_valid = sum(f[0] for f in flags)
            for k in range(nk):
                mo = min_overlaps[k, metric, m]
                args = [(ovs[i], gts[i]["alpha"], dts[i]["bbox"], dts[i]["alpha"], dts[i]["score"], flags[i][1], flags[i][2], flags[i][3])
                        for i in range(len(gts))]
                tps = np.concatenate([match(*a, metric, mo)[4] for a in args]) if args else np.zeros(0)
                ths = sample_thresholds(tps, total_valid)
                pr = np.zeros((len(ths), 4))
                for a in args:
                    for t, th in enumerate(ths):
                        tp, fp, fn, sim, _ = match(*a, metric, mo, thresh=th, count_fp=True, aos=aos)
                        pr[t, :3] += (tp, fp, fn)
                        if sim != -1:
                            pr[t, 3] += sim
                if tables is not None:
                    tables[(m, level, k)] = (list(ths), pr.copy(), total_valid)
                with np.errstate(divide="ignore", invalid="ignore"):
                    for t in range(len(ths)):
                        rec[m, level, k, t] = pr[t, 0] / (pr[t, 0] + pr[t, 2])
                        prec[m, level, k, t] = pr[t, 0] / (pr[t, 0] + pr[t, 1])
                        if aos:
                            ori[m, level, k, t] = pr[t, 3] / (pr[t, 0] + pr[t, 1])
                for t in range(len(ths)):                         # right-to-left running maximum
                    prec[m, level, k, t] = np.max(prec[m, level, k, t:])
                    rec[m, level, k, t] = np.max(rec[m, level, k, t:])
                    if aos:
                        ori[m, level, k, t] = np.max(ori[m, level, k, t:])
    return prec, rec, ori


def average_precision(curve, metric="R40"):
    idx = range(1, curve.shape[-1]) if metric == "R40" else range(0, curve.shape[-1], 4)   # eval.py:585-597: 40 / 11 points
    total = 0
    for i in idx:                                                 # left-to-right sum, like the reference
        total = total + curve[..., i]
    return total / (40 if metric == "R40" else 11) * 100


def official_result(gts, dts, classes=("Car", "Pedestrian", "Cyclist"), metric="R40"):
    """eval.py:648-741 -> (report text, result dict). `classes`: names or indices into Car/Pedestrian/Cyclist/Van/..."""
    names = {0: "Car", 1: "Pedestrian", 2: "Cyclist", 3: "Van", 4: "Person_sitting", 5: "Truck"}
    idx = {v: k for k, v in names.items()}
    classes = [idx[c] if isinstance(c, str) else c for c in classes]
    strict = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.7]] * 3)
    loose = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5]])
    mo = np.stack([strict, loose], 0)[:, :, classes]
    aos = False
    for d in dts:
        if d["alpha"].shape[0] != 0:
            aos = d["alpha"][0] != -10
            break
    p0, _, o0 = precision_curves(gts, dts, classes, 0, mo, aos)
    ap = {"bbox": average_precision(p0, metric), "aos": average_precision(o0, metric) if aos else None,
          "bev": average_precision(precision_curves(gts, dts, classes, 1, mo)[0], metric),
          "3d": average_precision(precision_curves(gts, dts, classes, 2, mo)[0], metric)}
    text, ret = "", {}
    for j, c in enumerate(classes):
        n = names[c]
        for i in range(2):
            text += "{} AP@{:.2f}, {:.2f}, {:.2f}:\n".format(n, *mo[i, :, j])
            text += "bbox AP:{:.4f}, {:.4f}, {:.4f}\n".format(*ap["bbox"][j, :, i])
            text += "bev  AP:{:.4f}, {:.4f}, {:.4f}\n".format(*ap["bev"][j, :, i])
            text += "3d   AP:{:.4f}, {:.4f}, {:.4f}\n".format(*ap["3d"][j, :, i])
            if aos:
                text += "aos  AP:{:.2f}, {:.2f}, {:.2f}\n".format(*ap["aos"][j, :, i])
                if i == 0:
                    for l, lv in enumerate(("easy", "moderate", "hard")):
                        ret["%s_aos/%s" % (n, lv)] = ap["aos"][j, l, 0]
            for l, lv in enumerate(("easy", "moderate", "hard")):
                ret["{}_3d_{:.2f}/{}".format(n, mo[i, 1, j], lv)] = ap["3d"][j, l, i]
                ret["{}_bev_{:.2f}/{}".format(n, mo[i, 2, j], lv)] = ap["bev"][j, l, i]
                ret["{}_image/{}".format(n, lv)] = ap["bbox"][j, l, 0]
    return text, ret
