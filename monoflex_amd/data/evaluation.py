"""KITTI result files and AP evaluation (reference data/datasets/evaluation/__init__.py:7-34,
kitti_object_eval_python/{evaluate,eval,rotate_iou,kitti_common}.py).

Same entry points as the reference -- `generate_kitti_3d_detection(prediction, predict_txt)` and
`evaluate_python(label_path, result_path, label_split_file, current_class, metric) -> (report text, result dict)` -- with
the work split differently: the host parses text and formats the report; the overlap matrices (2D, rotated BEV, 3D), the
per-class / per-difficulty ignore rules, the greedy matching at every one of the <= 41 score thresholds and the PR
accumulation run on the GPU (C ABI group 5, csrc/kitti_eval.hip), all images and all 54 (class, difficulty, metric, overlap
set) combinations at once.  The reference does this with numba CPU loops plus a numba.cuda kernel for the rotated IoU."""
import ctypes
import os
import re

import numpy as np
import torch

from .. import lib as L
from .encode import _upload

ID_TYPE_CONVERSION = {0: "Car", 1: "Pedestrian", 2: "Cyclist"}                             # evaluate.py:36-40
NAME_CODES = {"car": 0, "pedestrian": 1, "cyclist": 2, "van": 3, "person_sitting": 4, "truck": 5}
CLASS_TO_NAME = {0: "Car", 1: "Pedestrian", 2: "Cyclist", 3: "Van", 4: "Person_sitting", 5: "Truck"}
CODE_DONTCARE, CODE_OTHER, REC, PTS, MAX_DET = 6, 7, 16, 41, 64
LEVELS = ("easy", "moderate", "hard")


# ---- text ------------------------------------------------------------------------------------------------------------------
def generate_kitti_3d_detection(prediction, predict_txt):
    """(N,14) rows [cls, alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry, score] -> one KITTI result file: `Type 0 0 alpha ...`,
    values rounded to 4 decimals in float32 and written as the resulting python floats; an empty prediction gives one empty
    line (evaluate.py:34-52)."""
    rows = prediction.detach().cpu().numpy() if isinstance(prediction, torch.Tensor) else np.asarray(prediction)
    lines = []
    for p in rows.reshape(-1, 14) if len(rows) else []:
        p = p.round(4)
        lines.append(" ".join([ID_TYPE_CONVERSION[int(p[0])], "0", "0"] + [repr(v) for v in p[1:].tolist()]))
    with open(predict_txt, "w", newline="") as f:
        f.write("\n".join(lines) + "\n")


def parse_label_text(text):
    """One label / result file -> (n,16) records [code, truncated, occluded, alpha, bbox(4), l, h, w, x, y, z, ry, score]
    (kitti_common.py:294-332: a file whose first line is shorter than 15 characters is empty; scores are read when the first
    line has 16 fields). Name codes follow the evaluator's lower-cased comparisons; `DontCare` is matched exactly."""
    lines = text.splitlines(keepends=True)
    if len(lines) == 0 or len(lines[0]) < 15:
        return np.zeros((0, REC), dtype=np.float64)
    rows = [l.strip().split(" ") for l in lines]
    has_score = len(rows[0]) == 16
    out = np.zeros((len(rows), REC), dtype=np.float64)
    for i, r in enumerate(rows):
        code = CODE_DONTCARE if r[0] == "DontCare" else NAME_CODES.get(r[0].lower(), CODE_OTHER)
        v = [float(x) for x in r[1:15]]
        h, w, l = v[7], v[8], v[9]
        out[i] = [code, v[0], int(r[2]), v[2], v[3], v[4], v[5], v[6], l, h, w, v[10], v[11], v[12], v[13],
                  float(r[15]) if has_score else 0.0]
    return out


def read_label_folder(folder, image_ids=None):
    """kitti_common.py:334-349: records of `<folder>/%06d.txt` for the given ids (default: every six-digit file, sorted)."""
    if image_ids is None:
        image_ids = sorted(int(f[:-4]) for f in os.listdir(folder) if re.match(r"^\d{6}.txt$", f))
    out = []
    for idx in image_ids:
        with open(os.path.join(folder, "%06d.txt" % idx)) as f:
            out.append(parse_label_text(f.read()))
    return out


# ---- device evaluation -------------------------------------------------------------------------------------------------------
def pack_eval_inputs(gts, dts, classes, min_overlaps):
    """Ragged record lists -> the flat input arrays of mfx_kitti_eval_desc, their sizes, and the orientation flag."""
    if len(gts) != len(dts):
        raise ValueError("need one detection record array per ground-truth image")
    ng = np.array([len(g) for g in gts], dtype=np.int64)
    nd = np.array([len(d) for d in dts], dtype=np.int64)
    if nd.max(initial=0) > MAX_DET:
        raise ValueError("at most %d detections per image (got %d)" % (MAX_DET, nd.max()))
    aos = False
    for d in dts:                                                   # eval.py:676-681: alpha == -10 marks "no orientation"
        if len(d):
            aos = bool(d[0, 3] != -10)
            break
    arrays = dict(gt=np.concatenate(list(gts) + [np.zeros((0, REC))]), dt=np.concatenate(list(dts) + [np.zeros((0, REC))]),
                  gt_off=np.concatenate([[0], np.cumsum(ng)]).astype(np.int32), dt_off=np.concatenate([[0], np.cumsum(nd)]).astype(np.int32),
                  pair_off=np.concatenate([[0], np.cumsum(ng * nd)]).astype(np.int64), classes=np.asarray(classes, dtype=np.int32),
                  min_overlaps=np.ascontiguousarray(min_overlaps, dtype=np.float64))
    sizes = dict(B=len(gts), n_gt=int(ng.sum()), n_dt=int(nd.sum()), n_pairs=int((ng * nd).sum()), num_classes=len(classes),
                 num_k=int(arrays["min_overlaps"].shape[0]))
    return arrays, sizes, aos


def pr_table(gts, dts, classes, min_overlaps, device="cuda"):
    """[ (n_i,16) ] ground truths and detections, class codes, min_overlaps (num_k, 3, num_classes) ->
    (pr (nc,3,3,nk,41,4) float64, num_thresholds (nc,3,3,nk), overlaps (3,n_pairs), pair offsets, compute_aos)."""
    if torch.device(device).type != "cuda":
        raise RuntimeError("the KITTI evaluator runs on the GPU (HIP kernels); there is no CPU fallback")
    arrays, sz, aos = pack_eval_inputs(gts, dts, classes, min_overlaps)
    B, n_gt, n_dt, n_pairs, nc, nk = sz["B"], sz["n_gt"], sz["n_dt"], sz["n_pairs"], sz["num_classes"], sz["num_k"]
    n_comb = nc * 9 * nk
    lib = L.load()
    dev = _upload({k: v for k, v in arrays.items() if v.size}, device)
    f64 = dict(dtype=torch.float64, device=device)
    overlaps = torch.empty((3, max(n_pairs, 1)), **f64)
    tp_scores = torch.empty((n_comb, max(n_gt, 1)), **f64)
    thresholds = torch.empty((n_comb, PTS), **f64)
    pr = torch.empty((n_comb, PTS, 4), **f64)
    num_valid = torch.empty((nc, 3), dtype=torch.int32, device=device)
    num_th = torch.empty(n_comb, dtype=torch.int32, device=device)
    d = L.KittiEvalDesc()
    for k in arrays:
        setattr(d, k, dev[k].data_ptr() if k in dev else None)
    d.overlaps, d.tp_scores, d.thresholds, d.pr = overlaps.data_ptr(), tp_scores.data_ptr(), thresholds.data_ptr(), pr.data_ptr()
    d.num_valid_gt, d.num_thresholds = num_valid.data_ptr(), num_th.data_ptr()
    d.B, d.n_gt, d.n_dt, d.num_classes, d.num_k, d.compute_aos, d.n_pairs = B, n_gt, n_dt, nc, nk, int(aos), n_pairs
    with torch.cuda.device(device):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(lib.mfx_kitti_eval_overlaps(ctypes.byref(d), st), "mfx_kitti_eval_overlaps")
        L.check(lib.mfx_kitti_eval_match_pass1(ctypes.byref(d), st), "mfx_kitti_eval_match_pass1")
        ordered = torch.sort(tp_scores, dim=1, descending=True).values.contiguous()
        L.check(lib.mfx_kitti_eval_thresholds(ctypes.byref(d), ordered.data_ptr(), st), "mfx_kitti_eval_thresholds")
        L.check(lib.mfx_kitti_eval_match_pass2(ctypes.byref(d), st), "mfx_kitti_eval_match_pass2")
    shape = (nc, 3, 3, nk)
    return (pr.cpu().numpy().reshape(shape + (PTS, 4)), num_th.cpu().numpy().reshape(shape), overlaps[:, :n_pairs].cpu().numpy(),
            arrays["pair_off"], aos)


def _curves(pr, num_th):
    """tp/fp/fn/similarity table -> precision, orientation curves with the right-to-left running maximum (eval.py:551-565)."""
    prec, ori = np.zeros(pr.shape[:-1]), np.zeros(pr.shape[:-1])
    with np.errstate(divide="ignore", invalid="ignore"):
        for idx in np.ndindex(*num_th.shape):
            n = int(num_th[idx])
            p = pr[idx][:n]
            prec[idx][:n] = p[:, 0] / (p[:, 0] + p[:, 1])
            ori[idx][:n] = p[:, 3] / (p[:, 0] + p[:, 1])
            for t in range(n):                                      # maximum over the whole tail, zeros past n included
                prec[idx][t] = np.max(prec[idx][t:])
                ori[idx][t] = np.max(ori[idx][t:])
    return prec, ori


def _ap(curve, metric):
    idx = range(1, PTS) if metric == "R40" else range(0, PTS, 4)     # eval.py:585-597
    total = 0
    for i in idx:
        total = total + curve[..., i]
    return total / (40 if metric == "R40" else 11) * 100


def get_official_eval_result(gts, dts, current_classes, metric="R40", device="cuda"):
    """eval.py:648-741: (report text, {'Car_3d_0.70/moderate': AP, ...}) from record lists."""
    name_to_class = {v: k for k, v in CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    classes = [name_to_class[c] if isinstance(c, str) else int(c) for c in current_classes]
    if metric not in ("R40", "R11"):
        raise ValueError(metric)
    strict = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.7]] * 3)
    loose = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5]])
    mo = np.stack([strict, loose], axis=0)[:, :, classes]
    pr, num_th, _, _, aos = pr_table(gts, dts, classes, mo, device)
    prec, ori = _curves(pr, num_th)
    ap = _ap(prec, metric)                                          # [class, level, metric, k]
    ap_aos = _ap(ori[:, :, 0], metric) if aos else None
    text, ret = "", {}
    for j, c in enumerate(classes):
        n = CLASS_TO_NAME[c]
        for i in range(mo.shape[0]):
            text += "{} AP@{:.2f}, {:.2f}, {:.2f}:\n".format(n, *mo[i, :, j])
            for tag, m in (("bbox", 0), ("bev ", 1), ("3d  ", 2)):
                text += "{} AP:{:.4f}, {:.4f}, {:.4f}\n".format(tag, *ap[j, :, m, i])
            if aos:
                text += "aos  AP:{:.2f}, {:.2f}, {:.2f}\n".format(*ap_aos[j, :, i])
                if i == 0:
                    for l, lv in enumerate(LEVELS):
                        ret["%s_aos/%s" % (n, lv)] = ap_aos[j, l, 0]
            for l, lv in enumerate(LEVELS):                         # key naming as in the reference (eval.py:727-737)
                ret["{}_3d_{:.2f}/{}".format(n, mo[i, 1, j], lv)] = ap[j, l, 2, i]
                ret["{}_bev_{:.2f}/{}".format(n, mo[i, 2, j], lv)] = ap[j, l, 1, i]
                ret["{}_image/{}".format(n, lv)] = ap[j, l, 0, 0]
    return text, ret


def evaluate_python(label_path, result_path, label_split_file, current_class, metric="R40", score_thresh=-1, device="cuda"):
    """data/datasets/evaluation/__init__.py:31-34 -> evaluate.py:17-32. `score_thresh` > 0 drops detections scoring below it
    before the evaluation (kitti_common.py:191-202)."""
    with open(label_split_file) as f:
        ids = [int(line) for line in f.readlines()]
    dts = read_label_folder(result_path)
    if score_thresh > 0:
        dts = [d[d[:, 15] >= score_thresh] for d in dts]
    gts = read_label_folder(label_path, ids)
    return get_official_eval_result(gts, dts, current_class, metric=metric, device=device)
