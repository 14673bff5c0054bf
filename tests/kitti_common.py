"""Shared by the CPU (host shim) and GPU tests of the KITTI encoder: golden access, oracle batches, comparison rules."""
import os

import numpy as np

from oracle import kitti_encode_ref as K
from monoflex_amd import synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "kitti_encode.npz"))
NAMES = [str(n) for n in GOLD["names"]]
EXACT = ["cls_ids", "target_centers", "reg_mask", "trunc_mask", "reg_weight", "keypoints_depth_mask", "pad_size", "edge_len",
         "edge_indices", "occlusions", "truncations", "gt_bboxes", "dimensions", "locations", "rotys"]
CLOSE = ["hm", "keypoints", "offset_3D", "2d_bboxes", "alphas", "orientations"]


def golden_sample(name):
    w, h, flip, iseed = (int(v) for v in GOLD[name + "_meta"])
    text = str(GOLD[name + "_labels"])
    return (text.split("\n") if text else []), w, h, bool(flip), iseed


def fuzz_sample(seed):
    """Seeded label set in the style of the golden cases (sizes, flips and object counts vary with the seed)."""
    rs = np.random.RandomState(seed)
    w, h = [(1242, 375), (1224, 370), (1238, 374), (1241, 376), (1280, 384), (1000, 300)][rs.randint(6)]
    return S.synthetic_kitti_labels(seed, w, h, int(rs.randint(0, 41))), w, h, bool(rs.randint(2))


def oracle_fields(lines, w, h, flip):
    try:
        return K.encode_sample(lines, S.KITTI_P2, w, h, do_flip=flip)
    except (TypeError, ValueError, AssertionError, IndexError):
        return None                                              # inputs the reference itself fails on


def compare_fields(got, ref, tag, exact_close=False):
    """got/ref: {field: array} for one sample. Integer-valued fields identical; float fields to float32 round-off."""
    for k in EXACT:
        r = np.asarray(ref[k])
        assert np.array_equal(np.asarray(got[k]).astype(r.dtype).reshape(r.shape), r), (tag, k)
    for k in CLOSE:
        r = np.asarray(ref[k])
        g = np.asarray(got[k]).reshape(r.shape)
        if exact_close:
            assert np.array_equal(g, r), (tag, k, float(np.abs(g - r).max()))
        else:
            np.testing.assert_allclose(g, r, rtol=1e-6, atol=1e-6, err_msg="%s %s" % (tag, k))
    assert np.array_equal(np.asarray(got["hm"]).reshape(ref["hm"].shape) == 1.0, np.asarray(ref["hm"]) == 1.0), (tag, "hm peaks")
