"""oracle/kitti_eval_ref.py (result writer, label parsing, overlaps, matching, AP) against tests/golden/kitti_eval.npz = the
reference's evaluate.py / eval.py / rotate_iou.py / kitti_common.py executed in the build container (oracle/gen_golden.py eval)."""
import os

import numpy as np
import pytest

from oracle import kitti_eval_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "kitti_eval.npz"))
N_IMG = len([k for k in GOLD.files if k.startswith("det_")])


def annos():
    gts = [R.parse_annos(str(GOLD["labels_%d" % i])) for i in range(N_IMG)]
    dts = [R.parse_annos(str(GOLD["txt_%d" % i])) for i in range(N_IMG)]
    return gts, dts


def test_result_files_are_byte_identical():
    for i in range(N_IMG):
        assert R.result_text(GOLD["det_%d" % i]) == str(GOLD["txt_%d" % i]), i
    assert R.result_text(np.zeros((0, 14), np.float32)) == "\n"
    d = R.parse_annos(str(GOLD["txt_0"]))
    assert d["score"].shape == (len(GOLD["det_0"]),) and (d["score"] > 0).all()
    assert len(R.parse_annos("\n")["name"]) == 0                  # an image without detections


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_overlap_matrices_match_reference(metric):
    gts, dts = annos()
    n_pos = 0
    for i in range(N_IMG):
        ref = GOLD["ov%d_%d" % (metric, i)]
        got = R.image_overlaps(dts[i], gts[i], metric)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), (metric, i, float(np.abs(got - ref).max()))
        n_pos += int((ref > 0).sum())
    assert n_pos > 50


@pytest.mark.parametrize("metric", ["R40", "R11"])
def test_official_result_matches_reference(metric):
    gts, dts = annos()
    text, ret = R.official_result(gts, dts, ("Car", "Pedestrian", "Cyclist"), metric)
    keys = [str(k) for k in GOLD["keys_" + metric]]
    assert sorted(ret.keys()) == keys
    np.testing.assert_array_equal(np.array([float(ret[k]) for k in keys]), GOLD["values_" + metric])
    assert text == str(GOLD["result_" + metric])
    assert sum(v > 1.0 for v in GOLD["values_" + metric]) > 30     # the fixture is not degenerate


def test_fixture_has_no_borderline_overlaps():
    """No overlap within 2e-4 of a matching threshold: the float32 round-off of the device's rotated-box arithmetic (<= 1e-4
    on this fixture) cannot flip a match, so AP values must agree to round-off."""
    for m in range(3):
        for i in range(N_IMG):
            o = GOLD["ov%d_%d" % (m, i)]
            for th in (0.25, 0.5, 0.7):
                assert not ((np.abs(o - th) < 2e-4) & (o > 0)).any(), (m, i, th)
