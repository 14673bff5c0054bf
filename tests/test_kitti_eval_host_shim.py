"""CPU check of the evaluator's device functions (monoflex_amd/csrc/kitti_eval_math.h compiled for the host by the test-only
tests/shim/kitti_eval_host.cpp) and of the host half of monoflex_amd/data/evaluation.py: result files byte-identical to the
reference's, parsing, packing, curve / AP arithmetic and the report text against the reference goldens."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from monoflex_amd import lib as L
from monoflex_amd.data import evaluation as EV
from oracle import kitti_eval_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "kitti_eval.npz"))
N_IMG = len([k for k in GOLD.files if k.startswith("det_")])


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("shim") / "libkitti_eval_shim.so")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                        os.path.join(ROOT, "tests", "shim", "kitti_eval_host.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = ctypes.CDLL(so)
    lib.shim_kitti_eval.argtypes = [ctypes.POINTER(L.KittiEvalDesc)]
    lib.shim_kitti_eval.restype = None
    return lib


def records():
    gts = [EV.parse_label_text(str(GOLD["labels_%d" % i])) for i in range(N_IMG)]
    dts = [EV.parse_label_text(str(GOLD["txt_%d" % i])) for i in range(N_IMG)]
    return gts, dts


def shim_pr_table(shim, gts, dts, classes, mo):
    arrays, sz, aos = EV.pack_eval_inputs(gts, dts, classes, mo)
    n_comb = sz["num_classes"] * 9 * sz["num_k"]
    out = dict(overlaps=np.zeros((3, max(sz["n_pairs"], 1))), tp_scores=np.zeros((n_comb, max(sz["n_gt"], 1))),
               thresholds=np.zeros((n_comb, 41)), pr=np.full((n_comb, 41, 4), 7.0), num_valid_gt=np.zeros((sz["num_classes"], 3), np.int32),
               num_thresholds=np.zeros(n_comb, np.int32))
    d = L.KittiEvalDesc()
    for k, a in list(arrays.items()) + list(out.items()):
        setattr(d, k, a.ctypes.data)
    for k, v in sz.items():
        setattr(d, k, v)
    d.compute_aos = int(aos)
    shim.shim_kitti_eval(ctypes.byref(d))
    shape = (sz["num_classes"], 3, 3, sz["num_k"])
    return out["pr"].reshape(shape + (41, 4)), out["num_thresholds"].reshape(shape), out["overlaps"][:, :sz["n_pairs"]], arrays["pair_off"], aos


def test_result_writer_is_byte_identical(tmp_path):
    for i in range(N_IMG):
        f = tmp_path / ("%06d.txt" % i)
        EV.generate_kitti_3d_detection(torch.from_numpy(GOLD["det_%d" % i]), str(f))
        assert f.read_text() == str(GOLD["txt_%d" % i]), i
    got = EV.read_label_folder(str(tmp_path))
    assert len(got) == N_IMG and got[3].shape == (0, 16)
    assert np.allclose(got[0][:, 15], GOLD["det_0"][:, 13].round(4))


def test_parsing_matches_oracle():
    text = str(GOLD["labels_1"])
    rec, a = EV.parse_label_text(text), R.parse_annos(text)
    assert rec.shape == (len(a["name"]), 16)
    codes = {"car": 0, "pedestrian": 1, "cyclist": 2, "van": 3, "person_sitting": 4, "truck": 5}
    for r, name in zip(rec, a["name"]):
        assert r[0] == (6 if name == "DontCare" else codes.get(name.lower(), 7))
    assert np.array_equal(rec[:, 1], a["truncated"]) and np.array_equal(rec[:, 2], a["occluded"]) and np.array_equal(rec[:, 3], a["alpha"])
    assert np.array_equal(rec[:, 4:8], a["bbox"]) and np.array_equal(rec[:, 8:11], a["dimensions"])
    assert np.array_equal(rec[:, 11:14], a["location"]) and np.array_equal(rec[:, 14], a["rotation_y"]) and (rec[:, 15] == 0).all()
    assert EV.parse_label_text("\n").shape == (0, 16) and EV.parse_label_text("").shape == (0, 16)
    with pytest.raises(ValueError):
        EV.pack_eval_inputs([np.zeros((0, 16))], [np.zeros((65, 16))], [0], np.zeros((1, 3, 1)))


def test_shim_overlaps_match_reference(shim):
    gts, dts = records()
    _, _, ov, pair_off, _ = shim_pr_table(shim, gts, dts, [0, 1, 2], np.full((2, 3, 3), 0.5))
    for m in range(3):
        for i in range(N_IMG):
            ref = GOLD["ov%d_%d" % (m, i)]
            got = ov[m, pair_off[i]:pair_off[i + 1]].reshape(ref.shape)
            # rotated boxes: float32 arithmetic with cosf/sinf/sqrtf instead of the double-precision libm calls of the emulated
            # reference run -> near-parallel edge crossings move by a few 1e-5
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-4 if m else 0, err_msg="metric %d image %d" % (m, i))


@pytest.mark.parametrize("metric", ["R40", "R11"])
def test_shim_result_matches_reference(shim, metric, monkeypatch):
    gts, dts = records()
    monkeypatch.setattr(EV, "pr_table", lambda g, d, c, mo, device="cuda": shim_pr_table(shim, g, d, c, mo))
    text, ret = EV.get_official_eval_result(gts, dts, ["Car", "Pedestrian", "Cyclist"], metric=metric)
    keys = [str(k) for k in GOLD["keys_" + metric]]
    assert sorted(ret.keys()) == keys
    np.testing.assert_allclose(np.array([float(ret[k]) for k in keys]), GOLD["values_" + metric], rtol=1e-12, atol=1e-12)
    assert text == str(GOLD["result_" + metric])


def test_shim_subset_of_classes_and_no_orientation(shim, monkeypatch):
    gts, dts = records()
    monkeypatch.setattr(EV, "pr_table", lambda g, d, c, mo, device="cuda": shim_pr_table(shim, g, d, c, mo))
    _, car = EV.get_official_eval_result(gts, dts, "Car", metric="R40")
    i = [str(k) for k in GOLD["keys_R40"]].index("Car_3d_0.70/moderate")
    assert abs(car["Car_3d_0.70/moderate"] - GOLD["values_R40"][i]) < 1e-9 and not any(k.startswith("Ped") for k in car)
    for d in dts:
        d[:, 3] = -10
    text, ret = EV.get_official_eval_result(gts, dts, [0], metric="R40")
    assert "aos" not in text and not any("aos" in k for k in ret)
    ga, da = [R.parse_annos(str(GOLD["labels_%d" % i])) for i in range(N_IMG)], [R.parse_annos(str(GOLD["txt_%d" % i])) for i in range(N_IMG)]
    for a in da:
        a["alpha"][:] = -10
    otext, oret = R.official_result(ga, da, (0,), "R40")
    assert text == otext and all(abs(ret[k] - oret[k]) < 1e-9 for k in oret)


def _random_records(rs, n, with_score):
    """Random box records that exercise every ignore rule: all name codes, heights around the 25/40 px limits, occlusion
    0..3, truncation around 0.15/0.3/0.5, scores on a coarse grid (ties)."""
    r = np.zeros((n, 16))
    r[:, 0] = rs.choice([0, 0, 0, 1, 1, 2, 2, 3, 4, 5, 6, 7], n) if not with_score else rs.choice([0, 0, 1, 2], n)
    r[:, 1] = rs.choice([0.0, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5, 0.6], n)
    r[:, 2] = rs.randint(0, 4, n)
    r[:, 3] = rs.uniform(-3, 3, n)
    r[:, 4], r[:, 5] = rs.uniform(0, 1000, n), rs.uniform(0, 300, n)
    r[:, 6] = r[:, 4] + rs.uniform(5, 120, n)
    r[:, 7] = r[:, 5] + rs.choice([20.0, 25.0, 25.5, 39.0, 40.0, 40.5, 60.0], n)
    r[:, 8:11] = rs.uniform(0.5, 4, (n, 3))
    r[:, 11:14] = rs.uniform(-20, 60, (n, 3))
    r[:, 14] = rs.uniform(-3, 3, n)
    r[:, 15] = np.round(rs.uniform(0.05, 1.0, n), 2) if with_score else 0.0
    return r


def _as_annos(rec):
    names = np.array(["Car", "Pedestrian", "Cyclist", "Van", "Person_sitting", "Truck", "DontCare", "Tram"])
    return dict(name=names[rec[:, 0].astype(int)], truncated=rec[:, 1], occluded=rec[:, 2].astype(int), alpha=rec[:, 3], bbox=rec[:, 4:8],
                dimensions=rec[:, 8:11], location=rec[:, 11:14], rotation_y=rec[:, 14], score=rec[:, 15])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_matching_logic_on_arbitrary_overlaps(shim, seed):
    """Steps 2-4 (ignore rules, both greedy passes, threshold sampling, PR accumulation) on random boxes and RANDOM overlap
    matrices -- including values exactly at the matching thresholds and tied scores -- against the oracle's restatement of
    the reference loops: thresholds, tp, fp, fn identical, orientation similarity to 1e-12, for all 54 combinations."""
    rs = np.random.RandomState(seed)
    B = 120
    gts = [_random_records(rs, int(rs.randint(0, 12)), False) for _ in range(B)]
    dts = [_random_records(rs, int(rs.randint(0, 15)), True) for _ in range(B)]
    grid = np.array([0.0, 0.1, 0.25, 0.3, 0.5, 0.6, 0.7, 0.9])
    ovs = [[rs.choice(grid, (len(d), len(g))) * (rs.rand(len(d), len(g)) < 0.7) for g, d in zip(gts, dts)] for _ in range(3)]
    classes = [0, 1, 2]
    mo = np.stack([np.array([[0.7, 0.5, 0.5]] * 3), np.array([[0.7, 0.5, 0.5], [0.5, 0.25, 0.25], [0.5, 0.25, 0.25]])], 0)
    arrays, sz, aos = EV.pack_eval_inputs(gts, dts, classes, mo)
    assert aos
    n_comb = 54
    flat = np.zeros((3, max(sz["n_pairs"], 1)))
    for m in range(3):
        for b in range(B):
            flat[m, arrays["pair_off"][b]:arrays["pair_off"][b + 1]] = ovs[m][b].reshape(-1)
    out = dict(overlaps=flat, tp_scores=np.zeros((n_comb, max(sz["n_gt"], 1))), thresholds=np.zeros((n_comb, 41)),
               pr=np.zeros((n_comb, 41, 4)), num_valid_gt=np.zeros((3, 3), np.int32), num_thresholds=np.zeros(n_comb, np.int32))
    d = L.KittiEvalDesc()
    for k, a in list(arrays.items()) + list(out.items()):
        setattr(d, k, a.ctypes.data)
    for k, v in sz.items():
        setattr(d, k, v)
    d.compute_aos = 1
    shim.shim_kitti_eval_match_only.argtypes = [ctypes.POINTER(L.KittiEvalDesc)]
    shim.shim_kitti_eval_match_only.restype = None
    shim.shim_kitti_eval_match_only(ctypes.byref(d))
    ga, da = [_as_annos(g) for g in gts], [_as_annos(x) for x in dts]
    checked = 0
    for metric in range(3):
        tables = {}
        R.precision_curves(ga, da, classes, metric, mo, aos=(metric == 0), overlaps=ovs[metric], tables=tables)
        for (m, level, k), (ths, pr, nvalid) in tables.items():
            comb = ((m * 3 + level) * 3 + metric) * 2 + k
            assert out["num_valid_gt"][m, level] == nvalid
            assert out["num_thresholds"][comb] == len(ths), (metric, m, level, k)
            assert np.array_equal(out["thresholds"][comb, :len(ths)], np.array(ths))
            got = out["pr"][comb, :len(ths)]
            assert np.array_equal(got[:, :3], pr[:, :3]), (metric, m, level, k)
            assert np.allclose(got[:, 3], pr[:, 3], rtol=0, atol=1e-12)
            checked += len(ths)
    assert checked > 300


def _clip_area(r1, r2):
    """Exact intersection area of two rotated rectangles by Sutherland-Hodgman clipping in float64 (independent of the
    reference's vertex-collection algorithm)."""
    def corners(r):
        c, s = np.cos(r[4]), np.sin(r[4])
        pts = np.array([[-r[2] / 2, -r[3] / 2], [-r[2] / 2, r[3] / 2], [r[2] / 2, r[3] / 2], [r[2] / 2, -r[3] / 2]])
        return np.stack([c * pts[:, 0] + s * pts[:, 1] + r[0], -s * pts[:, 0] + c * pts[:, 1] + r[1]], 1)
    poly, clip = corners(r1), corners(r2)
    e1, e2 = clip[1] - clip[0], clip[2] - clip[1]
    if e1[0] * e2[1] - e1[1] * e2[0] < 0:
        clip = clip[::-1]
    for i in range(4):
        a, b = clip[i], clip[(i + 1) % 4]
        side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        out = []
        for j in range(len(poly)):
            p, q = poly[j], poly[(j + 1) % len(poly)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp >= 0) != (sq >= 0):
                out.append(p + (q - p) * (sp / (sp - sq)))
        poly = np.array(out).reshape(-1, 2)
        if len(poly) == 0:
            return 0.0
    x, y = poly[:, 0], poly[:, 1]
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def test_rotated_intersection_is_geometrically_right(shim):
    """The reference's vertex-collection / angular-sort / triangle-fan algorithm (float32) equals true polygon clipping on
    generic box pairs -- a sanity property beyond the golden overlaps."""
    rs = np.random.RandomState(3)
    n = 3000
    a = np.stack([rs.uniform(-2, 2, n), rs.uniform(-2, 2, n), rs.uniform(0.5, 5, n), rs.uniform(0.5, 5, n), rs.uniform(-3.2, 3.2, n)], 1).astype(np.float32)
    b = np.stack([rs.uniform(-2, 2, n), rs.uniform(-2, 2, n), rs.uniform(0.5, 5, n), rs.uniform(0.5, 5, n), rs.uniform(-3.2, 3.2, n)], 1).astype(np.float32)
    out = np.zeros(n, dtype=np.float32)
    shim.shim_rotated_intersections.restype = None
    shim.shim_rotated_intersections(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), n, out.ctypes.data_as(ctypes.c_void_p))
    want = np.array([_clip_area(a[i].astype(np.float64), b[i].astype(np.float64)) for i in range(n)])
    err = np.abs(out - want)
    assert (want > 0.1).sum() > 2000
    assert np.percentile(err, 99) < 1e-4 and err.max() < 5e-3, (float(np.percentile(err, 99)), float(err.max()))
    ora = np.array([R.rotated_intersection(a[i], b[i]) for i in range(300)])       # and the oracle agrees with the device functions
    assert np.abs(ora - out[:300]).max() < 1e-4


def test_evaluate_python_with_score_threshold(shim, monkeypatch, tmp_path):
    """The file-level entry point with score_thresh (evaluate.py:17-32, kitti_common.py:191-202) against the reference run."""
    monkeypatch.setattr(EV, "pr_table", lambda g, d, c, mo, device="cuda": shim_pr_table(shim, g, d, c, mo))
    label_dir, result_dir = tmp_path / "label_2", tmp_path / "data"
    label_dir.mkdir(); result_dir.mkdir()
    for i in range(N_IMG):
        (label_dir / ("%06d.txt" % i)).write_text(str(GOLD["labels_%d" % i]))
        EV.generate_kitti_3d_detection(GOLD["det_%d" % i], str(result_dir / ("%06d.txt" % i)))
    (tmp_path / "val.txt").write_text("".join("%06d\n" % i for i in range(N_IMG)))
    text, ret = EV.evaluate_python(str(label_dir), str(result_dir), str(tmp_path / "val.txt"), ["Car"], metric="R40", score_thresh=0.5)
    keys = [str(k) for k in GOLD["keys_thresh"]]
    assert sorted(ret) == keys and text == str(GOLD["result_thresh"])
    np.testing.assert_allclose(np.array([float(ret[k]) for k in keys]), GOLD["values_thresh"], rtol=1e-12, atol=1e-12)
    full, _ = EV.evaluate_python(str(label_dir), str(result_dir), str(tmp_path / "val.txt"), ("Car", "Pedestrian", "Cyclist"))
    assert full == str(GOLD["result_R40"])
