"""The oracle's restatement of the KITTI input pipeline / target encoding (oracle/kitti_encode_ref.py) against
tests/golden/kitti_encode.npz, the outputs of the reference's own KITTIDataset.__getitem__ (oracle/gen_golden.py kitti).
Integer / mask / index fields must be identical; float fields agree to float32 round-off."""
import os

import numpy as np
import pytest

from oracle import kitti_encode_ref as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "kitti_encode.npz"))
EXACT = ["cls_ids", "target_centers", "reg_mask", "trunc_mask", "reg_weight", "keypoints_depth_mask", "pad_size", "edge_len",
         "edge_indices", "occlusions", "truncations", "gt_bboxes", "dimensions", "locations", "rotys"]
CLOSE = ["hm", "keypoints", "offset_3D", "2d_bboxes", "alphas", "orientations"]


def sample(name):
    w, h, flip, iseed = (int(v) for v in GOLD[name + "_meta"])
    lines = str(GOLD[name + "_labels"]).split("\n") if str(GOLD[name + "_labels"]) else []
    return lines, w, h, bool(flip), iseed


@pytest.mark.parametrize("name", [str(n) for n in GOLD["names"]])
def test_targets_match_reference_dataset(name):
    from monoflex_amd.synthetic import KITTI_P2
    lines, w, h, flip, _ = sample(name)
    f = K.encode_sample(lines, KITTI_P2, w, h, do_flip=flip)
    for k in EXACT:
        ref = GOLD[name + "_" + k]
        assert np.array_equal(np.asarray(f[k]).astype(ref.dtype), ref), (name, k)
        assert np.asarray(f[k]).dtype == ref.dtype or k in ("edge_len", "pad_size"), (name, k, np.asarray(f[k]).dtype, ref.dtype)
    for k in CLOSE:
        ref = GOLD[name + "_" + k]
        assert np.asarray(f[k]).dtype == ref.dtype and f[k].shape == ref.shape, (name, k)
        np.testing.assert_allclose(f[k], ref, rtol=1e-6, atol=1e-6, err_msg="%s %s" % (name, k))
    np.testing.assert_allclose(f["P"], GOLD[name + "_P"], rtol=0, atol=1e-12)
    assert np.array_equal(f["hm"] == 1.0, GOLD[name + "_hm"] == 1.0)                  # peak pixels: exactly 1 at the same places


def test_fixture_covers_the_edge_cases():
    """The golden set must exercise: no objects, flipped samples, truncated (approximate-centre) objects, objects dropped
    by every early exit, and the 40-slot maximum."""
    names = [str(n) for n in GOLD["names"]]
    kept = {n: int(GOLD[n + "_reg_mask"].sum()) for n in names}
    assert min(kept.values()) == 0 and max(kept.values()) >= 20
    assert sum(int(GOLD[n + "_meta"][2]) for n in names) >= 4
    assert sum(int(GOLD[n + "_trunc_mask"].sum()) for n in names) >= 5
    n_lines = {n: len([l for l in str(GOLD[n + "_labels"]).split("\n") if l.split(" ")[0] in ("Car", "Pedestrian", "Cyclist")]) for n in names}
    assert any(n_lines[n] > kept[n] for n in names)
    assert max(n_lines.values()) <= 40 and max(len(str(GOLD[n + "_labels"]).split("\n")) for n in names) == 40
    sizes = {tuple(int(v) for v in GOLD[n + "_meta"][:2]) for n in names}
    assert len(sizes) == 4


@pytest.mark.parametrize("name", ["s00", "s01", "s06"])
def test_image_transform_matches_reference_pipeline(name):
    _, w, h, flip, iseed = sample(name)
    img = np.random.RandomState(iseed).randint(0, 256, (h, w, 3)).astype(np.uint8)
    x = K.transform_image(img, do_flip=flip)
    assert x.shape == (3, 384, 1280) and x.dtype == np.float32
    flat = x.astype(np.float64).ravel()
    s, a, q = GOLD[name + "_img_sum"]
    np.testing.assert_allclose([flat.sum(), np.abs(flat).sum(), (flat * flat).sum()], [s, a, q], rtol=1e-9)
    np.testing.assert_allclose(flat[GOLD[name + "_img_idx"]], GOLD[name + "_img_samples"], rtol=0, atol=1e-6)


def test_edge_indices_closed_form():
    ei, el = K.edge_indices(1242, 375, K.pad_size(1242, 375))
    assert el == 806 and tuple(ei[0]) == (5, 1) and tuple(ei[806]) == (5, 1) and (ei[807:] == 0).all()
    assert len({tuple(p) for p in ei[:806]}) == 806                                  # a closed walk: last point == first
