"""GPU parity of the KITTI AP evaluator (mfx_kitti_eval_* through the C ABI) against the reference goldens
(tests/golden/kitti_eval.npz) and known answers."""
import os

import numpy as np
import pytest
import torch

from monoflex_amd import synthetic as S
from monoflex_amd.data import evaluation as EV

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "kitti_eval.npz"))
N_IMG = len([k for k in GOLD.files if k.startswith("det_")])


def records():
    return ([EV.parse_label_text(str(GOLD["labels_%d" % i])) for i in range(N_IMG)],
            [EV.parse_label_text(str(GOLD["txt_%d" % i])) for i in range(N_IMG)])


def test_overlaps_match_reference():
    gts, dts = records()
    _, _, ov, pair_off, aos = EV.pr_table(gts, dts, [0, 1, 2], np.full((2, 3, 3), 0.5))
    assert aos
    for m in range(3):
        for i in range(N_IMG):
            ref = GOLD["ov%d_%d" % (m, i)]
            got = ov[m, pair_off[i]:pair_off[i + 1]].reshape(ref.shape)
            # bbox overlaps are float64 on both sides; rotated boxes are float32 with the device's cosf/sinf/sqrtf
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-4 if m else 1e-15, err_msg="metric %d image %d" % (m, i))


@pytest.mark.parametrize("metric", ["R40", "R11"])
def test_official_result_matches_reference(metric):
    gts, dts = records()
    text, ret = EV.get_official_eval_result(gts, dts, ["Car", "Pedestrian", "Cyclist"], metric=metric)
    keys = [str(k) for k in GOLD["keys_" + metric]]
    assert sorted(ret.keys()) == keys
    np.testing.assert_allclose(np.array([float(ret[k]) for k in keys]), GOLD["values_" + metric], rtol=1e-9, atol=1e-9)
    assert text == str(GOLD["result_" + metric])


def test_evaluate_python_through_files(tmp_path):
    label_dir, result_dir = tmp_path / "label_2", tmp_path / "data"
    label_dir.mkdir(); result_dir.mkdir()
    for i in range(N_IMG):
        (label_dir / ("%06d.txt" % i)).write_text(str(GOLD["labels_%d" % i]))
        EV.generate_kitti_3d_detection(torch.from_numpy(GOLD["det_%d" % i]).cuda(), str(result_dir / ("%06d.txt" % i)))
        assert (result_dir / ("%06d.txt" % i)).read_text() == str(GOLD["txt_%d" % i])
    split = tmp_path / "val.txt"
    split.write_text("".join("%06d\n" % i for i in range(N_IMG)))
    text, ret = EV.evaluate_python(str(label_dir), str(result_dir), str(split), ("Car", "Pedestrian", "Cyclist"), metric="R40")
    assert text == str(GOLD["result_R40"])
    i = [str(k) for k in GOLD["keys_R40"]].index("Car_3d_0.70/moderate")
    assert abs(ret["Car_3d_0.70/moderate"] - GOLD["values_R40"][i]) < 1e-9


def test_second_set_against_the_oracle():
    """30 other images (seeds chosen so that no rotated overlap lies within 1e-3 of a matching threshold): the report and every AP
    equal the CPU restatement's; an image set without any detection scores 0 everywhere."""
    from oracle import kitti_eval_ref as R
    labels = [S.synthetic_kitti_labels(3000 + i, 1242, 375, 4 + i % 12, z_range=(5, 38), occl_max=1) for i in range(30)]
    dets = [S.synthetic_detections(3500 + i, l, 1242, 375, recall=0.9) for i, l in enumerate(labels)]
    gts = [EV.parse_label_text("\n".join(l)) for l in labels]
    dts = [EV.parse_label_text(R.result_text(d)) for d in dets]
    ga, da = [R.parse_annos("\n".join(l)) for l in labels], [R.parse_annos(R.result_text(d)) for d in dets]
    otext, oret = R.official_result(ga, da, (0, 1, 2), "R40")
    text, ret = EV.get_official_eval_result(gts, dts, [0, 1, 2], metric="R40")
    assert sorted(ret) == sorted(oret)
    for k in oret:
        assert abs(float(ret[k]) - float(oret[k])) < 1e-9 or (np.isnan(ret[k]) and np.isnan(oret[k])), k
    assert text == otext and sum(v > 1 for v in oret.values()) > 20
    _, ret0 = EV.get_official_eval_result(gts, [np.zeros((0, 16))] * 30, [0, 1, 2], metric="R40")
    assert all(v == 0 or np.isnan(v) for v in ret0.values())


def test_errors():
    with pytest.raises(RuntimeError):
        EV.pr_table([np.zeros((0, 16))], [np.zeros((0, 16))], [0], np.zeros((1, 3, 1)), device="cpu")
    with pytest.raises(ValueError):
        EV.pr_table([np.zeros((1, 16))], [np.zeros((65, 16))], [0], np.zeros((1, 3, 1)))


def test_inference_loop_end_to_end(tmp_path):
    """Generated KITTI directory -> DeviceLoader -> detector (eval, B=2) -> result files -> device AP evaluation
    (engine/inference.py:66-126): files exist for every image, parse back, and the scores are finite numbers."""
    from PIL import Image
    from monoflex_amd.config import get_cfg
    from monoflex_amd.data import DeviceLoader, InferenceSampler, KITTIDataset
    from monoflex_amd.engine.inference import inference
    from monoflex_amd.model.detector import KeypointDetector
    for d in ("image_2", "label_2", "calib", "ImageSets"):
        (tmp_path / d).mkdir()
    P = np.asarray(S.KITTI_P2).reshape(-1)
    n = 3
    for i in range(n):
        Image.fromarray(np.random.RandomState(i).randint(0, 256, (375, 1242, 3)).astype(np.uint8)).save(tmp_path / "image_2" / ("%06d.png" % i))
        (tmp_path / "label_2" / ("%06d.txt" % i)).write_text("\n".join(S.synthetic_kitti_labels(70 + i, 1242, 375, 8, z_range=(5, 38), occl_max=1)))
        (tmp_path / "calib" / ("%06d.txt" % i)).write_text("P2: " + " ".join("%.12e" % v for v in P) + "\nP3: " + " ".join("%.12e" % v for v in P) + "\n")
    (tmp_path / "ImageSets" / "val.txt").write_text("".join("%06d\n" % i for i in range(n)))
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"), ["MODEL.COMPUTE_DTYPE", "bf16"])
    cfg.MODEL.PRETRAIN = False
    ds = KITTIDataset(cfg, str(tmp_path), is_train=False)
    assert ds.split == "val" and ds.flip_p == 0.0
    torch.manual_seed(0)
    model = KeypointDetector(cfg).cuda()
    sd = S.synthetic_state_dict(model.state_dict(), seed=0, cls_bias=-1.0)
    model.load_state_dict(sd)
    loader = DeviceLoader(ds, batch_size=2, sampler=InferenceSampler(len(ds)))
    ret_dicts, result, _ = inference(model, loader, "kitti_val", output_folder=str(tmp_path / "out"), metrics=("R40", "R11"))
    files = sorted(os.listdir(tmp_path / "out" / "data"))
    assert files == ["%06d.txt" % i for i in range(n)]
    rows = EV.read_label_folder(str(tmp_path / "out" / "data"))
    assert sum(len(r) for r in rows) > 0 and all(len(r) <= 50 for r in rows)
    assert len(ret_dicts) == 2 and "Car_3d_0.70/moderate" in ret_dicts[0] and result.startswith("Car AP@0.70, 0.70, 0.70:")
    assert all(np.isfinite(v) or np.isnan(v) for v in ret_dicts[0].values())
    # the loop keeps one batch in flight by default (rows of batch k fetched after batch k+1 was launched): byte-identical files to the
    # reference's strictly sequential loop, also with a last, smaller batch
    from monoflex_amd.engine.inference import compute_on_dataset
    (tmp_path / "seq").mkdir()
    timer = {}
    assert compute_on_dataset(model, loader, "cuda", str(tmp_path / "seq"), timer, overlap=False) == n and timer["inference_seconds"] > 0
    for f in files:
        assert (tmp_path / "seq" / f).read_bytes() == (tmp_path / "out" / "data" / f).read_bytes(), f


def test_eval_all_depths_walks_every_method(tmp_path):
    """`--eval_all_depths` (engine/inference.py:131-198): eight passes over a generated validation directory, one result folder per depth-solving
    method, `output_depth` restored afterwards; the 'soft' pass writes the files the plain evaluation writes, the other passes different ones;
    'oracle' reads the ground-truth fields of the val split (detector_infer.py:238-277)."""
    from PIL import Image
    from monoflex_amd.config import get_cfg
    from monoflex_amd.data import DeviceLoader, InferenceSampler, KITTIDataset
    from monoflex_amd.engine.inference import EVAL_DEPTH_METHODS, inference, inference_all_depths
    from monoflex_amd.model.detector import KeypointDetector
    for d in ("image_2", "label_2", "calib", "ImageSets"):
        (tmp_path / d).mkdir()
    P = np.asarray(S.KITTI_P2).reshape(-1)
    n = 3
    for i in range(n):
        Image.fromarray(np.random.RandomState(i).randint(0, 256, (375, 1242, 3)).astype(np.uint8)).save(tmp_path / "image_2" / ("%06d.png" % i))
        (tmp_path / "label_2" / ("%06d.txt" % i)).write_text("\n".join(S.synthetic_kitti_labels(70 + i, 1242, 375, 8, z_range=(5, 38), occl_max=1)))
        (tmp_path / "calib" / ("%06d.txt" % i)).write_text("P2: " + " ".join("%.12e" % v for v in P) + "\nP3: " + " ".join("%.12e" % v for v in P) + "\n")
    (tmp_path / "ImageSets" / "val.txt").write_text("".join("%06d\n" % i for i in range(n)))
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"), ["MODEL.COMPUTE_DTYPE", "bf16"])
    cfg.MODEL.PRETRAIN = False
    ds = KITTIDataset(cfg, str(tmp_path), is_train=False)
    torch.manual_seed(0)
    model = KeypointDetector(cfg).cuda()
    model.load_state_dict(S.synthetic_state_dict(model.state_dict(), seed=0, cls_bias=-1.0))
    loader = DeviceLoader(ds, batch_size=2, sampler=InferenceSampler(len(ds)))
    inference(model, loader, "kitti_val", output_folder=str(tmp_path / "out"))
    ret = inference_all_depths(model, loader, "kitti_val", output_folder=str(tmp_path / "out"))
    assert model.heads.post_processor.output_depth == "soft"
    assert tuple(ret) == EVAL_DEPTH_METHODS and all("Car_3d_0.70/moderate" in d for d in ret.values())
    texts = {}
    for m in EVAL_DEPTH_METHODS:
        folder = tmp_path / "out" / "eval_all_depths" / m
        assert sorted(os.listdir(folder)) == ["%06d.txt" % i for i in range(n)]
        texts[m] = b"".join((folder / f).read_bytes() for f in sorted(os.listdir(folder)))
    assert texts["soft"] == b"".join((tmp_path / "out" / "data" / ("%06d.txt" % i)).read_bytes() for i in range(n))
    # the methods decode different rows (with synthetic weights several keypoint estimates sit on the 100 m clamp and coincide)
    assert len(set(texts.values())) >= 4 and texts["soft"] != texts["mean"] and texts["direct"] != texts["hard"]
