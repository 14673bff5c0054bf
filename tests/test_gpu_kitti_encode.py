"""GPU parity of the KITTI input pipeline (mfx_kitti_encode_targets / mfx_kitti_preprocess_u8 through the C ABI) against
the reference goldens and the oracle: integer / mask / index fields identical, float fields to float32 round-off
(device exp/sin/cos/atan2 are not the host's libm), frames bit-exact."""
import os

import numpy as np
import pytest
import torch

from monoflex_amd import synthetic as S
from monoflex_amd.data import encode as E
from monoflex_amd.data.datasets import kitti_utils as KU
from oracle import kitti_encode_ref as K
from tests.kitti_common import GOLD, NAMES, compare_fields, fuzz_sample, golden_sample, oracle_fields

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = ("Car", "Pedestrian", "Cyclist")
GOLD_FIELDS = ("hm", "cls_ids", "target_centers", "reg_mask", "trunc_mask", "reg_weight", "keypoints_depth_mask", "pad_size", "edge_len",
               "edge_indices", "occlusions", "truncations", "gt_bboxes", "dimensions", "locations", "rotys", "keypoints", "offset_3D",
               "2d_bboxes", "alphas", "orientations")


def device_encode(samples, params=None, check=True):
    params = params or E.EncodeParams()
    out = E.encode_targets([KU.read_label_records(l, CLASSES) for l, _, _, _ in samples], [S.KITTI_P2] * len(samples),
                           [(w, h) for _, w, h, _ in samples], [f for _, _, _, f in samples], params, "cuda", check=check)
    return {k: v.cpu().numpy() for k, v in out.items()}


def test_targets_match_reference_goldens():
    out = device_encode([golden_sample(n)[:4] for n in NAMES])
    assert (out["status"] == 0).all()
    for b, n in enumerate(NAMES):
        compare_fields({k: v[b] for k, v in out.items()}, {k: GOLD[n + "_" + k] for k in GOLD_FIELDS}, n)
        np.testing.assert_allclose(out["P"][b], GOLD[n + "_P"], rtol=0, atol=1e-12)


def test_targets_match_oracle_on_fuzzed_labels():
    samples, refs, seed = [], [], 9000
    while len(samples) < 64:
        lines, w, h, flip = fuzz_sample(seed)
        seed += 1
        ref = oracle_fields(lines, w, h, flip)
        if ref is not None:
            samples.append((lines, w, h, flip)); refs.append(ref)
    out = device_encode(samples)                                  # one launch pair for the 64 samples
    for b in range(64):
        compare_fields({k: v[b] for k, v in out.items()}, refs[b], "fuzz%d" % b)


def test_other_input_size_and_filter_switch():
    small = E.EncodeParams(in_w=640, in_h=192)
    sl = S.synthetic_kitti_labels(5, 620, 187, 10)
    got = device_encode([(sl, 620, 187, True)], small)
    compare_fields({k: v[0] for k, v in got.items()}, K.encode_sample(sl, S.KITTI_P2, 620, 187, do_flip=True, in_w=640, in_h=192), "small")
    lines, w, h, flip, _ = golden_sample("s03")
    off = device_encode([(lines, w, h, flip)], E.EncodeParams(filter_enable=False))
    assert off["reg_mask"].sum() >= GOLD["s03_reg_mask"].sum()


def test_bad_inputs_raise_like_the_reference():
    line = "Car 0.50 0 1.50 -300.00 150.00 -100.00 250.00 1.50 1.60 3.90 -12.00 1.65 8.00 0.10"
    with pytest.raises(ValueError):
        device_encode([([line], 1242, 375, False)])
    out = device_encode([([line], 1242, 375, False), golden_sample("s00")[:4]], check=False)
    assert out["status"][0] == 2 and out["status"][1] == 0 and out["reg_mask"][0].sum() == 0
    with pytest.raises(RuntimeError):
        E.encode_targets([np.zeros((0, 14))], [S.KITTI_P2], [(1242, 375)], [0], E.EncodeParams(), "cpu")


def test_frames_bit_exact():
    frames, flips = [], []
    for n in ("s00", "s01", "s06", "s03"):
        _, w, h, flip, iseed = golden_sample(n)
        frames.append(np.random.RandomState(iseed).randint(0, 256, (h, w, 3)).astype(np.uint8)); flips.append(int(flip))
    frames.append(np.random.RandomState(1).randint(0, 256, (384, 1280, 3)).astype(np.uint8)); flips.append(1)       # no padding at all
    out = E.preprocess_images(frames, flips, E.EncodeParams(), "cuda").cpu().numpy()
    for b, f in enumerate(frames):
        assert np.array_equal(out[b], K.transform_image(f, do_flip=bool(flips[b]))), b
    for n, b in (("s00", 0), ("s01", 1), ("s06", 2)):                             # and against the reference pipeline's samples
        flat = out[b].astype(np.float64).ravel()
        np.testing.assert_allclose(flat[GOLD[n + "_img_idx"]], GOLD[n + "_img_samples"], rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        E.preprocess_images([np.zeros((400, 1300, 3), np.uint8)], [0], E.EncodeParams(), "cuda")


def test_do_train_replays_the_graphed_step(tmp_path):
    """engine.trainer.do_train (the reference's loop, engine/trainer.py:61-225) over DeviceLoader batches: the first batch captures
    the step as hipGraphs, the following ones are copied into the captured buffers and replayed; the learning rate the captured
    AdamW reads follows the warm-up + step schedule (device scalars), parameters move, the loss stays finite."""
    from PIL import Image
    from monoflex_amd.config import get_cfg
    from monoflex_amd.data import DeviceLoader, KITTIDataset
    from monoflex_amd.engine import trainer as TR
    from monoflex_amd.model.detector import KeypointDetector
    from monoflex_amd.solver import build_optimizer, build_scheduler
    for d in ("image_2", "label_2", "calib", "ImageSets"):
        (tmp_path / d).mkdir()
    P = np.asarray(S.KITTI_P2).reshape(-1)
    for i, (w, h) in enumerate([(1242, 375), (1224, 370), (1238, 374), (1242, 375)]):
        Image.fromarray(np.random.RandomState(i).randint(0, 256, (h, w, 3)).astype(np.uint8)).save(tmp_path / "image_2" / ("%06d.png" % i))
        (tmp_path / "label_2" / ("%06d.txt" % i)).write_text("".join(l + "\n" for l in S.synthetic_kitti_labels(60 + i, w, h, 5 + i)))
        (tmp_path / "calib" / ("%06d.txt" % i)).write_text("P2: " + " ".join("%.12e" % v for v in P) + "\nP3: " + " ".join("%.12e" % v for v in P) + "\n")
    (tmp_path / "ImageSets" / "train.txt").write_text("000000\n000001\n000002\n000003\n")
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.MODEL.PRETRAIN = False
    cfg.MODEL.COMPUTE_DTYPE = "bf16"
    cfg.SOLVER.MAX_ITERATION, cfg.SOLVER.LR_WARMUP, cfg.SOLVER.WARMUP_STEPS, cfg.SOLVER.STEPS = 6, True, 3, [4]
    cfg.SOLVER.SAVE_CHECKPOINT_INTERVAL, cfg.SOLVER.EVAL_INTERVAL = 1000, 0
    ds = KITTIDataset(cfg, str(tmp_path), is_train=True, augment=False)
    batches = list(DeviceLoader(ds, batch_size=2)) * 3                           # 6 iterations over two different batches
    torch.manual_seed(0)
    model = KeypointDetector(cfg).cuda().train()
    model.heads.loss_evaluator.log_as_float = False
    opt = build_optimizer(model, cfg)                                            # capturable on a GPU: device-scalar learning rates
    sched, warm = build_scheduler(opt, total_iters_each_epoch=2, optim_cfg=cfg.SOLVER)
    replays = []
    real_call = TR.GraphedTrainStep.__call__
    TR.GraphedTrainStep.__call__ = lambda self: replays.append(len(self.graphs)) or real_call(self)
    w0 = model.backbone.base.level2.tree1.conv1.weight.detach().clone()
    args = {"iteration": 0}
    try:
        loss = TR.do_train(cfg, False, model, batches, None, opt, sched, warm, None, "cuda", args)
    finally:
        TR.GraphedTrainStep.__call__ = real_call
    assert args["iteration"] == 6 and len(replays) == 6 and np.isfinite(loss)
    assert not torch.equal(w0, model.backbone.base.level2.tree1.conv1.weight)
    base = cfg.SOLVER.BASE_LR
    assert torch.is_tensor(opt.param_groups[0]["lr"]) and float(opt.param_groups[0]["lr"]) == pytest.approx(base * cfg.SOLVER.LR_DECAY, rel=1e-5)


def test_dataset_loader_feeds_a_training_step(tmp_path):
    """Generated KITTI directory -> DeviceLoader -> model(images, targets): the encoded batch drives the HIP training
    forward/backward unchanged, and equals the oracle's encoding of the same files."""
    from PIL import Image
    from monoflex_amd.config import get_cfg
    from monoflex_amd.data import DeviceLoader, KITTIDataset
    from monoflex_amd.model.detector import KeypointDetector
    for d in ("image_2", "label_2", "calib", "ImageSets"):
        (tmp_path / d).mkdir()
    P = np.asarray(S.KITTI_P2).reshape(-1)
    sizes = [(1242, 375), (1224, 370)]
    for i, (w, h) in enumerate(sizes):
        Image.fromarray(np.random.RandomState(i).randint(0, 256, (h, w, 3)).astype(np.uint8)).save(tmp_path / "image_2" / ("%06d.png" % i))
        (tmp_path / "label_2" / ("%06d.txt" % i)).write_text("".join(l + "\n" for l in S.synthetic_kitti_labels(60 + i, w, h, 9)))
        (tmp_path / "calib" / ("%06d.txt" % i)).write_text("P2: " + " ".join("%.12e" % v for v in P) + "\nP3: " + " ".join("%.12e" % v for v in P) + "\n")
    (tmp_path / "ImageSets" / "train.txt").write_text("000000\n000001\n")
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.MODEL.PRETRAIN = False
    ds = KITTIDataset(cfg, str(tmp_path), is_train=True, augment=False)
    img, tgt, idx = ds[1]
    assert img.shape == (3, 384, 1280) and img.is_cuda and idx == "000001" and tgt.get_field("hm").shape == (3, 96, 320)
    batch = next(iter(DeviceLoader(ds, batch_size=2)))
    assert batch["images"].tensors.shape == (2, 3, 384, 1280) and batch["img_ids"] == ("000000", "000001")
    for b, (w, h) in enumerate(sizes):
        ref = K.encode_sample(S.synthetic_kitti_labels(60 + b, w, h, 9), S.KITTI_P2, w, h)
        compare_fields({k: batch["targets"][b].get_field(k).cpu().numpy() for k in GOLD_FIELDS}, ref, "loader%d" % b)
        c = batch["targets"][b].get_field("calib")
        assert np.isclose(c.f_u, P[0]) and len(batch["targets"][b]) == int(ref["reg_mask"].sum())
    torch.manual_seed(0)
    model = KeypointDetector(cfg).cuda().train()
    loss_dict, log = model(batch["images"], list(batch["targets"]))
    total = sum(loss_dict.values())
    assert torch.isfinite(total) and float(total.detach()) > 0
    from monoflex_amd.engine.trainer import prepare_targets
    fast = prepare_targets(model, batch["targets"], "cuda", fields=batch["fields"])        # no per-image re-stacking
    slow = prepare_targets(model, batch["targets"], "cuda")
    assert torch.equal(fast.edge[0], slow.edge[0]) and torch.equal(fast.loss[0], slow.loss[0])
    assert all(torch.equal(v, slow.loss[1][k]) for k, v in fast.loss[1].items() if torch.is_tensor(v))
    total.backward()
    g = model.backbone.base.base_layer[0].weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
