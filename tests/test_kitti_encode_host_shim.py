"""CPU check of the encoder's device functions: monoflex_amd/csrc/kitti_encode_math.h is compiled for the host by
tests/shim/kitti_encode_host.cpp (test-only) and driven through the same mfx_kitti_desc / packing code as the GPU path,
then compared with the reference goldens and with the oracle on fuzzed label sets. Also covers the host-side packing,
label parsing and the dataset front on a generated KITTI directory (no GPU work)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from monoflex_amd import lib as L
from monoflex_amd import synthetic as S
from monoflex_amd.data import encode as E
from monoflex_amd.data.datasets import kitti_utils as KU
from oracle import kitti_encode_ref as K
from tests.kitti_common import GOLD, NAMES, compare_fields, fuzz_sample, golden_sample, oracle_fields

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NP_DTYPES = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8}


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("shim") / "libkitti_shim.so")
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "shim", "kitti_encode_host.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = ctypes.CDLL(so)
    lib.shim_kitti_encode.argtypes = [ctypes.POINTER(L.KittiDesc)]
    lib.shim_kitti_encode.restype = None
    lib.shim_kitti_preprocess.restype = None
    return lib


def run_shim(shim, samples, params=None):
    """samples: [(lines, w, h, flip)] -> {field: (B, ...) numpy}, through pack_inputs + the descriptor layout of the GPU path."""
    params = params or E.EncodeParams()
    recs = [KU.read_label_records(lines, ("Car", "Pedestrian", "Cyclist")) for lines, _, _, _ in samples]
    inp = E.pack_inputs(recs, [S.KITTI_P2] * len(samples), [(w, h) for _, w, h, _ in samples], [f for _, _, _, f in samples], params)
    dims, B = params.dims(), len(samples)
    out = {name: np.full((B,) + tuple(dims.get(s, s) for s in shape), 77, dtype=NP_DTYPES[dt])      # poison: every element must be written
           for name, (_, shape, dt) in E.TARGET_FIELDS.items()}
    d = L.KittiDesc()
    for k, a in inp.items():
        setattr(d, k, a.ctypes.data)
    for name, (member, _, _) in E.TARGET_FIELDS.items():
        setattr(d, member, out[name].ctypes.data)
    d.B, d.max_objs, d.in_w, d.in_h, d.down, d.num_classes = B, params.max_objs, params.in_w, params.in_h, params.down, params.num_classes
    d.filter_trunc, d.filter_size, d.edge_ratio = params.filter_trunc, params.filter_size, params.edge_ratio
    shim.shim_kitti_encode(ctypes.byref(d))
    return out


def test_descriptor_matches_the_c_struct():
    """ctypes layout == C layout: a marker written through each member must land where the C code reads it."""
    assert ctypes.sizeof(L.KittiDesc) == 29 * 8 + 6 * 4 + 3 * 8
    assert L.KittiDesc.B.offset == 29 * 8 and L.KittiDesc.filter_trunc.offset == 29 * 8 + 24
    members = [m for m, _ in L.KittiDesc._fields_[:29]]
    assert {m for m, _, _ in E.TARGET_FIELDS.values()} | {"records", "n_obj", "P", "img_wh", "flip"} == set(members)


def test_shim_matches_reference_goldens_as_one_batch(shim):
    samples = [golden_sample(n)[:4] for n in NAMES]
    out = run_shim(shim, samples)
    assert (out["status"] == 0).all()
    for b, n in enumerate(NAMES):
        ref = {k: GOLD[n + "_" + k] for k in ("hm", "cls_ids", "target_centers", "reg_mask", "trunc_mask", "reg_weight", "keypoints_depth_mask",
                                             "pad_size", "edge_len", "edge_indices", "occlusions", "truncations", "gt_bboxes", "dimensions",
                                             "locations", "rotys", "keypoints", "offset_3D", "2d_bboxes", "alphas", "orientations")}
        compare_fields({k: v[b] for k, v in out.items()}, ref, n)
        np.testing.assert_allclose(out["P"][b], GOLD[n + "_P"], rtol=0, atol=1e-12)


def test_shim_matches_oracle_on_fuzzed_labels(shim):
    samples, refs = [], []
    seed = 7000
    while len(samples) < 48:
        lines, w, h, flip = fuzz_sample(seed)
        seed += 1
        ref = oracle_fields(lines, w, h, flip)
        if ref is not None:
            samples.append((lines, w, h, flip)); refs.append(ref)
    for lo in range(0, len(samples), 16):
        out = run_shim(shim, samples[lo:lo + 16])
        assert (out["status"] == 0).all()
        for b in range(16):
            compare_fields({k: v[b] for k, v in out.items()}, refs[lo + b], "fuzz%d" % (lo + b))
    assert sum(int(r["trunc_mask"].sum()) for r in refs) > 20 and sum(int(r["reg_mask"].sum()) for r in refs) > 300


def test_shim_flags_inputs_the_reference_fails_on(shim):
    # a truncated object (centre left of the image) whose label box centre is outside the image as well
    line = "Car 0.50 0 1.50 -300.00 150.00 -100.00 250.00 1.50 1.60 3.90 -12.00 1.65 8.00 0.10"
    assert oracle_fields([line], 1242, 375, False) is None
    out = run_shim(shim, [([line], 1242, 375, False), (golden_sample("s00")[0], 1242, 375, False)])
    assert out["status"][0] == 2 and out["status"][1] == 0 and out["reg_mask"][0].sum() == 0
    with pytest.raises(IndexError):
        E.pack_inputs([np.zeros((41, 14))], [S.KITTI_P2], [(1242, 375)], [0], E.EncodeParams())


def test_filter_switch_and_other_sizes(shim):
    lines, w, h, flip, _ = golden_sample("s03")
    on = run_shim(shim, [(lines, w, h, flip)])
    off = run_shim(shim, [(lines, w, h, flip)], E.EncodeParams(filter_enable=False))
    assert off["reg_mask"].sum() >= on["reg_mask"].sum()
    small = E.EncodeParams(in_w=640, in_h=192)
    sl = S.synthetic_kitti_labels(5, 620, 187, 10)
    got = run_shim(shim, [(sl, 620, 187, True)], small)
    ref = K.encode_sample(sl, S.KITTI_P2, 620, 187, do_flip=True, in_w=640, in_h=192)
    compare_fields({k: v[0] for k, v in got.items()}, ref, "small")


def test_shim_image_transform(shim):
    frames, flips = [], []
    for n in ("s00", "s01", "s06"):
        _, w, h, flip, iseed = golden_sample(n)
        frames.append(np.random.RandomState(iseed).randint(0, 256, (h, w, 3)).astype(np.uint8)); flips.append(int(flip))
    offsets = np.cumsum([0] + [f.size for f in frames[:-1]]).astype(np.int64)
    pixels = np.concatenate([f.reshape(-1) for f in frames])
    wh = np.array([(f.shape[1], f.shape[0]) for f in frames], dtype=np.int32)
    out = np.full((3, 3, 384, 1280), 77, dtype=np.float32)
    mean, std = (ctypes.c_float * 3)(*K.PIXEL_MEAN), (ctypes.c_float * 3)(*K.PIXEL_STD)
    shim.shim_kitti_preprocess(pixels.ctypes.data_as(ctypes.c_void_p), offsets.ctypes.data_as(ctypes.c_void_p), wh.ctypes.data_as(ctypes.c_void_p),
                               np.asarray(flips, dtype=np.int32).ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                               3, 1280, 384, mean, std)
    for b, f in enumerate(frames):
        assert np.array_equal(out[b], K.transform_image(f, do_flip=bool(flips[b])))


def test_label_parsing_and_calibration(tmp_path):
    lines = S.synthetic_kitti_labels(3, 1242, 375, 12)
    rec = KU.read_label_records(lines, ("Car", "Pedestrian", "Cyclist"))
    objs = K.read_objects(lines)
    assert rec.shape == (len(objs), 14)
    for r, o in zip(rec, objs):
        assert r[0] == K.TYPE_ID[o.type] and r[1] == o.truncation and r[2] == o.occlusion
        assert tuple(r[3:7]) == (o.xmin, o.ymin, o.xmax, o.ymax) and tuple(r[7:10]) == (o.h, o.w, o.l) and r[13] == o.ry
        assert np.array_equal(r[10:13].astype(np.float32), o.t)
    with pytest.raises(ValueError):
        KU.parse_label_line("Car 0.0 0 1.0 2.0")
    with pytest.raises(KeyError):
        KU.parse_label_line("Spaceship " + " ".join(["0"] * 14))
    f = tmp_path / "000000.txt"
    P = np.asarray(S.KITTI_P2).reshape(-1)
    f.write_text("P2: " + " ".join("%.12e" % v for v in P) + "\nP3: " + " ".join("%.12e" % (v + 1) for v in P) + "\nR0_rect: 1 0 0 0 1 0 0 0 1\ncalib_time: 09-Jan-2012\n")
    c = KU.Calibration(str(f))
    assert np.allclose(c.P.reshape(-1), P) and c.f_u == P[0] and np.isclose(c.b_x, P[3] / -P[0])
    assert np.allclose(KU.Calibration(str(f), use_right_cam=True).P.reshape(-1), P + 1)
    fl = c.flipped(1242)
    assert np.isclose(fl.c_u, 1242 - c.c_u - 1) and np.isclose(fl.b_x, -c.b_x) and fl.f_u == c.f_u


def test_dataset_front_on_generated_directory(tmp_path):
    """File layout, split handling, flip coin, raw samples and the border walk -- everything up to the GPU launch."""
    import random
    from PIL import Image
    from monoflex_amd.config import get_cfg
    from monoflex_amd.data import KITTIDataset
    for d in ("image_2", "label_2", "calib", "ImageSets"):
        (tmp_path / d).mkdir()
    P = np.asarray(S.KITTI_P2).reshape(-1)
    for i, (w, h) in enumerate([(1242, 375), (1224, 370)]):
        Image.fromarray(np.random.RandomState(i).randint(0, 256, (h, w, 3)).astype(np.uint8)).save(tmp_path / "image_2" / ("%06d.png" % i))
        (tmp_path / "label_2" / ("%06d.txt" % i)).write_text("".join(l + "\n" for l in S.synthetic_kitti_labels(40 + i, w, h, 9)))
        (tmp_path / "calib" / ("%06d.txt" % i)).write_text("P2: " + " ".join("%.12e" % v for v in P) + "\nP3: " + " ".join("%.12e" % v for v in P) + "\n")
    (tmp_path / "ImageSets" / "train.txt").write_text("000000\n000001\n")
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    ds = KITTIDataset(cfg, str(tmp_path), is_train=True)
    assert len(ds) == 2 and ds.max_edge_length == 832 and ds.flip_p == 0.5
    random.seed(0)
    coins = [ds.load_raw(0).flip for _ in range(40)]
    assert 8 < sum(coins) < 32
    raw = KITTIDataset(cfg, str(tmp_path), is_train=True, augment=False).load_raw(1)
    assert raw.frame.shape == (370, 1224, 3) and raw.frame.dtype == np.uint8 and raw.flip is False and raw.original_idx == "000001"
    assert raw.records.shape[1] == 14 and set(raw.records[:, 0]) <= {0.0, 1.0, 2.0}
    ei, el = K.edge_indices(1224, 370, K.pad_size(1224, 370))
    walk = ds.get_edge_utils((1224, 370), K.pad_size(1224, 370))
    assert walk.shape[0] == el + 1 and np.array_equal(walk.numpy(), ei[:el + 1])
    with pytest.raises(FileNotFoundError):
        KITTIDataset(cfg, str(tmp_path), is_train=False)                     # no ImageSets/val.txt
    with pytest.raises(RuntimeError):
        ds.encode_batch([raw]) if not torch.cuda.is_available() else (_ for _ in ()).throw(RuntimeError("gpu present"))


def test_prepared_targets_from_stacked_fields_equal_the_per_image_path(shim):
    """engine.trainer.prepare_targets(fields=...) (the DeviceLoader fast path) builds the same loss / edge inputs as
    stacking the per-image ParamsLists, and the loss evaluated on both is identical (CPU tensors; no GPU work)."""
    from monoflex_amd.config import get_cfg
    from monoflex_amd.data.datasets.kitti_utils import Calibration
    from monoflex_amd.engine.trainer import prepare_targets
    from monoflex_amd.model.head.detector_loss import make_loss_evaluator
    from monoflex_amd.structures.params_3d import ParamsList
    samples = [golden_sample(n)[:4] for n in ("s00", "s03", "s07")]
    fields = {k: torch.from_numpy(v) for k, v in run_shim(shim, samples).items()}
    targets = []
    for b, (_, w, h, flip) in enumerate(samples):
        t = ParamsList(image_size=(1280, 384), is_train=True)
        for k in ("cls_ids", "target_centers", "keypoints", "keypoints_depth_mask", "dimensions", "locations", "reg_mask", "reg_weight",
                  "offset_3D", "2d_bboxes", "pad_size", "rotys", "trunc_mask", "alphas", "orientations", "hm", "edge_len", "edge_indices"):
            t.add_field(k, fields[k][b])
        c = Calibration.from_matrix(S.KITTI_P2)
        t.add_field("calib", c.flipped(w) if flip else c)
        targets.append(t)

    class _M:                                                      # the two attributes prepare_targets touches
        pass
    m = _M(); m.heads = _M()
    m.heads.loss_evaluator = make_loss_evaluator(get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml")))
    slow, fast = prepare_targets(m, targets, "cpu"), prepare_targets(m, targets, "cpu", fields=fields)
    assert torch.equal(slow.edge[0], fast.edge[0]) and torch.equal(slow.edge[1], fast.edge[1])
    assert torch.equal(slow.loss[0], fast.loss[0]) and set(slow.loss[1]) - {"ori_imgs"} == set(fast.loss[1])
    for k, v in fast.loss[1].items():
        if torch.is_tensor(v):
            assert torch.equal(v, slow.loss[1][k]) and v.dtype == slow.loss[1][k].dtype, k
    g = torch.Generator().manual_seed(0)
    pred = {"cls": torch.rand(3, 3, 96, 320, generator=g).clamp(1e-4, 1 - 1e-4), "reg": torch.randn(3, 50, 96, 320, generator=g) * 0.1}
    a, _ = m.heads.loss_evaluator(pred, slow)
    b, _ = m.heads.loss_evaluator(pred, fast)
    assert all(float(a[k]) == float(b[k]) for k in a)


@pytest.fixture
def cpu_encoders(shim, monkeypatch):
    """Stand in for the two GPU launches with the host build of the same device functions, so that the dataset / loader
    front (field lists, splits, calibration flips, batching, worker processes) can be exercised end to end on CPU."""
    import monoflex_amd.data.datasets.kitti as DK

    def encode(records, Ps, sizes, flips, params, device, check=True):
        inp = E.pack_inputs(records, Ps, sizes, flips, params)
        dims, B = params.dims(), len(records)
        out = {n: np.zeros((B,) + tuple(dims.get(s, s) for s in sh), dtype=NP_DTYPES[dt]) for n, (_, sh, dt) in E.TARGET_FIELDS.items()}
        d = L.KittiDesc()
        for k, a in inp.items():
            setattr(d, k, a.ctypes.data)
        for n, (member, _, _) in E.TARGET_FIELDS.items():
            setattr(d, member, out[n].ctypes.data)
        d.B, d.max_objs, d.in_w, d.in_h, d.down, d.num_classes = B, params.max_objs, params.in_w, params.in_h, params.down, params.num_classes
        d.filter_trunc, d.filter_size, d.edge_ratio = params.filter_trunc, params.filter_size, params.edge_ratio
        shim.shim_kitti_encode(ctypes.byref(d))
        return {k: torch.from_numpy(v) for k, v in out.items()}

    def frames(fr, flips, params, device, mean, std):
        return torch.from_numpy(np.stack([K.transform_image(f, bool(fl)) for f, fl in zip(fr, flips)]))
    monkeypatch.setattr(DK, "encode_targets", encode)
    monkeypatch.setattr(DK, "preprocess_images", frames)


def _kitti_dir(tmp_path, n=3, splits=("train", "val", "test")):
    from PIL import Image
    for d in ("image_2", "label_2", "calib", "ImageSets"):
        (tmp_path / d).mkdir()
    P = np.asarray(S.KITTI_P2).reshape(-1)
    sizes = [(1242, 375), (1224, 370), (1238, 374)]
    for i in range(n):
        w, h = sizes[i % 3]
        Image.fromarray(np.random.RandomState(i).randint(0, 256, (h, w, 3)).astype(np.uint8)).save(tmp_path / "image_2" / ("%06d.png" % i))
        (tmp_path / "label_2" / ("%06d.txt" % i)).write_text("".join(l + "\n" for l in S.synthetic_kitti_labels(40 + i, w, h, 9)))
        (tmp_path / "calib" / ("%06d.txt" % i)).write_text("P2: " + " ".join("%.12e" % v for v in P) + "\nP3: " + " ".join("%.12e" % v for v in P) + "\n")
    for s in splits:
        (tmp_path / "ImageSets" / (s + ".txt")).write_text("".join("%06d\n" % i for i in range(n)))
    return sizes


def test_dataset_and_loader_front_end_to_end_on_cpu(tmp_path, cpu_encoders):
    import random
    from monoflex_amd.config import get_cfg
    from monoflex_amd.data import DeviceLoader, InferenceSampler, IterationBatchSampler, KITTIDataset, TrainingSampler
    sizes = _kitti_dir(tmp_path)
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    train = KITTIDataset(cfg, str(tmp_path), is_train=True, device="cpu")
    img, tgt, idx = train[1]
    want = ["cls_ids", "target_centers", "keypoints", "keypoints_depth_mask", "dimensions", "locations", "calib", "reg_mask", "reg_weight",
            "offset_3D", "2d_bboxes", "pad_size", "rotys", "trunc_mask", "alphas", "orientations", "hm", "gt_bboxes", "occlusions",
            "truncations", "edge_len", "edge_indices"]
    assert tgt.fields() == want and img.shape == (3, 384, 1280) and idx == "000001" and tgt.size == (1280, 384)   # kitti.py:496-523 order
    # the flip coin flips image, labels and calibration together
    random.seed(3)
    raws = [train.load_raw(0) for _ in range(12)]
    assert {r.flip for r in raws} == {True, False}
    flipped = next(r for r in raws if r.flip)
    _, (t,), _, fields = train.encode_batch([flipped])
    ref = K.encode_sample(S.synthetic_kitti_labels(40, *sizes[0], 9), S.KITTI_P2, *sizes[0], do_flip=True)
    compare_fields({k: t.get_field(k).numpy() for k in ref if k in want and k != "calib"}, ref, "flipped")
    assert np.isclose(t.get_field("calib").c_u, sizes[0][0] - S.KITTI_P2[0][2] - 1) and np.allclose(fields["P"][0].numpy(), ref["P"])
    # val split: labels, no augmentation; test split: four fields only (kitti.py:287-299)
    val = KITTIDataset(cfg, str(tmp_path), is_train=False, device="cpu")
    want_objects = int(K.encode_sample(S.synthetic_kitti_labels(40, *sizes[0], 9), S.KITTI_P2, *sizes[0])["reg_mask"].sum())
    assert val.split == "val" and val.flip_p == 0 and int(val[0][1].get_field("reg_mask").sum()) == want_objects
    assert len(val[0][1]) == 0 and len(train[0][1]) in (want_objects, len(train[0][1]))    # ParamsList.__len__ counts objects in training mode only
    cfg2 = cfg.clone(); cfg2.DATASETS.TEST_SPLIT = "test"
    test = KITTIDataset(cfg2, str(tmp_path), is_train=False, device="cpu")
    assert test[2][1].fields() == ["pad_size", "calib", "edge_len", "edge_indices"] and int(test[2][1].get_field("edge_len")) == K.edge_indices(*sizes[2], K.pad_size(*sizes[2]))[1]
    # loaders: inference shard in order, iteration-based training batches from worker processes
    batches = list(DeviceLoader(val, batch_size=2, sampler=InferenceSampler(len(val))))
    assert [b["img_ids"] for b in batches] == [("000000", "000001"), ("000002",)] and batches[0]["images"].tensors.shape == (2, 3, 384, 1280)
    bs = IterationBatchSampler(TrainingSampler(len(train), seed=5), batch_size=2, num_iterations=3)
    out = list(DeviceLoader(train, batch_sampler=bs, num_workers=2))
    assert len(out) == 3 and all(len(b["targets"]) == 2 and b["fields"]["hm"].shape == (2, 3, 96, 320) for b in out)
    g = torch.Generator(); g.manual_seed(5)
    stream = torch.randperm(3, generator=g).tolist() + torch.randperm(3, generator=g).tolist()
    assert [i for b in out for i in b["img_ids"]] == ["%06d" % i for i in stream]
    with pytest.raises(NotImplementedError):
        c3 = cfg.clone(); c3.INPUT.HEATMAP_CENTER = "2D"
        KITTIDataset(c3, str(tmp_path), is_train=True)
