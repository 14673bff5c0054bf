"""GPU parity tests, end to end: the HIP KeypointDetector against (a) fixtures captured from the
reference's own Python (tests/golden) and (b) the CPU oracle on the same seeded inputs.

fp32 mode carries the north-star gate: |logits - reference| <= 1e-3 and identical top-K indices, plus the per-stage
goldens (six DLA levels, four DLAUp outputs, the 64-channel feature).  bf16 mode (the benchmarked mode) is run at the
benchmarked shape (B=8, 1280x384) against the same reference goldens with measured bounds; what it measures is written to
gpurun_out/bf16_vs_reference.json (committed as profiles/r02_bf16_vs_reference.json) and repeated in the bench line."""
import ast
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hip_model(cls_bias=-1.0, dtype="fp32", out_w=320, out_h=96):
    from monoflex_amd import synthetic as S
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.detector import KeypointDetector
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.MODEL.PRETRAIN = False
    cfg.DATASETS.TEST_SPLIT = "test"
    cfg.MODEL.COMPUTE_DTYPE = dtype
    cfg.INPUT.WIDTH_TRAIN, cfg.INPUT.HEIGHT_TRAIN = out_w * 4, out_h * 4
    m = KeypointDetector(cfg).eval()
    m.load_state_dict(S.synthetic_state_dict(m.state_dict(), seed=0, cls_bias=cls_bias))
    return m.to(DEV)


def _run(m, imgs, tgts):
    from monoflex_amd.structures.params_3d import make_test_target
    targets = [make_test_target(t) for t in tgts]
    tg = m.device_targets(targets, DEV)
    with torch.no_grad():
        det, topk, valid, hm = m.detect_device(imgs.to(DEV), *tg)
    torch.cuda.synchronize()
    return det.cpu(), topk.cpu(), valid.cpu(), hm.cpu()


def _stages(m, imgs):
    """The tensors the reference's forward hooks recorded (oracle/gen_golden.py:run_case): base level outputs, DLAUp outputs,
    the backbone feature -- as NHWC device tensors of the whole batch."""
    from monoflex_amd.model.backbone import dla_dcn
    bb = m.backbone
    fuse = dla_dcn.FUSE_F1[0]
    dla_dcn.FUSE_F1[0] = False            # the 16-bit modes run stem + level0 + level1 as one kernel (level0's map never exists): per-stage goldens need it
    try:
        with torch.no_grad():
            base = bb.base(imgs.to(DEV), bb.compute_dtype)
            up = bb.dla_up(list(base))
            y = [up[i] for i in range(bb.last_level - bb.first_level)]
            bb.ida_up(y, 0, len(y))
        torch.cuda.synchronize()
    finally:
        dla_dcn.FUSE_F1[0] = fuse
    out = {"base%d" % i: t for i, t in enumerate(base)}
    out.update({"dlaup%d" % i: t for i, t in enumerate(up)})
    out["feature"] = y[-1]
    return out


def _stage_errors(g, n, stages, image=0):
    """Per stage: max |sample - golden sample| / max |golden sample| over the 256 strided samples, and the relative error
    of the absolute sum over the whole map (image `image` of the batch against golden image n)."""
    errs = {}
    for name, t in stages.items():
        p = "img%d_%s_" % (n, name)
        if p + "samples" not in g:
            continue
        flat = t[image].float().permute(2, 0, 1).reshape(-1).cpu().double()            # the golden indexes the NCHW map
        assert int(np.prod(g[p + "shape"])) == flat.numel(), name
        want = g[p + "samples"].astype(np.float64)
        got = flat[torch.as_tensor(g[p + "idx"])].numpy()
        errs[name] = (float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-6)),
                      abs(float(flat.abs().sum()) - float(g[p + "abssum"])) / float(g[p + "abssum"]))
    return errs


def S_target():
    from monoflex_amd import synthetic as S
    return S.synthetic_target(320, 96)


def _golden_batch(meta, batch, height=384, width=1280):
    """A batch whose first images are the golden's own (meta["seeds"], in order); the rest are fresh frames."""
    from monoflex_amd import synthetic as S
    seeds = list(meta["seeds"])[:batch]
    seeds += [2000 + i for i in range(batch - len(seeds))]
    return torch.cat([S.synthetic_images(1, height, width, seed=s_) for s_ in seeds]), min(batch, len(meta["seeds"]))


def _check_against_golden(g, n, meta, hm, topk, det, valid, full):
    p = "img%d_" % n
    logits = hm[..., :3].permute(2, 0, 1)
    reg = hm[..., 8:58].permute(2, 0, 1)
    if full:
        dl = np.abs(logits.numpy() - g[p + "cls_logits"]).max()
        dr = np.abs(reg.numpy() - g[p + "reg"]).max()
    else:
        pix = torch.as_tensor(g[p + "pix"])
        dl = np.abs(logits.reshape(3, -1)[:, pix].numpy() - g[p + "cls_logits_at"]).max()
        dr = np.abs(reg.reshape(50, -1)[:, pix].numpy() - g[p + "reg_at"]).max()
    assert dl <= 1e-3 and dr <= 1e-3, "logits differ from the reference by %.3e / %.3e (bar 1e-3)" % (dl, dr)
    # the identical SEQUENCE of peaks, at every batch size and in every parity mode: the full-size goldens hold no pair of peaks closer than 4e-4 in
    # logit units among their top 51 (oracle/gen_golden.py MIN_GAP; r05's fixture had a 1.7e-6 pair and this check a tolerance for it)
    assert np.array_equal(topk[:, 1].numpy().astype(np.int64), g[p + "topk_index"]), "top-K indices differ"
    assert np.array_equal(topk[:, 2].numpy(), g[p + "topk_cls"])
    assert np.array_equal(topk[:, 3].numpy(), g[p + "topk_ys"]) and np.array_equal(topk[:, 4].numpy(), g[p + "topk_xs"])
    assert np.abs(topk[:, 0].numpy() - g[p + "topk_scores"]).max() < 1e-4
    res = det[valid.bool()].numpy()
    assert res.shape == g[p + "result"].shape
    assert np.allclose(res, g[p + "result"], rtol=2e-3, atol=2e-2), np.abs(res - g[p + "result"]).max()
    return dl, dr


# the two modes that carry the north-star gate (<= 1e-3 on logits, identical top-K): "fp32" = f32 MFMA, "fp16x2" = fp32 activations with
# split-precision (fp16 hi + lo) MFMA operands (csrc/common.h f32s_t) -- same tests, same bounds
PARITY_MODES = ["fp32", "fp16x2"]


@pytest.mark.parametrize("mode", PARITY_MODES)
def test_e2e_small_vs_reference_golden_fp32(mode):
    from monoflex_amd import synthetic as S
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_small.npz"))
    meta = ast.literal_eval(str(g["meta"]))
    ow, oh = meta["out_w"], meta["out_h"]
    m = _hip_model(meta["cls_bias"], mode, ow, oh)
    # both images in ONE batch: batched decode must equal the reference's per-image (B=1) decode
    imgs = torch.cat([S.synthetic_images(1, oh * 4, ow * 4, seed=s) for s in meta["seeds"]])
    det, topk, valid, hm = _run(m, imgs, [S.synthetic_target(ow, oh)] * len(meta["seeds"]))
    stages = _stages(m, imgs)
    for n in range(len(meta["seeds"])):
        _check_against_golden(g, n, meta, hm[n], topk[n], det[n], valid[n], full=True)
        errs = _stage_errors(g, n, stages, image=n)
        assert len(errs) == 11 and all(e[0] <= 2e-4 and e[1] <= 1e-4 for e in errs.values()), errs


@pytest.mark.parametrize("mode,batch", [("fp32", 1), ("fp32", 8), ("fp16x2", 1), ("fp16x2", 8)])
def test_e2e_full_vs_reference_golden_fp32(mode, batch):
    """Full-size frames against the reference's goldens (BASELINE configs[0]'s four seeded images): every golden image in the batch
    (one at B = 1, all four at B = 8 -- the benchmarked shape) under the same gate in both parity modes: logits <= 1e-3, the IDENTICAL top-K
    sequence, (N, 14) rows, 11 stage goldens."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_full.npz"))
    meta = ast.literal_eval(str(g["meta"]))
    m = _hip_model(meta["cls_bias"], mode)
    imgs, ngold = _golden_batch(meta, batch)
    det, topk, valid, hm = _run(m, imgs, [S_target()] * batch)
    stages = _stages(m, imgs)
    worst = {"dl": 0.0, "dr": 0.0}
    for n in range(ngold):
        assert float(g["img%d_top51_min_gap" % n]) >= 4e-4                 # the fixture's own property (logit units)
        dl, dr = _check_against_golden(g, n, meta, hm[n], topk[n], det[n], valid[n], full=False)
        worst["dl"], worst["dr"] = max(worst["dl"], float(dl)), max(worst["dr"], float(dr))
        errs = _stage_errors(g, n, stages, image=n)
        assert len(errs) == 11 and all(e[0] <= 2e-4 and e[1] <= 1e-4 for e in errs.values()), (n, errs)
        pix = torch.as_tensor(g["img%d_pix" % n])
        feat = stages["feature"][n].float().permute(2, 0, 1).reshape(64, -1)[:, pix].cpu().numpy()
        assert np.abs(feat - g["img%d_feature_at" % n]).max() <= 2e-4 * max(1.0, np.abs(g["img%d_feature_at" % n]).max())
    print("full-size %s B=%d vs reference (%d golden images): max |dlogit| %.2e, max |dreg| %.2e" % (mode, batch, ngold, worst["dl"], worst["dr"]))
    if mode == "fp16x2":
        from monoflex_amd import lib as L_
        assert L_.f16x2_range_ok(), "an activation left fp16's range on its way into an MFMA operand pair"      # the mode's one precondition
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "%s_b%d_vs_reference.json" % (mode, batch)), "w") as f:
        json.dump({"shape": "B=%d, 1280x384, %s; golden images %s of tests/golden/e2e_full.npz" % (batch, mode, meta["seeds"][:ngold]),
                   "max_abs_dlogit": worst["dl"], "max_abs_dreg": worst["dr"], "topk_identical_sequence": True, "golden_images_checked": ngold},
                  f, indent=1, sort_keys=True)


@pytest.mark.parametrize("mode", PARITY_MODES)
def test_e2e_full_default_class_bias_vs_reference_golden(mode):
    """the reference's DEFAULT class bias -log(1/0.01 - 1) (detector_predictor.py:43) at full size -- scores near 0.01, three peaks
    above the 0.2 detection threshold: the partial-detection path (detector_infer.py:106-113) on a real frame, identical top-K sequence included."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_full_default_bias.npz"))
    meta = ast.literal_eval(str(g["meta"]))
    assert abs(meta["cls_bias"] + 4.59512) < 1e-4 and 0 < g["img0_result"].shape[0] < 50
    m = _hip_model(meta["cls_bias"], mode)
    imgs, _ = _golden_batch(meta, 1)
    det, topk, valid, hm = _run(m, imgs, [S_target()])
    _check_against_golden(g, 0, meta, hm[0], topk[0], det[0], valid[0], full=False)


@pytest.mark.parametrize("mode", PARITY_MODES)
def test_c5_batch32_hipgraph_fp32_rows_equal_reference_golden(mode):
    """BASELINE configs[4] per-GPU shape: batch 32 captured in ONE hipGraph (DLA + DCN + heads + top-K + decode), replayed;
    images 0-3 of the batch are the four golden images: logits <= 1e-3, identical top-K sequence, (N,14) rows -- the batched, graphed
    decode equals the reference's batch-1 eager decode.  Two further images of the batch are checked against their own B=1 run."""
    from monoflex_amd import synthetic as S
    from monoflex_amd.structures.params_3d import make_test_target
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_full.npz"))
    meta = ast.literal_eval(str(g["meta"]))
    m = _hip_model(meta["cls_bias"], mode)
    B = 32
    imgs, ngold = _golden_batch(meta, B)
    imgs = imgs.to(DEV)
    tg = m.device_targets([make_test_target(S.synthetic_target(320, 96)) for _ in range(B)], DEV)
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.detect_device(imgs, *tg)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = m.detect_device(imgs, *tg)
        graph.replay(); graph.replay()
    torch.cuda.synchronize()
    det, topk, valid, hm = [t.cpu() for t in out]
    for n in range(ngold):                                   # all four golden images, the identical top-K sequence included
        _check_against_golden(g, n, meta, hm[n], topk[n], det[n], valid[n], full=False)
    for n in (13, 31):
        d1, t1, v1, h1 = _run(m, imgs[n:n + 1].cpu(), [S.synthetic_target(320, 96)])
        assert torch.equal(topk[n][:, 1], t1[0][:, 1]) and torch.equal(valid[n], v1[0])
        assert torch.allclose(det[n], d1[0], rtol=2e-3, atol=2e-2)          # other tile shapes at B=32: fp32 sums reorder


@pytest.mark.parametrize("mode", PARITY_MODES)
def test_e2e_vs_oracle_other_seeds_fp32(mode):
    """Fresh inputs (not in any fixture) against the CPU oracle, 96x192 frame, batch 3."""
    from monoflex_amd import synthetic as S
    from oracle import monoflex_ref as R
    ow, oh = 48, 24
    m = _hip_model(-1.0, mode, ow, oh)
    ref = R.KeypointDetectorRef().eval()
    ref.load_state_dict(S.synthetic_state_dict(ref.state_dict(), seed=0, cls_bias=-1.0))
    imgs = S.synthetic_images(3, oh * 4, ow * 4, seed=77)
    tgts = [S.synthetic_target(ow, oh)] * 3
    det, topk, valid, hm = _run(m, imgs, tgts)
    dec, maps = ref.detect(imgs, [dict(t, calib=R.Calib(t["P"])) for t in tgts])
    for b in range(3):
        assert np.array_equal(topk[b][:, 1].numpy().astype(np.int64), dec[b]["indexs"].numpy())
        assert float((hm[b][..., 8:58].permute(2, 0, 1) - maps["reg"][b]).abs().max()) < 1e-3
        res = det[b][valid[b].bool()]
        assert res.shape == dec[b]["result"].shape
        assert torch.allclose(res, dec[b]["result"], rtol=2e-3, atol=2e-2)


def _perf_mode_vs_reference(dtype, stage_bound, abssum_bound, dlogit_bound, dreg_bound, topk_agree, row_bound, min_matched):
    from monoflex_amd import synthetic as S
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_full.npz"))
    meta = ast.literal_eval(str(g["meta"]))
    m = _hip_model(meta["cls_bias"], dtype)
    imgs, _ = _golden_batch(meta, 8)
    tgts = [S.synthetic_target(320, 96)] * 8
    det, topk, valid, hm = _run(m, imgs, tgts)
    errs = _stage_errors(g, 0, _stages(m, imgs))
    pix = torch.as_tensor(g["img0_pix"])
    logits = hm[0][..., :3].permute(2, 0, 1).reshape(3, -1)[:, pix].numpy()
    reg = hm[0][..., 8:58].permute(2, 0, 1).reshape(50, -1)[:, pix].numpy()
    dl = float(np.abs(logits - g["img0_cls_logits_at"]).max())
    dr = float(np.abs(reg - g["img0_reg_at"]).max() / max(1.0, np.abs(g["img0_reg_at"]).max()))
    # a peak = (class, pixel): one pixel can rank for two classes, and their decoded rows differ (class-mean dimensions)
    mine = topk[0][:, 2].numpy().astype(np.int64) * (1 << 20) + topk[0][:, 1].numpy().astype(np.int64)
    ref = g["img0_topk_cls"].astype(np.int64) * (1 << 20) + g["img0_topk_index"].astype(np.int64)
    agree = len(set(mine.tolist()) & set(ref.tolist())) / 50.0
    # decoded rows of the peaks both sides found (matched by class and heat-map index), relative to the row magnitudes
    ref_rows = {int(i): r for i, r in zip(ref[:len(g["img0_result"])], g["img0_result"])}
    rows = det[0][valid[0].bool()].numpy()
    deltas = [np.abs(r - ref_rows[int(i)]) / np.maximum(np.abs(ref_rows[int(i)]), 1.0) for i, r in zip(mine, rows) if int(i) in ref_rows]
    row_delta = float(np.max(deltas)) if deltas else float("nan")
    report = {"shape": "B=8, 1280x384, %s, image 0 = tests/golden/e2e_full.npz (reference KeypointDetector)" % dtype,
              "stage_sample_rel_err": {k: v[0] for k, v in errs.items()}, "stage_abssum_rel_err": {k: v[1] for k, v in errs.items()},
              "max_abs_dlogit": dl, "max_rel_dreg": dr, "topk_index_agreement": agree, "matched_rows": len(deltas),
              "max_rel_row_delta_matched": row_delta}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "%s_vs_reference.json" % dtype), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print("%s B=8 vs reference:" % dtype, json.dumps(report))
    assert np.isfinite(hm.numpy()).all()
    assert len(errs) == 11
    assert all(e[0] <= stage_bound and e[1] <= abssum_bound for e in errs.values()), errs
    assert dl <= dlogit_bound and dr <= dreg_bound and agree >= topk_agree, (dl, dr, agree)
    assert len(deltas) >= min_matched and row_delta <= row_bound


def test_e2e_bf16_benchmarked_shape_vs_reference_golden():
    """The benchmarked mode at the benchmarked shape (BASELINE configs[1]: B=8, 1280x384, bf16) against the REFERENCE's
    goldens: image 0 of the batch is the golden image.  bf16 is not the north-star parity mode (that is fp32, above); this
    pins how far it is from the reference, stage by stage, with bounds of ~2x the deviation measured on MI355X, and writes the
    measured numbers out."""
    _perf_mode_vs_reference("bf16", BF16_STAGE_BOUND, BF16_ABSSUM_BOUND, BF16_DLOGIT_BOUND, BF16_DREG_BOUND, BF16_TOPK_AGREE, BF16_ROW_BOUND, 25)


def test_e2e_fp16_benchmarked_shape_vs_reference_golden():
    """The fp16 inference mode (IEEE-half activations, fp16 MFMA at the bf16 rate) at the benchmarked shape against the same
    reference goldens: three more mantissa bits per stored activation than bf16 -> every bound ~8x tighter."""
    _perf_mode_vs_reference("fp16", *FP16_BOUNDS)


# bounds of the bf16 mode against the reference goldens: ~2x the deviation measured on MI355X (profiles/r02_bf16_vs_reference.json)
# measured (r02, MI355X): stage samples <= 0.045 (DLAUp outputs; <= 0.011 in the DLA trunk), abs-sums <= 0.0048, |dlogit| 0.151,
# rel dreg 0.028, 34 of the reference's 50 peaks found again (68 %), their decoded rows within 10.3 %
BF16_STAGE_BOUND, BF16_ABSSUM_BOUND = 0.09, 0.01
BF16_DLOGIT_BOUND, BF16_DREG_BOUND, BF16_TOPK_AGREE, BF16_ROW_BOUND = 0.30, 0.056, 0.6, 0.21
# fp16 mode: (stage samples, abs-sums, |dlogit|, rel dreg, top-50 agreement, matched-row delta, matched rows), ~2x the deviation measured
# on MI355X (profiles/r03_fp16_vs_reference.json: stages <= 7.1e-3, abs-sums <= 8.7e-4, |dlogit| 0.029, rel dreg 3.6e-3, all 50 of
# the reference's peaks found again, rows within 3.7 %): 4-8x closer to the reference than bf16 at the same speed
FP16_BOUNDS = (0.015, 0.002, 0.06, 0.008, 0.9, 0.08, 45)


def test_forward_surface_matches_reference_contract():
    """model(images, targets) -> (result (N,14), eval_utils, visualize_preds) as model/detector.py:36-37."""
    from monoflex_amd import synthetic as S
    from monoflex_amd.structures.params_3d import make_test_target
    m = _hip_model(-1.0, "fp32", 32, 16)
    img = S.synthetic_images(1, 64, 128, seed=1000).to(DEV)
    result, eval_utils, vis = m(img, [make_test_target(S.synthetic_target(32, 16))])
    assert result.dim() == 2 and result.shape[1] == 14 and result.shape[0] <= 50
    assert vis["heat_map"].shape == (1, 3, 16, 32)
    with pytest.raises(RuntimeError):
        m(img.cpu(), [make_test_target(S.synthetic_target(32, 16))])       # no CPU fallback


@pytest.mark.parametrize("dtype", ["bf16", "fp16x2"])
def test_inference_step_is_bitwise_repeatable(dtype):
    """No kernel of the forward + decode path accumulates with atomics, so five launches on the same inputs must agree bit for bit in every written
    channel of the head map, in the top-K table and in the decoded boxes; a difference is a race or an uncovered hardware hazard in some kernel
    (r06: tools/probes/infer_repeat.py runs the same check over all modes and batch sizes)."""
    import bench
    from monoflex_amd import synthetic as S
    from monoflex_amd.structures.params_3d import make_test_target
    dev = torch.device("cuda:0")
    model, _, _ = bench.build_model(dtype, dev)
    B = 2
    images = bench.bench_images(B, 0, dev)
    tg = model.device_targets([make_test_target(S.synthetic_target(320, 96)) for _ in range(B)], dev)
    with torch.no_grad():
        det0, topk0, valid0, hm0 = [t.clone() for t in model.detect_device(images, *tg)]
        for _ in range(4):
            det, topk, valid, hm = model.detect_device(images, *tg)
            torch.cuda.synchronize()
            assert torch.equal(hm[..., :3], hm0[..., :3]) and torch.equal(hm[..., 8:58], hm0[..., 8:58])
            assert torch.equal(topk, topk0) and torch.equal(valid, valid0) and torch.equal(det[valid0.bool()], det0[valid0.bool()])
