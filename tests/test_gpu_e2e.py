"""GPU parity tests, end to end: the HIP KeypointDetector against (a) fixtures captured from the
reference's own Python (tests/golden) and (b) the CPU oracle on the same seeded inputs.

fp32 mode carries the north-star gate: |logits - reference| <= 1e-3 and identical top-K indices.
bf16 mode (the perf mode) reports its own deviation; asserted loosely."""
import ast
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
    assert np.array_equal(topk[:, 1].numpy().astype(np.int64), g[p + "topk_index"]), "top-K indices differ"
    assert np.array_equal(topk[:, 2].numpy(), g[p + "topk_cls"])
    assert np.array_equal(topk[:, 3].numpy(), g[p + "topk_ys"]) and np.array_equal(topk[:, 4].numpy(), g[p + "topk_xs"])
    assert np.abs(topk[:, 0].numpy() - g[p + "topk_scores"]).max() < 1e-4
    res = det[valid.bool()].numpy()
    assert res.shape == g[p + "result"].shape
    assert np.allclose(res, g[p + "result"], rtol=2e-3, atol=2e-2), np.abs(res - g[p + "result"]).max()
    return dl, dr


def test_e2e_small_vs_reference_golden_fp32():
    from monoflex_amd import synthetic as S
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_small.npz"))
    meta = ast.literal_eval(str(g["meta"]))
    ow, oh = meta["out_w"], meta["out_h"]
    m = _hip_model(meta["cls_bias"], "fp32", ow, oh)
    # both images in ONE batch: batched decode must equal the reference's per-image (B=1) decode
    imgs = torch.cat([S.synthetic_images(1, oh * 4, ow * 4, seed=s) for s in meta["seeds"]])
    det, topk, valid, hm = _run(m, imgs, [S.synthetic_target(ow, oh)] * len(meta["seeds"]))
    for n in range(len(meta["seeds"])):
        _check_against_golden(g, n, meta, hm[n], topk[n], det[n], valid[n], full=True)


def test_e2e_full_vs_reference_golden_fp32():
    from monoflex_amd import synthetic as S
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_full.npz"))
    meta = ast.literal_eval(str(g["meta"]))
    m = _hip_model(meta["cls_bias"], "fp32")
    imgs = S.synthetic_images(1, 384, 1280, seed=meta["seeds"][0])
    det, topk, valid, hm = _run(m, imgs, [S.synthetic_target(320, 96)])
    dl, dr = _check_against_golden(g, 0, meta, hm[0], topk[0], det[0], valid[0], full=False)
    print("full-size fp32 vs reference: max |dlogit| %.2e, max |dreg| %.2e" % (dl, dr))


def test_e2e_vs_oracle_other_seeds_fp32():
    """Fresh inputs (not in any fixture) against the CPU oracle, 96x192 frame, batch 3."""
    from monoflex_amd import synthetic as S
    from oracle import monoflex_ref as R
    ow, oh = 48, 24
    m = _hip_model(-1.0, "fp32", ow, oh)
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


def test_e2e_bf16_perf_mode_deviation():
    """bf16 perf mode: not a parity gate (SURVEY section 7 'hard parts'); report and bound the deviation."""
    from monoflex_amd import synthetic as S
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_full.npz"))
    meta = ast.literal_eval(str(g["meta"]))
    m = _hip_model(meta["cls_bias"], "bf16")
    imgs = S.synthetic_images(1, 384, 1280, seed=meta["seeds"][0])
    det, topk, valid, hm = _run(m, imgs, [S.synthetic_target(320, 96)])
    pix = torch.as_tensor(g["img0_pix"])
    logits = hm[0][..., :3].permute(2, 0, 1).reshape(3, -1)[:, pix].numpy()
    dl = np.abs(logits - g["img0_cls_logits_at"]).max()
    agree = len(set(topk[0][:, 1].numpy().astype(np.int64).tolist()) & set(g["img0_topk_index"].tolist())) / 50.0
    print("bf16 vs reference: max |dlogit| %.3e, top-K index agreement %.0f%%" % (dl, 100 * agree))
    assert np.isfinite(hm[..., :3].numpy()).all() and np.isfinite(hm[..., 8:58].numpy()).all()
    assert dl < 0.25 and agree >= 0.6


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
