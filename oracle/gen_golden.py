#!/usr/bin/env python
"""oracle/gen_golden.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

Imports the reference's own Python (read-only, from /root/reference) and records small golden
fixtures under tests/golden/ that pin oracle/monoflex_ref.py.  The reference never travels to
the GPU box; these .npz files (inputs are re-derivable from seeds, outputs are stored) do.

What is genuinely the reference here: model/backbone/dla_dcn.py (DLA-34 wiring, DLAUp/IDAUp),
model/backbone/DCNv2/dcn_v2.py (offset/mask split, autograd Function), model/head/
detector_predictor.py (9 branches, edge fusion), model/head/detector_infer.py, model/anno_encoder.py,
model/layers/utils.py (NMS, top-K, decode), structures/params_3d.py, data/datasets/kitti_utils.py
(Calibration), data/datasets/kitti.py (get_edge_utils).

What is NOT the reference (recorded in every fixture's `meta`):
  * `_ext` (native DCN sampling + GEMM): the reference's C++ needs <TH/TH.h>, which torch 2.10
    does not ship, so it is unbuildable here without a stand-in header.  The module handed to the
    reference's dcn_v2.py is oracle/dcn_v2_ref.c.  DCN arithmetic is therefore pinned by the
    reference's known-answer tests (tests/test_oracle_dcn.py), not by these fixtures.
  * yacs / inplace_abn / torchvision / cv2 / shapely / ... are absent: yacs.config.CfgNode is served
    by monoflex_amd.config.CfgNode, inplace_abn.InPlaceABN by BatchNorm2d(eps=1e-5)+leaky_relu(0.01)
    (upstream semantics; parity unpinned at that boundary), the rest by MagicMock (never executed).
  * two in-process patches for torch>=1.7: torch.cuda.FloatTensor asserts and
    integer `/` in select_topk (floor division under the reference's pinned torch 1.4).
"""
import os
import sys
import tempfile
import types
from unittest.mock import MagicMock

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from monoflex_amd import synthetic as S
from monoflex_amd.config import CfgNode
from oracle import dcn_ref

GOLD = os.path.join(REPO, "tests", "golden")


def use_reference_packages():
    """The reference's top-level packages (`model`, `config`, `data`, ...) are namespace packages (no __init__.py), and a regular package of the same
    name anywhere on sys.path wins over a namespace portion -- the repository root holds such packages since r04 (the reference's import names aliased
    to monoflex_amd.*, monoflex_amd/_alias.py).  Bind the names to the REFERENCE's directories explicitly, so that everything imported below really is
    the reference's code (asserted in build_reference)."""
    import importlib.machinery
    import importlib.util
    for name in ("config", "model", "solver", "engine", "data", "utils", "structures"):
        spec = importlib.machinery.PathFinder.find_spec(name, [REF])
        assert spec is not None and name not in sys.modules, name
        if spec.loader is None:                                   # namespace package: would lose to the repository's regular package of that name
            sys.modules[name] = importlib.util.module_from_spec(spec)
        # (a regular package of the reference -- config, solver -- is found first anyway once REF precedes the repository on sys.path)


def install_stubs():
    yacs = types.ModuleType("yacs"); yacs_cfg = types.ModuleType("yacs.config")
    yacs_cfg.CfgNode = CfgNode; yacs.config = yacs_cfg
    sys.modules["yacs"], sys.modules["yacs.config"] = yacs, yacs_cfg

    class InPlaceABN(nn.BatchNorm2d):
        def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu",
                     activation_param=0.01):
            super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)
            self.slope = activation_param

        def forward(self, x):
            return F.leaky_relu(super().forward(x), self.slope)
    abn = types.ModuleType("inplace_abn"); abn.InPlaceABN = InPlaceABN
    sys.modules["inplace_abn"] = abn

    ext = types.ModuleType("_ext")
    ext.dcn_v2_forward = dcn_ref.dcn_v2_forward
    ext.dcn_v2_backward = dcn_ref.dcn_v2_backward

    def _no_psroi(*a, **k):
        raise RuntimeError("psroi pooling is not on the MonoFlex path")
    ext.dcn_v2_psroi_pooling_forward = ext.dcn_v2_psroi_pooling_backward = _no_psroi
    sys.modules["_ext"] = ext

    for name in ['torchvision', 'torchvision.ops', 'torchvision.ops.roi_align', 'torchvision.transforms', 'cv2',
                 'shapely', 'shapely.geometry', 'pycocotools', 'pycocotools.mask', 'iopath', 'iopath.common',
                 'iopath.common.file_io', 'numba', 'numba.cuda', 'skimage', 'skimage.transform', 'fvcore',
                 'tensorboardX', 'torch.utils.tensorboard', 'fire', 'matplotlib', 'matplotlib.pyplot', 'PIL',
                 'PIL.Image']:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = MagicMock(name=name)

    torch.cuda.FloatTensor = torch.FloatTensor                      # model/layers/utils.py:83,84,93
    _td = torch.Tensor.__truediv__

    def _truediv(a, b):                                              # torch-1.4 integer '/' == floor
        if (not a.is_floating_point()) and isinstance(b, int):
            return torch.div(a, b, rounding_mode='floor')
        return _td(a, b)
    torch.Tensor.__truediv__ = _truediv
    use_reference_packages()


def build_reference(out_w, out_h):
    sys.path.insert(0, REF)
    os.chdir(REF)
    from config import cfg
    cfg.merge_from_file(os.path.join(REF, "runs", "monoflex.yaml"))
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.PRETRAIN = False
    cfg.DATASETS.TEST_SPLIT = "test"
    cfg.INPUT.WIDTH_TRAIN, cfg.INPUT.HEIGHT_TRAIN = out_w * 4, out_h * 4
    from model.detector import KeypointDetector
    import model.detector as _md
    assert os.path.abspath(_md.__file__).startswith(REF + os.sep), "not the reference's model package: " + _md.__file__
    model = KeypointDetector(cfg).eval()
    return cfg, model


def reference_target(tgt):
    from structures.params_3d import ParamsList
    from data.datasets.kitti_utils import Calibration
    f = tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False)
    P = tgt["P"].reshape(-1)
    f.write("P2: " + " ".join("%.12e" % v for v in P) + "\n")
    f.write("P3: " + " ".join("%.12e" % v for v in P) + "\n")
    f.write("R0_rect: 1 0 0 0 1 0 0 0 1\n")
    f.write("Tr_velo_to_cam: 1 0 0 0 0 1 0 0 0 0 1 0\n")
    f.close()
    calib = Calibration(f.name)
    os.unlink(f.name)
    t = ParamsList(image_size=tgt["size"], is_train=False)
    t.add_field("pad_size", tgt["pad_size"].numpy())
    t.add_field("calib", calib)
    t.add_field("edge_len", tgt["edge_len"])
    t.add_field("edge_indices", tgt["edge_indices"].numpy())
    return t


def checksum(t):
    t = t.detach().double().flatten()
    n = t.numel()
    idx = torch.linspace(0, n - 1, 256).long()
    return dict(sum=float(t.sum()), abssum=float(t.abs().sum()), sq=float((t * t).sum()),
                idx=idx.numpy(), samples=t[idx].float().numpy(), shape=np.array(list(t.shape)))


MIN_GAP = 4e-4      # smallest allowed distance, in LOGIT units, between neighbours among the top-51 reference peaks of a full-size golden image
                    # (= 1e-4 in score units at sigmoid's steepest point, the bar VERDICT r5 item 4 names; in logit units it also applies to the
                    # default-bias image, whose scores live near 0.01 where the sigmoid is 25 x flatter)


def top_gap(logits, k=51):
    """Smallest gap between consecutive peaks among the k best of the reference's NMS-ed heat map (model/layers/utils.py:39-58: sigmoid, clamp,
    3x3 max-pool, `hmax == heat`), in logit units (score gap / s(1 - s)).  k = 51: the 50 detections plus the first one left out -- a near-tie at the
    cut changes the SET, not only the order.  The HIP path's fp32-grade modes reproduce the reference's logits to ~2.5e-5: a 4e-4 gap is 16 x that."""
    heat = torch.clamp(torch.sigmoid(logits.float()), min=1e-4, max=1 - 1e-4)
    hmax = torch.nn.functional.max_pool2d(heat[None], 3, 1, 1)[0]
    sc = torch.topk((heat * (hmax == heat).float()).flatten(), k).values.double()
    mid = 0.5 * (sc[:-1] + sc[1:])
    return float(((sc[:-1] - sc[1:]) / (mid * (1 - mid))).min())


def run_case(name, out_w, out_h, seeds, cls_bias, store_full, n_images=None):
    """`n_images` set: `seeds` is a candidate stream; the first n_images whose top-51 reference scores are pairwise >= MIN_GAP apart are kept (SURVEY
    section 7 'hard parts': a golden image must not contain a near-tie -- the order of a 1.7e-6 pair is decided by the summation order of a 50-layer fp32
    network, not by the detector, and every tile shape would round it its own way)."""
    cfg, model = build_reference(out_w, out_h)
    sd = S.synthetic_state_dict(model.state_dict(), seed=0, cls_bias=cls_bias)
    model.load_state_dict(sd)
    import model.head.detector_predictor as dp
    import model.head.detector_infer as di
    rec = {}
    orig_sig, orig_topk = dp.sigmoid_hm, di.select_topk

    def sig(x):
        rec["cls_logits"] = x.detach().clone()
        return orig_sig(x)

    def topk(hm, K=100):
        r = orig_topk(hm, K=K)
        rec["topk"] = [t.detach().clone() for t in r]
        return r
    dp.sigmoid_hm, di.select_topk = sig, topk
    model.backbone.base.register_forward_hook(lambda m, i, o: rec.__setitem__("base", [t.detach().clone() for t in o]))
    model.backbone.dla_up.register_forward_hook(lambda m, i, o: rec.__setitem__("dla_up", [t.detach().clone() for t in o]))
    model.backbone.register_forward_hook(lambda m, i, o: rec.__setitem__("feature", o.detach().clone()))
    model.heads.predictor.register_forward_hook(
        lambda m, i, o: rec.__setitem__("pred", {k: v.detach().clone() for k, v in o.items()}))

    out = {}
    meta = dict(case=name, out_w=out_w, out_h=out_h, seeds=[], weight_seed=0, cls_bias=cls_bias,
                ext="oracle/dcn_v2_ref.c (reference _ext unbuildable: TH/TH.h)",
                inplace_abn="stub BatchNorm2d(eps=1e-5)+leaky_relu(0.01)", torch=torch.__version__)
    kept, n = [], -1
    for seed in seeds:
        if n_images is not None and len(kept) == n_images:
            break
        img = S.synthetic_images(1, out_h * 4, out_w * 4, seed=seed)
        tgt = S.synthetic_target(out_w, out_h)
        with torch.no_grad():
            result, eval_utils, vis = model(img, [reference_target(tgt)])
        if n_images is not None:
            gap = top_gap(rec["cls_logits"][0])
            if gap < MIN_GAP:
                print(name, "seed", seed, "rejected: smallest logit gap among the top-51 reference peaks %.2e < %.0e" % (gap, MIN_GAP))
                continue
            out["img%d_top51_min_gap" % len(kept)] = np.array(gap)
        kept.append(seed)
        n += 1
        p = "img%d_" % n
        for i, t in enumerate(rec["base"]):
            for k, v in checksum(t).items():
                out[p + "base%d_%s" % (i, k)] = v
        for i, t in enumerate(rec["dla_up"]):
            for k, v in checksum(t).items():
                out[p + "dlaup%d_%s" % (i, k)] = v
        for k, v in checksum(rec["feature"]).items():
            out[p + "feature_" + k] = v
        logits, reg = rec["cls_logits"][0], rec["pred"]["reg"][0]
        sc, ind, cl, ys, xs = [t[0] for t in rec["topk"]]
        assert len(torch.unique(sc)) == len(sc), "tie among top-K scores: pick another seed"
        if store_full:
            out[p + "feature"] = rec["feature"][0].numpy()
            out[p + "cls_logits"] = logits.numpy()
            out[p + "reg"] = reg.numpy()
        else:
            pix = torch.unique(torch.cat((torch.linspace(0, out_w * out_h - 1, 512).long(), ind)))
            out[p + "pix"] = pix.numpy()
            out[p + "cls_logits_at"] = logits.reshape(3, -1)[:, pix].numpy()
            out[p + "reg_at"] = reg.reshape(reg.shape[0], -1)[:, pix].numpy()
            out[p + "feature_at"] = rec["feature"][0].reshape(64, -1)[:, pix].numpy()
        out[p + "topk_scores"], out[p + "topk_index"] = sc.numpy(), ind.numpy()
        out[p + "topk_cls"], out[p + "topk_ys"], out[p + "topk_xs"] = cl.numpy(), ys.numpy(), xs.numpy()
        out[p + "result"] = result.numpy()
        print(name, "img", n, "detections", tuple(result.shape), "top score %.4f" % float(sc[0]))
    dp.sigmoid_hm, di.select_topk = orig_sig, orig_topk
    meta["seeds"] = [int(s_) for s_ in kept]
    if n_images is not None:
        assert len(kept) == n_images, "candidate seeds exhausted"
        meta["min_logit_gap_top51"] = MIN_GAP
    out["meta"] = np.array(repr(meta))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)


DEPTH_MODES = ["hard", "mean", "direct", "keypoints_avg", "keypoints_center", "keypoints_02", "keypoints_13"]


def run_decode_cases():
    """Reference PostProcessor alone on synthetic head maps (cheap): includes an image with zero
    detections (detector_infer.py:106-113) and one where only some of the 50 slots pass 0.2."""
    cfg, model = build_reference(320, 96)
    post = model.heads.post_processor
    out = {}
    for n, (seed, shift) in enumerate([(7, 0.0), (8, -2.25), (9, -6.0), (10, 1.0)]):
        g = torch.Generator().manual_seed(seed)
        logits = torch.randn(1, 3, 96, 320, generator=g) * 0.8 - 2.0 + shift
        cls = torch.sigmoid(logits).clamp(1e-4, 1 - 1e-4)
        reg = torch.randn(1, 50, 96, 320, generator=g) * 0.7
        tgt = S.synthetic_target(320, 96)
        result, _, _ = post({"cls": cls.clone(), "reg": reg.clone()}, [reference_target(tgt)], test=True)
        out["case%d_seed" % n], out["case%d_shift" % n] = np.array(seed), np.array(shift)
        out["case%d_result" % n] = result.numpy()
        print("decode case", n, tuple(result.shape))
        if n > 1:
            continue
        # the other `output_depth` settings (detector_infer.py:149-198; engine/inference.py:154 walks them) on the same maps
        for mode in DEPTH_MODES:
            post.output_depth = mode
            r_m, _, _ = post({"cls": cls.clone(), "reg": reg.clone()}, [reference_target(tgt)], test=True)
            out["case%d_result_%s" % (n, mode)] = r_m.numpy()
        # 'oracle' (get_oracle_depths, :238-277) reads ground truth: every third 'mean' detection becomes an object whose box is the detection's
        # shifted by (1.5, -1) px (IoU stays above 0.5 for all but the thinnest boxes) at 0.93 x its mean depth, plus two objects no detection meets
        mean_rows = torch.from_numpy(out["case%d_result_mean" % n])[::3]
        gt_boxes = torch.cat((mean_rows[:, 2:6] + torch.tensor([1.5, -1.0, 1.5, -1.0]), torch.tensor([[5., 5., 30., 40.], [1100., 200., 1180., 260.]])))
        gt_cls = torch.cat((mean_rows[:, 0], torch.tensor([0., 1.]))).long()
        gt_depth = torch.cat((mean_rows[:, 11] * 0.93, torch.tensor([20., 35.])))
        G = gt_boxes.shape[0]
        t_o = reference_target(tgt)
        reg_mask = torch.zeros(G + 3, dtype=torch.uint8); reg_mask[:G] = 1        # padded target rows, as the dataset makes them
        pad_rows = lambda t: torch.cat((t, t.new_zeros((3,) + tuple(t.shape[1:]))))
        t_o.add_field("reg_mask", reg_mask)
        t_o.add_field("cls_ids", pad_rows(gt_cls))
        t_o.add_field("gt_bboxes", pad_rows(gt_boxes))
        t_o.add_field("locations", pad_rows(torch.stack((torch.zeros(G), torch.zeros(G), gt_depth), dim=1)))
        post.output_depth = "oracle"
        r_o, _, _ = post({"cls": cls.clone(), "reg": reg.clone()}, [t_o], test=True)
        post.output_depth = "soft"
        out["case%d_result_oracle" % n] = r_o.numpy()
        out["case%d_gt_boxes" % n], out["case%d_gt_cls" % n], out["case%d_gt_depth" % n] = gt_boxes.numpy(), gt_cls.numpy(), gt_depth.numpy()
        differs = float((torch.from_numpy(out["case%d_result_mean" % n])[:, 11] != r_o[:, 11]).float().mean())
        print("decode case", n, "modes", DEPTH_MODES, "oracle rows that left the mean: %.2f" % differs)
        assert 0.1 < differs < 0.9
    out["meta"] = np.array(repr(dict(case="decode_only", torch=torch.__version__,
                                     maps="logits=randn*0.8-2+shift; cls=clamp(sigmoid); reg=randn*0.7 (same generator)")))
    np.savez_compressed(os.path.join(GOLD, "decode_only.npz"), **out)


def reference_train_target(tgt):
    from monoflex_amd.structures.params_3d import TRAIN_FIELDS
    t = reference_target(tgt)
    t.is_train = True
    for k in TRAIN_FIELDS:
        t.add_field(k, np.asarray(tgt[k]))
    t.add_field("ori_img", np.zeros((4, 4, 3), np.uint8))           # stacked by prepare_targets, visualisation only
    return t


LOSS_CASES = {   # name -> list of (target seed, n_obj or None, focal scale of P)
    "b2": [(1, None, 1.0), (2, None, 1.0)],
    "b3_empty_middle_mixed_calib": [(3, 5, 1.0), (4, 0, 1.1), (5, 7, 1.2)],
    "b1_many": [(6, 30, 1.0)],
}


def loss_case_inputs(name):
    """Seeded predictions + synthetic training targets of one loss case (shared with tests/test_loss_golden.py)."""
    spec = LOSS_CASES[name]
    tg = []
    for seed, n_obj, fs in spec:
        P = np.array(S.KITTI_P2, dtype=np.float64).reshape(3, 4).copy()
        P[0, 0] *= fs; P[1, 1] *= fs
        tg.append(S.synthetic_train_target(seed, n_obj=n_obj, P=P))
    B = len(spec)
    g = torch.Generator().manual_seed(100 + B)
    reg = torch.randn(B, 50, 96, 320, generator=g) * 0.6
    cls = torch.sigmoid(torch.randn(B, 3, 96, 320, generator=g) * 0.8 - 2.5).clamp(1e-4, 1 - 1e-4)
    return tg, cls, reg


def run_loss_cases():
    """Reference Loss_Computation (model/head/detector_loss.py) on seeded predictions and synthetic training targets:
    the 11 loss values, the log MAEs, and the gradient of the summed loss w.r.t. the regression map at the object
    centres (it is zero elsewhere) and w.r.t. the heat map (checksums)."""
    cfg, _ = build_reference(320, 96)
    import model.head.detector_loss as dl
    dl.get_iou_3d = lambda a, b: a.new_zeros(a.shape[0])            # shapely is absent; log-only (detector_loss.py:333)
    out = {}
    for name in LOSS_CASES:
        tg, cls, reg = loss_case_inputs(name)
        evaluator = dl.Loss_Computation(cfg)
        cls, reg = cls.clone().requires_grad_(), reg.clone().requires_grad_()
        loss_dict, log_dict = evaluator({"cls": cls, "reg": reg}, [reference_train_target(t) for t in tg])
        sum(loss_dict.values()).backward()
        for k, v in loss_dict.items():
            out["%s/loss/%s" % (name, k)] = np.float64(v.item())
        for k, v in log_dict.items():
            out["%s/log/%s" % (name, k)] = np.float64(v)
        cen = torch.stack([torch.as_tensor(t["target_centers"]) for t in tg]).long()          # (B,40,2)
        gr = reg.grad.permute(0, 2, 3, 1)
        bi = torch.arange(len(tg)).view(-1, 1).expand(cen.shape[:2])
        out["%s/grad_reg_at_centres" % name] = gr[bi, cen[..., 1], cen[..., 0]].numpy()        # (B,40,50)
        out["%s/grad_reg_abssum" % name] = np.float64(reg.grad.abs().double().sum())
        cs = checksum(cls.grad)
        out["%s/grad_cls_samples" % name], out["%s/grad_cls_idx" % name] = cs["samples"], cs["idx"]
        out["%s/grad_cls_sum" % name] = np.float64(cs["sum"])
        print("loss case", name, {k: round(v.item(), 4) for k, v in loss_dict.items()})
    out["meta"] = np.array(repr(dict(case="loss", torch=torch.__version__, patches="get_iou_3d -> zeros (shapely absent)",
                                     inputs="oracle/gen_golden.py:loss_case_inputs (seeded; regenerated by the test)")))
    np.savez_compressed(os.path.join(GOLD, "loss.npz"), **out)



def serialization_cases():
    """(model keys, loaded keys) pairs for the checkpoint key-alignment golden: the HIP model's real key list against
    (a) itself, (b) a DDP-saved copy, (c) a trunk-only ImageNet file with its classifier, (d) a file carrying the
    pretrain-grown `backbone.base.fc.*`, (e) ambiguous suffixes (longest wins), (f) suffixes that cut a component."""
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.detector import KeypointDetector
    cfg = get_cfg(os.path.join(REPO, "runs", "monoflex.yaml"))
    cfg.MODEL.PRETRAIN = False
    keys = list(KeypointDetector(cfg).state_dict().keys())
    trunk = [k[len("backbone.base."):] for k in keys if k.startswith("backbone.base.")]
    return {
        "self": (keys, keys),
        "ddp": (keys, ["module." + k for k in keys]),
        "imagenet_trunk": (keys, trunk + ["fc.weight", "fc.bias"]),
        "with_fc": (keys[:40], keys[:40] + ["backbone.base.fc.weight", "backbone.base.fc.bias"]),
        "longest_wins": (["a.b.conv1.weight", "a.c.conv1.weight", "conv1.weight", "x.bias"],
                         ["conv1.weight", "b.conv1.weight", "weight", "y.bias", "ias"]),
        "cut_component": (["m.bn1.weight", "m.1.weight", "m.bn1.bias"], ["1.weight", "n1.bias"]),
        "partial_prefix": (["p.q.w", "p.r.w"], ["module.q.w", "module.r.w", "s.w"]),
    }


def run_serialization_cases():
    """Mapping produced by the reference's own loader (utils/model_serialization.py:8-78) -> tests/golden/serialization.json.gz.
    Values are the loaded key names, so after the call each model entry names the loaded key it took (or None)."""
    import json
    from collections import OrderedDict
    sys.path.insert(0, REF)
    from utils import model_serialization as ref_ms
    out = {}
    for name, (mk, lk) in serialization_cases().items():
        model_sd = OrderedDict((k, None) for k in mk)
        loaded = ref_ms.strip_prefix_if_present(OrderedDict((k, k) for k in lk), prefix="module.")
        stripped = {v: k for k, v in loaded.items()}             # original name -> name after the prefix strip
        ref_ms.align_and_update_state_dicts(model_sd, loaded)
        out[name] = {"model_keys": mk, "loaded_keys": lk, "taken": [model_sd[k] for k in mk],
                     "stripped": [stripped[k] for k in lk]}
    import gzip
    with gzip.GzipFile(os.path.join(GOLD, "serialization.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(out).encode())
    print("serialization.json.gz:", {k: sum(t is not None for t in v["taken"]) for k, v in out.items()})



# ---- KITTI input pipeline + training-target encoding -----------------------------------------------
KITTI_SIZES = [(1242, 375), (1224, 370), (1238, 374), (1241, 376)]
KITTI_FIELDS = ["hm", "cls_ids", "target_centers", "keypoints", "keypoints_depth_mask", "dimensions", "locations", "reg_mask",
                "reg_weight", "offset_3D", "2d_bboxes", "pad_size", "rotys", "trunc_mask", "alphas", "orientations", "gt_bboxes",
                "occlusions", "truncations", "edge_len", "edge_indices"]


def kitti_cases():
    """(name, image size, #label lines, flip, label seed, image seed)."""
    cases = []
    for i in range(12):
        w, h = KITTI_SIZES[i % 4]
        n = [6, 0, 14, 28, 9, 1, 40, 11, 3, 22, 8, 17][i]
        cases.append(("s%02d" % i, w, h, n, i % 2 == 1 or i == 6, 500 + i, 900 + i))
    return cases


def run_kitti_cases():
    """Runs the reference's KITTIDataset.__getitem__ (+ its flip augmentation, transforms) on a generated KITTI directory
    -> tests/golden/kitti_encode.npz. torchvision is absent: `data.transforms.transforms.F` is a two-function stand-in with
    torchvision's documented semantics (to_tensor: HWC uint8 -> CHW float32 / 255; normalize: (x - mean) / std) -- the
    image half of the fixture is pinned to that, not to torchvision itself."""
    import random
    from PIL import Image
    sys.path.insert(0, REF)
    os.chdir(REF)
    np.int = int                                                        # kitti.py:434,436 under numpy >= 1.24
    from config import cfg
    cfg.merge_from_file(os.path.join(REF, "runs", "monoflex.yaml"))
    import data.transforms.transforms as T
    T.F = types.SimpleNamespace(
        to_tensor=lambda img: torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255),
        normalize=lambda t, mean, std: (t - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1))
    from data.transforms import build_transforms
    from data.datasets.kitti import KITTIDataset
    from data.augmentations.augmentations import Compose, RandomHorizontallyFlip
    root = tempfile.mkdtemp(prefix="kitti_fake_")
    for d in ("image_2", "label_2", "calib", "ImageSets"):
        os.makedirs(os.path.join(root, d))
    cases = kitti_cases()
    P = np.asarray(S.KITTI_P2, dtype=np.float64).reshape(-1)
    out = dict(names=np.array([c[0] for c in cases]))
    for i, (name, w, h, n, flip, lseed, iseed) in enumerate(cases):
        img = np.random.RandomState(iseed).randint(0, 256, (h, w, 3)).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, "image_2", "%06d.png" % i))
        lines = S.synthetic_kitti_labels(lseed, w, h, n)
        with open(os.path.join(root, "label_2", "%06d.txt" % i), "w") as f:
            f.write("".join(l + "\n" for l in lines))
        with open(os.path.join(root, "calib", "%06d.txt" % i), "w") as f:
            f.write("P2: " + " ".join("%.12e" % v for v in P) + "\n")
            f.write("P3: " + " ".join("%.12e" % v for v in P) + "\n")
            f.write("R0_rect: 1 0 0 0 1 0 0 0 1\nTr_velo_to_cam: 1 0 0 0 0 1 0 0 0 0 1 0\n")
        out[name + "_labels"] = np.array("\n".join(lines))
        out[name + "_meta"] = np.array([w, h, int(flip), iseed])
    with open(os.path.join(root, "ImageSets", "train.txt"), "w") as f:
        f.write("".join("%06d\n" % i for i in range(len(cases))))
    ds = KITTIDataset(cfg, root, is_train=True, transforms=build_transforms(cfg, True), augment=True)
    kept = []
    for i, (name, w, h, n, flip, lseed, iseed) in enumerate(cases):
        ds.augmentation = Compose([RandomHorizontallyFlip(1.0 if flip else 0.0)])
        random.seed(i)
        img, target, idx = ds[i]
        assert idx == "%06d" % i and tuple(img.shape) == (3, 384, 1280)
        for k in KITTI_FIELDS:
            out[name + "_" + k] = np.asarray(target.get_field(k))
        out[name + "_P"] = np.asarray(target.get_field("calib").P, dtype=np.float64)
        cs = checksum(img)
        out[name + "_img_sum"] = np.array([cs["sum"], cs["abssum"], cs["sq"]])
        out[name + "_img_idx"], out[name + "_img_samples"] = cs["idx"], cs["samples"]
        kept.append(int(target.get_field("reg_mask").sum()))
    out["meta"] = np.array("reference KITTIDataset.__getitem__ (data/datasets/kitti.py) + RandomHorizontallyFlip + "
                           "ToTensor/Normalize stand-in; numpy %s; P2 = monoflex_amd.synthetic.KITTI_P2" % np.__version__)
    np.savez_compressed(os.path.join(GOLD, "kitti_encode.npz"), **out)
    import shutil
    shutil.rmtree(root)
    print("kitti_encode.npz: objects kept per sample", kept)



# ---- KITTI result files + AP evaluation -------------------------------------------------------------
def install_numba_emulation():
    """numba is absent: `numba.jit` becomes the identity (the reference's CPU loops run as plain Python) and `numba.cuda`
    a small SIMT emulator -- every CUDA thread of a block is a Python thread, `syncthreads` a barrier, `shared.array` one
    array per block and call site, `local.array` a fresh array -- so the reference's own rotate_iou.py kernels execute
    unmodified. Arithmetic then follows numpy's scalar rules (float32 op python-float stays float32) where numba would
    widen to float64: overlaps agree with a real numba run to float32 round-off, which the fixture's metadata states."""
    import threading

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    numba = types.ModuleType("numba")
    numba.jit = jit
    numba.float32, numba.float64, numba.int32, numba.int64 = np.float32, np.float64, np.int32, np.int64
    tls = threading.local()

    class Dim:
        def __init__(self, which):
            self.which = which
        x = property(lambda self: getattr(tls, self.which)[0])
        y = property(lambda self: getattr(tls, self.which)[1])

    class Dev(np.ndarray):
        def copy_to_host(self, dst, stream=None):
            dst[...] = self

    class Stream:
        def auto_synchronize(self):
            import contextlib
            return contextlib.nullcontext()

    class Kernel:
        def __init__(self, fn):
            self.fn = fn

        def __getitem__(self, cfg):
            grid, block = cfg[0], cfg[1]
            grid = tuple(grid) if isinstance(grid, (tuple, list)) else (grid,)
            grid = grid + (1,) * (2 - len(grid))

            def launch(*args):
                for bx in range(int(grid[0])):
                    for by in range(int(grid[1])):
                        shared, barrier, errs = [], threading.Barrier(block), []

                        def run(tx):
                            tls.blockIdx, tls.threadIdx, tls.shared, tls.nshared, tls.barrier = (bx, by), (tx, 0), shared, 0, barrier
                            try:
                                self.fn(*args)
                            except BaseException as e:          # noqa: BLE001
                                errs.append(e); barrier.abort()
                        ts = [threading.Thread(target=run, args=(t,)) for t in range(block)]
                        [t.start() for t in ts]; [t.join() for t in ts]
                        if errs:
                            raise errs[0]
            return launch

    lock = threading.Lock()

    def shared_array(shape, dtype):
        with lock:
            if tls.nshared == len(tls.shared):
                tls.shared.append(np.zeros(shape, dtype=dtype))
            a = tls.shared[tls.nshared]
        tls.nshared += 1
        return a

    def cuda_jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return Kernel(a[0])
        device = k.get("device", False)
        return (lambda f: f) if device else (lambda f: Kernel(f))
    cuda = types.ModuleType("numba.cuda")
    cuda.jit = cuda_jit
    cuda.local = types.SimpleNamespace(array=lambda shape, dtype: np.zeros(shape, dtype=dtype))
    cuda.shared = types.SimpleNamespace(array=shared_array)
    cuda.syncthreads = lambda: tls.barrier.wait()
    cuda.blockIdx, cuda.threadIdx = Dim("blockIdx"), Dim("threadIdx")
    cuda.select_device = lambda i: None
    cuda.stream = lambda: Stream()
    cuda.to_device = lambda arr, stream=None: np.array(arr).view(Dev)
    numba.cuda = cuda
    sys.modules["numba"], sys.modules["numba.cuda"] = numba, cuda


def eval_cases():
    """(image size, label seed, #label lines, detection seed) per image of the evaluation fixture."""
    return [((1242, 375), 700 + i, [10, 14, 6, 12, 9, 16, 0, 11, 13, 8, 15, 12][i], 800 + i) for i in range(12)]


def run_eval_cases():
    """The reference's result writer (evaluate.py:34-54) and its R40 / R11 evaluation (evaluate.py:17-32, eval.py,
    rotate_iou.py, kitti_common.py:294-349) on a generated label_2 folder and synthetic detections
    -> tests/golden/kitti_eval.npz."""
    install_numba_emulation()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from data.datasets.evaluation.kitti_object_eval_python import evaluate as ref_eval
    root = tempfile.mkdtemp(prefix="kitti_eval_")
    label_dir, result_dir = os.path.join(root, "label_2"), os.path.join(root, "data")
    os.makedirs(label_dir); os.makedirs(result_dir)
    out = {}
    cases = eval_cases()
    for i, ((w, h), lseed, n, dseed) in enumerate(cases):
        lines = S.synthetic_kitti_labels(lseed, w, h, n, z_range=(5, 38), occl_max=1)    # mostly easy/moderate objects
        with open(os.path.join(label_dir, "%06d.txt" % i), "w") as f:
            f.write("\n".join(lines))
        det = S.synthetic_detections(dseed, lines, w, h, recall=0.9) if i != 3 else np.zeros((0, 14), np.float32)    # one image without detections
        ref_eval.generate_kitti_3d_detection(torch.from_numpy(det), os.path.join(result_dir, "%06d.txt" % i))
        out["labels_%d" % i] = np.array("\n".join(lines))
        out["det_%d" % i] = det
        out["txt_%d" % i] = np.array(open(os.path.join(result_dir, "%06d.txt" % i)).read())
    split = os.path.join(root, "val.txt")
    with open(split, "w") as f:
        f.write("".join("%06d\n" % i for i in range(len(cases))))
    from data.datasets.evaluation.kitti_object_eval_python import kitti_common as ref_kc
    from data.datasets.evaluation.kitti_object_eval_python import eval as ref_ev
    dt_annos, gt_annos = ref_kc.get_label_annos(result_dir), ref_kc.get_label_annos(label_dir, list(range(len(cases))))
    for m in range(3):                                                  # per-image (num_dt, num_gt) overlaps: bbox, bev, 3d
        ovs = ref_ev.calculate_iou_partly(dt_annos, gt_annos, m)[0]
        for i, o in enumerate(ovs):
            out["ov%d_%d" % (m, i)] = np.asarray(o, dtype=np.float64)
    for metric in ("R40", "R11"):
        result, ret = ref_eval.evaluate(label_dir, result_dir, split, current_class=["Car", "Pedestrian", "Cyclist"], metric=metric)
        out["result_" + metric] = np.array(result)
        out["keys_" + metric] = np.array(sorted(ret.keys()))
        out["values_" + metric] = np.array([float(ret[k]) for k in sorted(ret.keys())])
        print(metric, {k: round(float(v), 3) for k, v in sorted(ret.items()) if "moderate" in k})
    # (the COCO-style path of the reference, get_coco_eval_result, passes a float `num` to np.linspace and fails on numpy >= 1.18)
    result, ret = ref_eval.evaluate(label_dir, result_dir, split, current_class=["Car"], score_thresh=0.5, metric="R40")
    out["result_thresh"], out["keys_thresh"] = np.array(result), np.array(sorted(ret.keys()))
    out["values_thresh"] = np.array([float(ret[k]) for k in sorted(ret.keys())])
    out["meta"] = np.array("reference evaluate.py/eval.py/rotate_iou.py/kitti_common.py executed with numba emulated in Python "
                           "(identity jit; SIMT emulation of numba.cuda; numpy scalar arithmetic); numpy %s" % np.__version__)
    np.savez_compressed(os.path.join(GOLD, "kitti_eval.npz"), **out)
    import shutil
    shutil.rmtree(root)



# ---- G6: one training step of the reference model -----------------------------------------------------------
TRAIN_STEP = dict(out_w=96, out_h=32, batch=2, weight_seed=3, seed0=20)
TRAIN_GRAD_KEYS = ["backbone.base.base_layer.0.weight", "backbone.base.level2.tree1.conv1.weight", "backbone.base.level3.tree2.root.bn.weight",
                   "backbone.base.level5.root.conv.weight", "backbone.dla_up.ida_0.proj_1.conv.weight",
                   "backbone.dla_up.ida_0.proj_1.conv.conv_offset_mask.weight", "backbone.dla_up.ida_2.node_3.conv.bias",
                   "backbone.ida_up.up_1.weight", "backbone.ida_up.node_2.actf.0.bias", "heads.predictor.class_head.0.weight",
                   "heads.predictor.class_head.2.bias", "heads.predictor.reg_heads.2.0.weight", "heads.predictor.reg_features.7.1.weight",
                   "heads.predictor.trunc_heatmap_conv.0.weight", "heads.predictor.trunc_offset_conv.3.bias"]


def train_step_inputs():
    """Seeded images + synthetic training targets of the train-step fixture (same generator as tests/test_gpu_train.py)."""
    c = TRAIN_STEP
    tg = [S.synthetic_train_target(c["seed0"] + i, out_w=c["out_w"], out_h=c["out_h"], n_obj=3 + i) for i in range(c["batch"])]
    imgs = S.synthetic_images(c["batch"], c["out_h"] * 4, c["out_w"] * 4, seed=c["seed0"])
    return imgs, tg


def run_train_step_case():
    """The reference KeypointDetector in training mode (tools/plain_train_net.py path: model(images, targets) -> loss dict ->
    summed loss -> backward) on a 128x384 input, B=2: the 11 losses, the global gradient norm, the parameters without
    gradient, and checksums + strided samples of 15 parameter gradients spanning trunk, DCN, up-sampling and heads."""
    c = TRAIN_STEP
    cfg, model = build_reference(c["out_w"], c["out_h"])
    import model.head.detector_loss as dl
    dl.get_iou_3d = lambda a, b: a.new_zeros(a.shape[0])
    model.load_state_dict(S.synthetic_state_dict(model.state_dict(), seed=c["weight_seed"], cls_bias=-1.0))
    model.train()
    imgs, tg = train_step_inputs()
    loss_dict, log_dict = model(imgs, [reference_train_target(t) for t in tg])
    total = sum(loss_dict.values())
    total.backward()
    out = {"loss/" + k: np.float64(v.item()) for k, v in loss_dict.items()}
    out["total"] = np.float64(total.item())
    grads = {n: p.grad for n, p in model.named_parameters()}
    out["no_grad"] = np.array(sorted(n for n, g in grads.items() if g is None))
    out["grad_norm"] = np.float64(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values() if g is not None)).item())
    out["norms"] = np.array([float(grads[n].double().norm()) if grads[n] is not None else -1.0 for n in sorted(grads)])
    out["norm_names"] = np.array(sorted(grads))
    for n in TRAIN_GRAD_KEYS:
        cs = checksum(grads[n])
        out["g/%s/samples" % n], out["g/%s/idx" % n] = cs["samples"], cs["idx"]
        out["g/%s/sum" % n], out["g/%s/abssum" % n] = np.float64(cs["sum"]), np.float64(cs["abssum"])
    bn = model.backbone.base.base_layer[1]
    out["bn/base_layer.running_mean"], out["bn/base_layer.running_var"] = bn.running_mean.numpy().copy(), bn.running_var.numpy().copy()
    out["meta"] = np.array(repr(dict(case="train_step", torch=torch.__version__, config=c,
                                     patches="get_iou_3d -> zeros; _ext = oracle/dcn_v2_ref.c; InPlaceABN = BN + leaky_relu(0.01)")))
    np.savez_compressed(os.path.join(GOLD, "train_step.npz"), **out)
    print("train step:", {k[5:]: round(float(v), 4) for k, v in out.items() if k.startswith("loss/")}, "grad norm", float(out["grad_norm"]),
          "no grad:", len(out["no_grad"]))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    install_stubs()
    which = sys.argv[1:] or ["small", "full", "decode", "loss"]
    if "train" in which:
        run_train_step_case()
    if "eval" in which:
        run_eval_cases()
    if "kitti" in which:
        run_kitti_cases()
    if "serialization" in which:
        run_serialization_cases()
    if "loss" in which:
        run_loss_cases()
    if "small" in which:
        run_case("e2e_small", 32, 16, seeds=(1000, 1001), cls_bias=-1.0, store_full=True)
    if "decode" in which:
        run_decode_cases()
    if "full" in which:
        # BASELINE configs[0]'s four seeded images (1000 ..; a seed whose top-51 scores hold a near-tie is skipped) and one image at the
        # reference's default class bias -log(1/0.01 - 1) (detector_predictor.py:43: nothing passes the 0.2 threshold there -- the zero-detection path at full size)
        run_case("e2e_full", 320, 96, seeds=range(1000, 1400), cls_bias=-1.0, store_full=False, n_images=4)
        run_case("e2e_full_default_bias", 320, 96, seeds=range(1000, 1400), cls_bias=-float(np.log(1 / 0.01 - 1)), store_full=False, n_images=1)
