"""CPU-side tests (no GPU): C ABI surface, config, host logic."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    from monoflex_amd import build, lib as L
    path = build.build_lib()
    assert os.path.exists(path)
    header = open(os.path.join(ROOT, "include", "monoflex_hip.h")).read()
    declared = set(re.findall(r"\b(mfx_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    cdll = ctypes.CDLL(path)                       # loads without a GPU
    for name in declared:
        assert hasattr(cdll, name), name
    cdll.mfx_abi_version.restype = ctypes.c_int
    assert cdll.mfx_abi_version() == 3                                   # r06: mfx_dcn_desc.w_pair_f16 (see MFX_ABI_VERSION in the header)


def test_ctypes_struct_layout_matches_header_sizes(tmp_path):
    """Every descriptor struct of include/monoflex_hip.h as the C compiler lays it out (gcc on the header itself) against its
    ctypes mirror in monoflex_amd/lib.py."""
    import subprocess
    from monoflex_amd import lib as L
    pairs = [("mfx_conv_desc", L.ConvDesc), ("mfx_dcn_desc", L.DcnDesc), ("mfx_cat_desc", L.CatDesc), ("mfx_heads_desc", L.HeadsDesc),
             ("mfx_pack_desc", L.PackDesc), ("mfx_object_loss_cfg", L.ObjectLossCfg), ("mfx_head_sparse_desc", L.HeadSparseDesc),
             ("mfx_gram_desc", L.GramDesc), ("mfx_kitti_desc", L.KittiDesc), ("mfx_kitti_eval_desc", L.KittiEvalDesc)]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void) {\n%s\nreturn 0; }\n'
                   % (os.path.join(ROOT, "include", "monoflex_hip.h"),
                      "\n".join('printf("%%zu\\n", sizeof(%s));' % n for n, _ in pairs)))
    exe = str(tmp_path / "sizes")
    r = subprocess.run(["gcc", "-std=c99", "-o", exe, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sizes = [int(v) for v in subprocess.run([exe], capture_output=True, text=True).stdout.split()]
    for (name, mirror), size in zip(pairs, sizes):
        assert ctypes.sizeof(mirror) == size, (name, ctypes.sizeof(mirror), size)
    assert ctypes.sizeof(L.ConvDesc) == 8 * 8 + 22 * 4 + 16 + 8 + 8 + 8 + 8  # ... + statistics pointer, copies (+pad), done pointer, paired fragments (r04)


def test_reference_yaml_drives_the_config():
    from monoflex_amd.config import get_cfg
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"), ["MODEL.COMPUTE_DTYPE", "bf16"])
    assert cfg.MODEL.HEAD.REGRESSION_CHANNELS == [[4], [2], [20], [3], [3], [8, 8], [1], [1]]
    assert cfg.DATASETS.DETECT_CLASSES == ("Car", "Pedestrian", "Cyclist")
    assert cfg.TEST.DETECTIONS_THRESHOLD == 0.2 and cfg.MODEL.HEAD.OUTPUT_DEPTH == "soft"
    assert cfg.MODEL.COMPUTE_DTYPE == "bf16"
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.SEED = 1


def test_model_state_dict_keys_match_reference_naming():
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.detector import KeypointDetector
    from oracle import monoflex_ref as R
    m = KeypointDetector(get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml")))
    a, b = m.state_dict(), R.KeypointDetectorRef().state_dict()
    assert len(a) == 478 and set(a) == set(b)      # the oracle loads the reference's own state_dict (gen_golden.py)
    assert all(a[k].shape == b[k].shape for k in a)
    assert "backbone.dla_up.ida_0.proj_1.conv.conv_offset_mask.weight" in a
    assert "heads.predictor.trunc_offset_conv.3.bias" in a


def test_product_path_refuses_cpu_tensors():
    from monoflex_amd import ops
    with pytest.raises(RuntimeError, match="no CPU"):
        ops.maxpool2x2(torch.zeros(1, 4, 4, 16))


def test_weight_packing_layouts():
    from monoflex_amd import ops
    w = torch.arange(2 * 16 * 3 * 3, dtype=torch.float32).reshape(2, 16, 3, 3)
    p = ops.pack_conv(w, torch.float32, None, None, stride=1, pad=1)
    assert p.w.shape == (16, 160) and p.Ck == 16 and p.K_pad == 160 and p.Cout == 2      # K padded to 128 bytes
    # k = tap*Cin + c
    assert float(p.w[1, 5 * 16 + 3]) == float(w[1, 3, 1, 2])
    ws = torch.arange(16 * 3 * 7 * 7, dtype=torch.float32).reshape(16, 3, 7, 7)
    pb = ops.pack_stem(ws, torch.bfloat16, torch.ones(16), torch.zeros(16))
    assert pb.w.shape == (16, 256) and (pb.kh, pb.kw, pb.Ck, pb.dil_w) == (7, 4, 8, 2)
    # super tap (th=2, j=1): elements [kw=2: c0..3][kw=3: c0..3], 4th channel zero
    row = pb.w[5].float()
    base = (2 * 4 + 1) * 8
    assert float(row[base + 1]) == float(ws[5, 1, 2, 2].bfloat16()) and float(row[base + 4 + 2]) == float(ws[5, 2, 2, 3].bfloat16())
    assert float(row[base + 3]) == 0.0 and float(row[(2 * 4 + 3) * 8 + 4]) == 0.0      # pad channel, pad tap kw=7
    pf = ops.pack_stem(ws, torch.float32, torch.ones(16), torch.zeros(16))
    assert pf.w.shape == (16, 224) and (pf.kh, pf.kw, pf.Ck, pf.dil_w) == (7, 7, 4, 1)


def test_optimizer_groups_match_reference_contract():
    """solver.build_optimizer: AdamW betas (0.9, 0.99), weight decay 1e-5, bias parameters at 2x lr; the merged two-group form
    and the reference's literal per-parameter form cover the same 280 parameters (solver/__init__.py:10-62)."""
    import os
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.detector import KeypointDetector
    from monoflex_amd.solver import build_optimizer, build_scheduler
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.MODEL.PRETRAIN = False
    m = KeypointDetector(cfg)
    opt = build_optimizer(m, cfg)
    assert [len(g["params"]) for g in opt.param_groups] == [168, 112]
    assert opt.param_groups[0]["lr"] == cfg.SOLVER.BASE_LR and opt.param_groups[1]["lr"] == 2 * cfg.SOLVER.BASE_LR
    assert all(g["betas"] == (0.9, 0.99) and g["weight_decay"] == cfg.SOLVER.WEIGHT_DECAY for g in opt.param_groups)
    ref_form = build_optimizer(m, cfg, per_parameter_groups=True)
    assert len(ref_form.param_groups) == 280
    bias_names = {n for n, _ in m.named_parameters() if "bias" in n}
    assert sum(1 for g in ref_form.param_groups if g["lr"] == 2 * cfg.SOLVER.BASE_LR) == len(bias_names) == 112
    sched = build_scheduler(opt, cfg, iters_per_epoch=10)
    f = sched.lr_lambdas[0]
    assert f(0) == 1.0 and abs(f(cfg.SOLVER.DECAY_EPOCH_STEPS[0] * 10) - cfg.SOLVER.LR_DECAY) < 1e-12
    assert abs(f(cfg.SOLVER.DECAY_EPOCH_STEPS[1] * 10) - cfg.SOLVER.LR_DECAY ** 2) < 1e-12


def test_synthetic_train_targets_are_consistent():
    """The seeded training targets carry every field of SURVEY Appendix D with the documented shapes/dtypes, are reproducible,
    and are geometrically consistent: target centre + 3D offset = projected centre, multi-bin angles decode back to alpha."""
    import numpy as np
    from monoflex_amd import synthetic as S
    from monoflex_amd.structures.params_3d import TRAIN_FIELDS, make_train_target
    a, b = S.synthetic_train_target(7, n_obj=12), S.synthetic_train_target(7, n_obj=12)
    for k in TRAIN_FIELDS:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    t = make_train_target(a)
    assert len(t) == int(a["reg_mask"].sum()) > 0
    shapes = {"hm": (3, 96, 320), "cls_ids": (40,), "target_centers": (40, 2), "keypoints": (40, 10, 3), "keypoints_depth_mask": (40, 3),
              "dimensions": (40, 3), "locations": (40, 3), "orientations": (40, 8), "2d_bboxes": (40, 4), "offset_3D": (40, 2)}
    for k, shp in shapes.items():
        assert tuple(np.asarray(a[k]).shape) == shp, k
    valid = a["reg_mask"] > 0
    assert float(a["hm"].max()) == 1.0 and (a["hm"] >= 0).all()
    assert (np.abs(a["offset_3D"][valid & (a["trunc_mask"] == 0)]) <= 1.0 + 1e-6).all()          # sub-pixel offsets for inside objects
    centers = np.array([0, np.pi / 2, np.pi, -np.pi / 2])
    for i in np.nonzero(valid)[0]:
        o = a["orientations"][i]
        assert o[:4].sum() >= 1
        for j in np.nonzero(o[:4])[0]:
            d = (centers[j] + o[4 + j] - a["alphas"][i] + np.pi) % (2 * np.pi) - np.pi
            assert abs(d) < 1e-5


def test_ctypes_layouts_equal_the_c_compilers(tmp_path):
    """Every descriptor struct of include/monoflex_hip.h: sizeof and the offset of each member as laid out by gcc must
    equal the ctypes mirror in monoflex_amd/lib.py (a silent mismatch would hand kernels the wrong pointers)."""
    import subprocess
    from monoflex_amd import lib as L
    pairs = {"mfx_conv_desc": L.ConvDesc, "mfx_cat_desc": L.CatDesc, "mfx_dcn_desc": L.DcnDesc, "mfx_heads_desc": L.HeadsDesc,
             "mfx_kitti_desc": L.KittiDesc, "mfx_kitti_eval_desc": L.KittiEvalDesc, "mfx_gram_desc": L.GramDesc,
             "mfx_head_sparse_desc": L.HeadSparseDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "monoflex_hip.h"', 'int main(void) {']
    for cname, ct in pairs.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for field, _ in ct._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, field, cname, field))
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    r = subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr                              # also proves the header is plain C and names every member
    out = subprocess.run([exe], capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for line in filter(None, out):
        cname, field, value = line.split(" ")
        ct = pairs[cname]
        expect = ctypes.sizeof(ct) if field == "sizeof" else getattr(ct, field).offset
        assert int(value) == expect, (cname, field, value, expect)
        seen += 1
    assert seen == sum(len(ct._fields_) + 1 for ct in pairs.values())


def test_c_abi_argument_errors_need_no_gpu():
    """Error behaviour of the C boundary: bad arguments are rejected before any device work, with a code and a message
    (the reference raises through AT_ASSERTM / python asserts, SURVEY 8b 'Errors')."""
    from monoflex_amd import lib as L
    lib = L.load()
    null = ctypes.c_void_p(None)
    assert lib.mfx_set_option(b"no_such_option", 1) != 0 and b"unknown option" in lib.mfx_last_error()
    assert lib.mfx_kitti_encode_targets(None, null) != 0 and b"null descriptor" in lib.mfx_last_error()
    d = L.KittiDesc()
    d.B, d.max_objs, d.in_w, d.in_h, d.down, d.num_classes = 1, 40, 1281, 384, 4, 3         # width not divisible by the stride
    assert lib.mfx_kitti_encode_targets(ctypes.byref(d), null) != 0 and b"bad sizes" in lib.mfx_last_error()
    d.in_w = 1280                                                                             # sizes fine, pointers missing
    assert lib.mfx_kitti_encode_targets(ctypes.byref(d), null) != 0 and b"pointer" in lib.mfx_last_error()
    assert lib.mfx_kitti_preprocess_u8(None, None, None, None, None, 1, 1280, 384, None, None, null) != 0
    e = L.KittiEvalDesc()
    for fn in (lib.mfx_kitti_eval_overlaps, lib.mfx_kitti_eval_match_pass1, lib.mfx_kitti_eval_match_pass2):
        assert fn(None, null) != 0
        assert fn(ctypes.byref(e), null) != 0 and b"bad sizes" in lib.mfx_last_error()      # B == 0
    e.B, e.num_classes, e.num_k = 1, 3, 2
    assert lib.mfx_kitti_eval_overlaps(ctypes.byref(e), null) != 0 and b"pointer" in lib.mfx_last_error()
    assert lib.mfx_decode_topk(None, 0, 0, 0, 3, 1, 96, 320, 50, None, None, None, 0, null) != 0
    assert lib.mfx_heads_fused(None, null) != 0 and lib.mfx_conv2d_nhwc(None, null) != 0 and lib.mfx_dcn_nhwc(None, null) != 0
    with pytest.raises(RuntimeError, match="mfx_conv2d_nhwc failed"):
        L.check(lib.mfx_conv2d_nhwc(None, null), "mfx_conv2d_nhwc")


def test_product_code_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import oracle/; the package itself must not."""
    import re
    imp = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b|from\s+\.+\s*oracle\b)|importlib\.import_module\([\"\']oracle", re.M)
    offenders = []
    for d, _, files in os.walk(os.path.join(ROOT, "monoflex_amd")):
        for f in files:
            if f.endswith(".py") and imp.search(open(os.path.join(d, f), errors="ignore").read()):
                offenders.append(os.path.relpath(os.path.join(d, f), ROOT))
    assert offenders == [], offenders
    bench = open(os.path.join(ROOT, "bench.py")).read()
    spans = []
    for fn in ("def cpu_baseline", "def cpu_train_baseline"):           # the two `cpu_baseline` legs (inference / training step)
        start = bench.index(fn)
        spans.append((start, bench.index("\ndef ", start + 10)))
    hits = [m.start() for m in imp.finditer(bench)]
    assert hits and all(any(a <= h < b for a, b in spans) for h in hits), "bench.py may use the oracle only inside its cpu_baseline legs"
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    s0 = entry.index("def smoke")
    assert all(h >= s0 or "dcn_ref" in entry[h:h + 80] for h in [m.start() for m in imp.finditer(entry)])   # build() only compiles the checker
