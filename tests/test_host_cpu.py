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
    assert cdll.mfx_abi_version() == 1


def test_ctypes_struct_layout_matches_header_sizes():
    # sizes of the descriptor structs as laid out by the C compiler for this ABI (LP64)
    from monoflex_amd import lib as L
    assert ctypes.sizeof(L.ConvDesc) == 8 * 8 + 22 * 4 + 16
    assert ctypes.sizeof(L.DcnDesc) == 7 * 8 + 17 * 4 + 4 + 8 + 16   # 4 bytes of padding before the trailing pointers
    assert ctypes.sizeof(L.CatDesc) == 9 * 8 + 18 * 4 + 2 * 4 + 5 * 8 + 8 * 4
    assert ctypes.sizeof(L.HeadsDesc) == 8 * 8 + 8 * 4 + 32 * 4


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
