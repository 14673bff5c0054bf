"""Checkpoint interop: key alignment against the mapping produced by the reference's own loader
(tests/golden/serialization.json.gz, made by oracle/gen_golden.py), save/resume round trips in the reference's file
format, DDP prefixes, weights-only files, the pretrain-grown classifier keys, and the ImageNet DLA-34 import."""
import gzip
import json
import os
from collections import OrderedDict

import pytest
import torch

from monoflex_amd.config import get_cfg
from monoflex_amd.model.detector import KeypointDetector
from monoflex_amd.solver import build_optimizer, build_scheduler
from monoflex_amd.utils import model_serialization as MS
from monoflex_amd.utils.check_point import Checkpointer, DetectronCheckpointer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.MODEL.PRETRAIN = False
    return cfg


def _model(seed):
    torch.manual_seed(seed)
    return KeypointDetector(_cfg())


def test_key_alignment_matches_reference_loader():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "serialization.json.gz")) as f:
        cases = json.loads(f.read().decode())
    assert len(cases) >= 7
    for name, c in cases.items():
        loaded = MS.strip_prefix_if_present(OrderedDict((k, k) for k in c["loaded_keys"]), "module.")
        assert list(loaded.keys()) == c["stripped"], name
        model_sd = OrderedDict((k, None) for k in c["model_keys"])
        MS.align_and_update_state_dicts(model_sd, loaded)
        assert [model_sd[k] for k in c["model_keys"]] == c["taken"], name


def test_save_resume_round_trip(tmp_path):
    cfg = _cfg()
    a = _model(1)
    opt = build_optimizer(a, cfg)
    sched = build_scheduler(opt, cfg, iters_per_epoch=5)
    for p in a.parameters():                                     # one synthetic optimizer step so that AdamW state exists
        p.grad = torch.full_like(p, 1e-3)
    opt.step(); sched.step()
    ck = DetectronCheckpointer(cfg, a, opt, sched, save_dir=str(tmp_path))
    path = ck.save("model_final", iteration=7)
    assert open(tmp_path / "last_checkpoint").read() == path and ck.has_checkpoint()
    raw = torch.load(path, map_location="cpu")
    assert set(raw) == {"model", "optimizer", "scheduler", "iteration"} and len(raw["model"]) == 478

    b = _model(2)
    opt_b = build_optimizer(b, cfg)
    sched_b = build_scheduler(opt_b, cfg, iters_per_epoch=5)
    rest = DetectronCheckpointer(cfg, b, opt_b, sched_b, save_dir=str(tmp_path)).load("ignored.pth")   # last_checkpoint wins
    assert rest == {"iteration": 7}
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    sa, sb = opt.state_dict(), opt_b.state_dict()
    assert sa["param_groups"] == sb["param_groups"]
    for i in sa["state"]:
        assert torch.equal(sa["state"][i]["exp_avg"], sb["state"][i]["exp_avg"])
    assert sched_b.last_epoch == sched.last_epoch


def test_optimizer_state_can_be_skipped_and_missing_file_is_scratch(tmp_path):
    cfg = _cfg()
    cfg.SOLVER.LOAD_OPTIMIZER_SCHEDULER = False
    a = _model(3)
    opt = build_optimizer(a, cfg)
    Checkpointer(a, opt, save_dir=str(tmp_path)).save("it10", iteration=10)
    b = _model(4)
    opt_b = build_optimizer(b, cfg)
    rest = DetectronCheckpointer(cfg, b, opt_b, save_dir=str(tmp_path)).load()
    assert "optimizer" in rest and rest["iteration"] == 10 and len(opt_b.state_dict()["state"]) == 0
    assert DetectronCheckpointer(cfg, b, save_dir=str(tmp_path / "nothing")).load(None) == {}
    with pytest.raises(RuntimeError):
        DetectronCheckpointer(cfg, b, save_dir=str(tmp_path / "nothing")).load("http://example.invalid/model.pth")


def test_ddp_prefix_bare_state_dict_and_extra_classifier_keys(tmp_path):
    a, b = _model(5), _model(6)
    sd = OrderedDict(("module." + k, v) for k, v in a.state_dict().items())
    sd["module.backbone.base.fc.weight"] = torch.zeros(1000, 512, 1, 1)      # grown by the reference's pretrained init
    sd["module.backbone.base.fc.bias"] = torch.zeros(1000)
    f = str(tmp_path / "weights_only.pth")
    torch.save(sd, f)
    rest = DetectronCheckpointer(_cfg(), b, save_dir=str(tmp_path)).load(f, use_latest=False)
    assert rest == {}
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    bad = OrderedDict(a.state_dict())
    bad["backbone.base.base_layer.0.weight"] = torch.zeros(16, 3, 3, 3)
    with pytest.raises(RuntimeError):                            # strict load: shape mismatches are errors
        MS.load_state_dict(b, bad)


def test_packed_weight_cache_is_dropped_on_load():
    a, b = _model(7), _model(8)
    conv_holder = [m for m in b.modules() if hasattr(m, "_packs")][0]
    conv_holder._packs["sentinel"] = object()
    MS.load_state_dict(b, a.state_dict())
    assert "sentinel" not in conv_holder._packs


def test_imagenet_dla34_import(tmp_path, monkeypatch):
    from monoflex_amd.model.backbone.dla_dcn import dla34
    torch.manual_seed(11)
    src = dla34(pretrained=False)
    sd = OrderedDict(src.state_dict())
    sd["fc.weight"], sd["fc.bias"] = torch.randn(1000, 512, 1, 1), torch.randn(1000)
    f = str(tmp_path / "dla34-ba72cf86.pth")
    torch.save(sd, f)
    torch.manual_seed(12)
    dst = dla34(pretrained=f)
    assert all(torch.equal(x, dst.state_dict()[k]) for k, x in src.state_dict().items())
    monkeypatch.setenv("MONOFLEX_DLA34_WEIGHTS", f)
    cfg = _cfg()
    cfg.MODEL.PRETRAIN = True
    m = KeypointDetector(cfg)
    assert torch.equal(m.backbone.base.level5.root.conv.weight, src.level5.root.conv.weight)
    assert len(m.state_dict()) == 478                            # no classifier is grown on the detection model
    monkeypatch.delenv("MONOFLEX_DLA34_WEIGHTS")
    with pytest.raises(RuntimeError):
        KeypointDetector(cfg)
    del sd["level3.tree1.tree1.conv1.weight"]
    torch.save(sd, f)
    with pytest.raises(RuntimeError):                            # the trunk loads strictly
        dla34(pretrained=f)
