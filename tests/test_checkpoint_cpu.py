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


def _stepped(model, cfg, per_parameter_groups):
    opt = build_optimizer(model, cfg, per_parameter_groups=per_parameter_groups)
    sched = build_scheduler(opt, cfg, iters_per_epoch=5)
    g = torch.Generator().manual_seed(5)
    for p in model.parameters():
        p.grad = torch.randn(p.shape, generator=g) * 1e-3
    opt.step(); sched.step()
    return opt, sched


def test_reference_layout_optimizer_state_loads_into_merged_groups(tmp_path):
    """a checkpoint whose optimizer has the reference's one-group-per-parameter layout (solver/__init__.py:10-25:
    280 groups; 282 with backbone.base.fc.*) resumes into this build's two merged groups, mapped by parameter name."""
    cfg = _cfg()
    a = _model(1)
    opt_ref, sched_ref = _stepped(a, cfg, per_parameter_groups=True)          # the reference's literal layout
    ref_sd = opt_ref.state_dict()
    assert len(ref_sd["param_groups"]) == 280
    # grow it to the 282-group form: two extra groups for fc.weight / fc.bias right after the last backbone.base.* parameter
    names = [n for n, p in a.named_parameters() if p.requires_grad]
    last_base = max(i for i, n in enumerate(names) if n.startswith("backbone.base."))
    grown = {"state": {}, "param_groups": []}
    for j in range(282):
        src = j if j <= last_base else (None if j <= last_base + 2 else j - 2)
        if src is None:
            grown["param_groups"].append(dict(ref_sd["param_groups"][0], params=[j]))
            grown["state"][j] = {"step": torch.tensor(1.0), "exp_avg": torch.zeros(3), "exp_avg_sq": torch.zeros(3)}
        else:
            grown["param_groups"].append(dict(ref_sd["param_groups"][src], params=[j]))
            grown["state"][j] = ref_sd["state"][src]
    for tag, sd in (("280", ref_sd), ("282", grown)):
        path = str(tmp_path / ("ref_%s.pth" % tag))
        torch.save({"model": a.state_dict(), "optimizer": sd, "scheduler": sched_ref.state_dict(), "iteration": 3}, path)
        b = _model(2)
        opt_b = build_optimizer(b, cfg)
        sched_b = build_scheduler(opt_b, cfg, iters_per_epoch=5)
        assert len(opt_b.param_groups) == 2
        rest = DetectronCheckpointer(cfg, b, opt_b, sched_b, save_dir=str(tmp_path / "none")).load(path, use_latest=False)
        assert rest == {"iteration": 3}
        pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
        for n in ("backbone.base.level2.tree1.conv1.weight", "backbone.dla_up.ida_0.proj_1.conv.bias", "heads.predictor.class_head.2.bias"):
            sa, sb = opt_ref.state[pa[n]], opt_b.state[pb[n]]
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), (tag, n)
        assert opt_b.param_groups[0]["lr"] == cfg.SOLVER.BASE_LR and opt_b.param_groups[1]["lr"] == cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR
        assert sched_b.state_dict()["last_epoch"] == sched_ref.state_dict()["last_epoch"] and len(sched_b.base_lrs) == 2
        # the resumed optimizer continues exactly like the reference-layout one
        g = torch.Generator().manual_seed(9)
        for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            gr = torch.randn(p.shape, generator=g) * 1e-3
            p.grad, q.grad = gr.clone(), gr.clone()
        if tag == "280":
            opt_ref.step(); opt_b.step()
            for n in pa:
                assert torch.allclose(pa[n], pb[n], rtol=0, atol=1e-7), n


def test_saved_optimizer_state_is_in_the_reference_layout(tmp_path):
    """What this build writes, a reference-layout optimizer (280 groups) loads with plain load_state_dict()."""
    cfg = _cfg()
    a = _model(1)
    opt, sched = _stepped(a, cfg, per_parameter_groups=False)
    path = DetectronCheckpointer(cfg, a, opt, sched, save_dir=str(tmp_path)).save("m", iteration=1)
    raw = torch.load(path, map_location="cpu")
    assert len(raw["optimizer"]["param_groups"]) == 280 and len(raw["scheduler"]["base_lrs"]) == 280
    assert all(len(g["params"]) == 1 for g in raw["optimizer"]["param_groups"])
    b = _model(2)
    opt_ref = build_optimizer(b, cfg, per_parameter_groups=True)
    sched_ref = build_scheduler(opt_ref, cfg, iters_per_epoch=5)
    opt_ref.load_state_dict(raw["optimizer"])                   # what the reference's Checkpointer.load does (check_point.py:68)
    sched_ref.load_state_dict(raw["scheduler"])
    n = "backbone.base.level3.tree2.root.bn.bias"
    assert torch.equal(opt.state[dict(a.named_parameters())[n]]["exp_avg"], opt_ref.state[dict(b.named_parameters())[n]]["exp_avg"])
    lrs = {g["lr"] for g in opt_ref.param_groups}
    assert lrs == {cfg.SOLVER.BASE_LR, cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR}
    # on a GPU the learning rates are device scalars (capturable AdamW), so the scheduler's per-group lists hold tensors;
    # the file must hold the reference's plain floats (no tensor anywhere in the scheduler part)
    from monoflex_amd.utils import check_point as CP
    sd = dict(sched.state_dict())
    for k in ("base_lrs", "_last_lr"):
        sd[k] = [torch.tensor(float(v)) for v in sd[k]]
    out = CP.scheduler_state_to_reference(a, opt, sd)
    for k in ("base_lrs", "_last_lr"):
        assert len(out[k]) == 280 and all(type(v) is float for v in out[k]), k
    assert not any(torch.is_tensor(v) for v in raw["scheduler"].values())
    assert not any(torch.is_tensor(x) for v in raw["scheduler"].values() if isinstance(v, (list, tuple)) for x in v)
