"""Loss_Computation (static-shape, sync-free restatement) against fixtures recorded from the reference's own
model/head/detector_loss.py (oracle/gen_golden.py run_loss_cases): the 11 loss values, the log MAEs, and the gradient of
the summed loss w.r.t. both prediction maps.  The loss is device-agnostic torch code, so the golden check runs on CPU;
tests/test_gpu_train.py repeats one case on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def case_inputs(name):
    """Same seeded inputs as oracle/gen_golden.py:loss_case_inputs (restated here: gen_golden imports the reference)."""
    from monoflex_amd import synthetic as S
    cases = {"b2": [(1, None, 1.0), (2, None, 1.0)],
             "b3_empty_middle_mixed_calib": [(3, 5, 1.0), (4, 0, 1.1), (5, 7, 1.2)],
             "b1_many": [(6, 30, 1.0)]}
    tg = []
    for seed, n_obj, fs in cases[name]:
        P = np.array(S.KITTI_P2, dtype=np.float64).reshape(3, 4).copy()
        P[0, 0] *= fs
        P[1, 1] *= fs
        tg.append(S.synthetic_train_target(seed, n_obj=n_obj, P=P))
    B = len(tg)
    g = torch.Generator().manual_seed(100 + B)
    reg = torch.randn(B, 50, 96, 320, generator=g) * 0.6
    cls = torch.sigmoid(torch.randn(B, 3, 96, 320, generator=g) * 0.8 - 2.5).clamp(1e-4, 1 - 1e-4)
    return tg, cls, reg


def evaluator():
    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.head.detector_loss import Loss_Computation
    return Loss_Computation(get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml")))


def check_case(name, device):
    from monoflex_amd.structures.params_3d import make_train_target
    g = np.load(os.path.join(ROOT, "tests", "golden", "loss.npz"), allow_pickle=True)
    tg, cls, reg = case_inputs(name)
    cls, reg = cls.to(device).requires_grad_(), reg.to(device).requires_grad_()
    loss_dict, logs = evaluator()({"cls": cls, "reg": reg}, [make_train_target(t) for t in tg])
    sum(loss_dict.values()).backward()
    keys = [k.split("/")[-1] for k in g.files if k.startswith(name + "/loss/")]
    assert sorted(keys) == sorted(loss_dict.keys()) and len(keys) == 11
    for k in keys:
        ref = float(g["%s/loss/%s" % (name, k)])
        assert abs(float(loss_dict[k]) - ref) <= 2e-5 * max(1.0, abs(ref)), (k, float(loss_dict[k]), ref)
    for k in [k.split("/")[-1] for k in g.files if k.startswith(name + "/log/")]:
        ref = float(g["%s/log/%s" % (name, k)])
        assert abs(float(logs[k]) - ref) <= 1e-4 * max(1.0, abs(ref)), (k, float(logs[k]), ref)
    cen = torch.stack([torch.as_tensor(t["target_centers"]) for t in tg]).long()
    bi = torch.arange(len(tg)).view(-1, 1).expand(cen.shape[:2])
    gr = reg.grad.detach().cpu().permute(0, 2, 3, 1)
    want = g["%s/grad_reg_at_centres" % name]
    assert np.abs(gr[bi, cen[..., 1], cen[..., 0]].numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    assert abs(float(reg.grad.abs().double().sum()) - float(g["%s/grad_reg_abssum" % name])) <= 1e-4 * float(g["%s/grad_reg_abssum" % name])
    gc = cls.grad.detach().cpu().double().flatten()
    ws = g["%s/grad_cls_samples" % name]
    assert np.abs(gc[torch.as_tensor(g["%s/grad_cls_idx" % name])].float().numpy() - ws).max() <= 1e-5 * max(1.0, np.abs(ws).max())
    assert abs(float(gc.sum()) - float(g["%s/grad_cls_sum" % name])) <= 1e-4 * abs(float(g["%s/grad_cls_sum" % name]))


@pytest.mark.parametrize("name", ["b2", "b3_empty_middle_mixed_calib", "b1_many"])
def test_loss_matches_reference_golden(name):
    check_case(name, "cpu")


def test_loss_without_objects_is_finite():
    """All-empty batch: the reference raises; here every regression loss is 0 and the gradient is finite."""
    from monoflex_amd import synthetic as S
    from monoflex_amd.structures.params_3d import make_train_target
    tg = [S.synthetic_train_target(9, n_obj=0)]
    g = torch.Generator().manual_seed(1)
    reg = torch.randn(1, 50, 96, 320, generator=g).requires_grad_()
    cls = torch.sigmoid(torch.randn(1, 3, 96, 320, generator=g) - 3).clamp(1e-4, 1 - 1e-4).requires_grad_()
    ld, _ = evaluator()({"cls": cls, "reg": reg}, [make_train_target(t) for t in tg])
    sum(ld.values()).backward()
    assert all(torch.isfinite(v) for v in ld.values()) and float(ld["bbox_loss"]) == 0
    assert torch.isfinite(reg.grad).all() and torch.isfinite(cls.grad).all()
