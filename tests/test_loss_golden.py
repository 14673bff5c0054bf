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


# ---- the per-object loss kernel's arithmetic (csrc/object_loss_math.h), compiled for the host -------------------------------------
TERM_NAMES = ('bbox_loss', 'depth_loss', 'offset_loss', 'trunc_offset_loss', 'orien_loss', 'dims_loss', 'corner_loss', 'keypoint_loss',
              'keypoint_depth_loss', 'weighted_avg_depth_loss')
LOG_NAMES = ('2D_IoU', 'depth_loss', 'keypoint_depth_loss', 'depth_MAE', 'center_MAE', '02_MAE', '13_MAE', 'lower_MAE', 'hard_MAE',
             'soft_MAE', 'mean_MAE')


@pytest.fixture(scope="module")
def object_loss_shim(tmp_path_factory):
    import ctypes
    import subprocess
    from monoflex_amd import lib as L
    so = str(tmp_path_factory.mktemp("shim") / "libobject_loss_shim.so")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                        os.path.join(ROOT, "tests", "shim", "object_loss_host.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = ctypes.CDLL(so)
    lib.shim_object_loss.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(L.ObjectLossCfg),
                                                                              ctypes.c_void_p, ctypes.c_void_p]
    lib.shim_object_loss.restype = None
    return lib


def run_object_shim(shim, ev, reg_nchw, targets, ld=50, ch_off=0):
    """reg (B,50,H,W) -> (terms[10], logged[14], dreg (B,H,W,50) of the summed terms) through the host build of the kernel math."""
    import ctypes
    from monoflex_amd import lib as L
    _, tv = ev.prepare_targets(targets)
    rows = tv["object_rows"].contiguous()
    B, C, H, W = reg_nchw.shape
    reg = torch.zeros(B, H, W, ld)
    reg[..., ch_off:ch_off + C] = reg_nchw.permute(0, 2, 3, 1)
    reg = reg.contiguous()
    N = rows.shape[0]
    vals, G = torch.zeros(L.OBJ_VALUES), torch.zeros(N, L.OBJ_TERMS, 64)
    cfg = ev.object_loss_cfg()
    shim.shim_object_loss(reg.data_ptr(), B, H, W, ld, ch_off, rows.data_ptr(), N, ctypes.byref(cfg), vals.data_ptr(), G.data_ptr())
    dreg = torch.zeros(B, H, W, C)
    for n in range(N):
        if rows[n, 0] != 0:
            dreg[int(rows[n, 57]), int(rows[n, 3]), int(rows[n, 2])] += G[n].sum(0)[:C]
    return vals[:L.OBJ_TERMS], vals[L.OBJ_TERMS:], dreg, G, rows


@pytest.mark.parametrize("name", ["b2", "b3_empty_middle_mixed_calib", "b1_many"])
@pytest.mark.parametrize("layout", [(50, 0), (64, 8)])
def test_object_loss_kernel_math_matches_reference_golden(name, layout, object_loss_shim):
    """The forward-mode kernel arithmetic against the fixtures recorded from the reference's detector_loss.py: the ten regression
    terms, the logged means and the gradient at the object centres -- the same bounds as the tensor-op form above."""
    from monoflex_amd.structures.params_3d import make_train_target
    g = np.load(os.path.join(ROOT, "tests", "golden", "loss.npz"), allow_pickle=True)
    tg, cls, reg = case_inputs(name)
    ev = evaluator()
    terms, logged, dreg, G, rows = run_object_shim(object_loss_shim, ev, reg, [make_train_target(t) for t in tg], *layout)
    for i, k in enumerate(TERM_NAMES):
        ref = float(g["%s/loss/%s" % (name, k)])
        assert abs(float(terms[i]) - ref) <= 2e-5 * max(1.0, abs(ref)), (k, float(terms[i]), ref)
    have = dict(zip(LOG_NAMES, logged.tolist()))
    for k in [k.split("/")[-1] for k in g.files if k.startswith(name + "/log/")]:
        if k in have:
            ref = float(g["%s/log/%s" % (name, k)])
            assert abs(have[k] - ref) <= 1e-4 * max(1.0, abs(ref)), (k, have[k], ref)
    cen = torch.stack([torch.as_tensor(t["target_centers"]) for t in tg]).long()
    bi = torch.arange(len(tg)).view(-1, 1).expand(cen.shape[:2])
    want = g["%s/grad_reg_at_centres" % name]
    assert np.abs(dreg[bi, cen[..., 1], cen[..., 0]].numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    assert abs(float(dreg.abs().double().sum()) - float(g["%s/grad_reg_abssum" % name])) <= 1e-4 * float(g["%s/grad_reg_abssum" % name])
    assert float(G[rows[:, 0] == 0].abs().max() if (rows[:, 0] == 0).any() else 0.0) == 0.0           # empty slots: zero rows
    assert float(G[..., 50:].abs().max()) == 0.0


def test_object_loss_kernel_math_per_term_gradients_match_autograd(object_loss_shim):
    """Each term's OWN gradient row (what backward contracts with the incoming per-term gradients), against autograd of the
    tensor-op form, on a case with truncated objects, invisible keypoints and invalid keypoint-depth groups."""
    from monoflex_amd.structures.params_3d import make_train_target
    tg, cls, reg = case_inputs("b3_empty_middle_mixed_calib")
    ev = evaluator()
    ev.fused_object_loss = False
    targets = [make_train_target(t) for t in tg]
    terms, logged, dreg, G, rows = run_object_shim(object_loss_shim, ev, reg, targets)
    reg = reg.clone().requires_grad_()
    loss_dict, _ = ev({"cls": cls, "reg": reg}, targets)
    for i, k in enumerate(TERM_NAMES):
        gr, = torch.autograd.grad(loss_dict[k], reg, retain_graph=True, allow_unused=True)
        gr = torch.zeros_like(reg) if gr is None else gr
        want = torch.zeros(reg.shape[0], reg.shape[2], reg.shape[3], 50)
        for n in range(rows.shape[0]):
            if rows[n, 0] != 0:
                want[int(rows[n, 57]), int(rows[n, 3]), int(rows[n, 2])] += G[n, i, :50]
        err = (gr.permute(0, 2, 3, 1) - want).abs().max()
        assert float(err) <= 1e-5 * max(1.0, float(gr.abs().max())), (k, float(err))


def test_object_loss_kernel_math_without_objects(object_loss_shim):
    from monoflex_amd import synthetic as S
    from monoflex_amd.structures.params_3d import make_train_target
    reg = torch.randn(1, 50, 96, 320, generator=torch.Generator().manual_seed(1))
    terms, logged, dreg, G, rows = run_object_shim(object_loss_shim, evaluator(), reg, [make_train_target(S.synthetic_train_target(9, n_obj=0))])
    assert float(terms.abs().max()) == 0 and float(logged.abs().max()) == 0 and float(G.abs().max()) == 0
