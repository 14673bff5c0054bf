"""SURVEY 8c golden G6: one training step of the reference KeypointDetector (tests/golden/train_step.npz, made by
oracle/gen_golden.py train) against the oracle network in training mode + the loss module -- the 11 losses, the gradient-less
parameters, the global gradient norm and strided samples of 15 parameter gradients.  This pins the CPU side that the GPU
training tests (tests/test_gpu_train.py) compare the HIP kernels with."""
import os

import numpy as np
import pytest
import torch

from monoflex_amd import synthetic as S
from monoflex_amd.config import get_cfg
from monoflex_amd.model.head.detector_loss import make_loss_evaluator
from monoflex_amd.structures.params_3d import make_train_target
from oracle import monoflex_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "train_step.npz"))
CFG = dict(out_w=96, out_h=32, batch=2, weight_seed=3, seed0=20)          # oracle/gen_golden.py:TRAIN_STEP


@pytest.fixture(scope="module")
def step():
    c = CFG
    cfg = get_cfg(os.path.join(ROOT, "runs", "monoflex.yaml"))
    cfg.INPUT.WIDTH_TRAIN, cfg.INPUT.HEIGHT_TRAIN = c["out_w"] * 4, c["out_h"] * 4
    ref = R.KeypointDetectorRef()
    ref.load_state_dict(S.synthetic_state_dict(ref.state_dict(), seed=c["weight_seed"], cls_bias=-1.0))
    ref.train()
    tg = [S.synthetic_train_target(c["seed0"] + i, out_w=c["out_w"], out_h=c["out_h"], n_obj=3 + i) for i in range(c["batch"])]
    imgs = S.synthetic_images(c["batch"], c["out_h"] * 4, c["out_w"] * 4, seed=c["seed0"])
    ei = torch.stack([torch.as_tensor(t["edge_indices"]) for t in tg]).long()
    el = torch.as_tensor([int(t["edge_len"]) for t in tg]).long()
    om = ref.forward_maps(imgs, ei, el)
    loss_dict, _ = make_loss_evaluator(cfg)(om, [make_train_target(t) for t in tg])
    sum(loss_dict.values()).backward()
    return ref, loss_dict


def test_losses_match_reference_train_step(step):
    _, loss_dict = step
    names = [k[5:] for k in G.files if k.startswith("loss/")]
    assert sorted(loss_dict) == sorted(names) and len(names) == 11
    for k in names:
        a, b = float(loss_dict[k]), float(G["loss/" + k])
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (k, a, b)
    assert abs(float(sum(loss_dict.values())) - float(G["total"])) <= 2e-4 * float(G["total"])


def test_gradients_match_reference_train_step(step):
    ref, _ = step
    grads = {n: p.grad for n, p in ref.named_parameters()}
    dead = sorted(n for n, g in grads.items() if g is None)
    assert dead == sorted(G["no_grad"].tolist())
    assert len(G["no_grad"]) == 6 and all("level3.project" in n or "level4.project" in n for n in G["no_grad"].tolist())
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values() if g is not None)))
    assert abs(total - float(G["grad_norm"])) <= 1e-5 * float(G["grad_norm"])
    # every parameter's gradient norm (274 live ones), then element samples of 15 gradients.  Conv / DCN biases that feed a
    # BatchNorm have an analytically zero gradient: both sides hold round-off noise of ~1e-10 of the global norm there, so
    # an absolute slack of 1e-9 * global norm is granted on top of the 1e-4 relative bound.
    slack = 1e-9 * float(G["grad_norm"])
    for n, want in zip(G["norm_names"].tolist(), G["norms"].tolist()):
        if want < 0:
            continue
        got = float(grads[n].double().norm())
        assert abs(got - want) <= 1e-4 * want + slack, (n, got, want)
    keys = sorted({k.split("/")[1] for k in G.files if k.startswith("g/")})
    assert len(keys) == 15
    for n in keys:
        g = grads[n].detach().double().flatten()
        want = G["g/%s/samples" % n].astype(np.float64)
        got = g[torch.from_numpy(G["g/%s/idx" % n])].numpy()
        scale = float(np.abs(want).max())
        assert np.abs(got - want).max() <= 1e-4 * scale + slack, (n, float(np.abs(got - want).max()), scale)
        assert abs(float(g.abs().sum()) - float(G["g/%s/abssum" % n])) <= 1e-4 * float(G["g/%s/abssum" % n]) + slack * g.numel(), n


def test_bn_running_statistics_match_reference_train_step(step):
    ref, _ = step
    bn = ref.backbone.base.base_layer[1]
    assert np.allclose(bn.running_mean.numpy(), G["bn/base_layer.running_mean"], rtol=1e-5, atol=1e-6)
    assert np.allclose(bn.running_var.numpy(), G["bn/base_layer.running_var"], rtol=1e-5, atol=1e-6)
