"""The one-launch AdamW (csrc/adamw.hip, monoflex_amd/solver.MultiTensorAdamW) against torch.optim.AdamW's own fused step: same state layout, same
trajectory (reference solver/__init__.py:10-60 builds torch.optim.AdamW; engine/trainer.py:121 calls optimizer.step())."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 64, 3, 3), (27,), (5000,), (1,), (3, 7, 11), (256, 512, 1, 1), (4097,), (16, 3, 7, 7)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    # a parameter whose storage is not 16-byte aligned (a view one element into a buffer)
    buf = torch.randn(1000 + 1, generator=g).to(DEV)
    ps.append(torch.nn.Parameter(buf[1:]))
    return ps


def _opt(ps, capturable=True):
    lr = lambda v: torch.tensor(v, dtype=torch.float32, device=DEV)       # noqa: E731
    groups = [{"params": ps[::2], "lr": lr(3e-3)}, {"params": ps[1::2], "lr": lr(6e-3)}]
    return torch.optim.AdamW(groups, lr=lr(3e-3), weight_decay=1e-2, betas=(0.9, 0.99), fused=True, capturable=capturable)


def test_multi_tensor_adamw_follows_torch_fused_adamw():
    from monoflex_amd.solver import MultiTensorAdamW, optimizer_step
    a, b = _params(1), _params(1)
    oa, ob = _opt(a), _opt(b)
    assert MultiTensorAdamW.eligible(ob)
    found = torch.zeros((), dtype=torch.float32, device=DEV)
    oa.found_inf = ob.found_inf = found                                     # the loss scaler's hook (engine/trainer.LossScaler.attach)
    oa.grad_scale = None
    g = torch.Generator().manual_seed(9)
    for it in range(6):
        skip = it == 3                                                      # a skipped step: nothing may move, counters included
        found.fill_(1.0 if skip else 0.0)
        before = [p.detach().clone() for p in b]
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).to(DEV) * (10.0 ** (it - 3))
            pa.grad, pb.grad = gr.clone(), gr.clone()
        if it == 4:
            for o in (oa, ob):
                o.param_groups[1]["lr"].fill_(1e-3)                         # a scheduler writes the device scalar between steps
        oa.step()
        optimizer_step(ob)
        torch.cuda.synchronize()
        for i, (pa, pb) in enumerate(zip(a, b)):
            sa, sb = oa.state[pa], ob.state[pb]
            assert float(sa["step"]) == float(sb["step"]) == (it + 1 - (1 if it >= 3 else 0)), (it, i)
            for x, y, name in ((pa, pb, "param"), (sa["exp_avg"], sb["exp_avg"], "exp_avg"), (sa["exp_avg_sq"], sb["exp_avg_sq"], "exp_avg_sq")):
                err = float((x.detach() - y.detach()).abs().max() / x.detach().abs().max().clamp(min=1e-30))
                assert err < 2e-6, (it, i, name, err)
            if skip:
                assert torch.equal(pb, before[i])
    # the state is torch's own: it round-trips through state_dict() into a plain torch optimizer
    oc = _opt(_params(1))
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))
    assert all(torch.equal(oc.state[pc]["exp_avg"], ob.state[pb]["exp_avg"]) for pc, pb in zip(oc.param_groups[0]["params"], ob.param_groups[0]["params"]))


def test_multi_tensor_adamw_replays_from_a_graph():
    from monoflex_amd.solver import finish_capture, optimizer_step
    a, b = _params(2), _params(2)
    oa, ob = _opt(a), _opt(b)
    g = torch.Generator().manual_seed(3)
    grads = [torch.randn(p.shape, generator=g).to(DEV) for p in a]
    for pa, pb, gr in zip(a, b, grads):
        pa.grad, pb.grad = gr.clone(), gr.clone()
    optimizer_step(ob)                                                       # eager first step: state and tables exist
    optimizer_step(oa)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            optimizer_step(ob)
    finish_capture(ob)
    for it in range(3):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).to(DEV)
            pa.grad.copy_(gr)
            pb.grad.copy_(gr)                                                # (the captured launch reads the gradients at these addresses)
        optimizer_step(oa)
        graph.replay()
        torch.cuda.synchronize()
        for pa, pb in zip(a, b):
            assert torch.equal(pa, pb) and float(oa.state[pa]["step"]) == float(ob.state[pb]["step"]) == it + 2


def test_multi_tensor_adamw_is_not_taken_for_other_optimizers():
    from monoflex_amd.solver import MultiTensorAdamW
    ps = _params(4)
    assert not MultiTensorAdamW.eligible(torch.optim.SGD(ps, lr=0.1))
    assert not MultiTensorAdamW.eligible(torch.optim.AdamW(ps, lr=1e-3, fused=True))            # float lr, not capturable
    assert not MultiTensorAdamW.eligible(_opt([torch.nn.Parameter(torch.randn(4, dtype=torch.float64, device=DEV))]))
