"""GramRegHeadsFn forward + backward captured in a hipGraph vs the same call run eagerly (same inputs): outputs and gradients must agree."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch
from monoflex_amd import lib
from monoflex_amd.gram_heads import gram_reg_heads
from monoflex_amd.model.head.detector_predictor import InPlaceABN
lib.load()
DEV = "cuda"
for dt in (torch.float32, torch.bfloat16):
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, C, N = 2, 32, 96, 64, 256, 12
    ks, offs = (4, 20, 3), (0, 6, 26)
    rows = torch.zeros(N, 72); rows[:, 0] = 1.0
    rows[:, 57] = torch.randint(0, B, (N,), generator=g).float(); rows[:, 2] = torch.randint(0, W, (N,), generator=g).float(); rows[:, 3] = torch.randint(0, H, (N,), generator=g).float()
    rows = rows.to(DEV)
    x = (torch.randn(B, H, W, Cin, generator=g) * 0.8 + 0.1).to(dt).to(DEV).requires_grad_()
    wt = [(torch.randn(C, Cin, 3, 3, generator=g) / 24.0).to(DEV).requires_grad_() for _ in ks]
    abns = [InPlaceABN(C).to(DEV) for _ in ks]
    w2 = [(torch.randn(k, C, 1, 1, generator=g) * 0.1).to(DEV).requires_grad_() for k in ks]
    b2 = [(torch.randn(k, generator=g) * 0.1).to(DEV).requires_grad_() for k in ks]
    dout = torch.randn(N, 50, generator=g).to(DEV)
    erows = torch.randint(0, B * H * W, (6672,), generator=g).to(DEV)         # the full-size step's edge-row count
    dae = (torch.randn(6672, C, generator=g) * 0.01).to(DEV)
    leaves = [x] + wt + [a.weight for a in abns] + [a.bias for a in abns] + w2 + b2

    def run():
        for t in leaves:
            t.grad = None
        out, ae = gram_reg_heads(x, rows, abns, offs, 50, wt, [a.weight for a in abns], [a.bias for a in abns], w2, b2, sync=False,
                                 extra_branch=1, extra_rows=erows)
        ((out * dout).sum() + (ae.float() * dae).sum()).backward()
        return out, [t.grad for t in leaves]
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            o_e, g_e = run()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    o_e, g_e = o_e.clone(), [t.clone() for t in g_e]
    junk = [torch.randn(1 << 20, device=DEV) for _ in range(8)]        # perturb the allocator state
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        o_g, g_g = run()
    for rep in range(3):
        gr.replay(); torch.cuda.synchronize()
        rel = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp(min=1e-20))
        errs = [rel(o_g, o_e)] + [rel(a, b) for a, b in zip(g_g, g_e)]
        print(dt, "replay", rep, "max rel err out / grads: %.2e / %.2e" % (errs[0], max(errs[1:])), "finite:", all(bool(torch.isfinite(t).all()) for t in [o_g] + g_g))
