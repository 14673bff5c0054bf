"""The algebra behind monoflex_amd/gram_heads.py on the CPU (pure torch, no HIP library): the patch Gram matrix G and the patch sum m of a
feature map, assembled the way GramRegHeadsFn assembles them -- 5x5 autocorrelation blocks minus the Gram matrix / column sums of the patches
centred on the one-pixel frame -- equal the explicit im2col quantities, hence the batch statistics of every 3x3 conv of that map; and the
5x5 kernel the backward pass builds from d loss / dR is the gradient of the autocorrelation (checked against autograd)."""
import torch
import torch.nn.functional as F

from monoflex_amd import gram_heads as GH


def _autocorr5(x):
    """R[a][b][kh][kw] = sum_q x[q][a] x~[q + (kh-2, kw-2)][b] for NHWC x -- what mfx_conv_wgrad_oihw(x, dy=x, 5x5, pad 2) returns."""
    B, H, W, C = x.shape
    xp = F.pad(x.permute(0, 3, 1, 2), (2, 2, 2, 2))
    R = torch.zeros(C, C, 5, 5, dtype=x.dtype)
    for kh in range(5):
        for kw in range(5):
            sh = xp[:, :, kh:kh + H, kw:kw + W]                                   # x~[q + d]
            R[:, :, kh, kw] = torch.einsum("bhwa,bchw->ac", x, sh)
    return R


def _im2col(x):
    B, H, W, C = x.shape
    cols = F.unfold(x.permute(0, 3, 1, 2), 3, padding=1)                           # [B][C*9][H*W], row index c*9 + tap
    return cols.view(B, C, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 9 * C)   # [px][tap*C + c]


def test_gram_matrix_from_autocorrelation_and_frame_equals_im2col_gram():
    torch.manual_seed(3)
    B, H, W, C = 2, 5, 7, 4
    x = torch.randn(B, H, W, C, dtype=torch.float64)
    geo = GH._geometry(B, H, W, torch.device("cpu"))
    xf = x.view(-1, C)
    A_f = (xf[geo["idx_f"]] * geo["val_f"].unsqueeze(-1)).reshape(-1, 9 * C)
    R5 = _autocorr5(x)
    Gp = (geo["sel"].double() @ R5.permute(2, 3, 0, 1).reshape(25, C * C)).view(9, 9, C, C).permute(0, 2, 1, 3).reshape(9 * C, 9 * C)
    G = Gp - A_f.t() @ A_f
    m = xf.sum(0).repeat(9) - A_f.sum(0)
    A = _im2col(x)
    assert torch.allclose(G, A.t() @ A, atol=1e-10) and torch.allclose(m, A.sum(0), atol=1e-10)
    # -> the batch statistics of any 3x3 / pad 1 conv of x
    Wt = torch.randn(6, C, 3, 3, dtype=torch.float64)
    y = F.conv2d(x.permute(0, 3, 1, 2), Wt, None, 1, 1)
    Wk = Wt.permute(0, 2, 3, 1).reshape(6, 9 * C)
    assert torch.allclose(Wk @ m, y.sum((0, 2, 3)), atol=1e-9)
    assert torch.allclose(((Wk @ G) * Wk).sum(1), (y * y).sum((0, 2, 3)), atol=1e-8)


def test_frame_gradient_tables_invert_the_frame_gather():
    """The backward pass turns d(A_f) into a gradient of x with a GATHER (inv / ring_idx: fixed summation order): same result as autograd's
    scatter through the index."""
    torch.manual_seed(4)
    B, H, W, C = 2, 4, 6, 3
    x = torch.randn(B, H, W, C, dtype=torch.float64, requires_grad=True)
    geo = GH._geometry(B, H, W, torch.device("cpu"))
    A_f = (x.view(-1, C)[geo["idx_f"]] * geo["val_f"].unsqueeze(-1)).reshape(-1, 9 * C)
    g = torch.randn_like(A_f)
    (A_f * g).sum().backward()
    ext = torch.cat((g.reshape(-1, C), g.new_zeros(1, C)), 0)
    dx = torch.zeros(B * H * W, C, dtype=torch.float64).index_add_(0, geo["ring_idx"], ext[geo["inv"]].sum(1))
    assert torch.allclose(dx.view_as(x), x.grad, atol=1e-12)
    assert geo["ring_idx"].unique().numel() == geo["ring_idx"].numel()              # unique targets: the add is order-free


def test_autocorrelation_gradient_is_the_symmetrised_5x5_kernel():
    """d/dx of sum(R * dR) = cross-correlation of x with K[a][b][d] = dR[a][b][d] + dR[b][a][-d], pad 2 (GramRegHeadsFn.backward)."""
    torch.manual_seed(5)
    B, H, W, C = 1, 6, 5, 3
    x = torch.randn(B, H, W, C, dtype=torch.float64, requires_grad=True)
    dR = torch.randn(C, C, 5, 5, dtype=torch.float64)
    (_autocorr5(x) * dR).sum().backward()
    K = dR + dR.permute(1, 0, 2, 3).flip(2, 3)
    dx = F.conv2d(x.detach().permute(0, 3, 1, 2), K, None, 1, 2).permute(0, 2, 3, 1)
    assert torch.allclose(dx, x.grad, atol=1e-10)


def test_gram_reg_heads_node_end_to_end_on_cpu(monkeypatch):
    """The WHOLE GramRegHeadsFn (forward table, extra activation rows, running statistics, every gradient) on the CPU against the dense layers in
    fp64 torch, with the node's three library launches swapped for torch equivalents (the 5x5 autocorrelation, the column sums, the 5x5
    gradient conv): the node's algebra and index plumbing are checked by the CPU suite, the launches themselves by the -m gpu tests."""
    from types import SimpleNamespace
    from monoflex_amd.model.head.detector_predictor import InPlaceABN
    monkeypatch.setattr(GH, "_autocorr5", lambda x: _autocorr5(x.double()).float())
    monkeypatch.setattr(GH.AG, "_colsum", lambda t: t.reshape(-1, t.shape[-1]).double().sum(0).float())
    monkeypatch.setattr(GH.AG, "_c", lambda t: t.contiguous())
    monkeypatch.setattr(GH.ops, "pack_conv", lambda w, dt, scale, shift, stride, pad: SimpleNamespace(w=w, scale=scale, shift=shift, pad=pad))

    def conv2d(x, p):
        y = F.conv2d(x.permute(0, 3, 1, 2).double(), p.w.double(), None, 1, p.pad) * p.scale.double().view(1, -1, 1, 1) + p.shift.double().view(1, -1, 1, 1)
        return y.permute(0, 2, 3, 1).float().contiguous()
    monkeypatch.setattr(GH.ops, "conv2d", conv2d)

    g = torch.Generator().manual_seed(31)
    B, H, W, Cin, C, N, R2 = 2, 7, 9, 8, 16, 10, 23
    ks, offs = (3, 5), (0, 4)
    rows = torch.zeros(N, 72)
    rows[:, 0] = 1.0; rows[7:, 0] = 0.0
    rows[:, 57] = torch.randint(0, B, (N,), generator=g).float()
    rows[:, 2] = torch.randint(0, W, (N,), generator=g).float(); rows[:, 3] = torch.randint(0, H, (N,), generator=g).float()
    rows[0, 2:4] = torch.tensor([0.0, 0.0]); rows[1, 2:4] = torch.tensor([W - 1.0, H - 1.0]); rows[2] = rows[3]
    erows = torch.randint(0, B * H * W, (R2,), generator=g)
    x = torch.randn(B, H, W, Cin, generator=g)
    wt = [torch.randn(C, Cin, 3, 3, generator=g) / 8.0 for _ in ks]
    w2 = [torch.randn(k, C, 1, 1, generator=g) * 0.3 for k in ks]
    b2 = [torch.randn(k, generator=g) * 0.1 for k in ks]
    dout, dae = torch.randn(N, 10, generator=g), torch.randn(R2, C, generator=g) * 0.1
    abns = [InPlaceABN(C) for _ in ks]
    for a in abns:
        with torch.no_grad():
            a.weight.copy_(torch.rand(C, generator=g) + 0.5); a.bias.copy_(torch.randn(C, generator=g) * 0.3)
    # dense reference (fp64)
    xr = x.double().permute(0, 3, 1, 2).clone().requires_grad_()
    rw = [w.double().clone().requires_grad_() for w in wt]
    rw2 = [w.double().clone().requires_grad_() for w in w2]
    rb2 = [b.double().clone().requires_grad_() for b in b2]
    bns = []
    for a in abns:
        m = torch.nn.BatchNorm2d(C).double()
        with torch.no_grad():
            m.weight.copy_(a.weight.double()); m.bias.copy_(a.bias.double())
        bns.append(m)
    bi, cy, cx, valid = rows[:, 57].long(), rows[:, 3].long(), rows[:, 2].long(), rows[:, 0].double()
    tot, outs = 0, []
    for i, k in enumerate(ks):
        act = F.leaky_relu(bns[i](F.conv2d(xr, rw[i], None, 1, 1)), 0.01)
        o = F.conv2d(act, rw2[i], rb2[i]).permute(0, 2, 3, 1)[bi, cy, cx] * valid[:, None]
        outs.append(o)
        tot = tot + (o * dout[:, offs[i]:offs[i] + k].double()).sum()
        if i == 1:
            ae_ref = act.permute(0, 2, 3, 1).reshape(-1, C)[erows]
            tot = tot + (ae_ref * dae.double()).sum()
    tot.backward()
    # the node
    xd = x.clone().requires_grad_()
    dw = [w.clone().requires_grad_() for w in wt]
    dw2 = [w.clone().requires_grad_() for w in w2]
    db2 = [b.clone().requires_grad_() for b in b2]
    out, ae = GH.gram_reg_heads(xd, rows, abns, offs, 10, dw, [a.weight for a in abns], [a.bias for a in abns], dw2, db2, sync=False,
                                extra_branch=1, extra_rows=erows)
    ((out * dout).sum() + (ae * dae).sum()).backward()
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-20))      # noqa: E731
    assert rel(ae, ae_ref.detach()) < 1e-4
    assert rel(xd.grad, xr.grad.permute(0, 2, 3, 1)) < 2e-4, rel(xd.grad, xr.grad.permute(0, 2, 3, 1))
    for i, k in enumerate(ks):
        assert rel(out[:, offs[i]:offs[i] + k], outs[i].detach()) < 1e-4
        assert rel(dw[i].grad, rw[i].grad) < 2e-4 and rel(dw2[i].grad, rw2[i].grad) < 2e-4 and rel(db2[i].grad, rb2[i].grad) < 2e-4
        assert rel(abns[i].weight.grad, bns[i].weight.grad) < 2e-4 and rel(abns[i].bias.grad, bns[i].bias.grad) < 2e-4
        assert rel(abns[i].running_var, bns[i].running_var) < 1e-4 and rel(abns[i].running_mean, bns[i].running_mean) < 1e-4
        assert int(abns[i].num_batches_tracked) == 1
    assert float(out[rows[:, 0] == 0].abs().max()) == 0.0 and float(out[:, 3].abs().max()) == 0.0          # empty slots; the unused column
