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
