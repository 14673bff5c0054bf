"""Regression branches of the TRAINING step without their dense 256-channel maps (reference model/head/detector_predictor.py:125-169).

A regression branch is  conv3x3(64 -> 256, no bias) -> ABN(batch statistics, leaky 0.01) -> 1x1 heads,  and the loss reads its output at the
object centres only (layers/utils.py:120-145).  The dense map y = W * x is needed for exactly two things: the batch statistics of the ABN
and -- in the backward pass -- the statistics' own gradient, which reaches every pixel.  Both are functions of the PATCH GRAM MATRIX of the
shared input x (the backbone feature map):

    x_p(px) in R^576      the 3x3 x 64-channel patch around pixel px (zero outside the image)
    m = sum_px x_p(px),   G = sum_px x_p(px) x_p(px)^T                      (576, 576 x 576; ONE pair for all branches)
    sum_px y   = W m,     sum_px y^2 = diag(W G W^T)                         -> mean, variance of every branch's 256 channels
    y(r)       = W x_p(r)                                                    at the N object rows only

so the forward pass needs no dense conv, and the backward pass -- d loss / d G -- is ONE 5x5 64 -> 64 convolution of x for all branches
together instead of a dense BN pass, a 256 -> 64 data-gradient conv and a 64 -> 256 weight-gradient pass PER BRANCH (seven branches: about
2.2 ms of the 21 ms step at B = 8).  The arithmetic is exact, not an approximation: BN's backward is  dy = (gamma/sigma) (dz - mean(dz) - xhat
mean(dz xhat)),  dz is zero away from the objects, and the two mean terms are precisely the gradients that flow through `mean` and `var` here.

How G is built.  With Omega+ = the image grown by one pixel,  sum_{px in Omega+} x_p(px) x_p(px)^T  depends on the tap displacement only:
block (t1, t2) is the autocorrelation R[t2 - t1] = sum_q x[q] x[q + d]^T, 25 matrices of 64 x 64 = the weight gradient of a 5x5 convolution
with x as input AND as output gradient (one launch of the library's weight-gradient kernel).  G = that minus the Gram matrix of the patches
centred on the one-pixel FRAME around the image (B x 2 (H + W + 2) rows, a small GEMM) -- which is also the whole of the border handling.
The small algebra (576-wide matrices, N <= a few hundred object rows) runs as torch ops on fp32 tensors inside the node and is differentiated
by torch; the two dense pieces are library launches (mfx_conv_wgrad_oihw forward, a 5x5 mfx_conv2d_nhwc backward)."""
import ctypes

import torch
import torch.nn.functional as F_

from . import autograd as AG
from . import lib as L
from . import ops

_GEOM = {}


def _geometry(B, H, W, device):
    """Static index tables of one map shape: frame-pixel patches (gather indices + validity), their inverse for the gradient (per border pixel
    the <= 5 (frame pixel, tap) entries that touch it, so the scatter is a gather with a fixed summation order), the tap-pair -> displacement
    selection matrix."""
    key = (B, H, W, str(device))
    if key in _GEOM:
        return _GEOM[key]
    if len(_GEOM) >= 8:                                       # a training run sees one or two map shapes; never let the tables pile up
        _GEOM.clear()
    py = torch.cat((torch.full((W + 2,), -1), torch.full((W + 2,), H), torch.arange(H), torch.arange(H)))
    px = torch.cat((torch.arange(-1, W + 1), torch.arange(-1, W + 1), torch.full((H,), -1), torch.full((H,), W)))
    nf = py.numel()
    ty = torch.tensor([-1, -1, -1, 0, 0, 0, 1, 1, 1])
    tx = torch.tensor([-1, 0, 1, -1, 0, 1, -1, 0, 1])
    qy, qx = py.view(nf, 1) + ty.view(1, 9), px.view(nf, 1) + tx.view(1, 9)
    valid = (qy >= 0) & (qy < H) & (qx >= 0) & (qx < W)
    flat = (qy.clamp(0, H - 1) * W + qx.clamp(0, W - 1))
    b_off = (torch.arange(B) * H * W).view(B, 1, 1)
    idx_f = (flat.view(1, nf, 9) + b_off).reshape(B * nf, 9)
    val_f = valid.view(1, nf, 9).expand(B, nf, 9).reshape(B * nf, 9)
    # inverse: border pixel q <- entries (frame row, tap) with frame + tap == q
    entries = {}
    for r in range(nf):
        for t in range(9):
            if bool(valid[r, t]):
                entries.setdefault(int(flat[r, t]), []).append(r * 9 + t)
    ring = sorted(entries)
    width = max(len(v) for v in entries.values())
    F_rows = B * nf
    inv = torch.full((B, len(ring), width), F_rows * 9, dtype=torch.long)            # F_rows * 9 = the appended zero row
    for j, q in enumerate(ring):
        for k, e in enumerate(entries[q]):
            inv[:, j, k] = e + torch.arange(B) * nf * 9
    ring_idx = (torch.tensor(ring).view(1, -1) + (torch.arange(B) * H * W).view(B, 1)).reshape(-1)
    sel = torch.zeros(81, 25)
    for t1 in range(9):
        for t2 in range(9):
            dy, dx = int(ty[t2] - ty[t1]), int(tx[t2] - tx[t1])
            sel[t1 * 9 + t2, (dy + 2) * 5 + dx + 2] = 1.0
    vy = ((torch.arange(H).view(H, 1) + ty.view(1, 9)) >= 0) & ((torch.arange(H).view(H, 1) + ty.view(1, 9)) < H)       # [H][9]: tap row inside?
    vx = ((torch.arange(W).view(W, 1) + tx.view(1, 9)) >= 0) & ((torch.arange(W).view(W, 1) + tx.view(1, 9)) < W)
    g = dict(idx_f=idx_f.to(device), val_f=val_f.to(device), inv=inv.view(-1, width).to(device), ring_idx=ring_idx.to(device),
             sel=sel.to(device), ty=ty.to(device), tx=tx.to(device), nf=nf, vy=vy.to(device), vx=vx.to(device),
             tap_off=(ty * W + tx).to(device), ones_f=torch.ones(1, B * nf, device=device), lin=torch.tensor([float(H * W), float(W), 1.0], device=device),
             hi=torch.tensor([B - 1.0, H - 1.0, W - 1.0], device=device), cols=torch.tensor([57, 3, 2], device=device))
    _GEOM[key] = g
    return g


_STATIC = {}


def _static(abns, cws, ks, offs, ld_out, has_b, device):
    """Per head configuration: the ABNs' eps as one vector; where each branch's 1x1 weights / biases sit in the [ld_out x channels] matrix."""
    key = (tuple(float(a.eps) for a in abns), tuple(cws), tuple(ks), tuple(offs), ld_out, tuple(has_b), str(device))
    if key in _STATIC:
        return _STATIC[key]
    if len(_STATIC) >= 8:
        _STATIC.clear()
    nch = sum(cws)
    eps = torch.cat([torch.full((cw,), float(a.eps)) for cw, a in zip(cws, abns)]).to(device)
    pos, bpos, c0 = [], [], 0
    for cw, k, off, hb in zip(cws, ks, offs, has_b):
        r = torch.arange(k).view(k, 1) + off
        c = torch.arange(cw).view(1, cw) + c0
        pos.append((r * nch + c).reshape(-1))
        if hb:
            bpos.append(torch.arange(k) + off)
        c0 += cw
    st = dict(eps=eps, w2_pos=torch.cat(pos).to(device), b2_pos=torch.cat(bpos).to(device) if bpos else None)
    _STATIC[key] = st
    return st


class _AllReduceSum(torch.autograd.Function):
    """Sum over the SyncBN group, differentiable (the gradient of a sum over ranks is the sum of the ranks' gradients)."""

    @staticmethod
    def forward(ctx, t, group):
        import torch.distributed as dist
        ctx.group = group
        out = t + 0.0                                         # (a kernel, not a device-to-device copy node: see GramRegHeadsFn)
        dist.all_reduce(out, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        g = g + 0.0
        dist.all_reduce(g, group=ctx.group)
        return g, None


class _RowAffine(torch.autograd.Function):
    """z[r][c] = y[r][c] * s[c] + t[c].  Written out because autograd's gradient of the two broadcasts is a column sum over ALL rows (thousands,
    with the edge rows), i.e. one of torch's multi-block reductions -- see the note at `m` in GramRegHeadsFn.forward; here both are GEMVs."""

    @staticmethod
    def forward(ctx, y, s, t):
        ctx.save_for_backward(y, s)
        return y * s + t

    @staticmethod
    def backward(ctx, dz):
        y, s = ctx.saved_tensors
        ones = torch.full((1, y.shape[0]), 1.0, dtype=dz.dtype, device=dz.device)
        return dz * s, (ones @ (dz * y)).view(-1), (ones @ dz).view(-1)


def _autocorr5(x):
    """R[a][b][kh][kw] = sum_q x[q][a] * x~[q + (kh - 2, kw - 2)][b]  (fp32; x~ = x with zeros outside the image): the weight gradient of a
    5x5 / pad 2 convolution whose input and output gradient are both x."""
    B, H, W, C = x.shape
    dw = torch.empty(C, C, 5, 5, dtype=torch.float32, device=x.device)
    ws = ops._splitk_workspace(x.device)
    L.check(L.load().mfx_conv_wgrad_oihw(ops._ptr(x), ops._ptr(x), ops._ptr(dw), B, H, W, C, C, 5, 5, 1, 2, 2, H, W, C, C, C, C, ops._dt(x.dtype),
                                         ops._ptr(ws), ws.numel() * 4, ops._stream()), "mfx_conv_wgrad_oihw")
    return dw


class GramRegHeadsFn(torch.autograd.Function):
    """(x, rows, per-branch trunk weights / ABN gamma, beta / stacked 1x1 weights, biases) -> fp32 [N][ld_out] table of the branches' outputs at
    the object rows (branch i at columns [offs[i], offs[i] + k_i)), as SparseRegHeadsFn returns it -- without the dense trunk maps."""

    @staticmethod
    def forward(ctx, x, rows, abns, offs, ld_out, sync, extra_branch, extra_rows, *ts):
        nb = len(abns)
        ws_, gammas, betas, w2s, b2s = (ts[i * nb:(i + 1) * nb] for i in range(5))
        x = AG._c(x)
        B, H, W, C = x.shape
        M = B * H * W
        dev = x.device
        geo = _geometry(B, H, W, dev)
        xf = x.view(M, C)
        with torch.no_grad():
            R5 = _autocorr5(x)
            S0 = AG._colsum(x)
            A_f = (xf[geo["idx_f"]] * geo["val_f"].unsqueeze(-1).to(x.dtype)).reshape(-1, 9 * C).float()
            # object rows -> (image, cy, cx) clamped, flat pixel, per-tap validity from the static row / column tables
            byx = torch.minimum(rows.index_select(1, geo["cols"]).clamp_min(0.0), geo["hi"])
            pix = (byx @ geo["lin"]).long()
            yx = byx.long()
            val_o = geo["vy"][yx[:, 1]] & geo["vx"][yx[:, 2]]
            idx_o = (pix.view(-1, 1) + geo["tap_off"].view(1, 9)).clamp(0, M - 1)
            A_o = (xf[idx_o] * val_o.unsqueeze(-1).to(x.dtype)).reshape(-1, 9 * C).float()
            if extra_rows is not None:                               # a second row set (flat pixel indices): branch `extra_branch`'s ACTIVATION there
                ey, ex = (extra_rows // W) % H, extra_rows % W
                val_e = geo["vy"][ey] & geo["vx"][ex]
                idx_e = (extra_rows.view(-1, 1) + geo["tap_off"].view(1, 9)).clamp(0, M - 1)
                A_e = (xf[idx_e] * val_e.unsqueeze(-1).to(x.dtype)).reshape(-1, 9 * C).float()
        group = AG._sync_group(sync)
        leaves = [t.detach().requires_grad_(True) for t in ((R5, A_f, A_o, S0) + ((A_e,) if extra_rows is not None else ()))]
        params = [None if t is None else t.detach().requires_grad_(t.requires_grad) for t in ts]
        pw, pg, pb, pw2, pb2 = (params[i * nb:(i + 1) * nb] for i in range(5))
        with torch.enable_grad():
            R5l, A_fl, A_ol, S0l = leaves[:4]
            # weights as the matrix cores would see them in this compute mode (16-bit modes round them), K order = (tap, channel)
            Wc = torch.cat([w.float() for w in pw], 0)
            if x.dtype != torch.float32:
                Wc = Wc.to(x.dtype).float()
            Wk = Wc.permute(0, 2, 3, 1).reshape(Wc.shape[0], 9 * C)
            Gp = (geo["sel"] @ R5l.permute(2, 3, 0, 1).reshape(25, C * C)).view(9, 9, C, C).permute(0, 2, 1, 3).reshape(9 * C, 9 * C)
            G = Gp - A_fl.t() @ A_fl
            # (column sums as a GEMV, not A_fl.sum(0): torch's multi-block reductions zero their semaphores with a memset, and memset nodes are
            # not reliably ordered against kernel nodes when a hipGraph is replayed on this platform -- csrc/fill.h; measured: the sum came out
            # wrong from the second replay on)
            m = S0l.repeat(9) - (geo["ones_f"] @ A_fl).view(-1)
            sums = torch.cat((Wk @ m, ((Wk @ G) * Wk).sum(1)))
            Mt = M
            if group is not None:
                import torch.distributed as dist
                sums = _AllReduceSum.apply(sums, group)
                Mt = M * dist.get_world_size(group)
            nch = Wk.shape[0]
            mean = sums[:nch] / Mt
            var = (sums[nch:] / Mt - mean * mean).clamp_min(0.0)
            st = _static(abns, [w.shape[0] for w in pw], [w.shape[0] for w in pw2], offs, ld_out, [b is not None for b in pb2], dev)
            rstd = torch.rsqrt(var + st["eps"])
            gam, bet = torch.cat([g.float() for g in pg]), torch.cat([b.float() for b in pb])
            Y_o = A_ol @ Wk.t()
            sc = rstd * gam                                          # BN as one per-channel affine map
            sh = bet - mean * sc
            act = F_.leaky_relu(_RowAffine.apply(Y_o, sc, sh), 0.01)
            # all 1x1 heads as ONE [ld_out x channels] matrix: the branches' weights scattered to their (row block, column block) positions
            w2flat = torch.cat([w.float().reshape(-1) for w in pw2])
            # (torch.full, not torch.zeros: a zero fill of a fresh tensor is a memset node inside a hipGraph capture, and those are not
            # reliably ordered against kernel nodes on replay here -- csrc/fill.h)
            W2 = torch.full((ld_out * nch,), 0.0, dtype=torch.float32, device=dev).scatter_(0, st["w2_pos"], w2flat).view(ld_out, nch)
            out = act @ W2.t()
            if st["b2_pos"] is not None:
                b2flat = torch.cat([b.float() for b in pb2 if b is not None])
                out = out + torch.full((ld_out,), 0.0, dtype=torch.float32, device=dev).scatter_(0, st["b2_pos"], b2flat)
            out = out * (rows[:, 0] > 0).to(act.dtype).view(-1, 1)          # empty slots of the object table read as zero rows
            act_e = None
            if extra_rows is not None:
                c0 = sum(w.shape[0] for w in pw[:extra_branch]); c1 = c0 + pw[extra_branch].shape[0]
                Y_e = leaves[4] @ Wk[c0:c1].t()
                act_e = F_.leaky_relu(_RowAffine.apply(Y_e, sc[c0:c1], sh[c0:c1]), 0.01)
        with torch.no_grad():                                         # running statistics: momentum update with the unbiased variance
            unb = var * (float(Mt) / max(Mt - 1, 1))
            c0 = 0
            rms, rvs, means, vars_, nbts = [], [], [], [], []
            for w, a in zip(pw, abns):
                cw = w.shape[0]
                if a.track_running_stats and a.running_mean is not None:
                    rms.append(a.running_mean); rvs.append(a.running_var)
                    means.append(mean[c0:c0 + cw].to(a.running_mean.dtype)); vars_.append(unb[c0:c0 + cw].to(a.running_var.dtype))
                    if a.num_batches_tracked is not None:
                        nbts.append(a.num_batches_tracked)
                c0 += cw
            if rms:
                mom = abns[0].momentum if abns[0].momentum is not None else 0.1
                torch._foreach_lerp_(rms, means, mom)
                torch._foreach_lerp_(rvs, vars_, mom)
                if nbts:
                    torch._foreach_add_(nbts, 1)
        ctx.graph = (out, act_e, leaves, params)
        if extra_rows is not None:
            ctx.save_for_backward(x, idx_o, val_o, idx_e, val_e)
            return out.detach(), act_e.detach().to(x.dtype)
        ctx.save_for_backward(x, idx_o, val_o)
        return out.detach(), None

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout, dact_e):
        x, idx_o, val_o = ctx.saved_tensors[:3]
        out, act_e, leaves, params = ctx.graph
        ctx.graph = None
        B, H, W, C = x.shape
        M = B * H * W
        geo = _geometry(B, H, W, x.device)
        wanted = leaves + [p for p in params if p is not None and p.requires_grad]
        if act_e is not None:
            if dact_e is None:
                dact_e = torch.full_like(act_e, 0.0)
            grads = torch.autograd.grad([out, act_e], wanted, [dout.float(), dact_e.float()], allow_unused=True)
        else:
            grads = torch.autograd.grad(out, wanted, dout.float(), allow_unused=True)
        dR5, dA_f, dA_o, dS0 = (g if g is not None else torch.full_like(t, 0.0) for g, t in zip(grads[:4], leaves[:4]))   # (None: unused leaf)
        nl = len(leaves)
        dx = None
        if ctx.needs_input_grad[0]:
            # dense part: R[a][b][d] appears with x[q][a] x[q+d][b] -> dx[q][a] = sum_{d,b} (dR[a][b][d] + dR[b][a][-d]) x[q+d][b]: a 5x5 conv of x;
            # the gradient of S0 = sum_q x[q] is the same vector at every pixel: the conv's per-channel shift
            Kx = dR5 + dR5.permute(1, 0, 2, 3).flip(2, 3)
            # these are gradients of SUMS over every pixel: ~1e-9 .. 1e-6, below fp16's normal range.  The conv runs on Kx / max |Kx| and its
            # epilogue multiplies the maximum back in (device scalars, no host sync; row maxima first: single-block reductions, see above)
            amax = Kx.abs().reshape(C, -1).amax(1).amax().clamp_min(1e-30)
            p = ops.pack_conv(Kx / amax, x.dtype, amax.expand(C), dS0, stride=1, pad=2)
            dx = ops.conv2d(x, p)
            dxf = dx.view(M, C)
            ext = torch.cat((dA_f.reshape(-1, C), dA_f.new_full((1, C), 0.0)), 0)
            dxf.index_add_(0, geo["ring_idx"], ext[geo["inv"]].sum(1).to(dx.dtype))          # border pixels: unique indices, fixed order
            dAo = (dA_o.view(-1, 9, C) * val_o.unsqueeze(-1)).to(dx.dtype)
            if torch.are_deterministic_algorithms_enabled():        # lib.set_deterministic(True)
                for t in range(9):                                   # one tap at a time: distinct objects -> distinct pixels per call, fixed order
                    dxf.index_add_(0, idx_o[:, t], dAo[:, t])
            else:
                dxf.index_add_(0, idx_o.reshape(-1), dAo.reshape(-1, C))
            if act_e is not None and grads[4] is not None:
                idx_e, val_e = ctx.saved_tensors[3:5]
                dxf.index_add_(0, idx_e.reshape(-1), (grads[4].view(-1, 9, C) * val_e.unsqueeze(-1)).to(dx.dtype).reshape(-1, C))
        it = iter(grads[nl:])
        pg = [next(it) if (p is not None and p.requires_grad) else None for p in params]
        return (dx, None, None, None, None, None, None, None, *pg)


# ---- the same node as HIP kernels (csrc/gram_heads.hip): 16-bit activations on the GPU; everything else keeps the torch form above ----------
HIP_NODE = [__import__("os").environ.get("MFX_GRAM_HIP", "1") != "0"]


def _autocorr5_rows(x):
    """The displacement rows dy = -2, -1, 0 of the 5x5 autocorrelation: R[a][b][kh][kw] = sum_q x[q][a] * x~[q + (kh - 2, kw - 2)][b], kh in 0..2 --
    the weight gradient of a 3x5 convolution (padding 2 x 2, outputs on the input grid) whose input and output gradient are both x.  The rows
    dy = 1, 2 are R[b][a][-d] (the same products), so 15 instead of 25 taps are computed (csrc/gram_heads.hip gram_build_kernel)."""
    B, H, W, C = x.shape
    dw = torch.empty(C, C, 3, 5, dtype=torch.float32, device=x.device)
    ws = ops._splitk_workspace(x.device)
    L.check(L.load().mfx_conv_wgrad_oihw(ops._ptr(x), ops._ptr(x), ops._ptr(dw), B, H, W, C, C, 3, 5, 1, 2, 2, H, W, C, C, C, C, ops._dt(x.dtype),
                                         ops._ptr(ws), ws.numel() * 4, ops._stream()), "mfx_conv_wgrad_oihw")
    return dw


def _dense_wgrad(xm, dym, out):
    """out[o][c] = sum_r dym[r][o] * xm[r][c]  (fp32): the library's weight-gradient launch on two dense row-major 16-bit matrices."""
    R, Kc = xm.shape
    Co = dym.shape[1]
    ws = ops._splitk_workspace(xm.device)
    L.check(L.load().mfx_conv_wgrad_oihw(ops._ptr(xm), ops._ptr(dym), ops._ptr(out), 1, 1, R, Kc, Kc, 1, 1, 1, 0, 0, 1, R, Co, Co, Co, Kc, ops._dt(xm.dtype),
                                         ops._ptr(ws), ws.numel() * 4, ops._stream()), "mfx_conv_wgrad_oihw")
    return out


class GramRegHeadsHipFn(torch.autograd.Function):
    """GramRegHeadsFn with every torch op replaced by a kernel of csrc/gram_heads.hip or a library launch (autocorrelation, column sums, the three
    weight-gradient-shaped GEMMs, the 5x5 data-gradient conv): no `at::native` kernel, no vendor GEMM inside the node; the backward is written
    out by hand there.  SyncBN: the two small all-reduces ([sum y | sum y^2] forward, their gradients backward) stay torch.distributed calls."""

    @staticmethod
    def forward(ctx, x, rows, abns, offs, ld_out, sync, extra_branch, extra_rows, *ts):
        nb = len(abns)
        ws_, gammas, betas, w2s, b2s = (ts[i * nb:(i + 1) * nb] for i in range(5))
        x = AG._c(x)
        rows = AG._c(rows)
        B, H, W, C = x.shape
        M, dev, dt = B * H * W, x.device, x.dtype
        geo = _geometry(B, H, W, dev)
        N = rows.shape[0]
        Ne = int(extra_rows.numel()) if (extra_rows is not None and extra_branch >= 0) else 0
        F, CH = B * geo["nf"], 256 * nb
        f32 = dict(dtype=torch.float32, device=dev)
        d = L.GramDesc()
        d.x, d.rows = x.data_ptr(), rows.data_ptr()
        er = AG._c(extra_rows.long()) if Ne else None
        d.extra_rows = er.data_ptr() if Ne else None
        d.B, d.H, d.W, d.C, d.N, d.Ne, d.F, d.nbranch = B, H, W, C, N, Ne, F, nb
        d.extra_branch, d.ld_out, d.dtype = (extra_branch if Ne else -1), ld_out, ops._dt(dt)
        keep = [er]
        packs = [AG._pack_weight(w, dt, 0, 256, C, 1, 1, 1) for w in ws_]          # [256][576], k = tap * 64 + c: the step's batched packing
        w2c = [AG._c(w.detach().float().view(w.shape[0], -1)) for w in w2s]
        b2c = [None if b is None else AG._c(b.detach().float()) for b in b2s]
        for i in range(nb):
            assert packs[i].K_pad == 9 * C and packs[i].w.shape[0] == 256
            d.wk[i] = packs[i].w.data_ptr()
            d.gamma[i], d.beta[i] = gammas[i].data_ptr(), betas[i].data_ptr()
            d.w2[i] = w2c[i].data_ptr()
            d.b2[i] = b2c[i].data_ptr() if b2c[i] is not None else None
            d.eps[i], d.k[i], d.off[i] = float(abns[i].eps), int(w2c[i].shape[0]), int(offs[i])
            a = abns[i]
            if a.track_running_stats and a.running_mean is not None:
                d.run_mean[i], d.run_var[i] = a.running_mean.data_ptr(), a.running_var.data_ptr()
                d.nbt[i] = a.num_batches_tracked.data_ptr() if a.num_batches_tracked is not None else None
        d.momentum = float(abns[0].momentum if abns[0].momentum is not None else 0.1)
        A = torch.empty((F + N + Ne, 9 * C), dtype=dt, device=dev)
        Wkc, WkT = torch.empty((CH, 9 * C), dtype=dt, device=dev), torch.empty((9 * C, CH), dtype=dt, device=dev)
        d.A, d.Wkc, d.WkT = A.data_ptr(), Wkc.data_ptr(), WkT.data_ptr()
        lib_, st = L.load(), ops._stream
        L.check(lib_.mfx_gram_heads(ctypes.byref(d), 0, st()), "mfx_gram_heads(0)")
        R5, S0 = _autocorr5_rows(x), AG._colsum(x)
        P = _dense_wgrad(A[:F], A[:F], torch.empty((9 * C, 9 * C), **f32))
        csA = AG._colsum(A[:F])
        G, m, Tm = torch.empty((9 * C, 9 * C), **f32), torch.empty(9 * C, **f32), torch.empty((CH, 9 * C), **f32)
        sums, stat = torch.empty(2 * CH, **f32), torch.empty(5 * CH, **f32)
        d.R5, d.S0, d.P, d.csA, d.G, d.m, d.Tm, d.sums, d.stat = (t.data_ptr() for t in (R5, S0, P, csA, G, m, Tm, sums, stat))
        L.check(lib_.mfx_gram_heads(ctypes.byref(d), 1, st()), "mfx_gram_heads(1)")
        group = AG._sync_group(sync)
        Mt = M
        if group is not None:
            import torch.distributed as dist
            dist.all_reduce(sums, group=group)
            Mt = M * dist.get_world_size(group)
        d.Mt = float(Mt)
        Y, act, out = torch.empty((N, CH), **f32), torch.empty((N, CH), **f32), torch.empty((N, ld_out), **f32)
        Ye = torch.empty((max(Ne, 1), 256), **f32)
        act_e = torch.empty((max(Ne, 1), 256), dtype=dt, device=dev)
        d.Y, d.act, d.out, d.Ye, d.act_e = Y.data_ptr(), act.data_ptr(), out.data_ptr(), Ye.data_ptr(), act_e.data_ptr()
        covered = sorted((int(o), int(o) + int(w.shape[0])) for o, w in zip(offs, w2c))
        if covered[0][0] != 0 or covered[-1][1] != ld_out or any(a[1] != b[0] for a, b in zip(covered, covered[1:])):
            out.fill_(0.0)                                     # (columns no branch of this call owns: the caller overwrites them)
        L.check(lib_.mfx_gram_heads(ctypes.byref(d), 2, st()), "mfx_gram_heads(2)")
        ctx.desc, ctx.group, ctx.geo = d, group, geo
        ctx.keep = keep + [x, rows, packs, w2c, b2c, A, Wkc, WkT, R5, S0, P, csA, G, m, Tm, sums, stat, Y, act, Ye, act_e, list(gammas), list(betas)]
        ctx.shapes = ([w.shape for w in ws_], [w.shape for w in w2s], [None if b is None else b.shape for b in b2s], nb, Ne, F, CH, N)
        return out, (act_e[:Ne] if Ne else None)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout, dact_e):
        d, geo = ctx.desc, ctx.geo
        wshapes, w2shapes, b2shapes, nb, Ne, F, CH, N = ctx.shapes
        if ctx.keep is None:
            raise RuntimeError("GramRegHeadsHipFn: backward ran twice (retain_graph=True): the node releases its saved buffers after the first pass")
        x = ctx.keep[1]
        A = ctx.keep[6]
        Wkc = ctx.keep[7]
        dev, dt = x.device, x.dtype
        B, H, W, C = x.shape
        f32 = dict(dtype=torch.float32, device=dev)
        dout = AG._c(dout.float())
        dae = AG._c(dact_e.to(dt)) if (dact_e is not None and Ne) else None
        d.dout, d.dact_e = dout.data_ptr(), (dae.data_ptr() if dae is not None else None)
        dYh, dYl = torch.empty((max(N, 1), CH), dtype=dt, device=dev), torch.empty((max(N, 1), CH), dtype=dt, device=dev)
        dYeh, dYel = torch.empty((max(Ne, 1), 256), dtype=dt, device=dev), torch.empty((max(Ne, 1), 256), dtype=dt, device=dev)
        # one arena, cleared by ONE fill inside phase 3: [d sc | d sh] sums, the four scalars, every branch's d W2 and d b2 (all accumulated into)
        ks = [int(d.k[i]) for i in range(nb)]
        arena = torch.empty(2 * CH + 4 + sum(k * 256 for k in ks) + sum(k for k, sh_ in zip(ks, b2shapes) if sh_ is not None), **f32)
        dsum, scal, o = arena[:2 * CH], arena[2 * CH:2 * CH + 4], 2 * CH + 4
        dw2, db2 = [], []
        for k in ks:
            dw2.append(arena[o:o + k * 256].view(k, 256)); o += k * 256
        for k, sh_ in zip(ks, b2shapes):
            db2.append(None if sh_ is None else arena[o:o + k]); o += 0 if sh_ is None else k
        d.arena_bytes = arena.numel() * 4
        ds = torch.empty(2 * CH, **f32)
        dgam = [torch.empty(256, **f32) for _ in range(nb)]
        dbet = [torch.empty(256, **f32) for _ in range(nb)]
        dwt = [torch.empty((256, C, 3, 3), **f32) for _ in range(nb)]
        for i in range(nb):
            d.dgamma[i], d.dbeta[i], d.dw2[i], d.dwt[i] = dgam[i].data_ptr(), dbet[i].data_ptr(), dw2[i].data_ptr(), dwt[i].data_ptr()
            d.db2[i] = db2[i].data_ptr() if db2[i] is not None else None
        d.dYh, d.dYl, d.dYeh, d.dYel, d.dsum, d.ds, d.scal = (t.data_ptr() for t in (dYh, dYl, dYeh, dYel, dsum, ds, scal))
        lib_, st = L.load(), ops._stream
        L.check(lib_.mfx_gram_heads(ctypes.byref(d), 3, st()), "mfx_gram_heads(3)")
        if ctx.group is not None:
            import torch.distributed as dist
            dist.all_reduce(ds, group=ctx.group)
        Dw16, dm = torch.empty((CH, 9 * C), dtype=dt, device=dev), torch.empty(9 * C, **f32)
        d.Dw16, d.dm = Dw16.data_ptr(), dm.data_ptr()
        L.check(lib_.mfx_gram_heads(ctypes.byref(d), 4, st()), "mfx_gram_heads(4)")
        dGs = _dense_wgrad(Wkc, Dw16, torch.empty((9 * C, 9 * C), **f32))
        Kx, Gn16 = torch.empty((C, C, 5, 5), **f32), torch.empty((9 * C, 9 * C), dtype=dt, device=dev)
        cscale, cshift = torch.empty(C, **f32), torch.empty(C, **f32)
        dAf, dArows = torch.empty((max(F, 1), 9 * C), **f32), torch.empty((max(N + Ne, 1), 9 * C), **f32)
        d.dGs, d.Kx, d.Gn16, d.cscale, d.cshift, d.dAf, d.dArows = (t.data_ptr() for t in (dGs, Kx, Gn16, cscale, cshift, dAf, dArows))
        L.check(lib_.mfx_gram_heads(ctypes.byref(d), 5, st()), "mfx_gram_heads(5)")
        p = AG._pack_weight(Kx, dt, 0, C, C, 1, 2, 2)           # one library launch (a temporary: packed now)
        p.scale, p.shift = cscale, cshift
        dx = ops.conv2d(x, p)
        d.dx = dx.data_ptr()
        d.ring_inv, d.ring_idx = geo["inv"].data_ptr(), geo["ring_idx"].data_ptr()
        d.nring, d.ring_width = int(geo["ring_idx"].numel()), int(geo["inv"].shape[1])
        L.check(lib_.mfx_gram_heads(ctypes.byref(d), 6, st()), "mfx_gram_heads(6)")
        dwo = _dense_wgrad(A[F:F + N], dYh[:N], torch.empty((CH, 9 * C), **f32)) if N else torch.zeros((CH, 9 * C), **f32)
        dwe = _dense_wgrad(A[F + N:F + N + Ne], dYeh[:Ne], torch.empty((256, 9 * C), **f32)) if Ne else None
        d.dwo, d.dwe = dwo.data_ptr(), (dwe.data_ptr() if dwe is not None else None)
        L.check(lib_.mfx_gram_heads(ctypes.byref(d), 7, st()), "mfx_gram_heads(7)")
        ctx.keep = None
        g_w2 = [dw2[i].view(w2shapes[i]) for i in range(nb)]
        return (dx if ctx.needs_input_grad[0] else None, None, None, None, None, None, None, None, *dwt, *dgam, *dbet, *g_w2, *db2)


def gram_reg_heads(x, rows, abns, offs, ld_out, trunk_ws, gammas, betas, w2s, b2s, sync=True, extra_branch=-1, extra_rows=None):
    """-> (table [N][ld_out], activation rows of branch `extra_branch` at the flat pixel indices `extra_rows` in x's dtype, or None)."""
    def f32c(t):
        return t is not None and t.dtype == torch.float32 and t.is_contiguous()
    # everything the C entry (mfx_gram_heads) requires -- a configuration outside it takes the torch node instead of failing with MFX_ERR_ARG:
    # 256 trunk channels per branch, <= 32 outputs per 1x1 head with a 256-wide fp32 weight, fp32 contiguous ABN parameters / running statistics, ONE
    # momentum for all branches (the node applies abns[0]'s; eps is per branch)
    hip_ok = (HIP_NODE[0] and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.shape[-1] == 64 and not torch.are_deterministic_algorithms_enabled()
              and all(w.shape[0] == 256 and tuple(w.shape[1:]) == (64, 3, 3) for w in trunk_ws) and len(trunk_ws) <= 8
              and all(w.shape[0] <= 32 and w.shape[1] == 256 and w.numel() == w.shape[0] * 256 for w in w2s)
              and all(f32c(t) for t in list(gammas) + list(betas))
              and all(f32c(a.running_mean) and f32c(a.running_var) for a in abns)
              and len({a.momentum for a in abns}) == 1)
    if hip_ok:
        return GramRegHeadsHipFn.apply(x, rows, tuple(abns), tuple(offs), ld_out, sync, extra_branch, extra_rows, *trunk_ws, *gammas, *betas, *w2s, *b2s)
    return GramRegHeadsFn.apply(x, rows, tuple(abns), tuple(offs), ld_out, sync, extra_branch, extra_rows, *trunk_ws, *gammas, *betas, *w2s, *b2s)
