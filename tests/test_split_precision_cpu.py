"""Host side of the split-precision mode (MODEL.COMPUTE_DTYPE fp16x2, csrc/common.h f32s_t) on the CPU: the operand packers of
monoflex_amd/ops.py produce the layouts the kernels read, and the arithmetic the kernels perform on them -- three fp16 x fp16 products per
value pair, fp32 accumulation -- reproduces the fp32 dot product to ~2^-21 relative (emulated here with torch)."""
import torch

from monoflex_amd import ops


def _halves(t):
    """float32-typed split tensor -> (hi, lo) as float32 values, chunk layout [h h h h | l l l l] per 4 elements."""
    h = t.contiguous().view(torch.float16).view(-1, 8)
    return h[:, :4].float().reshape(t.shape), h[:, 4:].float().reshape(t.shape)


def test_split_chunks_reconstruct_the_value_and_three_products_reproduce_the_dot_product():
    g = torch.Generator().manual_seed(11)
    K = 4608
    x = torch.randn(64, K, generator=g) * 0.7
    w = torch.randn(64, K, generator=g) * 0.02
    s = ops.split_weight_scale(w)
    xh, xl = _halves(ops.split_chunks(x))
    wh, wl = _halves(ops.split_chunks(w * s))
    assert torch.equal(xh, x.half().float()) and float((xh + xl - x).abs().max()) <= 2.0 ** -21 * float(x.abs().max())
    assert 2 ** 11 <= float((w * s).abs().max()) < 2 ** 12
    ref = (x.double() * w.double()).sum(1)
    three = ((xh * wh).double() + (xh * wl).double() + (xl * wh).double()).sum(1) / s        # hi.hi + hi.lo + lo.hi (lo.lo is below fp32 resolution)
    plain = (x.half().float() * w.half().float()).double().sum(1)
    e3, e1 = float((three - ref).abs().max()), float((plain - ref).abs().max())
    assert e3 <= 2e-6 * float(ref.abs().max()) + 1e-7 and e1 > 50 * e3, (e3, e1)


def test_pair_steps_layout():
    """pair_steps(x, dim): K steps (2p, 2p+1) become [pair p][hi | lo], a chunk = [its two dwords of step 2p | of step 2p+1] -- one 8-element fp16
    MFMA operand of hi (lo) halves (csrc/heads.hip, csrc/conv_halo.hip PR instantiation)."""
    nf, steps, lanes = 3, 6, 64
    x = torch.arange(nf * steps * lanes * 4, dtype=torch.float32).view(nf, steps, lanes, 4)          # dwords: [hi0 hi1 | lo0 lo1] per chunk
    y = ops.pair_steps(x, 1)
    assert y.shape == (nf, steps // 2, 2, lanes, 4)
    for p in range(steps // 2):
        assert torch.equal(y[:, p, 0, :, :2], x[:, 2 * p, :, :2]) and torch.equal(y[:, p, 0, :, 2:], x[:, 2 * p + 1, :, :2])       # hi operand
        assert torch.equal(y[:, p, 1, :, :2], x[:, 2 * p, :, 2:]) and torch.equal(y[:, p, 1, :, 2:], x[:, 2 * p + 1, :, 2:])       # lo operand


def test_fragment_major_and_pack_conv_paired_fragments_on_cpu():
    """pack_conv in split precision: fragment-major copy [Cout/16][K/16 steps][4 kq][16 n][4] and, for Cin >= 32, its paired form; a weight can be
    read back from both (up to the power-of-two scale folded into `scale`)."""
    g = torch.Generator().manual_seed(12)
    Cout, Cin = 32, 32
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    p = ops.pack_conv(w, ops.F16X2, torch.ones(Cout), torch.zeros(Cout), stride=1, pad=1)
    assert p.split and p.w_frag is not None and p.w_frag_pair is not None
    K = 9 * Cin
    assert p.w_frag.shape == (Cout // 16, K // 16, 4, 16, 4) and p.w_frag_pair.shape == (Cout // 16, K // 32, 2, 4, 16, 4)
    s = 1.0 / float(p.scale[0])
    hi, lo = _halves(p.w)                                           # [Cout_pad][K_pad], k = tap * Cin + c
    back = ((hi + lo) / s)[:Cout, :K].view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    assert float((back - w).abs().max()) <= 2.0 ** -20 * float(w.abs().max())
    # element (n, k) sits in fragment n // 16, step k // 16, k-group (k % 16) // 4, row n % 16, position k % 4
    fh, _ = _halves(p.w_frag)
    n, k = 21, 137
    assert float(fh[n // 16, k // 16, (k % 16) // 4, n % 16, k % 4]) == float(hi[n, k])
    small = ops.pack_conv(torch.randn(16, 16, 3, 3, generator=g), ops.F16X2, None, None, stride=1, pad=1)
    assert small.w_frag is not None and small.w_frag_pair is None   # a 16-channel tap is one step: nothing to pair
