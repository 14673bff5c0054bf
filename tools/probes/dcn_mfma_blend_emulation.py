#!/usr/bin/env python
"""Lane-level emulation (numpy, CPU) of a candidate inner loop for the LDS-patch DCN kernel: the four-corner bilinear blend on the
matrix cores instead of 16 packed-fp16 VALU instructions per fragment ("DCN main in inference").

For one wave, one tap, 16 output pixels, C input channels, Cout output channels:

  blend (transposed GEMM, K = 64 gathered rows = 16 pixels x 4 corners, two K-steps of 32):
      T[c][p] = sum_k  Xg[k][c] * Wb[k][p]        k = 4 * pixel + corner;  Wb[k][p] = bilinear weight (x mask) if pixel(k) == p else 0
    A operand = Xg^T, read from the patch with ds_read_b64_tr_b16 (rows = gathered patch pixels at data-dependent addresses),
    B operand = Wb, built in registers from the lane's own pixel weights (two of eight entries non-zero in one K-group per K-step),
    D = T: lane (pixel l & 15, group l >> 4) holds channels 16*cb + 4*(l >> 4) + r, r = 0..3  -- which IS the B-operand layout of
  main GEMM (transposed):  Y[o][p] = sum_c  Wm[o][c] * T[c][p]     (K-step = 32 channels = D fragments 2u, 2u + 1;
    the K order inside a step follows the accumulators -- the weights are packed with the same permutation, as in heads.hip).

The emulation implements the two instructions' lane semantics as probed on gfx950 (tools/probes/tr_probe.hip,
cdna_hip_programming.md section 3) and checks the result against the direct formula.  It fixes the operand layouts before any HIP
is written; nothing here is part of the product path.
"""
import numpy as np

rng = np.random.default_rng(0)
LANES = 64


def mfma_16x16x32(A, B, C):
    """A[l][i] = A_mat[m = l & 15][k = 8 * (l >> 4) + i];  B[l][i] = B_mat[k = 8 * (l >> 4) + i][n = l & 15];
    C / D[l][r] = mat[m = 4 * (l >> 4) + r][n = l & 15]."""
    Am, Bm = np.zeros((16, 32)), np.zeros((32, 16))
    for l in range(LANES):
        for i in range(8):
            Am[l & 15, 8 * (l >> 4) + i] = A[l][i]
            Bm[8 * (l >> 4) + i, l & 15] = B[l][i]
    Dm = Am @ Bm
    D = np.array(C, dtype=np.float64).copy()
    for l in range(LANES):
        for r in range(4):
            D[l][r] += Dm[4 * (l >> 4) + r, l & 15]
    return D


def ds_read_b64_tr_b16(lds, addr):
    """lds: 1-D array of 16-bit elements, addr[l] = element index (a multiple of 4: 8-byte aligned).  Inside each 16-lane group,
    lanes 4j .. 4j + 3 supply row j as four pieces of four elements; lane with local index c receives column c of rows 0..3."""
    out = np.zeros((LANES, 4))
    for g in range(4):
        rows = np.zeros((4, 16))
        for j in range(4):
            for p in range(4):
                a = addr[16 * g + 4 * j + p]
                assert a % 4 == 0
                rows[j, 4 * p:4 * p + 4] = lds[a:a + 4]
        for c in range(16):
            out[16 * g + c] = rows[:, c]
    return out


def main(C=64, Cout=64, PW=32, PH=32):
    PS = C + 8                                           # patch pixel stride in elements (padded layout: + 16 bytes)
    patch = rng.standard_normal(PW * PH * PS)
    # 16 output pixels of one row: top-left corner (ry, rx) inside the patch, four bilinear weights (mask folded in)
    ry, rx = rng.integers(0, PH - 1, 16), rng.integers(0, PW - 1, 16)
    wgt = rng.random((16, 4))
    Wm = rng.standard_normal((Cout, C))                  # weights of this tap

    def corner_addr(p, j):                                # element address of corner j of pixel p (channel 0)
        return ((ry[p] + (j >> 1)) * PW + rx[p] + (j & 1)) * PS

    # ---- direct formula
    T_ref = np.zeros((C, 16))
    for p in range(16):
        for j in range(4):
            a = corner_addr(p, j)
            T_ref[:, p] += wgt[p, j] * patch[a:a + C]
    Y_ref = Wm @ T_ref

    # ---- lane-level emulation
    lane = np.arange(LANES)
    xl, kq = lane & 15, lane >> 4
    # blend: D fragment per 16-channel block
    Tfrag = []
    for cb in range(C // 16):
        acc = np.zeros((LANES, 4))
        for s in range(2):                                # K-step: pixels 8s .. 8s + 7
            A = np.zeros((LANES, 8))
            for t in range(2):                            # transposed read t: rows k_local = 8 * kq + 4 * t + j = pixel 8s + 2kq + t, corner j
                addr = np.zeros(LANES, dtype=np.int64)
                for l in range(LANES):
                    j, piece = (l & 15) >> 2, l & 3
                    pix = 8 * s + 2 * (l >> 4) + t
                    addr[l] = corner_addr(pix, j) + cb * 16 + piece * 4
                A[:, 4 * t:4 * t + 4] = ds_read_b64_tr_b16(patch, addr)
            B = np.zeros((LANES, 8))                       # lane (pixel xl, group kq): entries i = 4 * (pixel - (8s + 2kq)) + corner
            for l in range(LANES):
                d = xl[l] - (8 * s + 2 * kq[l])
                if d in (0, 1):
                    B[l, 4 * d:4 * d + 4] = wgt[xl[l]]
            acc = mfma_16x16x32(A, B, acc)
        Tfrag.append(acc)                                 # lane: channels 16*cb + 4*kq + r of pixel xl
    for cb in range(C // 16):
        for l in range(LANES):
            for r in range(4):
                assert abs(Tfrag[cb][l][r] - T_ref[16 * cb + 4 * kq[l] + r, xl[l]]) < 1e-9
    # main GEMM: K-step u = channels 32u .. 32u + 31 in accumulator order
    def kperm(u, q, i):                                   # channel held at position i of lane group q in K-step u
        return 32 * u + (4 * q + i if i < 4 else 16 + 4 * q + (i - 4))
    Y = []
    for ob in range(Cout // 16):
        acc = np.zeros((LANES, 4))
        for u in range(C // 32):
            A = np.zeros((LANES, 8))                       # pre-permuted weights: lane (o = 16*ob + xl, group kq)
            Bop = np.zeros((LANES, 8))
            for l in range(LANES):
                for i in range(8):
                    A[l, i] = Wm[16 * ob + xl[l], kperm(u, kq[l], i)]
                Bop[l, :4] = Tfrag[2 * u][l]
                Bop[l, 4:] = Tfrag[2 * u + 1][l]
            acc = mfma_16x16x32(A, Bop, acc)
        Y.append(acc)
    err = 0.0
    for ob in range(Cout // 16):
        for l in range(LANES):
            for r in range(4):
                err = max(err, abs(Y[ob][l][r] - Y_ref[16 * ob + 4 * kq[l] + r, xl[l]]))
    print("blend fragments exact; main GEMM max |err| %.2e  (lane holds output channels 16*ob + 4*(l>>4) + r of pixel l&15)" % err)
    assert err < 1e-9
    n_tr, n_mfma = (C // 16) * 2 * 2, (C // 16) * 2 + (Cout // 16) * (C // 32)
    print("per (16 pixels, tap): %d ds_read_b64_tr_b16, %d MFMAs (%d for the blend), no blend VALU" % (n_tr, n_mfma, (C // 16) * 2))


if __name__ == "__main__":
    main()
    main(C=32, Cout=64)
