#!/usr/bin/env python
"""Timing probes of the fused heads kernel at the bench shape (B=8 bf16), all variants interleaved in one process:
heads_persist 0 / 1 (one unit range per resident workgroup; bit-identical) / heads_dbg=1 (no weight stream in
the K loop) / 2 (no LDS pixel reads) / 3 (neither) -- the dbg variants compute wrong results and only bound what each stream costs.
  usage (GPU box): python tools/probes/heads_probe.py [rounds]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from monoflex_amd import lib, ops

L = lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
model, _, _ = bench.build_model("bf16", torch.device("cuda", 0))
feat = torch.randn(8, 96, 320, 64, device="cuda").relu().to(model.compute_dtype)
pk = model.heads.predictor._pack(feat.dtype)


def opt(**kv):
    for k, v in kv.items():
        lib.check(L.mfx_set_option(k.encode(), int(v)), "opt")


def run():
    return ops.heads_fused(feat, pk, planar_classes=3)


def timed(reps=10):
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


Z = dict(heads_persist=0, heads_dbg=0, heads_planes=0)
variants = [("one workgroup per tile", dict(Z, heads_persist=0)), ("persist", dict(Z, heads_persist=1)), ("persist+planes", dict(Z, heads_persist=1, heads_planes=1)),
            ("persist 1024", dict(Z, heads_persist=1024)),
            ("no weight stream", dict(Z, heads_dbg=1)), ("no lds reads", dict(Z, heads_dbg=2)), ("neither", dict(Z, heads_dbg=3))]
opt(**variants[0][1])
written = torch.zeros(pk.ld_out, dtype=torch.bool)            # the row's gaps between branches are never written
for o, c in zip(pk.ch_off, pk.c_out):
    written[o:o + c] = True
written = written.cuda()
ref = [t.clone() for t in run()]
for n, kv in variants[1:4]:
    opt(**kv)
    got = run()
    print("%s bit-identical to default:" % n, torch.equal(ref[0][..., written], got[0][..., written]) and torch.equal(ref[1], got[1]))
times = {n: [] for n, _ in variants}
for r in range(rounds):
    for n, kv in variants:
        opt(**kv)
        times[n].append(timed())
opt(**variants[1][1])
for n, _ in variants:
    t = sorted(times[n])
    print("%-18s median %.1f us  (min %.1f max %.1f)" % (n, t[len(t) // 2], t[0], t[-1]))
