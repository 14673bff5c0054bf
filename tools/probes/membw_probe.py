#!/usr/bin/env python
"""What a plain streaming write / read / copy achieves on this box (torch kernels, 10 launches per hipGraph replay): the yardstick for the kernels whose
floor is the bytes they must move (project-then-sample DCN: the projected map is written once and read once)."""
import torch
N = 10
def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * N) * 1e3
for mb in (17.7, 35.4, 70.8, 141.6, 283, 566):
    n = int(mb * 1e6 / 2)
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.randn(n, device="cuda").to(torch.bfloat16)
    tw = timed(lambda: a.fill_(1.0)); tr = timed(lambda: b.sum()); tc = timed(lambda: a.copy_(b))
    print("%.1f MB: fill %.1f us (%.2f TB/s)  sum-read %.1f us (%.2f TB/s)  copy %.1f us (%.2f TB/s moved)" % (mb, tw, mb / tw / 1e6 * 1e6 / 1e6 * 1, tr, mb / tr, tc, 2 * mb / tc))
