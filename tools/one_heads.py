#!/usr/bin/env python
"""Run the fused heads kernel alone (for rocprofv3 --pmc).  usage: one_heads.py [reps] [dtype]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from monoflex_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
model, _ = bench.build_model(dtype, "cuda")
feat = torch.randn(8, 96, 320, 64, device="cuda").relu().to(model.compute_dtype)
pk = model.heads.predictor._pack(feat.dtype)
for _ in range(reps):
    ops.heads_fused(feat, pk, planar_classes=3)
torch.cuda.synchronize()
