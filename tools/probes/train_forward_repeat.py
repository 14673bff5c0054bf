#!/usr/bin/env python
"""Is the TRAINING-mode forward pass repeatable?  Deterministic mode (fixed-order reductions), the small test network (tests/test_gpu_train_step.py: 128 x 64
output map, B = 2) and the full-size one: the same state and batch REPS times; the backbone feature map, the class logits, the gathered regression rows and
the summed loss are compared bit for bit with the first run.  A difference = a race / hazard in some training-forward kernel."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from monoflex_amd import lib as L, synthetic as S
from monoflex_amd.config import get_cfg
from monoflex_amd.engine.trainer import prepare_targets, total_loss
from monoflex_amd.model.detector import KeypointDetector
from monoflex_amd.structures.params_3d import make_train_target

DEV = "cuda"
REPS = int(os.environ.get("REPS", "60"))
L.load()
L.set_deterministic(os.environ.get("DET", "1") == "1")
for dtype in os.environ.get("MODES", "bf16 fp16 fp32").split():
    for (ow, oh, B) in ((128, 64, 2), (320, 96, 8)):
        cfg = get_cfg("runs/monoflex.yaml")
        cfg.MODEL.PRETRAIN = False
        cfg.MODEL.COMPUTE_DTYPE = dtype
        cfg.INPUT.WIDTH_TRAIN, cfg.INPUT.HEIGHT_TRAIN = ow * 4, oh * 4
        m = KeypointDetector(cfg)
        m.load_state_dict(S.synthetic_state_dict(m.state_dict(), seed=3, cls_bias=-1.0))
        m = m.to(DEV).train()
        m.heads.loss_evaluator.log_as_float = False
        tg = [make_train_target(S.synthetic_train_target(20 + i, out_w=ow, out_h=oh, n_obj=3 + i)).to(DEV) for i in range(B)]
        imgs = S.synthetic_images(B, oh * 4, ow * 4, seed=20).to(DEV)
        pt = prepare_targets(m, tg, DEV)
        keep = {}
        hooks = [m.backbone.register_forward_hook(lambda mod, i, o: keep.__setitem__("feat", o.detach().clone()))]
        orig = m.heads.predictor.forward

        def fwd(*a, **k):
            out = orig(*a, **k)
            keep["logits"] = out["cls_logits_nhwc"].detach().clone() if out.get("cls_logits_nhwc") is not None else None
            keep["rows"] = out["reg_rows"].detach().clone() if out.get("reg_rows") is not None else None
            return out
        m.heads.predictor.forward = fwd
        base, bad = None, {"feat": 0, "logits": 0, "rows": 0, "loss": 0}
        if os.environ.get("GRAPH", "0") == "1":
            # the same forward captured once and REPLAYED (kernels back to back, no launch gaps: other timing than the eager loop)
            from monoflex_amd import autograd as AG
            out = {}

            def fwd_once():
                AG.pack_all_weights()
                ld, _ = m(imgs, pt)
                out["loss"] = total_loss(ld).detach()
                out["feat"], out["logits"], out["rows"] = keep["feat"], keep["logits"], keep["rows"]
            with torch.no_grad():
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    fwd_once(); fwd_once()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    fwd_once()
                for r in range(REPS * 3):
                    g.replay()
                    torch.cuda.synchronize()
                    cur = {k: (v.clone() if v is not None else None) for k, v in out.items()}
                    if base is None:
                        base = cur
                        continue
                    for k in bad:
                        if base[k] is not None and not torch.equal(cur[k], base[k]):
                            bad[k] += 1
                            if bad[k] == 1:
                                d = (cur[k].float() - base[k].float()).abs()
                                print("   first difference in %s at replay %d: %d elements, max %.3e (values up to %.3e)" % (k, r, int((d > 0).sum()), float(d.max()), float(base[k].float().abs().max())), flush=True)
            print("%-5s %dx%d B=%d: of %d REPLAYS differ: %s" % (dtype, ow, oh, B, REPS * 3 - 1, bad), flush=True)
            continue
        with torch.no_grad():
            for r in range(REPS):
                from monoflex_amd import autograd as AG
                AG.pack_all_weights()
                ld, _ = m(imgs, pt)
                keep["loss"] = total_loss(ld).detach().clone()
                torch.cuda.synchronize()
                if base is None:
                    base = dict(keep)
                    continue
                for k in bad:
                    if base[k] is not None and not torch.equal(keep[k], base[k]):
                        bad[k] += 1
                        if bad[k] == 1:
                            d = (keep[k].float() - base[k].float()).abs()
                            print("   first difference in %s at repeat %d: %d elements, max %.3e (values up to %.3e)" % (k, r, int((d > 0).sum()), float(d.max()), float(base[k].float().abs().max())), flush=True)
        for h in hooks:
            h.remove()
        print("%-5s %dx%d B=%d: of %d repeats differ: %s" % (dtype, ow, oh, B, REPS - 1, bad), flush=True)
