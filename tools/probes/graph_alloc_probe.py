"""Does a captured hipGraph survive eager allocations between replays?  control (pure torch) / inference graph / pieces."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
dev = "cuda"

def churn():
    t = torch.empty(1 << 30, dtype=torch.uint8, device=dev); t.fill_(77); del t
    ts = [torch.full((1 << 20,), 3.0, device=dev) for _ in range(64)]; del ts
    torch.cuda.synchronize()

def capture(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out

def check(name, fn, outs=lambda o: o):
    g, out = capture(fn)
    g.replay(); torch.cuda.synchronize()
    ref = [t.clone() for t in outs(out)]
    churn()
    g.replay(); torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for a, b in zip(ref, outs(out)))
    print("%-34s replay after eager alloc/free identical: %s" % (name, ok), flush=True)

x = torch.randn(1 << 22, device=dev)
check("control: torch elementwise chain", lambda: ((x * 2 + 1).relu().sqrt(),))
import test_gpu_e2e as E
from monoflex_amd import synthetic as S, ops, lib as L
from monoflex_amd.structures.params_3d import make_test_target
m = E._hip_model(-1.0, sys.argv[1] if len(sys.argv) > 1 else "fp32", 96, 32)
imgs = S.synthetic_images(2, 128, 384, seed=1000).to(dev)
tg = m.device_targets([make_test_target(S.synthetic_target(96, 32)) for _ in range(2)], dev)
with torch.no_grad():
    check("stem + level0/1 (conv2d)", lambda: (m.backbone.base(imgs, m.compute_dtype)[1],))
    check("DLA base (all conv kinds)", lambda: tuple(m.backbone.base(imgs, m.compute_dtype)))
    check("backbone (DCN + upsample)", lambda: (m.backbone.forward_nhwc(imgs),))
    check("full detect_device", lambda: m.detect_device(imgs, *tg))

# ---- training pieces (fp32 atomics: compare loosely; garbage / NaN / faults are what we look for)
import test_gpu_train_step as TS
from monoflex_amd.solver import build_optimizer
def close(a, b):
    return bool(torch.isfinite(b).all()) and float((a - b).abs().max()) <= 2e-2 * max(1.0, float(a.abs().max()))
def check_train(name, make):
    fn, outs = make()
    g, out = capture(fn)
    g.replay(); torch.cuda.synchronize()
    ref = [t.detach().clone() for t in outs(out)]
    churn()
    g.replay(); torch.cuda.synchronize()
    now = [t.detach() for t in outs(out)]
    print("%-34s replay after eager alloc/free consistent: %s  (%s -> %s)" % (name, all(close(a, b) for a, b in zip(ref, now)),
          [round(float(t.float().abs().mean()), 4) for t in ref][:3], [round(float(t.float().abs().mean()), 4) for t in now][:3]), flush=True)
dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
mt = TS._model(dt)
timgs, ttg = TS._batch(mt)
def mk_fwd():
    def fn():
        with torch.no_grad():
            ld, _ = mt(timgs, ttg)
        return (sum(ld.values()),)
    return fn, (lambda o: o)
def mk_fwd_grad():
    def fn():
        ld, _ = mt(timgs, ttg)
        return (sum(ld.values()).detach(),)
    return fn, (lambda o: o)
def mk_fwd_bwd():
    def fn():
        ld, _ = mt(timgs, ttg)
        l = sum(ld.values())
        mt.zero_grad(set_to_none=True)
        l.backward()
        return (l.detach(), mt.heads.predictor.class_head[0].weight.grad, mt.backbone.base.level2.tree1.conv1.weight.grad)
    return fn, (lambda o: o)
check_train("train forward, no_grad", mk_fwd)
check_train("train forward, autograd graph kept", mk_fwd_grad)
check_train("train forward + backward", mk_fwd_bwd)

def mk_backbone():
    def fn():
        with torch.no_grad():
            return (mt.backbone.forward_nhwc(timgs).float(),)
    return fn, (lambda o: o)
def mk_base():
    def fn():
        with torch.no_grad():
            return tuple(t.float() for t in mt.backbone.base(timgs, mt.compute_dtype))
    return fn, (lambda o: o)
feat = mt.backbone.forward_nhwc(timgs).detach()
def mk_pred():
    def fn():
        with torch.no_grad():
            c, r = mt.heads.predictor.forward_train(feat, *ttg.edge)
        return (c, r)
    return fn, (lambda o: o)
maps = mt.heads.predictor(feat.permute(0, 3, 1, 2), ttg)
maps = {k: v.detach() for k, v in maps.items()}
def mk_loss():
    def fn():
        ld, _ = mt.heads.loss_evaluator(maps, ttg)
        return (sum(ld.values()),)
    return fn, (lambda o: o)
check_train("train: DLA base only", mk_base)
check_train("train: backbone only", mk_backbone)
check_train("train: predictor.forward_train", mk_pred)
check_train("train: loss evaluator", mk_loss)
