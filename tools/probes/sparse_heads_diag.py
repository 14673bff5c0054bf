import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_train_step as T
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
def run(sparse):
    m = T._model(dtype)
    m.heads.sparse_regression = sparse
    imgs, tg = T._batch(m, B=2)
    l, _ = m(imgs, tg)
    sum(l.values()).backward()
    return m, l
a, la = run(True); b, lb = run(False); c, lc = run(False)
pa, pb, pc = dict(a.named_parameters()), dict(b.named_parameters()), dict(c.named_parameters())
for n in pa:
    if not n.startswith("heads.predictor.reg_") or pa[n].grad is None: continue
    ga, gb, gc = pa[n].grad.flatten().double(), pb[n].grad.flatten().double(), pc[n].grad.flatten().double()
    r1 = float((ga - gb).norm() / gb.norm().clamp(min=1e-30)); r0 = float((gc - gb).norm() / gb.norm().clamp(min=1e-30))
    print("%-50s |g| %.3e  sparse-vs-dense %.3e   dense-vs-dense %.3e" % (n, float(gb.norm()), r1, r0))
