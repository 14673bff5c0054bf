import sys
sys.path.insert(0, "/root/repo")
import torch
from monoflex_amd import lib, autograd as AG
from monoflex_amd import gram_heads as GH
lib.load()
DEV = "cuda"
g = torch.Generator().manual_seed(5)
x = (torch.randn(2, 32, 96, 64, generator=g) * 0.8 + 0.1).to(DEV)
def run():
    R5 = GH._autocorr5(x)
    S0 = AG._colsum(x)
    return R5, S0
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        e = run()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
e = [t.clone() for t in e]
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    o = run()
for rep in range(3):
    gr.replay(); torch.cuda.synchronize()
    print("replay", rep, [float((a - b).abs().max() / b.abs().max()) for a, b in zip(o, e)], [float(a.abs().max()) for a in o])
