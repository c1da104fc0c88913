import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import load_golden
from test_golden_cpu import build_from_golden, CASES
from torch_points3d_amd.dense import Data
from oracle import tpk_ref

name = sys.argv[1] if len(sys.argv) > 1 else "small_ssg"
g = load_golden(name)
def run(dev, kernels):
    net = build_from_golden(g, CASES[name](), kernels, device=dev)
    grads = {}
    hooks = []
    def mk(tag):
        def h(mod, gin, gout):
            grads[tag] = [None if t is None else t.detach().cpu() for t in gin]
        return h
    for n, m in net.named_modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.BatchNorm2d, torch.nn.Conv1d, torch.nn.BatchNorm1d, torch.nn.LeakyReLU)):
            hooks.append(m.register_full_backward_hook(mk(n)))
    x_in = g["x"].to(dev).requires_grad_(True)
    out = net(Data(pos=g["pos"].to(dev), x=x_in))
    (out.x * g["cotangent"].to(dev)).sum().backward()
    grads["x_in"] = [x_in.grad.cpu()]
    return grads
a = run("cuda:0", None)
b = run("cpu", tpk_ref)
for k in b:
    for i, (ta, tb) in enumerate(zip(a[k], b[k])):
        if ta is None or tb is None: continue
        err = (ta - tb).abs().max().item(); sc = tb.abs().max().item()
        print("%-40s %d  maxabs %.3e  scale %.3e  rel %.2e" % (k, i, err, sc, err / (sc + 1e-30)))
