import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import load_golden
from test_golden_cpu import build_from_golden, CASES
from torch_points3d_amd.dense import Data
from oracle import tpk_ref
name = "small_ssg"
g = load_golden(name)
def run(dev, kernels):
    net = build_from_golden(g, CASES[name](), kernels, device=dev)
    cap = {}
    act = net.up_modules[2].nn[2][2]
    bn = net.up_modules[2].nn[2][1]
    act.register_forward_hook(lambda m, i, o: cap.update(act_in=i[0].detach().cpu(), act_out=o.detach().cpu()))
    bn.register_forward_hook(lambda m, i, o: cap.update(bn_in=i[0].detach().cpu()))
    act.register_full_backward_hook(lambda m, gi, go: cap.update(act_gi=gi[0].detach().cpu(), act_go=go[0].detach().cpu()))
    x_in = g["x"].to(dev).requires_grad_(True)
    out = net(Data(pos=g["pos"].to(dev), x=x_in))
    (out.x * g["cotangent"].to(dev)).sum().backward()
    return cap
a = run("cuda:0", None); b = run("cpu", tpk_ref)
for k in a:
    print(k, a[k].shape, (a[k]-b[k]).abs().max().item(), b[k].abs().max().item())
ma = a["act_in"] > 0; mb = b["act_in"] > 0
print("sign flips", (ma != mb).sum().item(), "of", ma.numel())
ratio_a = a["act_gi"] / a["act_go"]; ratio_b = b["act_gi"] / b["act_go"]
print("gpu ratio uniq", torch.unique(ratio_a.round(decimals=3))[:10])
print("cpu ratio uniq", torch.unique(ratio_b.round(decimals=3))[:10])
bad = (a["act_gi"] - b["act_gi"]).abs() > 1e-3
print("bad count", bad.sum().item())
i = bad.nonzero()[:5]
for r in i:
    r = tuple(r.tolist()); print(r, "in", a["act_in"][r].item(), b["act_in"][r].item(), "go", a["act_go"][r].item(), b["act_go"][r].item(), "gi", a["act_gi"][r].item(), b["act_gi"][r].item())
