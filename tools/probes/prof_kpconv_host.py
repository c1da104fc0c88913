import sys, os, cProfile, pstats, io
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from bench_kpconv import synthetic_cloud
from torch_points3d_amd.kpconv_blocks import PDData
from torch_points3d_amd.kpconv_unet import KPConv
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = KPConv("unet", input_nc=3, in_feat=64, in_grid_size=0.02, num_layers=4, output_nc=13).to(dev).eval()
pos, batch = synthetic_cloud(65536, 1, 0.02)
x = torch.cat([torch.ones(65536, 1), torch.randn(65536, 3)], 1)
pos, batch, x = pos.to(dev), batch.to(dev), x.to(dev)
def step():
    with torch.no_grad():
        return model(PDData(pos=pos, batch=batch, x=x))
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
