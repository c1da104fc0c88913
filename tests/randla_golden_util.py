"""Shared by the CPU and GPU tests against tests/golden/randla.npz (reference RandLANet classes; make_golden.py
`make_randla_case`): builds this package's modules with the fixture's weights and replays its random draws."""
import torch

F = 6


def sub_state(gold, prefix):
    """state_dict stored under '<prefix>/sd/<key>'"""
    p = prefix + "/sd/"
    return {k[len(p):]: v for k, v in gold.items() if k.startswith(p)}


def build_kernel(gold, tag, with_x, device):
    from torch_points3d_amd.randla import RandlaKernel
    cin = F if with_x else 3
    ker = RandlaKernel(point_pos_nn=[10, 8, F], attention_nn=[cin + F, 8, cin + F], global_nn=[cin + F, 8, 16])
    ker.load_state_dict(sub_state(gold, tag), strict=True)  # the reference's keys, all of them
    return ker.to(device)


def build_blocks(gold, device):
    """the two RandLANetRes down modules of conf/models/segmentation/randlanet.yaml (Randlanet_Res, FEAT = 6)"""
    from torch_points3d_amd.randla import RandLANetRes
    b0 = RandLANetRes(indim=3, outdim=32, ratio=[1, 1], point_pos_nn=[[10, 8, F], [10, 16, 16]],
                      attention_nn=[[2 * F, 8, 2 * F], [32, 64, 32]], down_conv_nn=[[2 * F, 8, 16], [32, 64, 32]],
                      index=0, nb_feature=F)
    b1 = RandLANetRes(indim=32, outdim=128, ratio=[0.5, 0.5], point_pos_nn=[[10, 16, 32], [10, 32, 64]],
                      attention_nn=[[64, 128, 64], [128, 256, 128]], down_conv_nn=[[64, 64, 64], [128, 128, 128]],
                      index=1, nb_feature=F)
    net = torch.nn.ModuleDict({"b0": b0, "b1": b1})
    net.load_state_dict(sub_state(gold, "blk"), strict=True)
    return net.to(device)


class Replay(object):
    """stands in for RandomSampler: hands out the indices the reference drew (torch.randint on its CPU generator)"""

    def __init__(self, draws, device):
        self.draws, self.device, self.i = draws, device, 0

    def __call__(self, pos, x=None, batch=None):
        out = self.draws[self.i % len(self.draws)].to(self.device)
        self.i += 1
        return out


def replay_draws(net, gold, device):
    draws = [gold["blk/draw%d" % i] for i in range(4)]
    replay = Replay(draws, device)
    for blk in (net["b0"], net["b1"]):
        blk._conv.conv1.sampler = replay
        blk._conv.conv2.sampler = replay
    return replay


def bound(fp32, fp64, floor=1e-5):
    """the test tolerance: 1e-5, or twice the reference pass's own distance to its float64 evaluation where that is
    larger (sums over 16 edges feeding BatchNorm over a few hundred rows)"""
    own = float((fp32.double().cpu() - torch.as_tensor(fp64)).abs().max())
    return max(floor, 2.0 * own)
