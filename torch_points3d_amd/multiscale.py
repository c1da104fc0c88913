"""Pre-computation of the sampling / neighbour-search chain of a partial-dense network.

What the reference's `MultiScaleTransform` produces per sample inside DataLoader workers
(torch_points3d/core/data_transform/transforms.py:579-654; the strategies come from
`models/base_architectures/unet.py:315-334`, here `KPConvUnet.get_spatial_ops()`): per block the query cloud with its
neighbour table, per strided block the interpolation table of the matching decoder stage -- the `precomputed=` inputs of
the blocks (`modules/KPConv/blocks.py:71-82`, `core/base_conv/partial_dense.py:124-133`).  With them a forward pass has no
sampling, no search and no host read: every shape is static and the pass can be captured into a HIP graph.

One engine, `LevelChain`, serves both forms.  A level is a pair of callables
    sample(parent) -> child cloud or None (None: the level keeps its parent's points)
    search(parent, child) -> (rows of child, max_num) table of parent rows
and a strided level may own `table(child, parent)`, the up-sampling table from the child back to its parent.
    * `MultiScaleTransform` (this file) binds the model's own strategy objects: HIP grid sampling, radius search, kNN on the
      already batched cloud, on the device;
    * `multiscale_cpu.MultiScaleTransformCPU` binds host strategies over libtp3d_cpu.so for forked DataLoader workers.
"""
import torch

from .kpconv_blocks import PDData


class LevelChain(object):
    """levels: [(sample, search)], one per block in forward order; up_tables: [table] consumed by the strided levels in
    that order (fewer tables than strided levels is the reference's "missing upsample blocks" error)."""

    def __init__(self, levels, up_tables):
        self.levels = list(levels)
        self.up_tables = list(up_tables)

    def run(self, root):
        """root: bag with pos (N,3) and batch (N,) -> (clouds per level, up-sampling tables innermost first)"""
        clouds, tables = [], []
        pending = iter(self.up_tables)
        parent = root
        for sample, search in self.levels:
            child = sample(parent) if sample is not None else None
            if child is None:
                child = _same_points(parent)
            elif self.up_tables:
                table = next(pending, None)
                if table is None:
                    raise ValueError("You are missing some upsample blocks in your network")
                tables.append(table(child, parent))
            child.idx_neighboors = search(parent, child)
            clouds.append(child)
            parent = child
        return clouds, tables[::-1]


def _same_points(bag):
    out = PDData(pos=bag.pos, batch=bag.batch)
    bounds = getattr(bag, "pos_bounds", None)
    if bounds is not None:
        out.pos_bounds = bounds  # (host floats of the ancestors' voxel extent: lets grid sampling skip a device reduction)
    return out


def attach(data, clouds, tables):
    """data + `multiscale` / `upsample`, the attribute names the blocks read (datasets/multiscale_data.py:9-60)"""
    out = data.shallow_copy() if hasattr(data, "shallow_copy") else PDData(**vars(data))
    out.multiscale, out.upsample = clouds, tables
    return out


class MultiScaleTransform(object):
    """Device form: `strategies` = {"sampler": [...], "neighbour_finder": [...], "upsample_op": [...]} as the model lists them."""

    def __init__(self, strategies):
        self.strategies = strategies
        self.num_layers = len(strategies["sampler"])
        levels = []
        for sampler, finder in zip(strategies["sampler"], strategies["neighbour_finder"]):
            levels.append((None if not sampler else (lambda parent, s=sampler: s(_same_points(parent))),
                           lambda parent, child, f=finder: f(parent.pos, child.pos, batch_x=parent.batch, batch_y=child.batch)))
        self.chain = LevelChain(levels, [u.precompute for u in strategies["upsample_op"]])

    def __call__(self, data):
        """data: pos (N,3) [, batch (N,)] on the device -> data + multiscale=[...], upsample=[...]"""
        batch = getattr(data, "batch", None)
        if batch is None:
            batch = torch.zeros(data.pos.shape[0], dtype=torch.long, device=data.pos.device)
        root = _same_points(PDData(pos=data.pos, batch=batch, pos_bounds=getattr(data, "pos_bounds", None)))
        with torch.no_grad():
            clouds, tables = self.chain.run(root)
        return attach(data, clouds, tables)

    def __repr__(self):
        return "{}({} levels)".format(self.__class__.__name__, self.num_layers)
