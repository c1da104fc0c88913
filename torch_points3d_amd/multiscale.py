"""Pre-computation of the sampling / neighbour-search chain of a partial-dense network, on the device.

Mirrors `MultiScaleTransform` (torch_points3d/core/data_transform/transforms.py:579-654): given the network's
`strategies` (lists of samplers, neighbour finders and up-samplers, one entry per block --
`models/base_architectures/unet.py:315-334` collects them, here `KPConvUnet.get_spatial_ops()`), it walks the levels
once and records, per block, the query cloud and its neighbour table and, per strided block, the interpolation table
of the matching decoder stage.  The result feeds the blocks' `precomputed=` arguments
(`modules/KPConv/blocks.py:71-82`, `core/base_conv/partial_dense.py:124-133`), after which a forward pass contains no
sampling, no search and no host read -- every shape is static, so it can be captured into a HIP graph.

The reference runs this per sample on the CPU inside DataLoader workers; here it runs on the (already batched) cloud
with the HIP grid sampling, radius search and kNN.
"""
import torch

from .kpconv_blocks import PDData


class MultiScaleTransform(object):
    def __init__(self, strategies):
        self.strategies = strategies
        self.num_layers = len(self.strategies["sampler"])

    def __call__(self, data):
        """data: pos (N,3) [, batch (N,)] on the device -> PDData(multiscale=[...], upsample=[...]) + data's attributes"""
        batch = getattr(data, "batch", None)
        if batch is None:
            batch = torch.zeros(data.pos.shape[0], dtype=torch.long, device=data.pos.device)
        precomputed = [PDData(pos=data.pos, batch=batch)]
        upsample = []
        upsample_index = 0
        with torch.no_grad():
            for index in range(self.num_layers):
                sampler = self.strategies["sampler"][index]
                neighbour_finder = self.strategies["neighbour_finder"][index]
                support = precomputed[index]
                new_data = PDData(pos=support.pos, batch=support.batch)
                if getattr(support, "pos_bounds", None) is not None:
                    new_data.pos_bounds = support.pos_bounds
                if sampler:
                    query = sampler(new_data)
                    if len(self.strategies["upsample_op"]):
                        if upsample_index >= len(self.strategies["upsample_op"]):
                            raise ValueError("You are missing some upsample blocks in your network")
                        upsampler = self.strategies["upsample_op"][upsample_index]
                        upsample_index += 1
                        upsample.append(upsampler.precompute(query, support))
                else:
                    query = new_data
                query.idx_neighboors = neighbour_finder(support.pos, query.pos, batch_x=support.batch,
                                                        batch_y=query.batch)
                precomputed.append(query)
        out = data.shallow_copy() if hasattr(data, "shallow_copy") else PDData(**vars(data))
        out.multiscale = precomputed[1:]
        upsample.reverse()  # innermost decoder stage first
        out.upsample = upsample
        return out

    def __repr__(self):
        return "{}".format(self.__class__.__name__)
