#!/usr/bin/env python
"""Drop-in proof of the HOST API, run in the build container only (needs /root/reference).

The reference's own callers of `torch_points_kernels.points_cpu` / `region_grow` are imported from the reference tree
and run on CPU tensors twice: with `torch_points_kernels` = THIS package (torch_points_kernels/: libtp3d_cpu.so) and
with the brute-force oracle (oracle/points_cpu_ref.py) bound in its place.  Outputs must be identical; inputs and
expected outputs are written to tests/golden/dropin_host.npz for tests/test_dropin_fixture_cpu.py (which runs anywhere:
nothing of the reference travels).

Callers exercised (reference file:line):
  core/data_transform/transforms.py  RandomWalkDropout :770-816 (ball_query mode 0, max_num, the walk itself in rw_mask),
      RandomSphereDropout :832-863 (mode 1, max_num=-1), FixedSphereDropout :866-903, SphereCrop :906-928,
      DensityFilter :1022-1053 (mode 0, max_num=-1), MultiScaleTransform :579-654 run with this package's host
      strategies (torch_points3d_amd.multiscale_cpu.host_strategies of a KPConv unet's get_spatial_ops());
  models/panoptic/pointgroup.py:101-115  the two region_grow calls of PointGroup._cluster.

Stand-ins for what is not installed (none of them is on the path under test): torch_geometric.data.Data = an attribute
bag with the methods these callers use (keys, [], clone, contiguous, num_nodes, __inc__, iteration); numba.jit = identity;
torch_scatter / torch_cluster / voxel_grid = numpy restatements (only RandomSphereDropout's centre sampling touches them).
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


class Data(object):
    """torch_geometric.data.Data as far as the transforms use it"""

    def __init__(self, **kw):
        for k, v in kw.items():
            if v is not None:
                setattr(self, k, v)

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None and not k.startswith("__") and not callable(v)]

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def __iter__(self):
        for k in sorted(self.keys):
            yield k, getattr(self, k)

    def __contains__(self, k):
        return k in self.keys

    @property
    def num_nodes(self):
        return self.pos.shape[0] if getattr(self, "pos", None) is not None else None

    def __inc__(self, key, value):
        return self.num_nodes if "index" in key else 0

    def contiguous(self, *keys):
        for k in self.keys:
            if torch.is_tensor(self[k]):
                self[k] = self[k].contiguous()
        return self

    def clone(self):
        out = self.__class__()
        for k in self.keys:
            v = self[k]
            setattr(out, k, v.clone() if torch.is_tensor(v) else v)
        return out


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install(points_cpu_module, region_grow_fn):
    """(re)binds the third-party names and loads a FRESH copy of the reference's transforms module over them"""
    from oracle import voxel_ref

    def _na(*a, **k):
        raise RuntimeError("stubbed third-party function called")

    def scatter_mean(src, index, dim=0, dim_size=None):
        n = int(index.max()) + 1 if dim_size is None else dim_size
        out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, src)
        cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype))
        return out / cnt.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))

    def scatter_add(src, index, dim=0, dim_size=None):
        n = int(index.max()) + 1 if dim_size is None else dim_size
        return torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, src)

    def grid_cluster(pos, size, start=None, end=None):
        return torch.from_numpy(voxel_ref.grid_cluster_key(voxel_ref.voxel_coords(pos.numpy(), float(size[0])), None))

    def voxel_grid(pos, batch, size, start=None, end=None):
        return torch.from_numpy(voxel_ref.grid_cluster_key(voxel_ref.voxel_coords(pos.numpy(), float(size)), batch.numpy()))

    def consecutive_cluster(src):
        c, p = voxel_ref.consecutive_cluster(src.numpy())
        return torch.from_numpy(c), torch.from_numpy(p)

    for name in [n for n in sys.modules if n.startswith(("torch_points3d", "torch_points_kernels", "torch_geometric"))]:
        del sys.modules[name]
    tg = _stub("torch_geometric")
    tg.nn = _stub("torch_geometric.nn", voxel_grid=voxel_grid, knn=_na, radius=_na, fps=_na, knn_interpolate=_na,
                  global_max_pool=_na, global_mean_pool=_na)
    tg.nn.pool = _stub("torch_geometric.nn.pool")
    _stub("torch_geometric.nn.pool.consecutive", consecutive_cluster=consecutive_cluster)
    _stub("torch_geometric.nn.pool.pool", pool_pos=_na, pool_batch=_na)
    tg.data = _stub("torch_geometric.data", Data=Data, Batch=Data)
    tg.transforms = _stub("torch_geometric.transforms", FixedPoints=object)
    _stub("torch_scatter", scatter_add=scatter_add, scatter_mean=scatter_mean, scatter_max=_na)
    _stub("torch_cluster", grid_cluster=grid_cluster)
    _stub("numba", jit=lambda *a, **k: (lambda f: f))
    oc = _stub("omegaconf", OmegaConf=object, DictConfig=dict, ListConfig=list)
    oc.listconfig = _stub("omegaconf.listconfig", ListConfig=type("ListConfig", (list,), {}))
    oc.dictconfig = _stub("omegaconf.dictconfig", DictConfig=type("DictConfig", (dict,), {}))
    tpk = _stub("torch_points_kernels", region_grow=region_grow_fn)
    tpk.points_cpu = points_cpu_module
    sys.modules["torch_points_kernels.points_cpu"] = points_cpu_module
    if REF not in sys.path:
        sys.path.append(REF)  # after ROOT: `torch_points_kernels` never resolves into the reference tree anyway
    # the reference's data_transform package __init__ pulls every transform family (torch_geometric.transforms checks,
    # sparse back-ends): bind the package name to its directory and import only the module under test
    import torch_points3d.core
    pkg = types.ModuleType("torch_points3d.core.data_transform")
    pkg.__path__ = [os.path.join(REF, "torch_points3d/core/data_transform")]
    sys.modules["torch_points3d.core.data_transform"] = pkg
    torch_points3d.core.data_transform = pkg
    _stub("torch_points3d.datasets.registration.pair", Pair=Data)
    import torch_points3d.core.data_transform.transforms as T
    return T


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def run_transforms(T):
    """every points_cpu caller of transforms.py on one seeded cloud -> {name: surviving point ids}"""
    g = torch.Generator().manual_seed(2024)
    n = 1500
    pos = torch.rand(n, 3, generator=g)
    pos[200:260] = pos[0:60] + 0.002 * torch.randn(60, 3, generator=g)  # a dense clump: DensityFilter keeps it
    out = {"pos": pos}

    def fresh():
        return Data(pos=pos.clone(), ids=torch.arange(n), x=torch.arange(n, dtype=torch.float32).view(-1, 1))

    seed_all(1)
    out["density_filter"] = T.DensityFilter(radius_nn=0.12, min_num=10)(fresh()).ids
    seed_all(2)
    out["random_walk_dropout"] = T.RandomWalkDropout(dropout_ratio=0.05, num_iter=400, radius=0.08, max_num=12)(fresh()).ids
    seed_all(3)
    out["random_sphere_dropout"] = T.RandomSphereDropout(num_sphere=4, radius=0.15, grid_size_center=0.05)(fresh()).ids
    seed_all(4)
    out["fixed_sphere_dropout"] = T.FixedSphereDropout(centers=[[0.5, 0.5, 0.5], [0.1, 0.2, 0.9]], radius=0.2)(fresh()).ids
    seed_all(5)
    out["sphere_crop"] = T.SphereCrop(radius=0.3)(fresh()).ids
    return out


def run_multiscale(T, strategies):
    """the reference's MultiScaleTransform.__call__ (:603-651) over host strategies of this package"""
    g = torch.Generator().manual_seed(77)
    pos = torch.rand(3000, 3, generator=g) * 0.5
    data = Data(pos=pos)
    ms = T.MultiScaleTransform(strategies)(data)
    out = {"ms/pos": pos}
    for i, scale in enumerate(ms.multiscale):
        out["ms/%d/pos" % i] = scale.pos
        out["ms/%d/idx" % i] = scale.idx_neighboors
    for i, up in enumerate(ms.upsample):
        out["ms/up%d/x_idx" % i], out["ms/up%d/y_idx" % i] = up.x_idx, up.y_idx
        out["ms/up%d/weights" % i] = up.weights
    return out


def pointgroup_inputs():
    g = torch.Generator().manual_seed(9)
    centres = torch.rand(6, 3, generator=g)
    which = torch.randint(0, 6, (900,), generator=g)
    pos = centres[which] + 0.01 * torch.randn(900, 3, generator=g)
    labels = which % 3
    batch = torch.sort(torch.randint(0, 2, (900,), generator=g))[0]
    offsets = 0.5 * (centres[which] - pos)
    return pos, labels, batch, offsets


def run_pointgroup(region_grow):
    """PointGroup._cluster (models/panoptic/pointgroup.py:101-115): region_grow on the raw and on the vote-shifted
    positions, stuff classes ignored, nsample=200 on the second call"""
    pos, labels, batch, offsets = pointgroup_inputs()
    stuff = torch.tensor([2])
    a = region_grow(pos, labels, batch, ignore_labels=stuff, radius=0.03)
    b = region_grow(pos + offsets, labels, batch, ignore_labels=stuff, radius=0.03, nsample=200)
    canon = lambda cl: sorted(tuple(sorted(c.tolist())) for c in cl)  # noqa: E731  (the reference reads clusters as sets)
    return {"pg/pos": pos, "pg/labels": labels, "pg/batch": batch, "pg/offsets": offsets}, canon(a), canon(b)


def main():
    from oracle import points_cpu_ref
    # the product modules bind THIS package's torch_points_kernels.points_cpu at import: import them before any stub of that
    # name exists
    from torch_points3d_amd.kpconv_unet import KPConv  # only its strategy PARAMETERS are read (no device work)
    from torch_points3d_amd.multiscale_cpu import HostGridSampler, HostKnnTable, HostRadiusFinder, host_strategies
    ref_mod = types.ModuleType("points_cpu_ref_as_points_cpu")
    ref_mod.ball_query = points_cpu_ref.ball_query

    # ---- oracle-bound run
    T = install(ref_mod, points_cpu_ref.region_grow)
    want = run_transforms(T)
    model = KPConv("unet", input_nc=3, in_feat=8, in_grid_size=0.02, num_layers=4, output_nc=4)
    params = model.get_spatial_ops()

    class OracleRadius(object):
        def __init__(self, r, m):
            self.r, self.m = r, m

        def __call__(self, x, y, batch_x=None, batch_y=None):
            from oracle import tpk_ref
            return tpk_ref.ball_query(self.r, self.m, x, y, mode="partial_dense", batch_x=batch_x, batch_y=batch_y)[0]

    class OracleSampler(object):
        def __init__(self, size):
            self.size = size

        def __call__(self, data):
            from oracle import voxel_ref
            b = getattr(data, "batch", None)
            out = voxel_ref.grid_sampling_mean(data.pos.numpy(), self.size, batch=None if b is None else b.numpy())
            d = Data(pos=torch.from_numpy(out["pos"]))
            if b is not None:
                d.batch = torch.from_numpy(out["batch"])
            return d

    class OracleKnn(object):
        def __init__(self, k):
            self.k = k

        def precompute(self, query, support):
            from oracle import tpk_ref
            idx, d2 = tpk_ref.knn(self.k, query.pos, support.pos)
            n = support.pos.shape[0]
            y_idx, x_idx = torch.arange(n).repeat_interleave(self.k), idx.reshape(-1)
            w = 1.0 / torch.clamp(d2.reshape(-1, 1), min=1e-16)
            return Data(num_nodes_=n, x_idx=x_idx, y_idx=y_idx, weights=w)

    oracle_strategies = {"sampler": [None if not s else OracleSampler(s._grid_size) for s in params["sampler"]],
                         "neighbour_finder": [OracleRadius(f._radius, f._max_num_neighbors) for f in params["neighbour_finder"]],
                         "upsample_op": [OracleKnn(u.k) for u in params["upsample_op"]]}
    want.update(run_multiscale(T, oracle_strategies))
    pg_in, want_a, want_b = run_pointgroup(points_cpu_ref.region_grow)

    # ---- THIS package
    import importlib
    for name in [n for n in sys.modules if n.startswith("torch_points_kernels")]:
        del sys.modules[name]
    product = importlib.import_module("torch_points_kernels")
    product_cpu = importlib.import_module("torch_points_kernels.points_cpu")
    assert os.path.dirname(product.__file__) == os.path.join(ROOT, "torch_points_kernels")
    T = install(product_cpu, product.region_grow)
    sys.modules["torch_points_kernels"] = product  # (install() rebinds a stub; the transforms only read .points_cpu)
    got = run_transforms(T)
    hs = host_strategies(params)
    assert any(isinstance(v, HostGridSampler) for v in hs["sampler"]) and isinstance(hs["upsample_op"][0], HostKnnTable) \
        and all(isinstance(v, HostRadiusFinder) for v in hs["neighbour_finder"])

    class AsData(object):  # host strategies return this package's bags; the reference's loop wants .clone()/.contiguous()
        def __init__(self, s):
            self.s = s

        def __call__(self, data):
            has_batch = getattr(data, "batch", None) is not None
            out = self.s(Data(pos=data.pos, batch=data.batch if has_batch else torch.zeros(data.pos.shape[0], dtype=torch.long)))
            return Data(pos=out.pos, batch=out.batch if has_batch else None)  # a sample without batch stays without

    class KnnAsData(object):
        def __init__(self, u):
            self.u = u

        def precompute(self, query, support):
            bag = lambda d: Data(pos=d.pos, batch=getattr(d, "batch", torch.zeros(d.pos.shape[0], dtype=torch.long)))  # noqa: E731
            t = self.u.precompute(bag(query), bag(support))
            return Data(num_nodes_=t.num_nodes, x_idx=t.x_idx, y_idx=t.y_idx, weights=t.weights)

    product_strategies = {"sampler": [None if s is None else AsData(s) for s in hs["sampler"]],
                          "neighbour_finder": hs["neighbour_finder"],
                          "upsample_op": [KnnAsData(u) for u in hs["upsample_op"]]}
    got.update(run_multiscale(T, product_strategies))
    _, got_a, got_b = run_pointgroup(product.region_grow)

    bad = [k for k in want if not (torch.equal(want[k], got[k]) if want[k].dtype != torch.float32 or "weights" not in k
                                   else torch.allclose(want[k], got[k], rtol=1e-6, atol=0))]
    assert not bad, "this package and the oracle disagree under the reference's callers: %s" % bad
    assert got_a == want_a and got_b == want_b, "region_grow clusters differ"
    arrays = {k: v.numpy() for k, v in want.items()}
    arrays.update({k: v.numpy() for k, v in pg_in.items()})
    for tag, cl in (("pg/raw", want_a), ("pg/votes", want_b)):
        arrays[tag + "/members"] = np.array([i for c in cl for i in c], dtype=np.int64)
        arrays[tag + "/starts"] = np.cumsum([0] + [len(c) for c in cl]).astype(np.int64)
    path = os.path.join(HERE, "dropin_host.npz")
    np.savez_compressed(path, **arrays)
    kept = {k: int(v.shape[0]) for k, v in want.items() if not k.startswith("ms/") and k != "pos"}
    print("reference callers agree on this package and on the oracle; wrote %s (%.1f KiB)" % (path, os.path.getsize(path) / 1024.0))
    print("  points kept of 1500:", kept)
    print("  multiscale levels:", [int(want["ms/%d/pos" % i].shape[0]) for i in range(len(params["sampler"]))])
    print("  PointGroup clusters: %d on positions, %d on votes" % (len(want_a), len(want_b)))


if __name__ == "__main__":
    main()
