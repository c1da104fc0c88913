"""ctypes binding of libtp3d_hip.so (C-ABI declared in include/tp3d_hip.h).

There is no CPU fallback: if the library is missing or a call fails, the error is raised to the caller.
"""
import collections
import ctypes
import os

import torch  # noqa: F401  (loads the HIP runtime the library resolves against)

from . import build as _build

_p = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_int64
_f = ctypes.c_float

# name -> argtypes; every function returns int (0 = ok). Mirrors include/tp3d_hip.h one to one.
SIGNATURES = {
    "tp3d_fps_f32": [_p, _i, _i, _i, _p, _p, _p],
    "tp3d_ball_query_dense_f32": [_p, _p, _i, _i, _i, _f, _i, _i, _p, _p, _p, ctypes.c_size_t, _p],
    "tp3d_ball_query_partial_dense_f32": [_p, _p, _p, _p, _l, _l, _f, _i, _i, _p, _p, _p, _i, _i, _p, ctypes.c_size_t,
                                          _i, _p],
    "tp3d_three_nn_f32": [_p, _p, _i, _i, _i, _p, _p, _p],
    "tp3d_three_interpolate_fwd_f32": [_p, _p, _p, _i, _i, _i, _i, _p, _p],
    "tp3d_three_interpolate_bwd_f32": [_p, _p, _p, _i, _i, _i, _i, _p, _p, ctypes.c_size_t, _p],
    "tp3d_group_fwd_f32": [_p, _p, _i, _i, _i, _i, _i, _p, _p],
    "tp3d_group_bwd_f32": [_p, _p, _i, _i, _i, _i, _i, _p, _p, ctypes.c_size_t, _p],
    "tp3d_group_concat_fwd_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p, _p],
    "tp3d_rows_scatter_bwd_f32": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, ctypes.c_size_t, _p],
    "tp3d_rows_scatter_invert": [_p, _p, _i, _i, _i, _i, _p, ctypes.c_size_t, _p],
    "tp3d_rows_scatter_apply_f32": [_p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, ctypes.c_size_t, _p],
    "tp3d_bn_stats_f32": [_p, _l, _i, _f, _f, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p],
    "tp3d_bn_act_f32": [_p, _p, _p, _p, _f, _l, _i, _p, _p],
    "tp3d_bn_act_maxpool_f32": [_p, _p, _p, _p, _f, _l, _i, _i, _p, _p, _p],
    "tp3d_bn_act_bwd_f32": [_p, _p, _p, _p, _p, _p, _p, _f, _l, _i, _i, _i, _p, _p, _p, _p, _p],
    "tp3d_interp_concat_fwd_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "tp3d_idw_weights_f32": [_p, _l, _p, _p],
    "tp3d_gemm_tn_f32": [_p, _p, _l, _i, _i, _p, _p, _p],
    "tp3d_gemm_tn_x3_f32": [_p, _p, _l, _i, _i, _i, _p, _p, _i, _p],
    "tp3d_gemm_tn_x3_act_f32": [_p, _p, _p, _p, _p, _f, _l, _i, _i, _i, _p, _p, _i, _p],
    "tp3d_gemm_tn_x3_act_red_f32": [_p, _p, _p, _p, _p, _p, _f, _p, _i, _l, _i, _i, _i, _p, _p, _p, _p, _i, _p],
    "tp3d_gemm_rows_narrow_f32": [_p, _p, _l, _i, _i, _p, _p, _i, _p],
    "tp3d_gemm_tn_bn_narrow_f32": [_p, _p, _p, _p, _p, _p, _p, _f, _p, _l, _i, _i, _p, _p, _i, _p],
    "tp3d_bn_bwd_reduce_f32": [_p, _p, _p, _p, _p, _p, _p, _f, _l, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p],
    "tp3d_gemm_rows_f32": [_p, _p, _l, _i, _i, _p, _p, _p, _p],
    "tp3d_gemm_rows_epi_f32": [_p, _p, _l, _i, _i, _p, _p, _p, _f, _p, _p, _p],
    "tp3d_gemm_rows_bnact_sp_f32": [_p, _p, _p, _p, _f, _p, _l, _i, _i, _p, _p, _p, _i, _p],
    "tp3d_gemm_rows_bnact_x3_f32": [_p, _p, _p, _p, _f, _p, _l, _i, _i, _p, _p, _p, _i, _p],
    "tp3d_gemm_rows_bnbwd_sp_f32": [_p, _p, _p, _p, _p, _p, _p, _f, _p, _l, _i, _i, _p, _i, _i, _i, _p, _p, _i, _i, _p],
    "tp3d_bn_finalize_f32": [_p, _i, _l, _i, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "tp3d_kpconv_bwd_features_f32": [_p, _p, _p, _p, _p, _l, _l, _i, _i, _i, _f, _i, _i, _p, _p, ctypes.c_size_t, _i, _p,
                                     ctypes.c_size_t, _p],
    "tp3d_knn_partial_dense_f32": [_p, _p, _p, _p, _i, _i, _l, _l, _i, _f, _p, _p, _p, ctypes.c_size_t, _p],
    "tp3d_knn_dense_f32": [_p, _p, _i, _i, _i, _i, _f, _p, _p, _p, ctypes.c_size_t, _p],
    "tp3d_knn_interpolate_fwd_f32": [_p, _p, _p, _p, _l, _i, _i, _i, _i, _p, _p, _p],
    "tp3d_nbr_maxpool_fwd_f32": [_p, _p, _l, _l, _i, _i, _p, _p, _p],
    "tp3d_nbr_maxpool_bwd_f32": [_p, _p, _p, _l, _l, _i, _i, _p, _p, ctypes.c_size_t, _i, _p],
    "tp3d_voxel_bounds_f32": [_p, _p, _l, _f, _p, _p],
    "tp3d_voxel_cluster_f32": [_p, _p, _l, _f, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t, _p],
    "tp3d_cluster_mean_f32": [_p, _p, _p, _l, _i, _p, _p],
    "tp3d_cluster_majority_i64": [_p, _p, _p, _l, _l, _l, _p, _p],
    "tp3d_kpconv_weighted_f32": [_p, _p, _p, _p, _p, _l, _l, _i, _i, _i, _f, _i, _i, _p, _p],
    "tp3d_gemm_skinny_f32": [_p, _p, _l, _i, _i, _i, _p, _p],
    "tp3d_gemm_skinny_bnact_f32": [_p, _p, _l, _i, _i, _i, _p, _p, _p, _f, _p, _p],
    "tp3d_randla_relpos_f32": [_p, _p, _p, _l, _i, _l, _p, _p],
    "tp3d_attn_pool_fwd_f32": [_p, _p, _p, _l, _i, _i, _i, _i, _p, _p],
    "tp3d_attn_pool_bwd_f32": [_p, _p, _p, _p, _l, _i, _i, _i, _i, _p, _p, _p],
    "tp3d_relation_rows_f32": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    # launch plans (host arithmetic; the last argument is a HOST int64 array)
    "tp3d_gemm_tn_plan": [_l, _i, _i, _p],
    "tp3d_gemm_tn_x3_plan": [_l, _i, _i, _p],
    "tp3d_gemm_rows_plan": [_l, _i, _i, _p],
    "tp3d_bn_plan": [_l, _i, _i, _p],
    "tp3d_scatter_plan": [_i, _i, _i, _i, _p],
}
MISC = ("tp3d_abi_version", "tp3d_strerror", "tp3d_last_hip_error", "tp3d_scatter_workspace_bytes",
        "tp3d_bn_workspace_floats", "tp3d_gemm_tn_workspace_floats", "tp3d_gemm_tn_x3_workspace_floats", "tp3d_gemm_tn_x3_serves", "tp3d_gemm_tn_x3_red_chunks", "tp3d_ball_query_workspace_bytes",
        "tp3d_gemm_rows_stat_floats", "tp3d_gemm_rows_stat_chunks", "tp3d_gemm_rows_sp_chunks", "tp3d_gemm_rows_x3_chunks", "tp3d_gemm_rows_bnbwd_sp_serves", "tp3d_gemm_tn_bn_narrow_serves", "tp3d_gemm_rows_narrow_chunks", "tp3d_gemm_tn_bn_narrow_workspace_floats", "tp3d_gemm_rows_workspace_floats", "tp3d_kpconv_bwd_workspace_bytes", "tp3d_voxel_workspace_bytes", "tp3d_knn_workspace_bytes", "tp3d_kpconv_grad_workspace_bytes")
ABI_VERSION = 36

_handle = None


class Tp3dError(RuntimeError):
    pass


def library_path():
    return _build.LIB_PATH


def load():
    """dlopen the library (building it first if the in-tree copy is missing or stale and hipcc exists)."""
    global _handle
    if _handle is not None:
        return _handle
    path = _build.LIB_PATH
    if not os.path.exists(path):
        _build.build_library()
    if not os.path.exists(path):
        raise Tp3dError("libtp3d_hip.so is missing at %s and could not be built" % path)
    h = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(h, name)  # AttributeError here = the .so does not match the header
        fn.argtypes = argtypes
        fn.restype = _i
    h.tp3d_abi_version.restype = _i
    h.tp3d_strerror.restype = ctypes.c_char_p
    h.tp3d_strerror.argtypes = [_i]
    h.tp3d_last_hip_error.restype = _i
    h.tp3d_scatter_workspace_bytes.restype = ctypes.c_size_t
    h.tp3d_scatter_workspace_bytes.argtypes = [_i, _i, _i, _i]
    h.tp3d_bn_workspace_floats.restype = ctypes.c_size_t
    h.tp3d_bn_workspace_floats.argtypes = [_l, _i]
    h.tp3d_gemm_tn_workspace_floats.restype = ctypes.c_size_t
    h.tp3d_gemm_tn_workspace_floats.argtypes = [_l, _i, _i]
    h.tp3d_gemm_tn_x3_workspace_floats.restype = ctypes.c_size_t
    h.tp3d_gemm_tn_x3_workspace_floats.argtypes = [_l, _i, _i]
    h.tp3d_gemm_tn_x3_serves.restype = ctypes.c_int
    h.tp3d_gemm_tn_x3_serves.argtypes = [_l, _i, _i]
    h.tp3d_gemm_tn_x3_red_chunks.restype = ctypes.c_int
    h.tp3d_gemm_tn_x3_red_chunks.argtypes = [_l, _i, _i]
    h.tp3d_kpconv_bwd_workspace_bytes.restype = ctypes.c_size_t
    h.tp3d_kpconv_bwd_workspace_bytes.argtypes = [_l, _l]
    h.tp3d_gemm_rows_stat_floats.restype = ctypes.c_size_t
    h.tp3d_gemm_rows_stat_floats.argtypes = [_l, _i]
    h.tp3d_gemm_rows_stat_chunks.restype = ctypes.c_int
    h.tp3d_gemm_rows_stat_chunks.argtypes = [_l, _i]
    h.tp3d_gemm_rows_narrow_chunks.restype = ctypes.c_int
    h.tp3d_gemm_rows_narrow_chunks.argtypes = [_l]
    h.tp3d_gemm_tn_bn_narrow_serves.restype = ctypes.c_int
    h.tp3d_gemm_tn_bn_narrow_serves.argtypes = [_l, _i, _i]
    h.tp3d_gemm_tn_bn_narrow_workspace_floats.restype = ctypes.c_size_t
    h.tp3d_gemm_tn_bn_narrow_workspace_floats.argtypes = [_l, _i, _i]
    h.tp3d_gemm_rows_bnbwd_sp_serves.restype = ctypes.c_int
    h.tp3d_gemm_rows_bnbwd_sp_serves.argtypes = [_l, _i, _i]
    h.tp3d_gemm_rows_x3_chunks.restype = ctypes.c_int
    h.tp3d_gemm_rows_x3_chunks.argtypes = [_l, _i, _i, _i]
    h.tp3d_gemm_rows_sp_chunks.restype = ctypes.c_int
    h.tp3d_gemm_rows_sp_chunks.argtypes = [_l, _i, _i, _i]
    h.tp3d_gemm_rows_workspace_floats.restype = ctypes.c_size_t
    h.tp3d_gemm_rows_workspace_floats.argtypes = [_l, _i, _i]
    h.tp3d_kpconv_grad_workspace_bytes.restype = ctypes.c_size_t
    h.tp3d_kpconv_grad_workspace_bytes.argtypes = [_l, _l, _i]
    h.tp3d_knn_workspace_bytes.restype = ctypes.c_size_t
    h.tp3d_knn_workspace_bytes.argtypes = [_i, _l, _i]
    h.tp3d_voxel_workspace_bytes.restype = ctypes.c_size_t
    h.tp3d_voxel_workspace_bytes.argtypes = [_l]
    h.tp3d_ball_query_workspace_bytes.restype = ctypes.c_size_t
    h.tp3d_ball_query_workspace_bytes.argtypes = [_i, _l, _i]
    if h.tp3d_abi_version() != ABI_VERSION:
        raise Tp3dError("libtp3d_hip.so ABI %d != binding ABI %d" % (h.tp3d_abi_version(), ABI_VERSION))
    _handle = h
    return h


class KernelTimer(object):
    """Optional per-entry-point device timing with HIP events on the launch stream (bench.py's roofline leg).

    Events are recorded on torch's current stream, which is the stream every launch is enqueued on, with the stream
    drained in front of every bracket, so the elapsed time covers exactly the kernels (and memsets) one C-ABI call
    enqueues plus one launch latency.
    """

    def __init__(self):
        self.records = []  # ((entry point, integer arguments), start_event, end_event)

    def summary(self):
        """{(entry point, integer size arguments): (launches, total_ms)} -- call after a device synchronize."""
        out = {}
        for tag, a, b in self.records:
            n, t = out.get(tag, (0, 0.0))
            out[tag] = (n + 1, t + a.elapsed_time(b))
        return out


_timer = None


def set_timer(timer):
    """Install (or remove, with None) a KernelTimer; returns the previous one."""
    global _timer
    prev, _timer = _timer, timer
    return prev


_fn_cache = {}
_post_call_hook = None


def set_post_call_hook(hook):
    """Install (or remove, with None) `hook(name, args)`, run after every C-ABI call: the guard-band checker of
    tests/canary.py uses it to name the entry point that wrote outside a buffer.  Returns the previous hook."""
    global _post_call_hook
    prev, _post_call_hook = _post_call_hook, hook
    return prev


def call(name, *args):
    """Invoke a C-ABI entry point; pointers travel as plain ints (ptr()), sizes as ints/floats."""
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    if _timer is not None:
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        # The stream is drained first: hipEventElapsedTime counts from the moment the START marker began to wait in the
        # queue, not from the end of the work in front of it, so a bracket recorded while earlier kernels are still
        # running also covers what is left of them (seen as a 5 us statistics kernel "taking" 0.2 ms behind a GEMM).
        torch.cuda.current_stream().synchronize()
        a.record()
        rc = fn(*args)
        b.record()
        # size arguments only (device addresses are far above 2**40)
        _timer.records.append(((name, tuple(v for v in args if type(v) is int and v < (1 << 40))), a, b))
    else:
        rc = fn(*args)
    if rc != 0:
        h = load()
        raise Tp3dError("%s failed: %s (code %d, hipError %d)" % (
            name, h.tp3d_strerror(rc).decode(), rc, h.tp3d_last_hip_error()))
    if _post_call_hook is not None:
        _post_call_hook(name, args)


class on_device(object):
    """`with on_device(dev):` -- switches the current HIP device only when it is not `dev` already (the common
    one-process-per-GPU case costs one integer compare instead of a context-manager round trip)."""

    __slots__ = ("idx", "prev")

    def __init__(self, dev):
        self.idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.prev = -1

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


_ws_cache = {}    # (device, stream, purpose) -> [buffer, handed out during a stream capture]
_ws_retired = []  # buffers a captured graph may still point at: replaced, never freed


def workspace(tag, nbytes, device):
    """Grow-only scratch buffer per (device, stream, purpose).  Kernels of one stream run in order and every entry
    point consumes its workspace before returning control to the stream's next kernel, so reuse is safe.

    A buffer handed out while the stream is being captured has its ADDRESS baked into the graph: it is never freed
    afterwards (a larger request retires it to `_ws_retired` instead of dropping it), because torch's capture stream is
    shared between graphs and a replay would otherwise write into memory the allocator has given to someone else."""
    key = (device.index, _raw_stream(device), tag)
    capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    slot = _ws_cache.get(key)
    if slot is None or slot[0].numel() < nbytes:
        if slot is not None and slot[1]:
            _ws_retired.append(slot[0])
        slot = _ws_cache[key] = [torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device), False]
    if capturing:
        slot[1] = True
    return slot[0]


def _raw_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


def scatter_workspace(B, L, nbins, with_weights, device):
    """Device scratch for the atomic-free scatter-add backward ops (size dictated by the library)."""
    nbytes = load().tp3d_scatter_workspace_bytes(B, L, nbins, int(with_weights))
    return workspace("scatter", nbytes, device), nbytes


GRID_MIN_POINTS = 2048  # BQ_GRID_MIN_POINTS of csrc/ball_query.hip


def ball_query_workspace(num_clouds, rows, max_cloud_points, device):
    """(buffer, nbytes) for the uniform-grid radius search, or (None, 0) when the brute-force kernels serve it."""
    if max_cloud_points < GRID_MIN_POINTS:
        return None, 0
    nbytes = load().tp3d_ball_query_workspace_bytes(num_clouds, rows, max_cloud_points)
    if nbytes == 0:
        return None, 0
    return workspace("grid", nbytes, device), nbytes


_inverse_cache = collections.OrderedDict()  # key -> [nbr, buffer, nbytes, event, stream, pinned]
INVERSE_CACHE_ENTRIES = 48


def _evict_inverse():
    """drop least-recently-used tables beyond the limit; a table a captured graph reads (pinned) is never dropped"""
    if len(_inverse_cache) < INVERSE_CACHE_ENTRIES:
        return
    for key in list(_inverse_cache):
        if len(_inverse_cache) < INVERSE_CACHE_ENTRIES:
            break
        if not _inverse_cache[key][5]:
            del _inverse_cache[key]


def neighbour_inverse(nbr, M, device):
    """(buffer, nbytes, ready, token) for the inverted form of the neighbour table `nbr` (Nq, Mn) over M support points.

    The buffer is kept per table (address, version counter, shape): the two backward kernels of a strided block share
    it, and with precomputed neighbour tables it is built once for the whole run (a captured training step then only
    reads it).  The entry keeps `nbr` alive, so an equal address means the same storage and an equal version counter
    the same content.  ready == 0: the caller's kernel builds the table; it then calls inverse_built(token).
    A table built on another stream is waited for through its event; inside a stream capture the caller is expected
    to have synchronised after warm-up (torch.cuda.graph does).  A table handed to a capture is PINNED: the graph
    replays read its address, so it is exempt from eviction; the others are evicted least recently used first."""
    key = (nbr.data_ptr(), nbr._version, tuple(nbr.shape), int(M), device.index)
    hit = _inverse_cache.get(key)
    cur = _raw_stream(device)
    capturing = torch.cuda.is_current_stream_capturing()
    if hit is not None and hit[3] is not None:
        _inverse_cache.move_to_end(key)
        if capturing:
            hit[5] = True
        elif hit[4] != cur:
            torch.cuda.current_stream(device).wait_event(hit[3])
        return hit[1], hit[2], 1, None
    nbytes = load().tp3d_kpconv_bwd_workspace_bytes(int(M), nbr.numel())
    if capturing:  # never publish a table whose build is only a node of a graph being recorded
        return workspace("nbr_inverse", nbytes, device), nbytes, 0, None
    buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    _evict_inverse()
    _inverse_cache[key] = [nbr, buf, nbytes, None, cur, False]
    return buf, nbytes, 0, key


def inverse_built(token, device):
    """Publish the table the caller's kernel has just enqueued (records the event other streams wait on)."""
    if token is None:
        return
    entry = _inverse_cache.get(token)
    if entry is not None:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        entry[3] = ev


def bn_workspace(M, C, device):
    n = load().tp3d_bn_workspace_floats(M, C)
    return workspace("bn", 4 * n, device)


def gemm_tn_workspace(M, N, K, device, x3=False):
    h = load()
    n = h.tp3d_gemm_tn_x3_workspace_floats(M, N, K) if x3 else h.tp3d_gemm_tn_workspace_floats(M, N, K)
    return workspace("gemm_tn", 4 * n, device)


def ptr(t):
    """Device pointer of a tensor as an int (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr(device):
    """hipStream_t of torch's current stream on `device`, so launches order with surrounding torch ops."""
    return _raw_stream(device)
