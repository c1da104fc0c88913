"""Cache policy of torch_points3d_amd._lib (no GPU): scratch buffers a captured graph may point at are retired, never
freed; inverted neighbour tables are evicted least-recently-used and never while a captured graph reads them."""
import torch

from torch_points3d_amd import _lib


def test_workspace_retires_buffers_handed_out_during_capture(monkeypatch):
    dev = torch.device("cpu")
    monkeypatch.setattr(_lib, "_raw_stream", lambda d: 7)
    monkeypatch.setattr(_lib, "_ws_cache", {})
    monkeypatch.setattr(_lib, "_ws_retired", [])
    a = _lib.workspace("t", 1000, dev)
    assert _lib.workspace("t", 500, dev) is a            # grow-only: a smaller request reuses it
    b = _lib.workspace("t", 5000, dev)
    assert b is not a and _lib._ws_retired == []         # eager buffers may simply be dropped
    # now the same during a "capture" (cuda device type + capturing stream)
    cap = {"on": True}
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: cap["on"])

    class Dev(object):  # a cuda-typed stand-in: workspace() only reads .index / .type and hands it to torch.empty
        index, type = 0, "cuda"
    real_empty = torch.empty
    monkeypatch.setattr(torch, "empty", lambda *a_, **k: real_empty(*a_, **{**k, "device": "cpu"}))
    c = _lib.workspace("g", 1000, Dev())
    cap["on"] = False
    d = _lib.workspace("g", 9000, Dev())                 # a later, larger request -- even outside a capture
    assert d is not c and any(r is c for r in _lib._ws_retired), "a buffer a graph points at was dropped"


def test_inverse_cache_is_lru_and_never_evicts_pinned_tables(monkeypatch):
    monkeypatch.setattr(_lib, "_inverse_cache", _lib.collections.OrderedDict())
    cache = _lib._inverse_cache
    for i in range(_lib.INVERSE_CACHE_ENTRIES):
        cache[i] = [None, None, 0, object(), 0, i in (0, 5)]  # entries 0 and 5 are read by a captured graph
    cache.move_to_end(1)  # 1 was used recently
    _lib._evict_inverse()
    assert len(cache) == _lib.INVERSE_CACHE_ENTRIES - 1
    assert 0 in cache and 5 in cache and 1 in cache and 2 not in cache  # oldest UNPINNED entry went
    for _ in range(10):
        cache[len(cache) + 1000] = [None, None, 0, object(), 0, False]
        _lib._evict_inverse()
    assert 0 in cache and 5 in cache
