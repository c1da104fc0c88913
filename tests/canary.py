"""Guard-band checker for the C-ABI entry points (test infrastructure, GPU only).

While a `Canary()` context is active every device tensor that Python code allocates through torch.empty /
torch.empty_like / torch.zeros / torch.zeros_like (which covers every output buffer and every `_lib.workspace`
scratch buffer of the product) is carved out of a larger allocation with a guard band of GUARD bytes, filled with a
pattern, on both sides.  After EVERY C-ABI call (`_lib.set_post_call_hook`) the device is synchronised and all live
guard bands are compared with the pattern: a kernel that writes outside any of these buffers is reported with the
entry point that did it, its size arguments, and the allocation site of the damaged buffer -- instead of a silent
corruption or a page fault somewhere later.

Enabled for a whole `pytest -m gpu` run with TP3D_TEST_CANARY=1 (tests/conftest.py); slow (a synchronise and two
small compares per live buffer per call), so it is a diagnostic mode, not the default.
"""
import traceback
import weakref

import torch

GUARD = 65536
PATTERN = 0xA5


class CanaryError(AssertionError):
    pass


class Canary(object):
    def __init__(self, guard=GUARD):
        self.guard = guard
        self.live = []  # (weakref(raw uint8 tensor), payload bytes, allocation site)
        self.calls = 0
        self._orig = {}

    # ---- guarded allocation -------------------------------------------------------------------------------
    def _site(self):
        for fr in reversed(traceback.extract_stack()[:-3]):
            if "canary.py" not in fr.filename:
                return "%s:%d (%s)" % (fr.filename.split("/repo/")[-1], fr.lineno, fr.name)
        return "?"

    def _alloc(self, shape, dtype, device, zero):
        dtype = dtype or torch.get_default_dtype()
        meta = self._orig["empty"](shape, dtype=dtype, device="meta")
        nbytes = meta.numel() * meta.element_size()
        g = self.guard
        pad = (-nbytes) % 256  # keep the rear guard's start (and the payload's end) 256-byte aligned
        raw = self._orig["empty"](nbytes + pad + 2 * g, dtype=torch.uint8, device=device)
        raw[:g] = PATTERN
        raw[g + nbytes:] = PATTERN
        out = raw[g:g + nbytes].view(dtype).view(meta.shape)
        if zero:
            out.zero_()
        self.live.append((weakref.ref(raw), nbytes, self._site()))
        out._tp3d_canary_raw = raw  # the view keeps its allocation (and its guards) alive
        return out

    @staticmethod
    def _is_cuda(device):
        return device is not None and torch.device(device).type == "cuda"

    def _patch(self):
        o = self._orig

        def shape_of(size):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                return tuple(size[0])
            return tuple(size)

        def empty(*size, dtype=None, device=None, **kw):
            if self._is_cuda(device) and not kw.get("pin_memory") and "out" not in kw:
                return self._alloc(shape_of(size), dtype, device, False)
            return o["empty"](*size, dtype=dtype, device=device, **kw)

        def zeros(*size, dtype=None, device=None, **kw):
            if self._is_cuda(device) and "out" not in kw:
                return self._alloc(shape_of(size), dtype, device, True)
            return o["zeros"](*size, dtype=dtype, device=device, **kw)

        def empty_like(t, dtype=None, device=None, **kw):
            dev = device if device is not None else t.device
            if self._is_cuda(dev) and t.layout == torch.strided:
                return self._alloc(tuple(t.shape), dtype or t.dtype, dev, False)
            return o["empty_like"](t, dtype=dtype, device=device, **kw)

        def zeros_like(t, dtype=None, device=None, **kw):
            dev = device if device is not None else t.device
            if self._is_cuda(dev) and t.layout == torch.strided:
                return self._alloc(tuple(t.shape), dtype or t.dtype, dev, True)
            return o["zeros_like"](t, dtype=dtype, device=device, **kw)

        for name, fn in (("empty", empty), ("zeros", zeros), ("empty_like", empty_like), ("zeros_like", zeros_like)):
            o[name] = getattr(torch, name)
            setattr(torch, name, fn)

    # ---- checking -----------------------------------------------------------------------------------------
    def check(self, name="(explicit check)", args=()):
        if torch.cuda.is_current_stream_capturing():
            return  # a graph is being recorded: nothing runs yet (and a synchronise would be illegal)
        self.calls += 1
        torch.cuda.synchronize()
        g = self.guard
        alive = []
        for ref, nbytes, site in self.live:
            raw = ref()
            if raw is None:
                continue
            alive.append((ref, nbytes, site))
            for side, band, base in (("before", raw[:g], -g), ("after", raw[g + nbytes:], nbytes)):
                bad = (band != PATTERN).nonzero()
                if bad.numel():
                    lo, hi = int(bad.min()), int(bad.max())
                    sizes = tuple(v for v in args if type(v) is int and v < (1 << 40))
                    raise CanaryError(
                        "%s%s wrote outside a buffer: %d guard bytes %s the %d-byte buffer allocated at %s were "
                        "overwritten (byte offsets %d..%d relative to the buffer start)"
                        % (name, sizes, int(bad.numel()), side, nbytes, site, base + lo, base + hi))
        self.live = alive

    def __enter__(self):
        from torch_points3d_amd import _lib
        self._lib = _lib
        # scratch buffers cached before the context was entered have no guards: start from fresh ones
        self._saved_ws = dict(_lib._ws_cache)
        _lib._ws_cache.clear()
        self._patch()
        self._prev_hook = _lib.set_post_call_hook(self.check)
        return self

    def __exit__(self, *exc):
        self._lib.set_post_call_hook(self._prev_hook)
        for name, fn in self._orig.items():
            setattr(torch, name, fn)
        self._lib._ws_cache.clear()
        self._lib._ws_cache.update(self._saved_ws)
        return False
