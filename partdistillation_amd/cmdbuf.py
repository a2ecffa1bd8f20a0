"""Command buffers (include/pd_cmdbuf.h, csrc/cmdbuf.hip): record the C-ABI calls of a region ONCE, replay them from C++ with one call.

    rec = cmdbuf.Recording(slots=[...every tensor the region reads or writes that it does not allocate itself...])
    with rec:
        outs = region(...)            # the ordinary Python faces run (and launch) as always; every pd_* call is noted
    ...
    rec.replay([...the same tensors of THIS step, same order, same shapes...])      # one C call: the same launches, new addresses

Inside `with rec:`
  * `lib.load()` hands out a proxy that performs each call and records (function, argument words);
  * torch.empty / zeros / empty_like / zeros_like / empty_strided on the GPU carve from the recording's persistent ARENA (zeros: a
    recorded pd_memset_async), so every buffer the region allocates has the same address at every replay — the tensors the region
    returns are arena views and stay valid until the next replay overwrites them;
  * every pointer argument must be (a) inside the arena, (b) inside one of the declared slots — recorded as slot + byte offset and
    re-based at replay —, (c) the current stream, or (d) host memory the recording keeps alive (ctypes objects passed by reference, pinned
    staging buffers handed out by fused.PinnedRing while recording).  Anything else raises: a pointer whose lifetime nobody vouches
    for would be replayed stale;
  * any ATen operator that is not a pure view raises too (it would run during the recording and silently not at replay).
A recording is per shape: the caller keys its cache by everything that decided the region's control flow.
"""
import ctypes
import struct

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import lib as _lib

MAX_ARGS = 48
LITERAL, STREAM = -1, -2
ENABLED = __import__("os").environ.get("PD_CMDBUF", "1") != "0"
DEBUG = __import__("os").environ.get("PD_CMDBUF_DEBUG", "0") != "0"


class PdCmd(ctypes.Structure):                                       # include/pd_cmdbuf.h
    _fields_ = [("fn", ctypes.c_int32), ("nargs", ctypes.c_int32), ("a", ctypes.c_uint64 * MAX_ARGS), ("kind", ctypes.c_int16 * MAX_ARGS)]


class RecorderError(RuntimeError):
    pass


_ACTIVE = None
_EVER = [0]            # recordings made in this process (TrainStep.capture refuses to run after one: see there)


def active():
    return _ACTIVE


def usable():
    """may a region be recorded / replayed right now?  Not while a hipGraph is being captured: a capture runs on its own stream (a new
    recording would allocate its arena and pin host buffers inside the capture), and a captured step is its own replay mechanism"""
    return ENABLED and _ACTIVE is None and not torch.cuda.is_current_stream_capturing()


class _Arena:
    CHUNK = 64 << 20

    def __init__(self, device, chunk=None):
        self.device, self.chunks, self.off = device, [], 0
        self.used = []                                              # bytes handed out of each chunk
        if chunk:
            self.CHUNK = chunk

    def alloc(self, nbytes):
        nbytes = max(int(nbytes), 1)
        al = (nbytes + 255) // 256 * 256
        if not self.chunks or self.off + al > self.chunks[-1].numel():
            _Purity.paused += 1
            try:
                self.chunks.append(_REAL["empty"](max(self.CHUNK, al), dtype=torch.uint8, device=self.device))
            finally:
                _Purity.paused -= 1
            self.used.append(0)
            self.off = 0
        c, o = self.chunks[-1], self.off
        self.off += al
        self.used[-1] = self.off
        return c, o

    def contains(self, ptr):
        for c in self.chunks:
            b = c.data_ptr()
            if b <= ptr < b + c.numel():
                return True
        return False

    def nbytes(self):
        return sum(c.numel() for c in self.chunks)


_REAL = {"empty": torch.empty, "zeros": torch.zeros, "empty_like": torch.empty_like, "zeros_like": torch.zeros_like,
         "empty_strided": torch.empty_strided}
_ITEM = {}


def _itemsize(dt):
    return dt.itemsize


def _is_cuda(device):
    if device is None:
        return False
    return (device.type if isinstance(device, torch.device) else str(device).split(":")[0]) == "cuda"


def _extent_bytes(t):
    if t.numel() == 0:
        return 0
    return (sum((s - 1) * st for s, st in zip(t.shape, t.stride())) + 1) * t.element_size()


class SlotRef:
    """placeholder, inside a recording's outputs, for "the tensor given for slot i at this replay" """
    __slots__ = ("i",)

    def __init__(self, i):
        self.i = i


class Fresh:
    """wrapper for tensors in a region's outputs that must be handed out as NEW tensor objects at every replay (gradients returned to
    autograd: AccumulateGrad adopts a gradient without a copy only if nobody else holds the tensor object)"""
    __slots__ = ("items",)

    def __init__(self, items):
        self.items = list(items)


def unalias_grads(params, owns):
    """call BEFORE a region whose returned parameter gradients live in persistent memory (a recording's arena, the fused ResNet
    plan's arena) overwrites that memory: AccumulateGrad adopts such a gradient as `.grad` WITHOUT a copy, so a `.grad` that survived
    the previous step (gradient accumulation over micro-batches, zero_grad(set_to_none=False), a stock torch optimizer — anything but
    engine/flat_params.py, which drops `.grad` after gathering it) still points into the memory about to be rewritten.  Every such
    `.grad` is moved to storage of its own first; autograd then accumulates the new gradient into it as usual (g1 + g2, not 2 * g2).
    owns(ptr) -> bool tells whether an address belongs to the persistent memory in question."""
    for p in params:
        g = getattr(p, "grad", None)
        if g is not None and g.is_cuda and g.numel() and owns(g.data_ptr()):
            p.grad = g.clone()


_ALL_CACHES = []                                   # every LRU of the process (drop_all)


def drop_all():
    """forget every cached plan / recording of the process (their arenas return to the allocator once nothing else holds them)"""
    for c in _ALL_CACHES:
        c.clear()


class LRU(dict):
    """a bounded cache of plans / recordings (each owns persistent arenas of 10^8 .. 10^10 bytes per input shape: training with
    ResizeShortestEdge + RandomCrop, or inference on arbitrary image sizes, must not accumulate one arena set per shape).  get() marks
    an entry as the most recently used; put() drops the least recently used entries beyond `cap` and hands them to `on_evict`."""

    def __init__(self, cap, on_evict=None):
        super().__init__()
        self.cap, self.on_evict = cap, on_evict
        _ALL_CACHES.append(self)

    def get(self, key, default=None):
        if key in self:
            v = self.pop(key)
            self[key] = v
            return v
        return default

    def put(self, key, value):
        old = self.pop(key, None)
        if old is not None and old is not value and self.on_evict is not None:
            self.on_evict(key, old, self)                     # a re-recorded region: whatever pins the replaced entry's arenas goes with it
        self[key] = value
        while len(self) > self.cap:
            k = next(iter(self))
            v = self.pop(k)
            if self.on_evict is not None:
                self.on_evict(k, v, self)


def require_stable(ptr, what):
    """inside a recorded region: `ptr` is about to be written into HOST memory the replay re-reads (a descriptor table) — it must be a
    persistent address (an arena), not a slot"""
    rec = _ACTIVE
    if rec is None or not ptr:
        return
    ok = rec._in_arena(ptr)
    if not ok:
        for i in rec._pinned:
            b, n = rec._slot_ranges[i]
            if b <= ptr < b + max(n, 1):
                ok = True
                break
    if not ok:
        raise RecorderError(f"{rec.name}: {what} (0x{ptr:x}) goes into a host-side table but is not arena memory: copy the tensor inside the region")


class _Purity(TorchDispatchMode):
    """only pure views may run inside a recorded region"""
    paused = 0

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if _Purity.paused:
            return func(*args, **kwargs)
        # host-side work (descriptor tables, python scalars -> CPU tensors) is not the recorder's business
        flat = [a for a in args if isinstance(a, torch.Tensor)] + [a for a in kwargs.values() if isinstance(a, torch.Tensor)]
        for a in args:
            if isinstance(a, (list, tuple)):
                flat += [x for x in a if isinstance(x, torch.Tensor)]
        if (flat and not any(t.is_cuda for t in flat)) or (not flat and not _is_cuda(kwargs.get("device"))):
            return func(*args, **kwargs)
        rets = func._schema.returns
        ok = len(rets) > 0 and all(r.alias_info is not None and not r.alias_info.is_write for r in rets)
        name = str(func)
        if not ok and not name.startswith(("aten.sym_", "prim.", "aten.is_", "aten.stride", "aten.size", "aten.storage_offset", "aten.numel", "aten.dim")):
            raise RecorderError(f"{name} ran inside a recorded region: it would execute now and be MISSING at replay (hoist it out of the region "
                                "or express it with a pd_* call)")
        return func(*args, **kwargs)


class host_ops:
    """`with cmdbuf.host_ops():` — ATen calls on HOST tensors (pinning a staging buffer, filling a table) inside a recorded region"""

    def __enter__(self):
        _Purity.paused += 1

    def __exit__(self, *a):
        _Purity.paused -= 1
        return False


class _Proxy:
    def __init__(self, rec, real):
        self._rec, self._real, self._cache = rec, real, {}

    def __getattr__(self, name):
        w = self._cache.get(name)
        if w is None:
            fn = getattr(self._real, name)
            idx = int(self._real.pd_cmd_fn_index(name.encode()))
            if idx < 0:                                             # host-only query (workspace sizes ...): nothing to replay
                w = fn
            else:
                rec, argtypes = self._rec, _lib.SIGNATURES[name][1]

                def w(*args, _fn=fn, _idx=idx, _name=name, _at=argtypes):
                    rc = _fn(*args)
                    if rc == 0:
                        rec._record(_idx, _name, _at, args)
                    return rc
            self._cache[name] = w
        return w


class Recording:
    def __init__(self, slots, name="region", stable=(), pinned=()):
        """stable: recordings whose ARENAS this region may point into (the forward recording whose saved activations the backward
        region reads): persistent addresses, recorded as they are.
        pinned: indices of slots whose ADDRESS the caller guarantees for the life of the recording (parameters: views of the flat
        parameter buffers): only those may be written into host-side tables; matches() checks the guarantee."""
        self.name = name
        self._stable = list(stable)
        self._pinned = {int(i): slots[int(i)].data_ptr() for i in pinned}
        self.slot_meta = [(tuple(t.shape), tuple(t.stride()), t.dtype) for t in slots]
        self._slot_ranges = [(t.data_ptr(), _extent_bytes(t)) for t in slots]
        self._rec_slots = list(slots)                               # alive during the recording only
        dev = next((t.device for t in slots if t.is_cuda), torch.device("cuda", torch.cuda.current_device()))
        self.arena = _Arena(dev)
        self.zero_arena = _Arena(dev, 8 << 20)                               # torch.zeros / zeros_like: cleared by ONE memset per chunk at replay start
        self._cmds, self._keep, self._host = [], [], []
        self._stream = None
        self.cmds = None
        self.outputs = None
        self.generation = 0

    # ------------------------------------------------------------------ recording
    def host_static(self, t):
        """host memory (a pinned staging tensor) the recording owns from now on: its address may appear in recorded calls"""
        self._keep.append(t)
        self._host.append((t.data_ptr(), max(t.numel() * t.element_size(), 1)))

    def _alloc(self, shape, dtype, strides=None, zero=False):
        dtype = dtype or torch.get_default_dtype()
        n = 1
        for s in shape:
            n *= int(s)
        arena = self.zero_arena if zero else self.arena
        if zero and strides is None:
            # every zero-filled buffer of the region is a FRESH piece of the zero arena (never handed out twice), so clearing the
            # whole arena once at the start of a replay equals clearing each buffer where the region asked for it: ~40 memset launches
            # per step become one or two.  This first execution clears the piece right here (unrecorded).
            c, o = arena.alloc(n * _itemsize(dtype))
            _lib.check(_lib.real().pd_memset_async(c.data_ptr() + o, 0, max(n * _itemsize(dtype), 1), self._stream))
            return c[o:o + n * _itemsize(dtype)].view(dtype).view(tuple(shape))
        if strides is None:
            c, o = self.arena.alloc(n * _itemsize(dtype))
            return c[o:o + n * _itemsize(dtype)].view(dtype).view(tuple(shape))
        ext = (sum((int(s) - 1) * int(st) for s, st in zip(shape, strides)) + 1) if n else 0
        c, o = self.arena.alloc(ext * _itemsize(dtype))
        return c[o:o + max(ext, 1) * _itemsize(dtype)].view(dtype).as_strided(tuple(shape), tuple(strides))

    def owns(self, ptr):
        """is `ptr` inside THIS recording's arenas (memory the next replay rewrites)?"""
        return self.arena.contains(ptr) or self.zero_arena.contains(ptr)

    def _in_arena(self, ptr):
        return (self.arena.contains(ptr) or self.zero_arena.contains(ptr)
                or any(r.arena.contains(ptr) or r.zero_arena.contains(ptr) for r in self._stable))

    def _classify(self, ptr, last, name, i):
        if ptr == 0:
            return LITERAL, 0
        if self._in_arena(ptr):
            return LITERAL, ptr
        for s, (b, n) in enumerate(self._slot_ranges):
            if b <= ptr < b + max(n, 1):
                return s, ptr - b
        if last and ptr == self._stream:
            return STREAM, 0
        for b, n in self._host:
            if b <= ptr < b + n:
                return LITERAL, ptr
        raise RecorderError(f"{self.name}: argument {i} of {name} points to memory the recording knows nothing about (0x{ptr:x}): "
                            "declare the tensor as a slot, allocate it inside the region, or keep it with host_static()")

    def _record(self, idx, name, argtypes, args):
        if len(args) > MAX_ARGS:
            raise RecorderError(f"{name}: more than {MAX_ARGS} arguments")
        c = PdCmd()
        c.fn, c.nargs = idx, len(args)
        for i, (v, tp) in enumerate(zip(args, argtypes)):
            kind = LITERAL
            if tp is ctypes.c_void_p or tp is ctypes.c_char_p:
                if v is None:
                    w = 0
                elif isinstance(v, int):
                    kind, w = self._classify(v, i == len(args) - 1, name, i)
                else:                                              # a ctypes object passed by reference: host memory, kept alive
                    self._keep.append(v)
                    obj = getattr(v, "_obj", v)
                    w = ctypes.addressof(obj)
            elif tp is ctypes.c_float:
                w = struct.unpack("<I", struct.pack("<f", float(v)))[0]
            elif tp is ctypes.c_double:
                w = struct.unpack("<Q", struct.pack("<d", float(v)))[0]
            else:
                w = int(v) & 0xFFFFFFFFFFFFFFFF
            c.a[i], c.kind[i] = w, kind
        self._cmds.append(c)

    def __enter__(self):
        global _ACTIVE
        if _ACTIVE is not None:
            raise RecorderError("recordings do not nest")
        _EVER[0] += 1
        self._stream = _lib.current_stream()
        real = _lib.load()
        self._proxy = _Proxy(self, real)
        rec = self

        def empty(*size, dtype=None, device=None, memory_format=None, **kw):
            if not _is_cuda(device) or kw.get("pin_memory"):
                return _REAL["empty"](*size, dtype=dtype, device=device, **({"memory_format": memory_format} if memory_format is not None else {}), **kw)
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
            if memory_format is torch.channels_last and len(shape) == 4:
                b, ch, h, w = shape
                return rec._alloc(shape, dtype, (h * w * ch, 1, w * ch, ch))
            return rec._alloc(shape, dtype)

        def zeros(*size, dtype=None, device=None, **kw):
            if not _is_cuda(device):
                return _REAL["zeros"](*size, dtype=dtype, device=device, **kw)
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
            return rec._alloc(shape, dtype, zero=True)

        def empty_like(t, dtype=None, memory_format=None, **kw):
            if not t.is_cuda:
                return _REAL["empty_like"](t, dtype=dtype, **kw)
            if memory_format is torch.channels_last and t.dim() == 4:
                return empty(tuple(t.shape), dtype=dtype or t.dtype, device=t.device, memory_format=torch.channels_last)
            if memory_format is None and not t.is_contiguous():
                dense = sorted(range(t.dim()), key=lambda d: -t.stride(d))
                st, acc = [0] * t.dim(), 1
                for d in reversed(dense):
                    st[d] = acc
                    acc *= t.shape[d]
                return rec._alloc(tuple(t.shape), dtype or t.dtype, st)
            return rec._alloc(tuple(t.shape), dtype or t.dtype)

        def zeros_like(t, dtype=None, **kw):
            if not t.is_cuda:
                return _REAL["zeros_like"](t, dtype=dtype, **kw)
            if t.is_contiguous():
                return rec._alloc(tuple(t.shape), dtype or t.dtype, zero=True)
            o = empty_like(t, dtype=dtype)
            _lib.check(rec._proxy.pd_memset_async(o.data_ptr(), 0, _extent_bytes(o), rec._stream))
            return o

        def empty_strided(size, stride, dtype=None, device=None, **kw):
            if not _is_cuda(device):
                return _REAL["empty_strided"](size, stride, dtype=dtype, device=device, **kw)
            return rec._alloc(tuple(size), dtype, tuple(stride))

        torch.empty, torch.zeros, torch.empty_like, torch.zeros_like, torch.empty_strided = empty, zeros, empty_like, zeros_like, empty_strided
        self._mode = _Purity()
        self._mode.__enter__()
        _ACTIVE = self
        _lib._PROXY = self._proxy
        return self

    def __exit__(self, et, ev, tb):
        global _ACTIVE
        _ACTIVE = None
        _lib._PROXY = None
        self._mode.__exit__(et, ev, tb)
        torch.empty, torch.zeros, torch.empty_like, torch.zeros_like, torch.empty_strided = (_REAL[k] for k in ("empty", "zeros", "empty_like", "zeros_like", "empty_strided"))
        if DEBUG:
            import sys
            print(f"[cmdbuf] recorded {self.name}: {len(self._cmds)} calls, {len(self.slot_meta)} slots, arena {self.arena.nbytes() >> 20} MiB, "
                  f"error={et.__name__ if et else None}", file=sys.stderr, flush=True)
        if et is None:
            pre = []
            idx = int(_lib.real().pd_cmd_fn_index(b"pd_memset_async"))
            for c, used in zip(self.zero_arena.chunks, self.zero_arena.used):   # the zero arena: cleared first thing at every replay
                k = PdCmd()
                k.fn, k.nargs = idx, 4
                k.a[0], k.a[1], k.a[2], k.a[3] = c.data_ptr(), 0, used, 0
                k.kind[0], k.kind[1], k.kind[2], k.kind[3] = LITERAL, LITERAL, LITERAL, STREAM
                pre.append(k)
            self._cmds = pre + self._cmds
            self.cmds = (PdCmd * len(self._cmds))(*self._cmds)
            self._bases = (ctypes.c_uint64 * max(len(self.slot_meta), 1))()
        self._cmds = None
        return False

    def finish(self, outs):
        """after the `with` block: keep what the region returned; -> the same structure (this first execution's tensors)"""
        self.set_outputs(outs)
        slots, self._rec_slots = self._rec_slots, None
        return self._materialize(slots)

    # ------------------------------------------------------------------ outputs
    def set_outputs(self, outs):
        """what the region returned: any nesting of tuples / lists / dicts of tensors and plain values.  Tensors must live in an
        arena (they are handed out again at every replay) or BE one of the slot tensors (replaced by that replay's tensor)."""
        slots = self._rec_slots

        def walk(o):
            if isinstance(o, Fresh):
                return Fresh([walk(x) for x in o.items])
            if isinstance(o, torch.Tensor):
                for i, t in enumerate(slots):
                    if o is t:
                        return SlotRef(i)
                if o.is_cuda and o.numel() and not self._in_arena(o.data_ptr()):
                    raise RecorderError(f"{self.name}: the region returns a tensor that is neither in the arena nor one of its slots (a view of a slot? "
                                        "return the slot itself or copy it inside the region)")
                return o
            if isinstance(o, tuple):
                return tuple(walk(x) for x in o)
            if isinstance(o, list):
                return [walk(x) for x in o]
            if isinstance(o, dict):
                return {k: walk(v) for k, v in o.items()}
            return o
        self._outs = walk(outs)
        self._has_refs = True

    def _materialize(self, slots):
        def walk(o):
            if isinstance(o, SlotRef):
                return slots[o.i]
            if isinstance(o, Fresh):
                return [x.detach() if isinstance(x, torch.Tensor) else walk(x) for x in o.items]
            if isinstance(o, tuple):
                return tuple(walk(x) for x in o)
            if isinstance(o, list):
                return [walk(x) for x in o]
            if isinstance(o, dict):
                return {k: walk(v) for k, v in o.items()}
            return o
        return walk(self._outs)

    # ------------------------------------------------------------------ replay
    def matches(self, slots):
        return (len(slots) == len(self.slot_meta) and all((tuple(t.shape), tuple(t.stride()), t.dtype) == m for t, m in zip(slots, self.slot_meta))
                and all(slots[i].data_ptr() == p for i, p in self._pinned.items()))

    def replay(self, slots):
        if not self.matches(slots):
            raise RecorderError(f"{self.name}: the tensors given to replay() differ in number, shape, stride or dtype from the recorded ones")
        b = self._bases
        for i, t in enumerate(slots):
            b[i] = t.data_ptr()
        _lib.check(_lib.real().pd_cmd_replay(self.cmds, len(self.cmds), b, len(slots), _lib.current_stream()))
        self.generation += 1
        return self._materialize(slots)
