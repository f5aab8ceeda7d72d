"""ctypes binding of libglowtts_hip.so (C ABI in include/glowtts_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this module
raises.  PyTorch is used only for device memory and streams (tensor.data_ptr(), current stream)."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libglowtts_hip.so")
# tools/ab.sh (A/B measurements of launch heuristics) points this at the experiment build `make -C glow_tts_amd/csrc tools`, whose
# GLOWTTS_* switches are live; the product library above has them compiled to their defaults and reads no environment variable.
if os.environ.get("GLOWTTS_LIB_PATH"):
    LIB_PATH = os.environ["GLOWTTS_LIB_PATH"]
_lib = None
ABI_VERSION = 7                          # GLOWTTS_ABI_VERSION of include/glowtts_hip.h

c_int, c_float, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class GlowTTSHipError(RuntimeError):
    pass


def lib():
    """Loads the shared library (once).  Raises if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GlowTTSHipError(
                f"{LIB_PATH} not found: build it with `make -C glow_tts_amd/csrc` "
                "(or __graft_entry__.build()); the HIP path has no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        _declare(L)
        if L.glowtts_abi_version() != ABI_VERSION:       # struct layouts in decoder.py mirror include/glowtts_hip.h of exactly this version
            raise GlowTTSHipError(f"{LIB_PATH} has ABI version {L.glowtts_abi_version()}, this package binds version {ABI_VERSION}: rebuild it")
        _lib = L
    return _lib


def _declare(L):
    L.glowtts_abi_version.restype = c_int
    L.glowtts_device_arch.argtypes = [ctypes.c_char_p, c_int]
    L.glowtts_mas_dp_f32.argtypes = [c_void_p] * 5 + [c_int] * 3 + [c_float, c_void_p]
    L.glowtts_mas_path_from_idx.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    L.glowtts_mas_f32.argtypes = [c_void_p] * 5 + [c_int] * 3 + [c_float, c_void_p]
    L.glowtts_pack_weight.argtypes = [c_void_p] + [c_int] * 7 + [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p]
    L.glowtts_conv_cl.argtypes = [c_void_p, c_void_p]


def check(rc, what):
    if rc != 0:
        raise GlowTTSHipError(f"{what} failed with code {rc}")


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GlowTTSHipError("expected a device tensor (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise GlowTTSHipError("expected a contiguous tensor")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream (as an integer for ctypes).  Called once per kernel launch: the raw accessor avoids
    building a torch.cuda.Stream object each time (eager launches are host-bound)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# --------------------------------------------------------------------------------------------------------------------
# Pinned host staging for small tables that the device reads (weight-gradient / optimizer job tables, hyper-parameter words).
#
# (1) Under hipGraph capture a host-to-device copy becomes a memcpy NODE that re-reads its pinned source at every replay, and the
#     tables hold pointers into that graph's private buffers: every capture therefore needs a pinned buffer of its OWN (a buffer shared
#     between captures would make an earlier graph replay with a later graph's pointers).  Pinned memory cannot be allocated while a
#     stream is capturing, so the eager steps that precede every capture keep a few spare buffers per size (`staged_upload` in eager
#     mode), and a capture takes ownership of one of them for good.
# (2) Eager copies from a pinned buffer that the host rewrites every step go through a ring with one event per slot (`PinnedRing`),
#     so that the host running ahead of the stream never overwrites words an in-flight copy has not read yet.
# --------------------------------------------------------------------------------------------------------------------
_PERSIST = {}


def persistent(device, key, factory):
    """A small device tensor that lives as long as the process (the constant 1 a backward pass is seeded with, the zeroed completion counters of the
    last-workgroup reductions), created on first use OUTSIDE a capture; under a capture that is the first use: None (the caller takes its unfused path)."""
    import torch
    k = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), key)
    t = _PERSIST.get(k)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        t = _PERSIST[k] = factory()
    return t


def one(device):
    """The constant 1.0 (a 0-d fp32 device tensor): `LossTerms.backward` seeds the loss terms with THIS tensor, and a loss node whose forward launch already wrote
    the gradients for that seed recognises it by identity and launches nothing (alignment.PriorLoss / DurationMSE)."""
    import torch
    return persistent(device, "one", lambda: torch.ones((), device=device))


def counter(device, key, owner=None):
    """The zeroed uint32 a last-workgroup reduction counts its finished workgroups in (the kernel leaves it zero).  One launch at a time may use it: `owner` (a
    dict that lives with the module issuing the launches - MLE_Loss, the encoder's cache) keeps one per module; without an owner it is one per process and
    device."""
    import torch
    if owner is None:
        return persistent(device, ("counter", key), lambda: torch.zeros(1, dtype=torch.int32, device=device))
    k = ("counter", key, str(device))
    t = owner.get(k)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        t = owner[k] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


_SPARE = {}            # nbytes -> [pinned uint8 tensors not owned by any captured graph]
_SPARE_CAP = 4         # same-size tables per step (e.g. the optimizer's and the gradient-norm table of one parameter list)
_OWNED = []            # buffers handed to captures when no sink is active (live as long as the process)
_SINKS = []
DEFER_UPLOADS = True    # (A/B switch of bench.py --no-defer-uploads: False = the copy nodes of rounds 2-5)


class pinned_sink:
    """`with pinned_sink(lst):` around one or more hipGraph captures - pinned buffers taken by captures inside the block are appended to `lst` (keep it alive with the
    graphs).  Round 6, `defer` (default): a table uploaded under capture does NOT become a memcpy node of the graph.  Its contents never change between replays (the
    pointers in it are the graph's own buffers), so the device copy is a slice of an arena allocated when the block is entered (outside the captures), filled ONCE,
    eagerly, when the block exits - before the first replay.
    A memcpy node in front of a launch was two dependency hops on that chain in every replay (~15-25 us at the head of the decoder's backward, ~10 in front of the
    gradient norm, more on the encoder's side)."""

    ARENA_BYTES = 2 << 20

    def __init__(self, keep, defer=None):
        self.keep, self.defer, self.pending = keep, bool(DEFER_UPLOADS if defer is None else defer), []
        self.arena, self.used = None, 0

    def __enter__(self):
        _SINKS.append(self)
        if self.defer and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            # the tables' device memory is allocated HERE, outside the captures: a block allocated under capture may be one that an earlier launch of the same
            # capture used as scratch and freed - every replay would then overwrite a table that is only uploaded once
            self.arena = torch.empty(self.ARENA_BYTES, dtype=torch.uint8, device=torch.cuda.current_device())
            self.keep.append(self.arena)
        return self.keep

    def take(self, n, device):
        """n bytes of the arena (256-byte aligned), or None when it is exhausted / on another device."""
        if self.arena is None or self.arena.device != torch.device(device) or self.used + n > self.arena.numel():
            return None
        t = self.arena[self.used:self.used + n]
        self.used += (n + 255) & ~255
        return t

    def __exit__(self, *exc):
        _SINKS.pop()
        if self.pending and exc[0] is None:
            for pinned, dst in self.pending:
                dst.copy_(pinned, non_blocking=True)
            torch.cuda.synchronize()
        self.pending = []
        return False


def staged_upload(raw, device):
    """bytes -> device uint8 tensor.  Eager: synchronous copy from a private buffer (the host may run ahead of the stream) and one more
    spare pinned buffer of this size is put aside; capturing: the table is filled once when the enclosing `pinned_sink` exits (see there), or - no sink - by a
    copy node that reads a pinned buffer that from now on belongs to the graph."""
    n = len(raw)
    src = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    if torch.cuda.is_current_stream_capturing():
        pool = _SPARE.get(n)
        if not pool:
            raise GlowTTSHipError("run at least one eager step of this shape before capturing a hipGraph (no spare pinned job table of "
                                  f"{n} bytes; pinned memory cannot be allocated during capture)")
        pinned = pool.pop()
        pinned.copy_(src)
        sink = _SINKS[-1] if _SINKS else None
        (sink.keep if sink is not None else _OWNED).append(pinned)
        dst = sink.take(n, device) if (sink is not None and sink.defer) else None
        if dst is not None:
            sink.pending.append((pinned, dst))
            return dst
        return pinned.to(device, non_blocking=True)
    pool = _SPARE.setdefault(n, [])
    if len(pool) < _SPARE_CAP:
        pool.append(torch.empty(n, dtype=torch.uint8).pin_memory())
    return src.to(device)


def refill_spares():
    """Tops every spare pool up to its cap (call OUTSIDE a capture, right before one): a capture takes pinned buffers of the sizes the eager
    steps have used for good, and a later capture of another input shape - whose warm-up passes skip the optimizer - needs them again."""
    if torch.cuda.is_current_stream_capturing():
        return
    for n, pool in _SPARE.items():
        while len(pool) < _SPARE_CAP:
            pool.append(torch.empty(n, dtype=torch.uint8).pin_memory())


class PinnedRing:
    """`slots` pinned buffers of `numel` elements reused round-robin; a slot is rewritten only after the copy that last read it has
    completed (one event per slot, which in steady state has long fired)."""

    def __init__(self, numel, dtype=torch.float32, slots=8):
        self.bufs = [torch.empty(numel, dtype=dtype).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.i = 0

    def push(self, values, dst):
        """dst.copy_(values) through the next slot, asynchronously on the current stream; `values`: a CPU tensor of the slot's shape."""
        i = self.i
        self.i = (i + 1) % len(self.bufs)
        if self.events[i] is not None:
            self.events[i].synchronize()
        self.bufs[i].copy_(values)
        dst.copy_(self.bufs[i], non_blocking=True)
        if self.events[i] is None:
            self.events[i] = torch.cuda.Event()
        self.events[i].record()
        return dst
