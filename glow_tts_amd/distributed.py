"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference is single-process (Train.py:27-35); SURVEY.md 8e lists what a
sharded step must keep consistent:
  * gradients: SUM all-reduce of one flat fp32 bucket per parameter group, scaled so that the result equals the
    gradient of the single-process *global-batch* loss (the MLE loss divides by the number of frames, Modules.py:1026,
    so ranks are weighted by their frame counts, not averaged);
  * ActNorm data-dependent init (Modules.py:698-711): the [2C+1] batch statistics are summed over ranks.
"""
import weakref

import torch
import torch.distributed as dist


SINGLE_RANK_IS_DIST = False      # bench.py --force-dist: run the collective code path with a 1-rank group (exercises RCCL on a single-GPU box)


def is_dist():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or SINGLE_RANK_IS_DIST)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Collectives and hipGraph capture in one process.
#
# ProcessGroupNCCL hands every eager collective's Work to its watchdog thread, which polls `hipEventQuery(work.ncclEndEvent_)` every 100 ms
# until the work has completed.  HIP (ROCm 7) answers such a query with hipErrorCapturedEvent - and the watchdog then aborts the process -
# when the STREAM THE EVENT WAS LAST RECORDED ON is capturing at that moment, whether or not the record itself was captured.  torch >= 2.7
# runs `async_op=False` collectives on the CALLER'S CURRENT stream and records the end event there; a step captured within the next 100 ms on
# that stream (or on a stream the captured autograd pass hops to: the AccumulateGrad nodes stay on the stream of the warm-up passes) then
# kills the process from the watchdog thread (round 4: `bench.py --force-dist` on a fresh box, the ActNorm-init all-reduces of the warm-up
# step followed by the capture).  Two rules, both enforced here:
#   1. every eager collective of this package is issued with `async_op=True` + `wait()`: it runs on the process group's OWN stream, which no
#      capture ever joins (no collective is issued under capture: `_check_not_capturing`), so the watchdog only ever holds events of that stream;
#   2. before any capture `before_capture()` drains the watchdog's list (`ProcessGroup._wait_for_pending_works`), so that it holds nothing at all.
# ---------------------------------------------------------------------------------------------------------------------------------------
def _check_not_capturing(what):
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise RuntimeError(f"glow_tts_amd.distributed.{what} was called while the current stream is capturing a hipGraph: collectives stay outside "
                           "the captured step (compute their results before the capture and pass them in)")


def _collective(fn, tensor, *args, **kwargs):
    """One eager collective (fn = dist.all_reduce / dist.broadcast ...) on the process group's own stream; returns when the caller's current
    stream is ordered behind it."""
    _check_not_capturing(fn.__name__)
    work = fn(tensor, *args, async_op=True, **kwargs)
    if work is not None:
        work.wait()
    return tensor


def before_capture():
    """Call in front of every hipGraph capture of a process that has a process group: waits for the device, then until the backend's watchdog
    has retired every Work it holds (a no-op without a process group / for backends without a watchdog)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    pg = dist.distributed_c10d._get_default_group()
    wait = getattr(pg, "_wait_for_pending_works", None)
    if wait is not None:
        try:
            wait()
        except (RuntimeError, NotImplementedError):
            pass


def barrier():
    """dist.barrier() on the process group's stream + a device synchronise (bench.py's window brackets)."""
    if dist.is_available() and dist.is_initialized():
        _check_not_capturing("barrier")
        if torch.cuda.is_available() and dist.get_backend() == "nccl":
            t = torch.zeros(1, device=torch.device("cuda", torch.cuda.current_device()))
            _collective(dist.all_reduce, t)
        else:
            dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def broadcast_parameters(model, src=0):
    """Identical replicas at start: rank `src`'s parameters and buffers to every rank (tensors made contiguous first: RCCL rejects
    strided ones, e.g. a loaded checkpoint's column-major 4x4 flow weights)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            if not t.data.is_contiguous():
                t.data = t.data.contiguous()
            _collective(dist.broadcast, t.data, src)


def actnorm_stats_allreduce(stats):
    if is_dist():
        _collective(dist.all_reduce, stats, op=dist.ReduceOp.SUM)


def global_frame_weight(local_frames):
    """local_frames: 0-d tensor (sum of this rank's mel lengths).  Returns local / global: the factor that turns this
    rank's mean-over-local-frames MLE loss into its share of the global-batch loss."""
    if not is_dist():
        return torch.ones((), device=local_frames.device)
    tot = local_frames.detach().clone().to(torch.float32)
    _collective(dist.all_reduce, tot, op=dist.ReduceOp.SUM)
    return local_frames.to(torch.float32) / tot


def global_token_extent(local_max):
    """local_max: 0-d tensor (this rank's longest text).  Returns the longest text of the GLOBAL batch (all-reduce MAX): the extent over which
    the reference's duration MSE (Train.py:210, a mean over B x longest text of the batch) is taken when the batch is sharded."""
    if not is_dist():
        return local_max
    t = local_max.detach().clone().to(torch.float32)
    _collective(dist.all_reduce, t, op=dist.ReduceOp.MAX)
    return t


def global_step_scalars(local_frames, local_max_tokens):
    """Both per-step scalars of a sharded `Train_Step` with ONE collective (a SUM all-reduce of 1 + world floats: the frame count, and every
    rank's longest text in its own slot): -> (global_frame_weight(local_frames), global_token_extent(local_max_tokens)), 0-d device tensors."""
    if not is_dist():
        return torch.ones((), device=local_frames.device), local_max_tokens
    frames = local_frames.detach().to(torch.float32)
    v = torch.zeros(1 + dist.get_world_size(), dtype=torch.float32, device=frames.device)
    v[0] = frames
    v[1 + dist.get_rank()] = local_max_tokens.detach().to(torch.float32)
    _collective(dist.all_reduce, v, op=dist.ReduceOp.SUM)
    return frames / v[0], v[1:].max()


def global_batch_weight(local_utterances, device=None):
    """local / global utterance count: the factor for losses that are means over the padded batch (the duration MSE, Train.py:211) when
    ranks hold different numbers of utterances (equal shards: 1 / world)."""
    if not is_dist():
        return 1.0
    t = torch.tensor([float(local_utterances)], device=device)
    _collective(dist.all_reduce, t, op=dist.ReduceOp.SUM)
    return float(local_utterances) / float(t.item())


# Gradient ARENAS: storages whose gradient views are separated by private, zeroed alignment pads (decoder.DecoderFunction.backward packs each weight
# class at a 16-byte boundary).  Only for these may FlatGradReducer reduce the whole span, pads included, in place; between gradient views of any
# other storage a gap may hold somebody else's data.  storage data_ptr -> weak reference to one of the arena's views (alive <=> the storage is).
_ARENAS = {}


def register_arena(*views):
    """Declares the storage under `views` (views that tile it, up to alignment pads the caller owns and keeps zero) as one arena: `FlatGradReducer` may then
    all-reduce the whole span in place.  The registration holds weak references to EVERY view given - it stays valid as long as any of them is alive (one
    particular wrapper could be dropped while the others kept the storage) - and is only taken when a process group exists."""
    if not views or not is_dist():
        return
    sptr = views[0].untyped_storage().data_ptr()
    _ARENAS[sptr] = [weakref.ref(v) for v in views]
    if len(_ARENAS) > 64:                                       # (eager data-parallel steps allocate new arenas every step: drop the dead ones)
        for k in [k for k, refs in _ARENAS.items() if not any(r() is not None for r in refs)]:
            _ARENAS.pop(k, None)


def _is_arena(storage_ptr):
    refs = _ARENAS.get(storage_ptr)
    if refs is None:
        return False
    for r in refs:
        t = r()
        if t is not None and t.untyped_storage().data_ptr() == storage_ptr:
            return True
    _ARENAS.pop(storage_ptr, None)
    return False


class FlatGradReducer:
    """All-reduces the gradients of `params` with few, large collectives (xGMI is point-to-point: few large messages beat many small
    ones) and IN PLACE: `p.grad` keeps its address, which a replayed hipGraph of the step relies on.
      * gradients that already tile one contiguous storage - the decoder's stacked weight classes, whose leaf gradients are views of
        one stacked tensor (decoder.LeafStack) - are reduced directly on that storage: no copy at all (>= 95 % of the bytes);
      * the remaining small tensors are concatenated into flat buckets, reduced, and copied back.
    Buckets follow reverse parameter order, i.e. the order the backward pass produces them."""

    def __init__(self, params, bucket_bytes=64 << 20, direct_bytes=8 << 20, static=False, frozen=False):
        """static: the gradients keep their addresses from call to call (a replayed hipGraph of the step writes them in place): the plan and the
        flat buckets are built on the first call and kept - no allocation per step; every call still checks (a data_ptr per parameter) that no gradient
        appeared, vanished or moved.  frozen: the caller guarantees that (GraphedTrainStep: the gradients of a captured shape are the graph's own buffers) -
        no per-step Python over hundreds of parameters at all."""
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes, self.direct_bytes, self.static, self.frozen = bucket_bytes, direct_bytes, static or frozen, frozen
        self._kept = self._kept_sig = None

    def _signature(self):
        return tuple(g.data_ptr() if (g := p.grad) is not None else 0 for p in self.params)

    def _plan(self):
        """-> (direct: list of flat tensors aliasing gradient storages, buckets: list of lists of gradient tensors)."""
        groups = {}
        for p in reversed(self.params):
            g = p.grad
            if g is None:
                g = p.grad = torch.zeros_like(p)
            groups.setdefault(g.untyped_storage().data_ptr(), []).append(g)
        direct, small = [], []
        for sptr, gs in groups.items():
            gs = sorted(gs, key=lambda t: t.storage_offset())
            # the gradients tile one span of the storage exactly; alignment gaps of a few elements only inside a registered arena (its pads are
            # private and zero: `register_arena`)
            slack = 4 if _is_arena(sptr) else 1
            covered = all(t.is_contiguous() for t in gs) and all(0 <= b.storage_offset() - (a.storage_offset() + a.numel()) < slack for a, b in zip(gs, gs[1:]))
            total = gs[-1].storage_offset() + gs[-1].numel() - gs[0].storage_offset()
            if covered and len({t.dtype for t in gs}) == 1 and total * gs[0].element_size() >= self.direct_bytes:
                direct.append(torch.empty(0, dtype=gs[0].dtype, device=gs[0].device).set_(gs[0].untyped_storage(), gs[0].storage_offset(), (total,)))
            else:
                small.extend(gs)
        buckets, cur, size = [], [], 0
        for g in small:
            cur.append(g)
            size += g.numel() * g.element_size()
            if size >= self.bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
        if cur:
            buckets.append(cur)
        return direct, buckets

    def _views(self, flat, bucket):
        return [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in bucket]), bucket)]

    def begin(self):
        """Issues the SUM all-reduces (asynchronously, on the process group's stream) and returns the state `finish` needs.  Work that does
        not touch these gradients may be launched in between: it overlaps the exchange."""
        if not is_dist():
            return None
        _check_not_capturing("FlatGradReducer.begin")
        if self.static and not self.frozen and self._kept is not None and self._signature() != self._kept_sig:
            self._kept = None                                  # a gradient appeared, vanished or moved since the plan was made: plan again
        if self.static and self._kept is not None:
            direct, buckets, flats, views = self._kept
            for b, v in zip(buckets, views):
                torch._foreach_copy_(v, b)
        else:
            direct, buckets = self._plan()
            flats = [torch.cat([g.reshape(-1) for g in b]) for b in buckets]
            views = [self._views(f, b) for f, b in zip(flats, buckets)]
            if self.static:
                self._kept, self._kept_sig = (direct, buckets, flats, views), self._signature()
        works = [dist.all_reduce(f, op=dist.ReduceOp.SUM, async_op=True) for f in direct + flats]
        return direct, buckets, flats, views, works

    def finish(self, state, average=False):
        if state is None:
            return
        direct, buckets, flats, views, works = state
        for w in works:
            w.wait()
        if average:
            ws = dist.get_world_size()
            for f in direct + flats:
                f.div_(ws)
        for b, v in zip(buckets, views):
            torch._foreach_copy_(b, v)

    def reduce(self, average=False):
        """SUM (or mean) all-reduce of every parameter gradient, in place."""
        self.finish(self.begin(), average)
