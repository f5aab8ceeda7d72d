"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference is single-process (Train.py:27-35); SURVEY.md 8e lists what a
sharded step must keep consistent:
  * gradients: SUM all-reduce of one flat fp32 bucket per parameter group, scaled so that the result equals the
    gradient of the single-process *global-batch* loss (the MLE loss divides by the number of frames, Modules.py:1026,
    so ranks are weighted by their frame counts, not averaged);
  * ActNorm data-dependent init (Modules.py:698-711): the [2C+1] batch statistics are summed over ranks.
"""
import torch
import torch.distributed as dist


SINGLE_RANK_IS_DIST = False      # bench.py --force-dist: run the collective code path with a 1-rank group (exercises RCCL on a single-GPU box)


def is_dist():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or SINGLE_RANK_IS_DIST)


def broadcast_parameters(model, src=0):
    """Identical replicas at start: rank `src`'s parameters and buffers to every rank (tensors made contiguous first: RCCL rejects
    strided ones, e.g. a loaded checkpoint's column-major 4x4 flow weights)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            if not t.data.is_contiguous():
                t.data = t.data.contiguous()
            dist.broadcast(t.data, src)


def actnorm_stats_allreduce(stats):
    if is_dist():
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)


def global_frame_weight(local_frames):
    """local_frames: 0-d tensor (sum of this rank's mel lengths).  Returns local / global: the factor that turns this
    rank's mean-over-local-frames MLE loss into its share of the global-batch loss."""
    if not is_dist():
        return torch.ones((), device=local_frames.device)
    tot = local_frames.detach().clone().to(torch.float32)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return local_frames.to(torch.float32) / tot


def global_token_extent(local_max):
    """local_max: 0-d tensor (this rank's longest text).  Returns the longest text of the GLOBAL batch (all-reduce MAX): the extent over which
    the reference's duration MSE (Train.py:210, a mean over B x longest text of the batch) is taken when the batch is sharded."""
    if not is_dist():
        return local_max
    t = local_max.detach().clone().to(torch.float32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def global_batch_weight(local_utterances, device=None):
    """local / global utterance count: the factor for losses that are means over the padded batch (the duration MSE, Train.py:211) when
    ranks hold different numbers of utterances (equal shards: 1 / world)."""
    if not is_dist():
        return 1.0
    t = torch.tensor([float(local_utterances)], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(local_utterances) / float(t.item())


class FlatGradReducer:
    """All-reduces the gradients of `params` with few, large collectives (xGMI is point-to-point: few large messages beat many small
    ones) and IN PLACE: `p.grad` keeps its address, which a replayed hipGraph of the step relies on.
      * gradients that already tile one contiguous storage - the decoder's stacked weight classes, whose leaf gradients are views of
        one stacked tensor (decoder.LeafStack) - are reduced directly on that storage: no copy at all (>= 95 % of the bytes);
      * the remaining small tensors are concatenated into flat buckets, reduced, and copied back.
    Buckets follow reverse parameter order, i.e. the order the backward pass produces them."""

    def __init__(self, params, bucket_bytes=64 << 20, direct_bytes=8 << 20, static=False):
        """static: the gradients keep their addresses from call to call (a replayed hipGraph of the step writes them in place): the plan and the
        flat buckets are built on the first call and kept - no per-step Python over hundreds of parameters, no allocation."""
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes, self.direct_bytes, self.static = bucket_bytes, direct_bytes, static
        self._kept = None

    def _plan(self):
        """-> (direct: list of flat tensors aliasing gradient storages, buckets: list of lists of gradient tensors)."""
        groups = {}
        for p in reversed(self.params):
            g = p.grad
            if g is None:
                g = p.grad = torch.zeros_like(p)
            groups.setdefault(g.untyped_storage().data_ptr(), []).append(g)
        direct, small = [], []
        for gs in groups.values():
            gs = sorted(gs, key=lambda t: t.storage_offset())
            # the gradients tile one span of the storage (alignment gaps of a few elements allowed: decoder.DecoderFunction's arenas pad classes to
            # 16 bytes and zero the pad)
            covered = all(t.is_contiguous() for t in gs) and all(0 <= b.storage_offset() - (a.storage_offset() + a.numel()) < 4 for a, b in zip(gs, gs[1:]))
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
        if self.static and self._kept is not None:
            direct, buckets, flats, views = self._kept
            for b, v in zip(buckets, views):
                torch._foreach_copy_(v, b)
        else:
            direct, buckets = self._plan()
            flats = [torch.cat([g.reshape(-1) for g in b]) for b in buckets]
            views = [self._views(f, b) for f, b in zip(flats, buckets)]
            if self.static:
                self._kept = (direct, buckets, flats, views)
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
