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


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


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


class FlatGradReducer:
    """All-reduces the gradients of `params` through a few large flat buckets (xGMI is point-to-point: few, large
    collectives beat many small ones).  Buckets are filled in reverse parameter order, i.e. in the order the
    backward pass produces them (decoder flows last-to-first, then the encoder)."""

    def __init__(self, params, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []
        cur, size = [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)

    def reduce(self, average=False):
        """SUM (or mean) all-reduce of every parameter gradient.  Per bucket: one concatenation kernel, one asynchronous
        all-reduce, and the parameters' .grad become VIEWS of the reduced flat buffer (no copy back)."""
        if not is_dist():
            return
        ws = dist.get_world_size()
        flats, works = [], []
        for b in self.buckets:
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in b])
            flats.append(flat)
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for b, flat, w in zip(self.buckets, flats, works):
            w.wait()
            if average:
                flat.div_(ws)
            off = 0
            for p in b:
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n
