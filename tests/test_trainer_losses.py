"""Host-side loss arithmetic of the training entry point (CPU).  The reference averages the duration MSE over B x (longest text of the batch)
elements (Train.py:210 on tensors its collater padded to the batch maximum, Datasets.py:225-250); this package pads the token axis to a shape
bucket, and `trainer.duration_loss` must still return the reference's number and gradient (a plain MSELoss over the bucketed tensors
is smaller by max_len / bucket_len, a batch-dependent factor)."""
import torch

from glow_tts_amd.trainer import duration_loss


def test_duration_loss_is_independent_of_the_padding_bucket():
    g = torch.Generator().manual_seed(0)
    lengths = torch.tensor([17, 9, 23, 4])
    B, tmax = len(lengths), int(lengths.max())
    mask = (torch.arange(tmax)[None] < lengths[:, None]).float()
    d = (torch.randn(B, 1, tmax, generator=g) * mask[:, None]).requires_grad_(True)
    t = torch.randn(B, 1, tmax, generator=g) * mask[:, None]
    want = torch.nn.MSELoss()(d, t)                                  # the reference: tensors padded to the batch maximum
    want.backward()
    for bucket in (tmax, 32, 64):
        dp = torch.zeros(B, 1, bucket)
        dp[:, :, :tmax] = d.detach()
        dp.requires_grad_(True)
        tp = torch.zeros(B, 1, bucket)
        tp[:, :, :tmax] = t
        got = duration_loss(dp, tp, lengths)
        got.backward()
        assert torch.allclose(got, want, rtol=1e-6, atol=0), (bucket, got.item(), want.item())
        assert torch.allclose(dp.grad[:, :, :tmax], d.grad, rtol=1e-6, atol=1e-9) and float(dp.grad[:, :, tmax:].abs().sum()) == 0.0
        if bucket > tmax:                                           # what the old code computed: too small by tmax / bucket
            assert abs(torch.nn.MSELoss()(dp, tp).item() - want.item() * tmax / bucket) < 1e-6


def _dp_worker(rank, world, port, q):
    """Each rank holds a ragged shard; the frame-weighted / extent-normalised shard losses must sum to the single-process global-batch loss."""
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from glow_tts_amd.distributed import before_capture, global_frame_weight, global_step_scalars, global_token_extent
    g = torch.Generator().manual_seed(0)
    lengths = torch.tensor([17, 9, 23, 4])                           # global batch; rank 0 holds the SHORT texts
    tmax = int(lengths.max())
    mask = (torch.arange(tmax)[None] < lengths[:, None]).float()
    d = torch.randn(4, 1, tmax, generator=g) * mask[:, None]
    t = torch.randn(4, 1, tmax, generator=g) * mask[:, None]
    want = torch.nn.MSELoss()(d, t)                                  # Train.py:210 on the global batch
    own = [1, 3] if rank == 0 else [0, 2]
    ext = global_token_extent(lengths[own].max())
    # the Trainer's per-step form: both scalars from ONE collective
    frames = torch.tensor(100.0 * (rank + 1))
    fw2, ext2 = global_step_scalars(frames, lengths[own].max())
    assert float(ext2) == float(ext) and abs(float(fw2) - float(global_frame_weight(frames))) < 1e-7 and abs(float(fw2) - (rank + 1) / 3.0) < 1e-6
    before_capture()                                                 # (a backend without a watchdog: returns at once)
    got = duration_loss(d[own][:, :, :32], t[own][:, :, :32], lengths[own], ext) / world
    tot = got.clone()
    dist.all_reduce(tot)
    wrong = duration_loss(d[own], t[own], lengths[own]) / world      # every rank normalising by ITS longest text (round 3)
    wsum = wrong.clone()
    dist.all_reduce(wsum)
    q.put((rank, float(ext), float(tot), float(want), float(wsum)))
    dist.destroy_process_group()


def test_duration_loss_of_ragged_shards_sums_to_the_global_batch_loss():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29671
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    for rank, ext, tot, want, wsum in res:
        assert ext == 23.0
        assert abs(tot - want) < 1e-6 * max(1.0, abs(want)), (tot, want)
        assert abs(wsum - want) > 1e-3 * abs(want)                    # the per-rank extent was NOT the global-batch loss
