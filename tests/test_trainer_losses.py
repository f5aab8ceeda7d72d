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
