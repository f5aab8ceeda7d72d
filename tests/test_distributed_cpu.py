"""CPU suite, world_size = 2, gloo: the data-parallel host logic (glow_tts_amd/distributed.py) reproduces the
single-process global-batch gradient: frame-weighted loss, SUM all-reduce through the flat buckets, ActNorm statistics
summed over ranks.  (The HIP kernels themselves need a GPU; this covers the N > 1 path the driver runs on 8 GPUs.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _toy_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 6))


def _toy_batch():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(6, 10, 6, generator=g)                       # 6 "utterances", up to 10 "frames"
    lengths = torch.tensor([10, 4, 7, 9, 2, 8])
    return x, lengths


def _nll_sum(model, x, lengths):
    mask = (torch.arange(x.shape[1])[None, :] < lengths[:, None]).float().unsqueeze(-1)
    return (((model(x) - x) ** 2) * mask).sum()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from glow_tts_amd.distributed import FlatGradReducer, actnorm_stats_allreduce, global_frame_weight
    model = _toy_model()
    x, lengths = _toy_batch()
    sl = slice(0, 2) if rank == 0 else slice(2, 6)                # uneven shards: 14 vs 26 frames
    xs, ls = x[sl], lengths[sl]
    local_frames = ls.sum()
    # local mean-over-frames loss (what MLE_Loss computes, Modules.py:1026), turned into this rank's share of the global loss
    loss = _nll_sum(model, xs, ls) / local_frames * global_frame_weight(local_frames)
    loss.backward()
    # tiny thresholds: several buckets for the small gradients AND the zero-copy path for the larger ones
    red = FlatGradReducer(list(model.parameters()), bucket_bytes=16, direct_bytes=128)
    direct, buckets = red._plan()
    assert len(buckets) > 1 and len(direct) >= 1
    ptrs = [p.grad.data_ptr() for p in model.parameters()]
    if rank == 0:
        red.reduce(average=False)
    else:                                                         # the split form bench.py uses to overlap the exchange with later launches
        pending = red.begin()
        red.finish(pending)
    assert ptrs == [p.grad.data_ptr() for p in model.parameters()], "the reduction must be in place (hipGraph replays rely on it)"
    stats = torch.tensor([float(rank + 1), 2.0, float(local_frames)])
    actnorm_stats_allreduce(stats)
    # identical replicas at start, also for strided parameters (the 4x4 flow weights come column-major out of torch.linalg.qr; RCCL rejects those)
    from glow_tts_amd.distributed import broadcast_parameters
    torch.manual_seed(100 + rank)
    m2 = torch.nn.Linear(4, 4)
    m2.weight.data = torch.randn(4, 4).t()                        # non-contiguous
    m2.register_buffer("buf", torch.full((3,), float(rank)))
    assert not m2.weight.data.is_contiguous()
    broadcast_parameters(m2)
    torch.manual_seed(100)
    ref = torch.nn.Linear(4, 4)
    ref_w = torch.randn(4, 4).t()
    assert m2.weight.data.is_contiguous() and torch.equal(m2.weight.data, ref_w) and torch.equal(m2.bias.data, ref.bias.data) and float(m2.buf[0]) == 0.0
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in model.parameters()], "stats": stats}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_global_batch(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    model = _toy_model()
    x, lengths = _toy_batch()
    (_nll_sum(model, x, lengths) / lengths.sum()).backward()                # single process, global batch
    for g, p in zip(got["grads"], model.parameters()):
        assert torch.allclose(g, p.grad, atol=1e-6, rtol=1e-5)
    assert torch.allclose(got["stats"], torch.tensor([3.0, 4.0, float(lengths.sum())]))


def test_single_process_is_a_noop():
    from glow_tts_amd.distributed import FlatGradReducer, global_frame_weight, is_dist
    assert not is_dist()
    m = _toy_model()
    m(torch.randn(3, 6)).sum().backward()
    before = [p.grad.clone() for p in m.parameters()]
    FlatGradReducer(list(m.parameters())).reduce()
    assert all(torch.equal(a, p.grad) for a, p in zip(before, m.parameters()))
    assert float(global_frame_weight(torch.tensor(5))) == 1.0


def test_leaf_stack_aliases_leaves_and_routes_grads():
    """decoder.LeafStack: the stacked tensor shares storage with the leaf Parameters (no per-step copy), the backward hands every
    leaf its slice, in-place updates of a leaf are seen by the stack, and replaced leaves are re-pointed."""
    from glow_tts_amd.decoder import LeafStack
    torch.manual_seed(0)
    leaves = [torch.nn.Parameter(torch.randn(3, 2)) for _ in range(6)]
    before = [p.detach().clone() for p in leaves]
    st = LeafStack(leaves, (2, 3))
    t = st.tensor()
    assert t.shape == (2, 3, 3, 2) and all(torch.equal(p.detach(), b) for p, b in zip(leaves, before))
    assert all(p.data_ptr() == st.flat.data_ptr() + i * 6 * 4 for i, p in enumerate(leaves))
    w = torch.randn(2, 3, 3, 2)
    (t * w).sum().backward()
    assert all(torch.equal(p.grad, w.view(6, 3, 2)[i]) for i, p in enumerate(leaves))
    with torch.no_grad():
        leaves[4].add_(1.0)                                   # optimizer-style in-place update
    assert torch.equal(st.tensor().detach().view(6, 3, 2)[4], before[4] + 1.0)
    leaves[2].data = leaves[2].data.clone()                   # e.g. model.to(...): the leaf left the flat storage
    t2 = st.tensor()
    assert leaves[2].data_ptr() == st.flat.data_ptr() + 2 * 6 * 4 and torch.equal(t2.detach().view(6, 3, 2)[2], before[2])
