"""Stress test of the last-workgroup hand-over (returning device-scope atomics instead of a release fence; csrc/loss_ops.hip prior_loss_kernel, csrc/dur_ops.hip
dur_proj_bwd_kernel): many launches beside a stream that keeps every CU and the L2s busy; every result must be bit-identical to the first."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from glow_tts_amd import _lib, alignment as A  # noqa: E402
from glow_tts_amd.encoder import ROW_PAD, DurProj  # noqa: E402


def main(iters=400):
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    B, C, Tx, Ty = 32, 80, 120, 800
    g = torch.Generator().manual_seed(1)
    tl = torch.randint(60, Tx + 1, (B,), generator=g); tl[0] = Tx
    ml = (torch.randint(300, Ty + 1, (B,), generator=g) // 2) * 2; ml[0] = Ty
    idx = torch.full((B, Ty), -1, dtype=torch.int32)
    for b in range(B):
        cuts = torch.sort(torch.randperm(int(ml[b]) - 1, generator=g)[: int(tl[b]) - 1] + 1).values
        idx[b, : ml[b]] = torch.bucketize(torch.arange(int(ml[b])), cuts, right=True).int()
    mean, ls = torch.randn(B, C, Tx, generator=g), torch.randn(B, C, Tx, generator=g) * 0.3
    z, ld = torch.randn(B, C, Ty, generator=g), torch.randn(B, generator=g)
    c = lambda t: t.to(dev)
    idx_d, tl_d, ml_d = c(idx), c(tl), c(ml)
    Cd, Tp = 256, Tx + 2 * ROW_PAD
    d = torch.randn(B * Tp, Cd, generator=g).to(dev)
    w, bias = (torch.randn(1, Cd, 1, generator=g) * 0.1).to(dev), torch.randn(1, generator=g).to(dev)
    mask = (torch.arange(Tx)[None] < tl[:, None]).float().unsqueeze(1).to(dev)
    gout = torch.randn(B, 1, Tx, generator=g).to(dev)
    noise_stream = torch.cuda.Stream()
    big = torch.randn(64 * 1024 * 1024, device=dev)
    first, owner, owner2 = None, {}, {}
    bad = 0
    for it in range(iters):
        with torch.cuda.stream(noise_stream):                     # HBM / L2 traffic and CU pressure beside the launches under test
            big.mul_(1.0000001)
            torch.mm(big[: 4096 * 4096].view(4096, 4096), big[4096 * 4096: 2 * 4096 * 4096].view(4096, 4096))
        m, l, zz, dd = (c(t).requires_grad_(True) for t in (mean, ls, z, ld))
        mm, ms, tg, _ = A.ExpandPair.apply(m, l, idx_d, tl_d, None)
        A.tag_prior(mm, ms, m, l, idx_d)
        mle = A.mle_loss(zz, mm, ms, dd, ml_d, 2, C, owner=owner)
        A.LossTerms([mle]).backward()
        dc, wc, bc = d.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        DurProj.apply(dc, wc, bc, mask, owner2).backward(gout)
        res = [t.detach().clone() for t in (mle, m.grad, l.grad, zz.grad, wc.grad, bc.grad)]
        if first is None:
            torch.cuda.synchronize()
            first = res
            # reference of the first: the unfused launches
            A.FUSED["seeded"] = False
            m2, l2, z2, d2 = (c(t).requires_grad_(True) for t in (mean, ls, z, ld))
            mm2, ms2, _, _ = A.ExpandPair.apply(m2, l2, idx_d, tl_d, None)
            A.tag_prior(mm2, ms2, m2, l2, idx_d)
            ref = A.mle_loss(z2, mm2, ms2, d2, ml_d, 2, C)
            ref.backward()
            A.FUSED["seeded"] = True
            assert torch.equal(ref.detach(), first[0]) and torch.equal(m2.grad, first[1]) and torch.equal(z2.grad, first[3]), "fused != unfused"
        else:
            for a, b in zip(first, res):
                if not torch.equal(a, b):
                    bad += 1
                    print("iteration", it, "differs:", (a - b).abs().max().item())
                    break
    torch.cuda.synchronize()
    print(f"stress_handover: {iters} iterations, {bad} mismatches; counters {[int(v.item()) for o in (owner, owner2) for v in o.values()]}")
    assert bad == 0


main(int(sys.argv[1]) if len(sys.argv) > 1 else 400)
