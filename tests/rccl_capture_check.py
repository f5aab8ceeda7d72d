"""Child process of tests/test_gpu_dp_rccl.py: the data-parallel step over RCCL with ONE rank (the test box has one GPU), in the order that killed
round 4's `bench.py --force-dist` on a fresh box - collectives issued on the stream of the warm-up passes, a capture within the watchdog's next
100-ms poll (glow_tts_amd/distributed.py, "Collectives and hipGraph capture in one process").

  capture:  ten NEW batch shapes in one process, each preceded - without any synchronisation - by the collectives a data-parallel step issues
            (ActNorm statistics, the per-step scalars) on the step's own stream; every shape is three captured graphs around the gradient exchange
            (graph_step.GraphedTrainStep).  The trajectory must equal the plain single-process GraphedTrainStep's on the same batches (a one-rank
            all-reduce is the identity, the frame weight is 1).
  trainer:  `Trainer.Train()` (Train.py:563-590) on a synthetic pattern directory with a one-rank RCCL group: graphed data-parallel Train_Steps over two
            shape buckets, evaluation, checkpoint, inference epoch.
"""
import os
import sys
import tempfile

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
what = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[2])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from glow_tts_amd import distributed as gd   # noqa: E402

if what == "capture":
    from test_gpu_model import build, load_case          # noqa: E402
    from glow_tts_amd.graph_step import GraphedTrainStep  # noqa: E402
    from glow_tts_amd.modules import MLE_Loss             # noqa: E402
    from glow_tts_amd.optim import Modified_Noam_Scheduler, RAdam   # noqa: E402
    sd, _, r = load_case("tiny_vanilla.npz")
    t = lambda k: torch.from_numpy(r[k]).cuda()
    b0 = (t("tokens"), t("token_lengths"), t("mels"), t("mel_lengths"))
    Tt, Tm = b0[0].shape[1], b0[2].shape[2]
    shapes = []
    for i in range(10):                                  # ten distinct (tokens, frames) shapes, lengths clamped to fit
        tt, tm = Tt - (i % 3), Tm - 2 * i
        shapes.append((b0[0][:, :tt].contiguous(), b0[1].clamp(max=tt), b0[2][:, :, :tm].contiguous(), b0[3].clamp(max=tm)))
    seq = shapes + shapes[::-1] + shapes[::3]

    def run(dp):
        gd.SINGLE_RANK_IS_DIST = dp
        m = build("Vanilla", "f32", sd)
        mle = MLE_Loss(m.hp)
        if dp:
            m.actnorm_allreduce = gd.actnorm_stats_allreduce

        def loss_fn(mm, tokens, tl, mels, ml, *scal):
            z, mean, std, ld, dur, durt, _, _ = mm(tokens, tl, mels, ml, None, None, None)
            w = scal[0] if scal else 1.0
            return mle(z=z, mean=mean, std=std, log_dets=ld, lengths=ml) * w + torch.nn.functional.mse_loss(dur, durt)
        opt = RAdam(m.parameters(), lr=1e-3, eps=1e-6, weight_decay=1e-6)
        step = GraphedTrainStep(m, loss_fn, warmup=2, optimizer=opt, scheduler=Modified_Noam_Scheduler(opt, base=4000), max_grad_norm=5.0)
        losses, junk = [], torch.zeros(321, device="cuda")
        for b in seq:
            if dp:
                # the hazard, on purpose: collectives with the STEP'S stream current (where the warm-up passes and, through autograd's AccumulateGrad
                # nodes, part of the captured backward run), then straight into the call that may capture - no synchronisation in between
                with torch.cuda.stream(step.stream):
                    gd.actnorm_stats_allreduce(junk)
                    scal = gd.global_step_scalars(b[3].sum(), b[1].max())
                torch.cuda.current_stream().wait_stream(step.stream)
                b = tuple(b) + (scal[0],)
            losses.append(float(step(*b).detach()))        # (read at once: a shape's loss lives in ONE static tensor, rewritten by its next replay)
        torch.cuda.synchronize()
        assert step.steps_taken == len(seq) and len(step.graphs) == 10
        if dp:
            assert all(e["tail"] is not None and e["opt"] is not None for e in step.graphs.values())
        return losses, [p.detach().clone() for p in m.parameters()]

    la, pa = run(False)
    lb, pb = run(True)
    if os.environ.get("RCCL_CHECK_VERBOSE"):
        print("[rccl] per-step loss differences:", " ".join(f"{abs(x - y):.1e}" for x, y in zip(la, lb)))
        print("[rccl] losses:", " ".join(f"{x:.4f}" for x in la))
    for i, (x, y) in enumerate(zip(la, lb)):
        assert abs(x - y) <= 5e-5 * max(1.0, abs(x)), (i, x, y)
    for i, (x, y) in enumerate(zip(pa, pb)):
        assert (x - y).abs().max().item() <= 1e-4 * max(1.0, x.abs().max().item()), i
    print(f"[rccl] 10 shapes x 3 captured graphs behind un-synchronised collectives, {len(seq)} data-parallel steps == the plain graphed trajectory "
          f"(last loss {lb[-1]:.5f} vs {la[-1]:.5f})")
    print("RCCL CAPTURE OK")
elif what == "trainer":
    from test_gpu_entrypoints import _make_dataset        # noqa: E402
    from glow_tts_amd.hparams import Recursive_Parse      # noqa: E402
    from glow_tts_amd.trainer import Trainer              # noqa: E402
    gd.SINGLE_RANK_IS_DIST = True
    with tempfile.TemporaryDirectory() as root:
        hp = Recursive_Parse(_make_dataset(root, "Vanilla"))
        hp.Train.Evaluation_Interval = 4
        torch.manual_seed(0)
        tr = Trainer(steps=0, hp=hp)
        assert tr.dp and tr.reducer is not None
        tr.Train()
        torch.cuda.synchronize()
        assert tr.steps >= hp.Train.Max_Step
        entries = list(tr._graphed.graphs.values())
        assert entries and all(e["tail"] is not None and e["early"] is not None and e["opt"] is not None for e in entries)
        assert all(torch.isfinite(p).all() for p in tr.model_Dict["GlowTTS"].parameters())
        got = sorted(os.listdir(os.path.join(hp.Inference_Path, "Step-5", "NPY")))
        assert got == [f"P{i}.npy" for i in range(4)], got
        assert any(f.startswith("S_") for f in os.listdir(hp.Checkpoint_Path))
        print(f"[rccl] Trainer.Train() data parallel over a one-rank RCCL group: {tr.steps} graphed steps over {len(entries)} shape bucket(s), evaluation, "
              "checkpoint and inference epoch done")
    print("RCCL TRAINER OK")
else:
    raise SystemExit(f"unknown check {what!r}")
gd.barrier()
dist.destroy_process_group()
