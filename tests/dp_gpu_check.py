"""Two data-parallel ranks on ONE device with the REAL (tiny golden) model - child of tests/test_gpu_model.py::test_two_ranks_on_one_gpu.
Checks, per SURVEY 8e: (1) ActNorm data-dependent init from statistics summed over ranks == the reference's global-batch init;
(2) uneven shards, frame-weighted losses, SUM all-reduce -> every gradient equals the reference's single-process global-batch gradient
(eager, then as the two-graph overlap form bench.py runs: body graph | deferred 1x1 weight-gradient tail graph, early + tail reducers);
(3) dropout streams differ between ranks.  Backend: RCCL ("nccl") when it accepts two ranks on one device, else gloo - the reason is logged.
usage: python tests/dp_gpu_check.py <backend> <port> <logfile>       (spawns its two ranks itself)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def probe(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    assert float(t[0]) == world
    dist.destroy_process_group()


def worker(rank, world, port, backend, log):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from helpers import load_case, tiny_hp_dict
    from glow_tts_amd import _lib, decoder as D
    from glow_tts_amd.distributed import FlatGradReducer, actnorm_stats_allreduce, global_batch_weight, global_frame_weight
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS, MLE_Loss
    say = (lambda *a: (print(*a, flush=True), open(log, "a").write(" ".join(str(x) for x in a) + "\n"))) if rank == 0 else (lambda *a: None)
    sd, grads, r = load_case("tiny_vanilla.npz")
    d = np.load(os.path.join(HERE, "golden", "tiny_vanilla.npz"))
    hp = tiny_hp_dict("Vanilla")
    hp["HIP_Precision"] = "f32"
    t = lambda k: torch.from_numpy(r[k]).cuda()
    sl = slice(0, 1) if rank == 0 else slice(1, 3)                                   # uneven shards of the 3-utterance golden batch
    shard = tuple(t(k)[sl].contiguous() for k in ("tokens", "token_lengths", "mels", "mel_lengths"))
    mle_fn = MLE_Loss(Recursive_Parse(hp))

    def make(sd_):
        m = GlowTTS(Recursive_Parse(hp))
        m.load_state_dict(sd_)
        m.actnorm_allreduce = actnorm_stats_allreduce
        return m.cuda()

    # (1) ActNorm init from the global batch (Modules.py:698-711): the fixture keeps the pre-init parameters
    sd0 = dict(sd)
    for k in d.files:
        if k.startswith("sd_before/"):
            sd0[k[len("sd_before/"):]] = torch.from_numpy(d[k])
    m0 = make(sd0).train()
    for mod in m0.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m0.hp.Decoder.Affine_Coupling.WaveNet.Dropout_Rate = 0.0
    m0.hp.Encoder.Prenet.Dropout_Rate = m0.hp.Encoder.Transformer.Dropout_Rate = m0.hp.Encoder.Duration_Predictor.Dropout_Rate = 0.0
    with torch.no_grad():
        m0(*shard, None, None, None)
    torch.cuda.synchronize()
    worst = 0.0
    for f in range(3):
        for nm in ("logs", "bias"):
            k = f"layer_Dict.Decoder.layer_Dict.Flows.{f}.layers.0.{nm}"
            worst = max(worst, (dict(m0.named_parameters())[k].detach().cpu() - sd[k]).abs().max().item())
    assert worst < 3e-4, worst
    say(f"[dp] ActNorm init from rank-summed statistics == reference global-batch init (max |diff| {worst:.2e})")

    # (2) gradients == single-process global batch
    model = make(sd).eval()
    for f in model.layer_Dict["Decoder"].layer_Dict["Flows"]:
        f.layers[0].initialized = True
    wf = global_frame_weight(shard[3].sum())
    wb = global_batch_weight(shard[0].shape[0], device="cuda")

    def fwd_bwd():
        z, mm, ms, ld, dur, durt, _, _ = model(*shard, None, None, None)
        loss = mle_fn(z=z, mean=mm, std=ms, log_dets=ld, lengths=shard[3]) * wf + torch.nn.functional.mse_loss(dur, durt) * wb
        model.zero_grad(set_to_none=True)
        loss.backward()
        return loss.detach()

    def check(tag):
        torch.cuda.synchronize()
        worst = ("", 0.0)
        for k, p in model.named_parameters():
            want = grads.get(k)
            if want is None:
                continue
            err = (p.grad.cpu() - want).abs().max().item() / (want.abs().max().item() + 1e-5)
            worst = max(worst, (k, err), key=lambda x: x[1])
            assert err < 5e-3, (tag, k, err)
        say(f"[dp] {tag}: all {len(grads)} gradients == reference single-process global-batch gradients (worst rel {worst[1]:.2e})")

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd_bwd()
        FlatGradReducer(list(model.parameters())).reduce(average=False)
    torch.cuda.current_stream().wait_stream(side)
    check("eager step + flat all-reduce")
    with torch.cuda.stream(side):
        fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    keep = []
    body, tail_g = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with _lib.pinned_sink(keep):
        with D.defer_tail_wgrads():
            with torch.cuda.graph(body, capture_error_mode="thread_local"):
                fwd_bwd()
        with torch.cuda.graph(tail_g, pool=body.pool(), capture_error_mode="thread_local"):
            D.flush_tail_wgrads()
    tail_ids = {id(p) for p in model._dec_stacks.tail_leaves()}
    early = FlatGradReducer([p for p in model.parameters() if id(p) not in tail_ids])
    tail = FlatGradReducer([p for p in model.parameters() if id(p) in tail_ids])
    for rep in range(2):
        body.replay()
        pending = early.begin()
        tail_g.replay()
        tail.reduce(average=False)
        early.finish(pending)
        check(f"two-graph overlap step, replay {rep}")

    # (2b) the product form: glow_tts_amd.graph_step.GraphedTrainStep in data-parallel mode (what Trainer.Train_Step runs under torchrun): three
    # graphs around the exchange, clip + RAdam + Noam schedule on the reduced gradients.  Four steps on uneven shards must leave every rank
    # with the parameters of a single-process eager run on the global batch.
    from glow_tts_amd.graph_step import GraphedTrainStep
    from glow_tts_amd.optim import Modified_Noam_Scheduler, RAdam, clip_grad_norm_
    full = tuple(t(k) for k in ("tokens", "token_lengths", "mels", "mel_lengths"))

    def fresh():
        m = make(sd).eval()
        for f in m.layer_Dict["Decoder"].layer_Dict["Flows"]:
            f.layers[0].initialized = True
        o = RAdam(m.parameters(), lr=1e-3, eps=1e-6, weight_decay=1e-6)
        return m, o, Modified_Noam_Scheduler(o, base=4000)

    def dp_loss(m, tokens, tl, mels, ml, wf_, wb_):
        z, mm, ms, ld, dur, durt, _, _ = m(tokens, tl, mels, ml, None, None, None)
        return mle_fn(z=z, mean=mm, std=ms, log_dets=ld, lengths=ml) * wf_ + torch.nn.functional.mse_loss(dur, durt) * wb_
    mg, og, sg = fresh()
    gstep = GraphedTrainStep(mg, dp_loss, warmup=2, optimizer=og, scheduler=sg, max_grad_norm=5.0)
    wb_t = torch.tensor(wb, device="cuda")
    for _ in range(4):
        gstep(*shard, wf.clone(), wb_t)
    torch.cuda.synchronize()
    assert gstep.steps_taken == 4 and len(gstep.graphs) == 1 and gstep.graphs[next(iter(gstep.graphs))]["tail"] is not None
    mr, orr, sr = fresh()
    for _ in range(4):
        mr.zero_grad(set_to_none=True)
        dp_loss(mr, *full, 1.0, 1.0).backward()
        clip_grad_norm_(list(mr.parameters()), 5.0)
        orr.step(); sr.step()
    torch.cuda.synchronize()
    worst = ("", 0.0)
    for (k, pa), pb in zip(mr.named_parameters(), mg.parameters()):
        err = (pa - pb).abs().max().item() / max(1.0, pa.abs().max().item())
        worst = max(worst, (k, err), key=lambda x: x[1])
        assert err < 2e-4, (k, err)
        assert orr.state[pa]["step"] == og.state[pb]["step"] == 4
    flat = torch.cat([p_.detach().reshape(-1) for p_ in mg.parameters()])
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi), "replicas diverged"
    say(f"[dp] GraphedTrainStep (3 graphs around the exchange, clip + RAdam): 4 steps == single-process global-batch run (worst rel {worst[1]:.2e}), replicas identical")

    # (3) dropout streams differ per rank (same inputs, per-rank seeds)
    torch.manual_seed(4321 + rank)
    mt = make(sd).train()
    for f in mt.layer_Dict["Decoder"].layer_Dict["Flows"]:
        f.layers[0].initialized = True
    full = tuple(t(k) for k in ("tokens", "token_lengths", "mels", "mel_lengths"))
    with torch.no_grad():
        z = mt(*full, None, None, None)[0]
    both = [torch.empty_like(z) for _ in range(world)]
    dist.all_gather(both, z.contiguous())
    assert (both[0] - both[1]).abs().max() > 1e-3
    say(f"[dp] dropout streams differ between ranks (max |z0 - z1| = {(both[0] - both[1]).abs().max().item():.3f})")
    dist.barrier()
    say("DP GPU CHECK OK backend=" + backend)
    dist.destroy_process_group()


if __name__ == "__main__":
    backend, port, log = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    if backend == "probe":
        mp.spawn(probe, args=(2, port), nprocs=2, join=True)
        print("NCCL PROBE OK")
    else:
        mp.spawn(worker, args=(2, port, backend, log), nprocs=2, join=True)
