#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): mel-frames/sec of a Glow-TTS training step (forward incl. log-prior + MAS +
losses, backward, gradient all-reduce when N > 1) on LJSpeech-shaped synthetic batches, B = 32 utterances per GPU,
T_tokens = 120, T_mel = 800 (BASELINE config 2: Vanilla, 1xMI355X, bf16), plus MAS us/utterance.

    python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches it through torch.distributed.run (one rank per GPU, RCCL).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_FRAME_FWD_BWD = 71.3e6      # SURVEY.md 8d / BASELINE.md: 57.05 GFLOP per 800-frame utterance (reference FlopCounterMode)


def synthetic_batch(B, Tt, Tm, mel_dim, seed, device, ragged=False):
    g = torch.Generator().manual_seed(seed)
    tokens = torch.randint(0, 35, (B, Tt), generator=g)
    mels = (torch.randn(B, mel_dim, Tm, generator=g) * 1.5).clamp(-4, 4)
    if ragged:
        ml = 2 * torch.randint(300, 501, (B,), generator=g).clamp(max=Tm // 2)
        tl = torch.round(0.15 * ml).long().clamp(max=Tt)
        ml[0], tl[0] = Tm, Tt
    else:
        ml = torch.full((B,), Tm, dtype=torch.long)
        tl = torch.full((B,), Tt, dtype=torch.long)
    for b in range(B):
        tokens[b, tl[b]:] = 1
        mels[b, :, ml[b]:] = -4.0
    return tokens.to(device), tl.to(device), mels.to(device), ml.to(device)


def build_model(precision, device):
    import yaml
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS, MLE_Loss
    with open(os.path.join(REPO, "glow_tts_amd", "Hyper_Parameters.default.yaml")) as f:
        hp = yaml.safe_load(f)
    hp["Mode"] = "Vanilla"
    hp["HIP_Precision"] = precision
    hp = Recursive_Parse(hp)
    torch.manual_seed(0)
    model = GlowTTS(hp).to(device).train()
    # the reference zero-initialises the coupling's End conv (identity flow at step 0); give it small random values
    # so the benchmarked step does the arithmetic of a model in training, not of the degenerate init
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for n, p in model.named_parameters():
            if n.endswith("layer_Dict.End.weight"):
                p.copy_((torch.randn(p.shape, generator=g) * 0.01).to(device))
    return model, MLE_Loss(hp), hp


def train_step(model, mle_loss, batch, reducer=None, world=1):
    """Train.py:193-227 up to (and including) backward, plus the gradient all-reduce when data parallel.
    Data parallel: every rank scales its MLE loss (a mean over ITS frames, Modules.py:1026) by local/global frames and its
    duration MSE (a mean over its padded [B,1,Tt]) by 1/world, and gradients are SUMMED: the result is the gradient of the
    single-process loss on the global batch (tests/test_distributed_cpu.py)."""
    from glow_tts_amd.distributed import global_frame_weight
    tokens, tl, mels, ml = batch
    z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, _, _ = model(tokens, tl, mels, ml, None, None, None)
    mle = mle_loss(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=ml)
    length = torch.nn.functional.mse_loss(log_dur, log_dur_t)                            # Train.py:203-211
    if reducer is not None:
        loss = mle * global_frame_weight(ml.sum()) + length / world
    else:
        loss = mle + length
    model.zero_grad(set_to_none=True)
    loss.backward()
    if reducer is not None:
        reducer.reduce(average=False)
    return mle + length


TRAFFIC_BF16_B32 = 21.1e6    # FETCH_SIZE x 2 + WRITE_SIZE of conv_dma_kernel at B = 32 (profiles/r01_conv_pmc.txt)


def dominant_kernel_roofline(precision, B, T, iters=30):
    """The WaveNet In_i k=5 conv (Modules.py:861), the kernel that carries most of the FLOPs: timed alone with HIP events
    on the launch stream.  Algorithmic FLOPs per launch = 2 * rows * 384 * 192 * 5."""
    from glow_tts_amd import ops
    dev = "cuda"
    H, k = 192, 5
    R = B * (T + 4)
    prec = ops.BF16 if precision == "bf16" else ops.F32
    bf = prec == ops.BF16                           # bf16 mode: state and gates are stored as bf16 (what the training step runs)
    a = torch.randn(R, H, device=dev)
    w = torch.randn(2 * H, H, k, device=dev) / (H * k) ** 0.5
    pw = ops.pack_weight(w, perm=ops.PERM_PAIR, perm_h=H, precision=prec)
    bias = torch.zeros(2 * H, device=dev)
    if bf:
        a = a.to(torch.bfloat16)
    G = torch.empty(R, 2 * H, device=dev, dtype=a.dtype)
    io = (ops.IO_A_BF16 | ops.IO_OUT0_BF16) if bf else 0
    run = lambda: ops.conv_cl(a, pw, H, R, pad=2, epi=ops.EPI_GATE, h=H, n=2 * H, rows_per_utt=T + 4, bias=bias, out0=G, ld0=2 * H, io_flags=io)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) / 1e3 / iters
    flops = 2.0 * B * T * (2 * H) * H * k              # valid rows only
    peak = 2500.0 if precision == "bf16" else 157.3
    ach = flops / sec / 1e12
    name = "conv_dma_kernel<EPI_GATE, 5>" if bf else "conv_cl_kernel<float, EPI_GATE, 5>"
    return {"bound": "mfma", "kernel": name + " (WaveNet In_i k=5, 192->384)", "achieved": round(ach, 1), "peak": peak,
            "unit": "TFLOP/s", "frac": round(ach / peak, 4), "us_per_launch": round(sec * 1e6, 2),
            # HBM bytes per launch of this kernel at this shape from separate rocprofv3 --pmc passes (FETCH_SIZE x 2 per the gfx950
            # correction + WRITE_SIZE; profiles/r01_conv_pmc.txt), not re-measured live; algorithmic bytes are 15.6e6
            "traffic": TRAFFIC_BF16_B32 if (precision == "bf16" and B == 32 and T == 400) else None}


def mas_us_per_utt(B, Tx, Ty, iters=30):
    from glow_tts_amd import alignment
    g = torch.Generator().manual_seed(1234)
    v = (torch.randn(B, Ty, Tx, generator=g) * 30 - 100).cuda()
    tx = torch.full((B,), Tx, dtype=torch.long, device="cuda")
    ty = torch.full((B,), Ty, dtype=torch.long, device="cuda")
    from glow_tts_amd.monotonic_align import path_from_idx
    for _ in range(3):
        path_from_idx(alignment.maximum_path_t(v, tx, ty), Tx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        path_from_idx(alignment.maximum_path_t(v, tx, ty), Tx)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters / B


def cpu_baseline(seconds_budget=25.0):
    """The oracle (CPU restatement of the reference, oracle/glowtts_ref.py + oracle/mas_ref.c) timed on this host:
    BASELINE config 1 = Vanilla, B = 8, T_tokens = 120, T_mel = 800, fp32, forward + losses + backward."""
    from oracle import glowtts_ref as O
    import yaml
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS
    cores = os.cpu_count() or 1
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    torch.set_num_threads(cores)
    with open(os.path.join(REPO, "glow_tts_amd", "Hyper_Parameters.default.yaml")) as f:
        hpd = yaml.safe_load(f)
    hpd["Mode"] = "Vanilla"
    torch.manual_seed(0)
    model = GlowTTS(Recursive_Parse(hpd))
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cfg = O.Cfg()
    B, Tt, Tm = 8, 120, 800
    tokens, tl, mels, ml = synthetic_batch(B, Tt, Tm, 80, 1234, "cpu")
    times = []
    t_start = time.time()
    while len(times) < 4 and (time.time() - t_start < seconds_budget or len(times) < 2):
        t0 = time.time()
        out = O.forward_train(sd, cfg, tokens, tl, mels, ml)
        mle, length = O.train_losses(out, ml, cfg)
        for v in sd.values():
            v.grad = None
        (mle + length).backward()
        times.append(time.time() - t0)
    best = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": round(B * Tm / best, 1), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch fp32 + C MAS), Vanilla B=8 T_mel=800 T_tok=120, fwd+losses+bwd, best of {len(times) - 1} steps after 1 warm-up, {best:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--ragged", action="store_true", help="Set V (ragged lengths) instead of Set F (fixed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to smoke-test the "
                    "multi-rank code path on a single-GPU box together with GLOWTTS_BENCH_ONE_DEVICE=1)")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch every kernel eagerly instead of replaying the "
                    "captured hipGraph of the step (default: graph replay; at B = 32 the eager step is bound by ~17 ms of host launch work)")
    ap.set_defaults(graph=True)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("GLOWTTS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    from glow_tts_amd.distributed import FlatGradReducer, actnorm_stats_allreduce
    model, mle_loss, hp = build_model(args.precision, dev)
    reducer = None
    if world > 1:
        import torch.distributed as dist
        for p in model.parameters():                       # identical replicas
            dist.broadcast(p.data, 0)
        model.actnorm_allreduce = actnorm_stats_allreduce
        reducer = FlatGradReducer(list(model.parameters()))
    torch.manual_seed(4321 + rank)                              # dropout streams differ per rank; the replicas' weights do not
    B, Tt, Tm = args.batch, int(os.environ.get("GLOWTTS_BENCH_TT", "120")), 800          # (the env override is for experiments only)
    batch = synthetic_batch(B, Tt, Tm, 80, 1234 + rank, dev, ragged=args.ragged)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # A step = forward + losses + backward as ONE captured hipGraph (static shapes; the dropout seed is re-drawn on the
    # device inside the graph), followed - when data parallel - by the flat-bucket gradient all-reduce.
    from glow_tts_amd.distributed import global_frame_weight
    mode = "eager"
    graph = tail_graph = early = tail = None
    side = torch.cuda.Stream() if args.graph else None
    if args.graph:
        # every eager step that precedes the capture runs on the side stream too: a backward that ran on the default stream
        # first makes the captured AccumulateGrad nodes hop streams, which crashes hipStreamEndCapture on ROCm 7.2
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, args.warmup - 1)):
                loss = train_step(model, mle_loss, batch, reducer, world)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    else:
        for _ in range(max(1, args.warmup - 1)):               # also runs the ActNorm data-dependent init
            loss = train_step(model, mle_loss, batch, reducer, world)
    if args.graph:
        try:
            wfr = global_frame_weight(batch[3].sum()) if world > 1 else None      # constant for a fixed batch

            def fwd_bwd():
                tokens, tl, mels, ml = batch
                z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, _, _ = model(tokens, tl, mels, ml, None, None, None)
                mle = mle_loss(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=ml)
                length = torch.nn.functional.mse_loss(log_dur, log_dur_t)
                total = mle * wfr + length / world if world > 1 else mle + length
                model.zero_grad(set_to_none=True)
                total.backward()
                return (mle + length).detach()

            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    fwd_bwd()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            if world > 1 and os.environ.get("GLOWTTS_DP_OVERLAP", "1") != "0":
                # Data parallel: the step is two graphs.  The first ends with the k-tap weight gradients (71 of the 114 MB); their
                # all-reduce - with the encoder's and the ActNorm / 1x1 gradients - then runs under the second graph, which holds the
                # 1x1 weight-gradient groups and the weight-norm backward of their classes; the 15 MB those produce are reduced last.
                from glow_tts_amd import decoder as D
                with D.defer_tail_wgrads():
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        static_loss = fwd_bwd()
                tail_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(tail_graph, pool=graph.pool(), capture_error_mode="thread_local"):
                    D.flush_tail_wgrads()
                tail_ids = {id(p) for p in model._dec_stacks.tail_leaves()}
                early = FlatGradReducer([p for p in model.parameters() if id(p) not in tail_ids])
                tail = FlatGradReducer([p for p in model.parameters() if id(p) in tail_ids])
            else:
                with torch.cuda.graph(graph, capture_error_mode="thread_local" if world > 1 else "global"):
                    static_loss = fwd_bwd()
            graph.replay()
            if tail_graph is not None:
                tail_graph.replay()
            torch.cuda.synchronize()
            mode = "hipgraph"
        except Exception as exc:                               # noqa: BLE001 - fall back to eager launches, say so
            import traceback
            traceback.print_exc()
            print(f"[bench] graph capture failed ({type(exc).__name__}); running eagerly", file=sys.stderr)
            graph = tail_graph = None
            torch.cuda.synchronize()

    def one_step():
        if graph is not None:
            graph.replay()
            if tail_graph is not None:
                pending = early.begin()
                tail_graph.replay()
                tail.reduce(average=False)
                early.finish(pending)
            elif reducer is not None:
                reducer.reduce(average=False)
            return static_loss
        return train_step(model, mle_loss, batch, reducer, world)

    one_step()
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        loss = one_step()
    barrier()
    elapsed = time.time() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if world > 1:
        # every gradient must have gone through the exchange: reduced gradients are identical on all ranks, unreduced ones are not
        # (different utterances and dropout streams per rank)
        import torch.distributed as dist
        cs = torch.stack([p.grad.double().sum() for p in model.parameters() if p.grad is not None])
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit(f"[bench] rank {rank}: {int((lo != hi).sum())} gradient tensors differ between ranks after the all-reduce")
    frames = int(batch[3].sum().item())
    if world > 1:
        import torch.distributed as dist
        ft = torch.tensor([frames], device=dev, dtype=torch.float64)
        dist.all_reduce(ft)
        frames = int(ft.item())
    if rank == 0:
        value = frames * args.steps / elapsed
        out = {
            "metric": "mel-frames/sec (train fwd+bwd)", "value": round(value, 1), "unit": "mel-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE config 2: Vanilla single-speaker, LJSpeech-shaped synthetic (80-mel, 800 frames, 120 tokens), "
                                   f"batch={B}/GPU, {'ragged Set V' if args.ragged else 'fixed Set F'}, forward+losses+backward"
                                   + (", RCCL grad all-reduce" if world > 1 else ""),
                       "global_batch": B * world, "mel_frames": Tm, "tokens": Tt, "parallelism": f"dp{world}"},
            "loss": round(float(loss.item()), 4), "launch_mode": mode,
            "model_tflops": round(value * FLOP_PER_FRAME_FWD_BWD / 1e12, 2),
            "mas_us_per_utt": round(mas_us_per_utt(B, Tt, Tm), 3),
            "roofline": dominant_kernel_roofline(args.precision, B, Tm // 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
