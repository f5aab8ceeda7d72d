#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): mel-frames/sec of a Glow-TTS training step (forward incl. log-prior + MAS +
losses, backward, gradient all-reduce when N > 1) on LJSpeech-shaped synthetic batches, plus MAS us/utterance.

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]
For N > 1 it runs one rank per GPU over RCCL: launched through torch.distributed.run (WORLD_SIZE set) it joins that job, launched plainly
(`python bench.py --gpus N`) it spawns the N ranks itself and refuses when the node has fewer GPUs.  Prints ONE JSON line on rank 0.

--config selects the BASELINE.json workload (default 2, the one the metric is quoted on):
    2  Vanilla single-speaker, bf16, 32 utterances per GPU, 800 frames / 120 tokens
    3  LUT speaker embedding (109 speakers, VCTK-shaped), 64 utterances over 2 GPUs  = 32 per GPU
    4  GE2E speaker-embedding mode (pre-computed unit-norm d-vectors), 128 over 8 GPUs = 16 per GPU
    5  PE / GST prosody-encoder mode, 256 over 8 GPUs = 32 per GPU, plus a long-form inverse-flow leg (`inverse_flow` in the JSON line:
       GlowTTS.inference through glow_tts_amd.graph_infer.GraphedInference, 200 tokens -> ~2000 frames per utterance)
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
if os.environ.get("GLOWTTS_PKG_ROOT"):          # A/B against another checkout of the package (tools only): its directory goes first on sys.path
    sys.path.insert(0, os.environ["GLOWTTS_PKG_ROOT"])

FLOP_PER_FRAME_FWD_BWD = 71.3e6      # SURVEY.md 8d / BASELINE.md: 57.05 GFLOP per 800-frame utterance (reference FlopCounterMode)
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    2: dict(mode="Vanilla", spk_type="LUT", batch=32, name="BASELINE config 2: Vanilla single-speaker"),
    3: dict(mode="SE", spk_type="LUT", batch=32, name="BASELINE config 3: LUT speaker embedding (109 speakers, VCTK-shaped; 64 utterances over 2 GPUs)"),
    4: dict(mode="SE", spk_type="GE2E", batch=16, name="BASELINE config 4: GE2E speaker-embedding mode (pre-computed d-vectors; 128 utterances over 8 GPUs)"),
    5: dict(mode="PE", spk_type="LUT", batch=32, name="BASELINE config 5: PE/GST prosody-encoder mode (256 utterances over 8 GPUs) + long-form inverse flow"),
}


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


def conditioning_inputs(cfg, B, seed, device, hp):
    """(speakers, mels_for_ge2e) of GlowTTS.forward for the config: LUT ids ~ U{0..108}; GE2E: unit-norm [B, 256] from N(0,1)."""
    g = torch.Generator().manual_seed(seed + 77)
    if cfg["mode"] != "SE":
        return None, None
    if cfg["spk_type"] == "LUT":
        return torch.randint(0, int(hp.Speaker_Embedding.Num_Speakers), (B,), generator=g).to(device), None
    v = torch.randn(B, int(hp.Speaker_Embedding.Embedding_Size), generator=g)
    return None, (v / v.norm(dim=1, keepdim=True)).to(device)


def build_model(precision, device, mode="Vanilla", spk_type="LUT"):
    import yaml
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS, MLE_Loss
    with open(os.path.join(REPO, "glow_tts_amd", "Hyper_Parameters.default.yaml")) as f:
        hp = yaml.safe_load(f)
    hp["Mode"] = mode
    hp["Speaker_Embedding"]["Type"] = spk_type
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


def forward_losses(model, mle_loss, batch, cond):
    tokens, tl, mels, ml = batch
    z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, _, _ = model(tokens, tl, mels, ml, cond[0], cond[1], None)
    from glow_tts_amd.alignment import duration_mse
    length = duration_mse(log_dur, log_dur_t)                                            # Train.py:203-211 MSELoss (one launch per direction; as Trainer._losses)
    mle = mle_loss(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=ml)
    return mle, length


def train_step(model, mle_loss, batch, cond, reducer=None, world=1, opt=None):
    """Train.py:193-233 (Train_Step): forward, losses, backward, the gradient all-reduce when data parallel, then - opt = (optimizer,
    scheduler, max_grad_norm) - clip_grad_norm_, RAdam and the Noam schedule.
    Data parallel: every rank scales its MLE loss (a mean over ITS frames, Modules.py:1026) by local/global frames and its
    duration MSE (a mean over its padded [B,1,Tt]) by 1/world, and gradients are SUMMED: the result is the gradient of the
    single-process loss on the global batch (tests/test_distributed_cpu.py)."""
    from glow_tts_amd.distributed import global_frame_weight
    from glow_tts_amd import alignment
    seeds = None
    if reducer is not None:
        seeds = [global_frame_weight(batch[3].sum()), torch.full((), 1.0 / world, device=batch[2].device)]
        alignment.SEEDS["mle"], alignment.SEEDS["rest"] = seeds
    try:
        mle, length = forward_losses(model, mle_loss, batch, cond)
    finally:
        alignment.SEEDS["mle"] = alignment.SEEDS["rest"] = None
    model.zero_grad(set_to_none=True)
    alignment.LossTerms([mle, length], seeds).backward()        # Train.py:213-216 `loss = MLE + Length; loss.backward()` (as Trainer._losses: seeded per term)
    if reducer is not None:
        reducer.reduce(average=False)
    if opt is not None:
        from glow_tts_amd.optim import clip_grad_norm_
        clip_grad_norm_(list(model.parameters()), opt[2])
        opt[0].step()
        opt[1].step()
    return mle + length


# ------------------------------------------------------------------------------------------------------------------------------------
# The two hottest kernels of the step, alone: the WaveNet In_l k = 5 conv with the gate epilogue (Modules.py:861-870) and its data
# gradient (the single largest kernel by time in profiles/*_kernel_stats.csv).  Used by the roofline leg below and by tools/pmc_conv.py.
# ------------------------------------------------------------------------------------------------------------------------------------
def hot_kernel_cases(precision, B, T):
    """-> {name: dict(run, flops, alg_bytes, kernel)}; flops / bytes per launch count VALID rows only."""
    from glow_tts_amd import ops
    dev = "cuda"
    H, k = 192, 5
    R = B * (T + 4)
    prec = ops.BF16 if precision == "bf16" else ops.F32
    bf = prec == ops.BF16                           # bf16 mode: state, gates and gate gradients are stored as bf16 (what the step runs)
    es = 2 if bf else 4
    dt = torch.bfloat16 if bf else torch.float32
    w = torch.randn(2 * H, H, k, device=dev) / (H * k) ** 0.5
    bias = torch.zeros(2 * H, device=dev)
    rowmask = torch.ones(R, device=dev)
    flops = 2.0 * B * T * (2 * H) * H * k
    wbytes = 2 * H * H * k * es
    cases = {}
    # forward: hs [R, H] -> gates [R, 2H] (+ bf16 tanh*sigmoid product [R, H])
    a = torch.randn(R, H, device=dev).to(dt)
    pw = ops.pack_weight(w, perm=ops.PERM_PAIR, perm_h=H, precision=prec)
    G = torch.empty(R, 2 * H, device=dev, dtype=dt)
    acts = torch.empty(R, H, device=dev, dtype=torch.bfloat16) if bf else None
    io = (ops.IO_A_BF16 | ops.IO_OUT0_BF16) if bf else 0
    cases["in_fwd"] = dict(
        run=lambda: ops.conv_cl(a, pw, H, R, pad=2, epi=ops.EPI_GATE, h=H, n=2 * H, rows_per_utt=T + 4, bias=bias, out0=G, ld0=2 * H,
                                out1=acts, ld1=H, io_flags=io),
        flops=flops, alg_bytes=B * T * (H * es + 2 * H * es + (H * 2 if bf else 0)) + wbytes,
        kernel=("conv_dma_kernel<EPI_GATE, 5>" if bf else "conv_cl_kernel<float, EPI_GATE, 5>") + " (WaveNet In_l k=5 + gate, 192->384)")
    # data gradient: d ins [R, 2H] (PAIR-packed) -> d h [R, H] = (conv^T + d h_next) * mask
    dins = torch.randn(R, 2 * H, device=dev).to(dt)
    pwt = ops.pack_weight(w, transpose=True, perm=ops.PERM_PAIR, perm_h=H, precision=prec)
    dnext = torch.randn(R, H, device=dev).to(dt)
    dh = torch.empty(R, H, device=dev, dtype=dt)
    io2 = (ops.IO_A_BF16 | ops.IO_IN0_BF16 | ops.IO_OUT0_BF16) if bf else 0
    cases["in_dgrad"] = dict(
        run=lambda: ops.conv_cl(dins, pwt, 2 * H, R, pad=2, epi=ops.EPI_LINEAR, flags=ops.F_MASK | ops.F_ADD_IN0, n=H, rows_per_utt=T + 4,
                                rowmask=rowmask, in0=dnext, ldi0=H, out0=dh, ld0=H, io_flags=io2),
        flops=flops, alg_bytes=B * T * (2 * H * es + H * es + H * es) + wbytes,
        kernel=("conv_dma_kernel<EPI_LINEAR, 5>" if bf else "conv_cl_kernel<float, EPI_LINEAR, 5>") + " (WaveNet In_l data gradient, 384->192)")
    if bf:
        cases["wn_fwd"] = fused_forward_case(B, T)
    return cases


def fused_forward_case(B, T):
    """The fused coupling-network forward (csrc/wavenet_fused.hip: Start .. End + affine coupling of ONE flow, kept activations written),
    launched alone through the C ABI.  Algorithmic FLOPs per valid row: 2 x (80 x 192 + 4 x 192 x 384 x 5 + 3 x 192 x 384 + 192 x 192 +
    192 x 160) = 3 557 376 (the halo rows the kernel recomputes are NOT counted); algorithmic bytes per valid row: 640 read (flow input)
    + 320 (x_b') + 4 x (384 + 768 + 384) (kept x_l, gate pairs, tanh * sigmoid, bf16) + 384 (skip sum, bf16 - round 5: its fp32 rows are no
    longer kept, 768 before) + 640 (m, logs) written, + the 3.6 MB weight image once."""
    import ctypes
    from glow_tts_amd import _lib, decoder as D, ops
    dev = "cuda"
    g = torch.Generator().manual_seed(5)
    cfgd = D.DecoderConfig(80, 1, 2, 4, 192, 4, 5, ops.BF16)
    H, C, Lw, k = 192, 160, 4, 5
    rn = lambda *sh, s=1.0: (torch.randn(*sh, generator=g) * s).to(dev)
    W = {"an_logs": rn(1, C, s=0.1), "an_bias": rn(1, C, s=0.1), "inv_w": torch.eye(4).unsqueeze(0).to(dev),
         "w_start": rn(1, H, C // 2, 1, s=(C // 2) ** -0.5), "b_start": rn(1, H, s=0.05),
         "w_in": rn(1, Lw, 2 * H, H, k, s=(H * k) ** -0.5), "b_in": rn(1, Lw, 2 * H, s=0.05),
         "w_rs": rn(1, Lw - 1, 2 * H, H, 1, s=H ** -0.5), "b_rs": rn(1, Lw - 1, 2 * H, s=0.05),
         "w_rs_last": rn(1, H, H, 1, s=H ** -0.5), "b_rs_last": rn(1, H, s=0.05), "w_end": rn(1, C, H, 1, s=0.02), "b_end": rn(1, C, s=0.02)}
    prep = D._Prepared(cfgd, W, need_bwd=False)
    if prep.wn_img is None:
        raise RuntimeError("fused coupling-network kernel not selected (decoder.TUNE['fused_wn'] off?)")
    R = B * (T + 4)
    buf = D._Buffers(cfgd, prep, R, dev)
    mels = rn(B, 80, 2 * T)
    _, rowmask, _ = D.squeeze_rows(cfgd, mels, torch.full((B,), 2 * T, device=dev), out=buf.x[0])
    buf.xmid[0].copy_(buf.x[0])
    acts = buf.acts(0, Lw, rowmask)
    seed = torch.tensor([7], device=dev, dtype=torch.int32)
    dims = D._dims(cfgd, B, T, 0.05, seed, 0)
    Lb = D._L()
    Lb.glowtts_wavenet_fwd.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    keep = (prep, buf, rowmask, seed, W)

    def run(keep=keep):
        _lib.check(Lb.glowtts_wavenet_fwd(ctypes.byref(dims), ctypes.byref(prep.params[0]), ctypes.byref(acts), buf.xmid[0].data_ptr(),
                                          buf.x[1].data_ptr(), 0, 1, _lib.stream()), "glowtts_wavenet_fwd")
    per_row = 2.0 * (80 * H + Lw * H * 2 * H * k + (Lw - 1) * H * 2 * H + H * H + H * C)
    return dict(run=run, flops=B * T * per_row, alg_bytes=B * T * (640 + 320 + Lw * (384 + 768 + 384) + 384 + 640) + (36 * Lw + 2) * 24576,
                kernel="wn_fwd_kernel<drop> (fused coupling network of one flow: Start + 4 x [In_l k=5 + gate + Res_Skip_l] + End + coupling)")


# Launches per training step of the roofline's kernels.  Filled from the library's own launch log (glowtts_launch_count) around the capture pass of
# the timed step - `step_launch_counts()` -; the constants are only what an eager (--no-graph) run reports (12 flows x 4 layers).
CALLS_PER_STEP = {"wn_fwd": 12, "wn_bwd": 0, "in_dgrad": 48, "in_fwd": 0}
CALLS_SOURCE = ["constants (eager run)"]


def launch_log_reset():
    from glow_tts_amd import _lib
    _lib.lib().glowtts_launch_log_reset()


def step_launch_counts(n_flows=12, n_layers=4):
    """Reads the launch log after the step's capture pass (exactly one pass of every captured graph was issued since launch_log_reset())."""
    import ctypes
    from glow_tts_amd import _lib
    L = _lib.lib()
    L.glowtts_launch_count.restype = ctypes.c_int64
    L.glowtts_launch_count.argtypes = [ctypes.c_char_p]
    n = lambda cls: int(L.glowtts_launch_count(cls.encode()))
    wn_bwd = n("wn_bwd<")
    CALLS_PER_STEP.update({"wn_fwd": n("wn_fwd<"), "wn_bwd": wn_bwd, "in_fwd": n("conv_dma<GATE,5>"),
                           "in_dgrad": n_layers * (n_flows - wn_bwd)})          # (the decoder flows that are not on the fused backward: one per layer)
    CALLS_SOURCE[0] = "glowtts_launch_count over the capture pass of the timed step"


def blob_sha(path):
    """git's blob hash of a file (sha1 over "blob <size>\\0" + bytes): the JSON line names the exact committed evidence file a field was read from."""
    import hashlib
    with open(path, "rb") as f:
        data = f.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def in_step_durations():
    """Average launch durations INSIDE the replayed step from the newest committed per-step kernel statistics (profiles/r*_step_kernel_stats.csv, a
    rocprofv3 --kernel-trace of `bench.py --profile-run`): evidence read from a file, named as such - not measured by this run."""
    import csv
    pdir = os.path.join(REPO, "profiles")
    cands = sorted(f for f in os.listdir(pdir) if f.endswith("_step_kernel_stats.csv") and f.startswith("r")) if os.path.isdir(pdir) else []
    if not cands:
        return {}, None
    out = {}
    with open(os.path.join(pdir, cands[-1])) as f:
        for row in csv.DictReader(f):
            nm = row.get("Name", "")
            for key, pat in (("wn_fwd", "wn_fwd_kernel<true, false, true"), ("wn_bwd", "wn_bwd_kernel<true, false"), ("in_dgrad", "conv_dma_kernel<0, 5, 2"),
                             ("in_fwd", "conv_dma_kernel<1, 5, 2")):
                if pat in nm and key not in out:
                    out[key] = (float(row["AverageNs"]) * 1e-3, float(row["CallsPerStep"]))
    return out, "profiles/" + cands[-1] + "@" + blob_sha(os.path.join(pdir, cands[-1]))[:12]


def time_kernel(run, iters=30):
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / iters


def measured_traffic():
    """HBM bytes per launch from the rocprofv3 --pmc passes of tools/pmc_conv.sh (FETCH_SIZE doubled per the gfx950 note of the guide,
    WRITE_SIZE calibrated on a fill of known size in the same pass) - the newest profiles/r*_conv_pmc.json, or nothing."""
    pdir = os.path.join(REPO, "profiles")
    cands = sorted(f for f in os.listdir(pdir) if f.endswith("_conv_pmc.json")) if os.path.isdir(pdir) else []
    if not cands:
        return None, None
    with open(os.path.join(pdir, cands[-1])) as f:
        return json.load(f), "profiles/" + cands[-1] + "@" + blob_sha(os.path.join(pdir, cands[-1]))[:12]


def sustained_mfma_clock():
    """GHz the chip holds under a dense bf16 MFMA loop (glowtts_mfma_clock_probe: 256 workgroups x 4 waves, ~0.4 ms), and the dense bf16
    rate that clock allows: CUs x 4 SIMDs x 32768 FLOP per 32-clk MFMA."""
    import ctypes
    from glow_tts_amd import _lib
    L = _lib.lib()
    L.glowtts_mfma_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    out = torch.zeros(ncu, 2, dtype=torch.int64, device="cuda")
    khz = ctypes.c_int(0)
    for _ in range(2):
        _lib.check(L.glowtts_mfma_clock_probe(out.data_ptr(), ncu, 20000, ctypes.byref(khz), _lib.stream()), "mfma_clock_probe")
    torch.cuda.synchronize()
    o = out.double().sum(0)
    ghz = float(o[0] / o[1]) * khz.value * 1e-6
    return ghz, ncu * 4 * 32768 / 32 * ghz * 1e-3            # GHz, TFLOP/s


def roofline(precision, B, T, step_tflops):
    """The dominant kernel by time in the step (launch duration x launches per step; profiles/*_kernel_stats.csv): `achieved` = algorithmic
    FLOPs per launch (valid rows only: fused_forward_case / 2 x rows x 384 x 192 x 5 for the In_l data gradient) / its launch duration, timed here with HIP events over back-to-back launches on the launch stream (the
    rocprofv3 average inside the running step, where launches are not back to back, is a few percent longer: profiles/)."""
    peak = PEAK_TFLOPS[precision]
    cases = hot_kernel_cases(precision, B, T)
    pmc, src = measured_traffic()
    rows = {}
    for name, c in cases.items():
        sec = time_kernel(c["run"])
        ach = c["flops"] / sec / 1e12
        tr = pmc.get(name, {}).get("traffic") if (pmc and pmc.get("shape") == [B, T] and pmc.get("precision") == precision) else None
        rows[name] = {"kernel": c["kernel"], "achieved": round(ach, 1), "frac": round(ach / peak, 4), "us_per_launch": round(sec * 1e6, 2),
                      "algorithmic_bytes": int(c["alg_bytes"]), "traffic": tr}
    instep, instep_src = in_step_durations()
    for name, r in rows.items():
        r["launches_per_step"] = CALLS_PER_STEP.get(name, 0)
        r["us_per_step"] = round(r["us_per_launch"] * r["launches_per_step"], 1)
        if name in instep and B == 32 and precision == "bf16":
            r["in_step_us_per_launch"] = round(instep[name][0], 2)
            r["frac_in_step"] = round(cases[name]["flops"] / (instep[name][0] * 1e-6) / 1e12 / peak, 4)
    top_name = max(rows, key=lambda k: rows[k]["us_per_step"])        # the dominant kernel BY TIME IN THE STEP
    top = rows[top_name]
    out = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": top["frac"],
           "us_per_launch": top["us_per_launch"], "traffic": top["traffic"], "algorithmic_bytes": top["algorithmic_bytes"],
           "traffic_source": src if top["traffic"] is not None else None,
           "timing": "HIP events on the launch stream, 30 back-to-back launches of the kernel alone",
           "launches_per_step_source": CALLS_SOURCE[0],
           "step_frac": round(step_tflops / peak, 4), "kernels": rows}
    if "frac_in_step" in top:
        # the same kernel's average duration inside the replayed step (both streams busy, lower clock): the smaller, stricter fraction
        out.update(frac_in_step=top["frac_in_step"], in_step_us_per_launch=top["in_step_us_per_launch"], in_step_source=instep_src,
                   evidence_note="traffic / frac_in_step / in_step_us_per_launch are READ from the committed profiles/ files named in traffic_source / in_step_source "
                                 "(path@git-blob-hash), not measured by this run; achieved / frac / us_per_launch are measured here")
    if precision == "bf16":
        # context, not the graded fraction: the clock (and with it the matrix rate) the chip actually sustains under MFMA load
        ghz, sus = sustained_mfma_clock()
        out["sustained_mfma_clock_ghz"] = round(ghz, 3)
        out["peak_sustained"] = round(sus, 1)
        out["frac_of_sustained"] = round(top["achieved"] / sus, 4)
    return out


def mas_us_per_utt(B, Tx, Ty, iters=30, ragged=False):
    """Transposed DP + dense path (what the training step runs), us per utterance.  ragged: Set V lengths (synthetic_batch's)."""
    from glow_tts_amd import alignment
    g = torch.Generator().manual_seed(1234)
    v = (torch.randn(B, Ty, Tx, generator=g) * 30 - 100).cuda()
    if ragged:
        ty = (2 * torch.randint(300, 501, (B,), generator=g).clamp(max=Ty // 2)).clamp(max=Ty)
        tx = torch.round(0.15 * ty).long().clamp(min=1, max=Tx)
        ty[0], tx[0] = Ty, Tx
        tx, ty = tx.cuda(), ty.cuda()
    else:
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


def mas_keys(B, Tx, Ty):
    """MAS timings beside the headline `mas_us_per_utt` (fixed-length Set F at the bench batch): ragged Set V, the latency of a
    single utterance, and the throughput at B = 256; 200 x 1000 is the reference's maximum text / mel size (Hyper_Parameters.yaml:94-99)."""
    return {"set_v_us_per_utt": round(mas_us_per_utt(B, Tx, Ty, ragged=True), 3),
            "latency_us_b1": round(mas_us_per_utt(1, Tx, Ty), 2),
            "b256_us_per_utt": round(mas_us_per_utt(256, Tx, Ty, iters=10), 3),
            "max_size_200x1000_us_per_utt": round(mas_us_per_utt(B, 200, 1000), 3)}


def cpu_baseline():
    """The oracle (CPU restatement of the reference, oracle/glowtts_ref.py + oracle/mas_ref.c / mas_ref.maximum_path_python) timed on this
    host: BASELINE config 1 = Vanilla, B = 8, T_tokens = 120, T_mel = 800, fp32, forward + losses + backward.  Bounded to ~30 s: the intra-op
    thread count is swept on a 2-utterance sample, the full B = 8 step is then timed at the best count; MAS is timed separately in both
    variants of the reference (compiled core.pyx semantics, Python loop of Modules.py:951-980)."""
    import numpy as np
    from oracle import glowtts_ref as O
    from oracle import mas_ref
    import yaml
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS
    logical = os.cpu_count() or 1
    physical = logical
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        pass
    with open(os.path.join(REPO, "glow_tts_amd", "Hyper_Parameters.default.yaml")) as f:
        hpd = yaml.safe_load(f)
    hpd["Mode"] = "Vanilla"
    torch.manual_seed(0)
    model = GlowTTS(Recursive_Parse(hpd))
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cfg = O.Cfg()
    Tt, Tm = 120, 800

    def step(batch):
        tokens, tl, mels, ml = batch
        t0 = time.time()
        out = O.forward_train(sd, cfg, tokens, tl, mels, ml)
        mle, length = O.train_losses(out, ml, cfg)
        for v in sd.values():
            v.grad = None
        (mle + length).backward()
        return time.time() - t0, out

    small = synthetic_batch(2, Tt, Tm, 80, 1234, "cpu")
    full = synthetic_batch(8, Tt, Tm, 80, 1234, "cpu")
    cands = sorted({n for n in (8, 16, 32, physical) if n <= physical} or {physical})
    torch.set_num_threads(cands[0])
    step(small)                                        # warm-up (allocator, oneDNN primitives)
    sweep = {}
    for n in cands:
        torch.set_num_threads(n)
        sweep[n] = min(step(small)[0], step(small)[0])
    best_n = min(sweep, key=sweep.get)

    def timed(n_threads, reps=5):                      # >= 5 timed steps after a warm-up, median
        torch.set_num_threads(n_threads)
        step(full)
        ts, out = [], None
        for _ in range(reps):
            t, out = step(full)
            ts.append(t)
        return statistics.median(ts), min(ts), max(ts), out
    med, lo, hi, out = timed(best_n)
    at8 = timed(8) if best_n != 8 and physical >= 8 else (med, lo, hi, out)
    # MAS alone on the same batch's log-prior matrix: C restatement of core.pyx (kernel only; with the wrapper's host copies), Python loop
    logp = out["logp"].detach()
    tmask, mmask = O.mask_from_lengths(full[1], Tt), O.mask_from_lengths(full[3], Tm)
    amask = (tmask.unsqueeze(-1) * mmask.unsqueeze(2)).squeeze(1)
    v = (logp * amask).numpy().astype(np.float32)
    tx, ty = full[1].numpy().astype(np.int32), full[3].numpy().astype(np.int32)
    t0 = time.time(); mas_ref.maximum_path_c(v, tx, ty); c_kernel = time.time() - t0
    t0 = time.time(); O.mas(logp, amask); c_wrapped = time.time() - t0
    t0 = time.time(); mas_ref.maximum_path_python(v[:2], tx[:2], ty[:2]); py_utt = (time.time() - t0) / 2
    # config 1 as BASELINE.json states it ("Python MAS"): the same step with the Python-loop search of Modules.py:951-980, TIMED (3 steps)
    torch.set_num_threads(best_n)
    c_mas = O.mas

    def python_mas(value, mask, max_neg_val=-1e9):
        vv = (value * mask).detach().numpy().astype(np.float32)
        txs = mask.sum(1)[:, 0].numpy().astype(np.int32)
        tys = mask.sum(2)[:, 0].numpy().astype(np.int32)
        return torch.from_numpy(np.ascontiguousarray(mas_ref.maximum_path_python(vv, txs, tys))).to(value.dtype)
    py_steps = []
    try:
        O.mas = python_mas
        for _ in range(3):
            py_steps.append(step(full)[0])
    finally:
        O.mas = c_mas
    py_step = statistics.median(py_steps)
    return {"value": round(8 * Tm / med, 1), "unit": "mel-frames/s", "cores": best_n, "kind": "port",
            "sample": f"oracle (torch fp32 + C MAS = core.pyx semantics), Vanilla B=8 T_mel=800 T_tok=120, fwd+losses+bwd, MEDIAN of 5 steps after 1 "
                      f"warm-up at {best_n} intra-op threads ({med:.3f} s/step, min {lo:.3f}, max {hi:.3f}); thread sweep on a B=2 sample, s/step: "
                      + ", ".join(f"{n}: {t:.2f}" for n, t in sweep.items()) + f"; host has {physical} physical / {logical} logical cores",
            "value_at_8_threads": round(8 * Tm / at8[0], 1), "s_per_step_at_8_threads": round(at8[0], 3),
            "value_python_mas": round(8 * Tm / py_step, 1), "s_per_step_python_mas": round(py_step, 3),
            "python_mas_sample": "the same step with Modules.py:951-980's Python-loop search, median of 3 timed steps",
            "mas_us_per_utt": {"c_kernel": round(c_kernel / 8 * 1e6, 1), "c_with_wrapper_copies": round(c_wrapped / 8 * 1e6, 1),
                               "python_loop": round(py_utt * 1e6, 1)}}


def parity_sample(model, cfg, batch, cond, n=4):
    """Part of the cpu_baseline leg (the only place bench.py touches the oracle; the checker, never the thing measured): the benchmarked model - its
    CURRENT weights, eval mode (no dropout draw), the arithmetic mode of the timed step - on the first `n` utterances of the timed batch against
    oracle/glowtts_ref.py on the same state dict: the fraction of valid frames whose aligned token differs from the fp32 oracle's path, max |z - oracle|
    over valid frames, |NLL - oracle|.  In `HIP_Precision: f32` the alignment is bit-exact (tests/test_gpu_benchmarked_sizes.py); in bf16 - the mode
    BASELINE config 2 names - near-tied paths of a barely trained model move a few percent of the frames."""
    from oracle import glowtts_ref as O
    from glow_tts_amd.modules import MLE_Loss
    tokens, tl, mels, ml = (t[:n] for t in batch)
    spk = cond[0][:n] if cond[0] is not None else (cond[1][:n] if cond[1] is not None else None)
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            z, mm, ms, ld, _, _, attn, _ = model(tokens, tl, mels, ml, None if cond[0] is None else cond[0][:n], None if cond[1] is None else cond[1][:n], None)
            nll = MLE_Loss(model.hp)(z=z, mean=mm, std=ms, log_dets=ld, lengths=ml)
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    finally:
        model.train(was_training)
    hpd = {"Mode": model.hp.Mode, "Speaker_Embedding": {"Type": model.hp.Speaker_Embedding.Type}}
    import yaml
    with open(os.path.join(REPO, "glow_tts_amd", "Hyper_Parameters.default.yaml")) as f:
        full = yaml.safe_load(f)
    full["Mode"] = hpd["Mode"]
    full["Speaker_Embedding"]["Type"] = hpd["Speaker_Embedding"]["Type"]
    ocfg = O.Cfg.from_yaml_dict(full)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        o = O.forward_train(sd, ocfg, tokens.cpu(), tl.cpu(), mels.cpu(), ml.cpu(), None if spk is None else spk.cpu())
        onll, _ = O.train_losses(o, ml.cpu(), ocfg)
    mmask = O.mask_from_lengths(ml.cpu(), mels.shape[2])
    a, oa = attn.cpu(), o["attn"]
    differ = ((a[:, :, :oa.shape[2]] != oa).any(1).float() * mmask[:, 0, :oa.shape[2]]).sum().item() / max(1.0, mmask.sum().item())
    # the moved frames in the oracle's own terms: the total score of this path on the oracle's fp32 score matrix against the oracle's optimum, per frame and relative
    # to the mean score magnitude (tests/test_gpu_benchmarked_sizes.py check_bf16: <= 1e-3 is the bar there) - near-ties move frames without losing score
    logp = o["logp"]
    frames = oa.sum((1, 2)).clamp_min(1)
    spread = (logp * oa).abs().sum((1, 2)) / frames
    deficit = (((logp * oa).sum((1, 2)) - (logp * a[:, :, :logp.shape[2]]).sum((1, 2))) / frames / spread.clamp_min(1e-6)).max().item()
    return {"utterances": int(n), "frames_aligned_differently": round(differ, 5), "path_score_deficit": float(f"{deficit:.3e}"),
            "z_max_err": round(float(((z.cpu() - o["z"]) * mmask).abs().max()), 6), "nll_abs_err": round(abs(float(nll) - float(onll)), 7),
            "note": "benchmarked model (current weights, eval mode, this run's arithmetic mode) vs oracle/glowtts_ref.py (fp32 CPU) on the first utterances "
                    "of the timed batch; alignment bit-exactness is a property of the MAS operator on identical fp32 scores and of the f32 mode"}


def f32_key(args):
    """north_star states its tolerance in fp32: the same step in `HIP_Precision: f32` (exact fp32 MFMA, v_mfma_f32_32x32x2_f32, the arithmetic the
    1e-3 / 1e-4 parity tests run in), timed by a child process so that its model and graph do not share this one's memory."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--precision", "f32", "--steps", "5", "--warmup", "3", "--windows", "0", "--no-cpu-baseline",
           "--no-f32-key", "--config", str(args.config)] + (["--batch", str(args.batch)] if args.batch else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return {"ms_per_step": d["ms_per_step"], "value": d["value"], "model_tflops": d["model_tflops"],
                "step_frac": round(d["model_tflops"] / PEAK_TFLOPS["f32"], 4), "peak": PEAK_TFLOPS["f32"],
                "note": "whole Train_Step in exact-fp32 MFMA arithmetic; frac against the 157.3 TFLOP/s fp32 matrix peak"}
    except Exception as exc:                                    # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"}


def inverse_flow_leg(model, hp, dev, B, seed):
    """BASELINE config 5's long-form inference: GlowTTS.inference (Modules.py:128-204) on 200-token utterances stretched to ~2000 mel
    frames each (length_scale), replayed through GraphedInference; PE mode takes the prosody reference mels."""
    from glow_tts_amd.graph_infer import GraphedInference
    model.eval()
    Tt = 200
    tokens, tl, ref_mels, ref_ml = synthetic_batch(B, Tt, 800, 80, seed, dev)
    gi = GraphedInference(model, mel_buckets=(1024, 2048, 2560))
    pe = dict(mels_for_prosody=ref_mels, mel_lengths_for_prosody=ref_ml) if "Prosody_Encoder" in model.layer_Dict else {}
    # per-utterance length scales that stretch whatever the duration predictor says at this point of the (synthetic) training to ~2000
    # frames: the leg measures the inverse flow at a fixed long-form size, not the state of the duration predictor
    with torch.no_grad():
        pred = model.inference_front(tokens, tl, pe.get("mels_for_prosody"), pe.get("mel_lengths_for_prosody"), None, None, 1.0)[3]
    kw = dict(noise_scale=0.667, length_scale=(2000.0 / pred.clamp(min=1).float()).to(dev), **pe)
    for _ in range(3):
        mels, lengths, _ = gi(tokens, tl, **kw)
    torch.cuda.synchronize()
    n = 10
    t0 = time.time()
    for _ in range(n):
        mels, lengths, _ = gi(tokens, tl, **kw)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    frames = int(lengths.sum().item())
    model.train()
    return {"metric": "mel-frames/sec (inverse flow, GlowTTS.inference)", "value": round(frames / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
            "utterances": B, "tokens": Tt, "mel_frames_per_utterance": round(frames / B, 1), "launch_mode": "hipgraph (2 graphs per call)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config (default 2: the one the metric is quoted on)")
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default: the config's per-GPU share)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--ragged", action="store_true", help="Set V (ragged lengths) instead of Set F (fixed)")
    ap.add_argument("--windows", type=int, default=10, help="extra timed windows of --steps steps after the reported one (median / spread keys)")
    ap.add_argument("--tokens", type=int, default=120, help="padded token length (experiments)")
    ap.add_argument("--frames", type=int, default=800, help="padded mel length (experiments; the metric is quoted on 800)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-run", action="store_true", help="for rocprofv3 --kernel-trace: skip the MAS and roofline legs (their launches would be counted "
                    "into the per-step kernel statistics); use with --windows 0 --no-cpu-baseline --no-f32-key")
    ap.add_argument("--no-f32-key", action="store_true", help="skip the extra `f32` key (the same step in HIP_Precision f32, timed in a child process)")
    ap.add_argument("--no-optimizer", action="store_true", help="time forward + losses + backward only (round 1's definition of the step); "
                    "default: the whole Train_Step of Train.py:193-233 including clip_grad_norm_, RAdam and the Noam schedule")
    ap.add_argument("--timeline", action="store_true", help="diagnostics: stamp kernels inside the captured step (decoder flows, encoder milestones); "
                    "prints when each stream reached them in one replay (stderr) - adds ~40 tiny launches to the step")
    ap.add_argument("--sync-each", action="store_true", help="experiment: synchronise after every step (no replay is enqueued ahead) and report the host time of a replay call")
    ap.add_argument("--graph-execs", type=int, default=1, help="experiment: capture the step n times and replay the instances round-robin")
    ap.add_argument("--tune", action="append", default=[], help="A/B measurements: key=value entries of glow_tts_amd.decoder.TUNE (e.g. wgrad_wide=0)")
    ap.add_argument("--one-device", action="store_true", help="all ranks on device 0 (multi-rank smoke test on a single-GPU box)")
    ap.add_argument("--no-overlap", action="store_true", help="data parallel: one graph + one gradient exchange instead of the two-graph overlap")
    ap.add_argument("--dp-skip-reduce", action="store_true", help="diagnosis (WRONG results for N > 1): the data-parallel step's graphs without the gradient "
                    "exchange between them - what splitting the step into graphs costs by itself")
    ap.add_argument("--force-dist", action="store_true", help="run the data-parallel code path (process group, two-graph overlap, all-reduces) "
                    "even with one rank: exercises RCCL on a single-GPU box")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to smoke-test the "
                    "multi-rank code path on a single-GPU box together with --one-device)")
    ap.add_argument("--no-defer-uploads", action="store_true", help="A/B: job tables as memcpy nodes inside the captured graphs (rounds 2-5) instead of one eager upload after the capture")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch every kernel eagerly instead of replaying the "
                    "captured hipGraph of the step (default: graph replay; at B = 32 the eager step is bound by host launch work)")
    ap.set_defaults(graph=True)
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: become the launcher - one rank per GPU under torch.distributed.run (RCCL over xGMI),
        # rendezvous on 127.0.0.1; rank 0's JSON line is this process's last line of output.  Refuses when the node has fewer devices.
        import subprocess
        if not args.one_device and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node has {torch.cuda.device_count()} visible GPU(s)")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (args.gpus == 1 and world == 1):
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: the two must agree (n_gpus in the JSON line is the "
                         "number of ranks the process group really has)")
    if not args.one_device and world > torch.cuda.device_count():
        raise SystemExit(f"WORLD_SIZE={world} but only {torch.cuda.device_count()} visible GPU(s) (use --one-device for a single-GPU smoke test)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dp = world > 1 or args.force_dist
    if dp:
        import torch.distributed as dist
        from glow_tts_amd import distributed as _gd
        _gd.SINGLE_RANK_IS_DIST = args.force_dist
        if args.force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    from glow_tts_amd import _lib, decoder as _dec
    _lib.DEFER_UPLOADS = not args.no_defer_uploads
    from glow_tts_amd import conv_fn as _cf, ops as _ops
    for kv in args.tune:                                      # decoder.TUNE, or the encoder's block-function switches (conv_fn.FUSE)
        k, v = kv.split("=")
        from glow_tts_amd import alignment as _al
        d = _dec.TUNE if k in _dec.TUNE else _cf.FUSE if k in _cf.FUSE else _al.FUSED
        d[k] = type(d[k])(int(v))
    if args.timeline:
        _dec.STAMPS["buf"] = torch.zeros(4096, dtype=torch.int64, device=dev)
    from glow_tts_amd.distributed import FlatGradReducer, actnorm_stats_allreduce
    model, mle_loss, hp = build_model(args.precision, dev, cfg["mode"], cfg["spk_type"])
    reducer = None
    if dp:
        import torch.distributed as dist
        from glow_tts_amd.distributed import broadcast_parameters
        broadcast_parameters(model)                          # identical replicas
        model.actnorm_allreduce = actnorm_stats_allreduce
        reducer = FlatGradReducer(list(model.parameters()))
    opt = None
    if not args.no_optimizer:                                   # Train.py:88-100 (Model_Generate): RAdam + Modified_Noam_Scheduler, hyper-parameters of the yaml
        from glow_tts_amd.optim import Modified_Noam_Scheduler, RAdam
        optimizer = RAdam(model.parameters(), lr=hp.Train.Learning_Rate.Initial, betas=(hp.Train.ADAM.Beta1, hp.Train.ADAM.Beta2),
                          eps=hp.Train.ADAM.Epsilon, weight_decay=hp.Train.Weight_Decay)
        opt = (optimizer, Modified_Noam_Scheduler(optimizer, base=hp.Train.Learning_Rate.Base), hp.Train.Gradient_Norm)
    torch.manual_seed(4321 + rank)                              # dropout streams differ per rank; the replicas' weights do not
    B, Tt, Tm = args.batch or cfg["batch"], args.tokens, args.frames
    batch = synthetic_batch(B, Tt, Tm, 80, 1234 + rank, dev, ragged=args.ragged)
    cond = conditioning_inputs(cfg, B, 1234 + rank, dev, hp)

    def barrier():
        if dp:
            _gd.barrier()                                       # on the process group's own stream (glow_tts_amd.distributed: collectives and capture)
        torch.cuda.synchronize()

    # A step = forward + losses + backward as ONE captured hipGraph (static shapes; the dropout seed is re-drawn on the
    # device inside the graph), followed - when data parallel - by the flat-bucket gradient all-reduce.
    from glow_tts_amd.distributed import global_frame_weight
    mode = "eager"
    graph = tail_graph = opt_graph = early = tail = None
    extra_graphs, rr = [], [0]
    keep = []                                                   # pinned job tables owned by the captured graphs
    side = torch.cuda.Stream() if args.graph else None
    if args.graph:
        # every eager step that precedes the capture runs on the side stream too: a backward that ran on the default stream
        # first makes the captured AccumulateGrad nodes hop streams, which crashes hipStreamEndCapture on ROCm 7.2
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, args.warmup - 1)):
                loss = train_step(model, mle_loss, batch, cond, reducer, world, opt)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    else:
        for _ in range(max(1, args.warmup - 1)):               # also runs the ActNorm data-dependent init
            loss = train_step(model, mle_loss, batch, cond, reducer, world, opt)
    if args.graph:
        try:
            wfr = global_frame_weight(batch[3].sum()) if dp else None      # constant for a fixed batch

            params = [p for p in model.parameters() if p.requires_grad]

            def clip_and_update():                              # Train.py:228-232; the clip coefficient stays on the device
                from glow_tts_amd.optim import grad_norm_and_coef
                _, coef = grad_norm_and_coef(params, opt[2])
                opt[0].step(grad_scale=coef)

            inv_world = torch.full((), 1.0 / world, device=batch[2].device) if dp else None

            def fwd_bwd():
                from glow_tts_amd import alignment
                alignment.SEEDS["mle"], alignment.SEEDS["rest"] = (wfr, inv_world) if dp else (None, None)
                try:
                    mle, length = forward_losses(model, mle_loss, batch, cond)
                finally:
                    alignment.SEEDS["mle"] = alignment.SEEDS["rest"] = None
                model.zero_grad(set_to_none=True)
                alignment.LossTerms([mle, length], [wfr, inv_world] if dp else None).backward()      # (as Trainer._losses: no sum node in front of the backward)
                if opt is not None and not dp:
                    clip_and_update()
                return (mle + length).detach()

            def uncount_capture_pass():                         # a capture pass advances the optimizer's step counters without running a kernel
                for p in params:
                    st = opt[0].state.get(p)
                    if p.grad is not None and st is not None and "step" in st:
                        st["step"] -= 1

            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    fwd_bwd()
                    if opt is not None and not dp:
                        opt[1].step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if dp:
                _gd.before_capture()                            # the watchdog holds no Work when the captures begin (glow_tts_amd.distributed)
            launch_log_reset()
            graph = torch.cuda.CUDAGraph()
            with _lib.pinned_sink(keep):
                if dp and not args.no_overlap:
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
                    early = FlatGradReducer([p for p in model.parameters() if id(p) not in tail_ids], static=True)
                    tail = FlatGradReducer([p for p in model.parameters() if id(p) in tail_ids], static=True)
                else:
                    with torch.cuda.graph(graph, capture_error_mode="thread_local" if dp else "global"):
                        static_loss = fwd_bwd()
                    if opt is not None and not dp:
                        uncount_capture_pass()
                    for _ in range(max(0, args.graph_execs - 1) if not dp else 0):
                        gx = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gx, capture_error_mode="global"):
                            fwd_bwd()
                        if opt is not None:
                            uncount_capture_pass()
                        extra_graphs.append(gx)
                if opt is not None and dp:
                    # data parallel: the update reads the REDUCED gradients, so it is its own graph behind the exchange
                    opt_graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(opt_graph, pool=graph.pool(), capture_error_mode="thread_local"):
                        clip_and_update()
                    uncount_capture_pass()
            if not extra_graphs:
                step_launch_counts()                            # one pass of every graph of the step has been issued since the reset
            if opt is not None and not dp:
                opt[0].advance_host()
            graph.replay()
            if opt is not None and not dp:
                opt[1].step()
            if tail_graph is not None:
                tail_graph.replay()
            torch.cuda.synchronize()
            mode = "hipgraph"
        except Exception as exc:                               # noqa: BLE001 - fall back to eager launches, say so
            import traceback
            traceback.print_exc()
            print(f"[bench] graph capture failed ({type(exc).__name__}); running eagerly", file=sys.stderr)
            graph = tail_graph = opt_graph = None
            torch.cuda.synchronize()

    def one_step():
        if graph is not None:
            if opt is not None and not dp:
                opt[0].advance_host()                           # this step's hyper-parameter words, stream-ordered before the replay
            if extra_graphs:
                rr[0] = (rr[0] + 1) % (len(extra_graphs) + 1)
                (graph if rr[0] == 0 else extra_graphs[rr[0] - 1]).replay()
            else:
                graph.replay()
            if tail_graph is not None:
                pending = early.begin() if not args.dp_skip_reduce else None
                tail_graph.replay()
                if not args.dp_skip_reduce:
                    tail.reduce(average=False)
                    early.finish(pending)
            elif reducer is not None and not args.dp_skip_reduce:
                reducer.reduce(average=False)
            if opt_graph is not None:
                opt[0].advance_host()
                opt_graph.replay()
            if opt is not None:
                opt[1].step()
            return static_loss
        return train_step(model, mle_loss, batch, cond, reducer, world, opt)

    host_replay = []
    own = [0.0]                                                 # this rank's own time of the last window (before the MAX over ranks)

    def timed_window():
        barrier()
        t0 = time.time()
        for _ in range(args.steps):
            if args.sync_each:
                h0 = time.time()
                out = one_step()
                host_replay.append(time.time() - h0)
                torch.cuda.synchronize()
            else:
                out = one_step()
        barrier()
        el = time.time() - t0
        own[0] = el
        if dp:
            import torch.distributed as dist
            t = torch.tensor([el], device=dev)
            _gd._collective(dist.all_reduce, t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    one_step()
    elapsed, loss = timed_window()                              # the reported window: exactly --steps steps
    own_elapsed = own[0]
    if args.sync_each and rank == 0:
        print(f"[sync-each] host time of one step's enqueue (replay call): median {sorted(host_replay)[len(host_replay) // 2] * 1e6:.0f} us", file=sys.stderr)
    extra = [timed_window()[0] for _ in range(max(0, args.windows))]
    if args.timeline and rank == 0:
        # the captured pass stamped last: its slots are the last occurrence of every name; one more replay fills them
        # (four replays enqueued back to back: the stamps are those of the LAST one, whose launches were enqueued while its predecessors ran -
        # a single replay on an idle GPU exposes the host's enqueue order: the branch enqueued second then starts ~200 us late)
        torch.cuda.synchronize()
        for _ in range(4):
            one_step()
        torch.cuda.synchronize()
        names, buf = _dec.STAMPS["names"], _dec.STAMPS["buf"].cpu()
        last = {n: i for i, n in enumerate(names)}
        ev = sorted(((int(buf[i]), n) for n, i in last.items() if int(buf[i]) > 0))
        t0 = ev[0][0]
        print("[timeline] us since the first stamp of one replayed step (wall counter, 10 ns ticks):", file=sys.stderr)
        prev = t0
        for t, n in ev:
            print(f"[timeline] {(t - t0) / 100.0:9.1f}  (+{(t - prev) / 100.0:7.1f})  {n}", file=sys.stderr)
            prev = t
    if dp:
        # every gradient must have gone through the exchange: reduced gradients are identical on all ranks, unreduced ones are not
        # (different utterances and dropout streams per rank)
        import torch.distributed as dist
        cs = torch.stack([p.grad.double().sum() for p in model.parameters() if p.grad is not None])
        lo, hi = cs.clone(), cs.clone()
        _gd._collective(dist.all_reduce, lo, op=dist.ReduceOp.MIN)
        _gd._collective(dist.all_reduce, hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit(f"[bench] rank {rank}: {int((lo != hi).sum())} gradient tensors differ between ranks after the all-reduce")
        if opt is not None:                                     # ... and the replicas must still hold the same weights after all those updates
            ps = torch.stack([p.detach().double().sum() for p in model.parameters()])
            lo, hi = ps.clone(), ps.clone()
            _gd._collective(dist.all_reduce, lo, op=dist.ReduceOp.MIN)
            _gd._collective(dist.all_reduce, hi, op=dist.ReduceOp.MAX)
            if not torch.equal(lo, hi):
                names = [k for (k, _), bad in zip(model.named_parameters(), (lo != hi).tolist()) if bad]
                raise SystemExit(f"[bench] rank {rank}: {len(names)} parameter tensors differ between ranks after the optimizer steps, e.g. {names[:4]} .. {names[-2:]}; "
                                 f"largest relative difference of the sums {((hi - lo).abs() / (lo.abs() + 1e-9)).max().item():.2e}")
    fwd_bwd_only = None
    if opt is not None and not dp and graph is not None and args.windows > 0:      # (--windows 0 = profiling runs: every traced step is a full Train_Step)
        # round 1's definition of the step (forward + losses + backward, no update), for continuity: a second graph of the same model
        try:
            def fwd_bwd_noopt():
                from glow_tts_amd import alignment
                mle, length = forward_losses(model, mle_loss, batch, cond)
                model.zero_grad(set_to_none=True)
                alignment.LossTerms([mle, length]).backward()
                return (mle + length).detach()
            g2, keep2 = torch.cuda.CUDAGraph(), []
            with torch.cuda.stream(side):
                fwd_bwd_noopt()
            torch.cuda.synchronize()
            with _lib.pinned_sink(keep2):
                with torch.cuda.graph(g2):
                    fwd_bwd_noopt()
            keep.extend(keep2)
            g2.replay()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(args.steps):
                    g2.replay()
                torch.cuda.synchronize()
                ts.append((time.time() - t0) / args.steps)
            fwd_bwd_only = statistics.median(ts)
        except Exception as exc:                               # noqa: BLE001 - an extra, never fatal
            print(f"[bench] forward+backward-only leg skipped ({type(exc).__name__}: {exc})", file=sys.stderr)
    frames = int(batch[3].sum().item())
    dist_check = None
    if dp:
        import torch.distributed as dist
        ft = torch.tensor([frames], device=dev, dtype=torch.float64)
        _gd._collective(dist.all_reduce, ft)
        frames = int(ft.item())
        # self-check of the first real multi-GPU run: how many ranks the process group REALLY has (an all-reduce of ones), and the spread of
        # the per-rank step time over the reported window (`elapsed` above is the MAX, as the contract asks)
        ones = torch.ones(1, device=dev)
        _gd._collective(dist.all_reduce, ones)
        mine = torch.tensor([own_elapsed], device=dev, dtype=torch.float64)
        lo_t, hi_t = mine.clone(), mine.clone()
        _gd._collective(dist.all_reduce, lo_t, op=dist.ReduceOp.MIN)
        _gd._collective(dist.all_reduce, hi_t, op=dist.ReduceOp.MAX)
        dist_check = {"rccl_ranks_seen": int(ones.item()), "world_size": dist.get_world_size(), "backend": dist.get_backend(),
                      "ms_per_step_rank_min": round(1e3 * float(lo_t) / args.steps, 3), "ms_per_step_rank_max": round(1e3 * float(hi_t) / args.steps, 3)}
    inv = inverse_flow_leg(model, hp, dev, B, 99 + rank) if (args.config == 5 and rank == 0 and not args.profile_run) else None      # (kernel-trace runs: the step's launches only)
    if rank == 0:
        value = frames * args.steps / elapsed
        ms = [1e3 * e / args.steps for e in [elapsed] + extra]
        tflops = value * FLOP_PER_FRAME_FWD_BWD / 1e12
        out = {
            "metric": "mel-frames/sec (train fwd+bwd)", "value": round(value, 1), "unit": "mel-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{cfg['name']}, LJSpeech-shaped synthetic (80-mel, {Tm} frames, {Tt} tokens), "
                                   f"batch={B}/GPU, {'ragged Set V' if args.ragged else 'fixed Set F'}, "
                                   + ("forward+losses+backward" if opt is None else "Train_Step = forward+losses+backward+clip_grad_norm+RAdam+Noam schedule")
                                   + (f", {'RCCL' if args.backend == 'nccl' else args.backend} grad all-reduce" if dp else ""),
                       "baseline_config": args.config, "mode": cfg["mode"] + ("/" + cfg["spk_type"] if cfg["mode"] == "SE" else ""),
                       "global_batch": B * world, "mel_frames": Tm, "tokens": Tt, "parallelism": f"dp{world}"},
            "loss": round(float(loss.item()), 4), "launch_mode": mode, "optimizer_in_step": opt is not None,
            "windows": {"n": len(ms), "steps_each": args.steps, "ms_per_step_median": round(statistics.median(ms), 3),
                        "ms_per_step_min": round(min(ms), 3), "ms_per_step_max": round(max(ms), 3)},
            "model_tflops": round(tflops, 2),
        }
        if not args.profile_run:                              # (kernel-trace runs: only the step's own launches, so calls per step come out as integers)
            out["mas_us_per_utt"] = round(mas_us_per_utt(B, Tt, Tm), 3)
            out["mas"] = mas_keys(B, Tt, Tm)
            out["roofline"] = roofline(args.precision, B, Tm // 2, tflops / world)
        if fwd_bwd_only is not None:
            out["fwd_bwd_only"] = {"ms_per_step": round(1e3 * fwd_bwd_only, 3), "value": round(frames / fwd_bwd_only, 1),
                                   "note": "forward + losses + backward without the parameter update (the step round 1 reported)"}
        if inv is not None:
            out["inverse_flow"] = inv
        if dist_check is not None:
            out["distributed"] = dist_check
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            try:
                out["parity_sample"] = parity_sample(model, cfg, batch, cond)
            except Exception as exc:                            # noqa: BLE001 - a reported extra, never fatal for the timing line
                out["parity_sample"] = {"error": f"{type(exc).__name__}: {exc}"}
        if args.precision == "bf16" and not dp and not args.no_f32_key and args.config == 2 and opt is not None:
            out["f32"] = f32_key(args)
    # The JSON line must be the LAST line of the job's stdout.  RCCL writes a banner ("Librccl path : ...") through C stdio, which a pipe
    # buffers until exit, i.e. behind Python's own output: every rank flushes its C buffers, the ranks meet once more, then rank 0 prints.
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:                                        # noqa: BLE001
        pass
    sys.stdout.flush()
    if dp:
        import torch.distributed as dist
        _gd.barrier()
        dist.destroy_process_group()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:                                    # noqa: BLE001
            pass
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
