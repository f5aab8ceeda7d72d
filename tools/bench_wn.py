"""Times ONE flow's coupling network alone at the bench shape (B = 32, 800 frames -> 12 928 rows): the fused kernel (wn_fwd) against the ten
per-conv launches, forward with kept activations.  With GLOWTTS_LIB_PATH = tools build, GLOWTTS_WN_ABL selects a timing ablation of the fused
kernel (1: no weight DMAs, 2: no MFMAs, 4: no kept stores; sums combine)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import full_width_state
from glow_tts_amd import decoder as D, _lib

B, TM = int(os.environ.get("WN_B", "32")), int(os.environ.get("WN_TM", "800"))
DROP = float(os.environ.get("WN_DROP", "0.05"))
g = torch.Generator().manual_seed(0)
cfg, sd = full_width_state(1, g)
dc = D.DecoderConfig(cfg.mel_dim, 1, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, 1)
P = {k: v.cuda() for k, v in sd.items()}
W = dict(zip(D.WEIGHT_KEYS, [w.contiguous() for w in D.stack_decoder_weights(P, dc)]))
mels = torch.randn(B, 80, TM, generator=g).cuda()
ml = torch.full((B,), TM).cuda()
seed = torch.tensor([7], device="cuda", dtype=torch.int32)
L = D._L()
res = {}
for fused in (True, False):
    D.TUNE["fused_wn"] = fused
    with torch.no_grad():
        prep = D._Prepared(dc, W, need_bwd=False)
        R = B * (TM // 2 + 4)
        buf = D._Buffers(dc, prep, R, "cuda")
        _, rowmask, T = D.squeeze_rows(dc, mels, ml, out=buf.x[0])
        acts = buf.acts(0, dc.L, rowmask)
        tl = torch.zeros(512 * 12 * 32, dtype=torch.int64, device="cuda")
        if os.environ.get("GLOWTTS_WN_ABL") in ("16", "32", "96", "160", "288") and fused:
            acts.skip_bf = tl.data_ptr()
        dims = D._dims(dc, B, T, DROP, seed if DROP > 0 else None, 0)
        run = lambda: _lib.check(L.glowtts_flow_forward(ctypes.byref(dims), ctypes.byref(prep.params[0]), ctypes.byref(acts), _lib.stream()), "flow_forward")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        n = int(os.environ.get("ITERS", "30"))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            with torch.cuda.graph(gr, stream=st):
                for _ in range(n):
                    run()
            gr.replay()
            e0.record(st)
            gr.replay()
            e1.record(st)
        torch.cuda.synchronize()
        fwd_us = e0.elapsed_time(e1) * 1e3 / n
        if fused:
            tl_fused = tl
        # ---- backward of the same flow (data gradients + ActNorm / 1x1 backward; weight gradients deferred as in the training step) ----
        D.TUNE["fused_wn_bwd"] = bool(fused)
        prepb = D._Prepared(dc, W, need_bwd=True)
        ldo, ldin, H, C = prepb.ldo, prepb.ldin, dc.H, dc.C
        dx = torch.randn(R, C, device="cuda") * 0.1
        gb = D.FlowGrads()
        douts = torch.zeros(R, ldo, device="cuda"); douts_bf = torch.zeros(R, ldo, device="cuda", dtype=torch.bfloat16)
        dins = torch.zeros(dc.L, R, ldin, device="cuda", dtype=torch.bfloat16); dskip = torch.empty(R, H, device="cuda", dtype=torch.bfloat16)
        dh0 = torch.empty(R, H, device="cuda"); dhn = torch.empty(dc.L, R, H, device="cuda", dtype=torch.bfloat16)
        scratch = torch.empty(L.glowtts_actnorm_stats_scratch_floats(R, C), device="cuda"); dld = torch.zeros(B, device="cuda")
        gb.dx, gb.dlogdet, gb.douts, gb.dskip, gb.douts_bf = dx.data_ptr(), dld.data_ptr(), douts.data_ptr(), dskip.data_ptr(), douts_bf.data_ptr()
        gb.scratch, gb.defer_wgrad = scratch.data_ptr(), 1
        tlb = torch.zeros(512 * 64, dtype=torch.int64, device="cuda")
        if fused and int(os.environ.get("GLOWTTS_WN_BWD_ABL", "0")) & 64:
            gb.dcond = tlb.data_ptr()
        for l in range(dc.L):
            gb.dh[l] = dh0.data_ptr() if l == 0 else dhn[l].data_ptr()
            gb.dins[l] = dins[l].data_ptr()
        runb = lambda: _lib.check(L.glowtts_flow_backward(ctypes.byref(dims), ctypes.byref(prepb.params[0]), ctypes.byref(acts), ctypes.byref(gb), _lib.stream()), "flow_backward")
        for _ in range(3):
            runb()
        torch.cuda.synchronize()
        gr2 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(gr2, stream=st):
                for _ in range(n):
                    runb()
            gr2.replay()
            e0.record(st)
            gr2.replay()
            e1.record(st)
        torch.cuda.synchronize()
        res[("fused" if fused else "per-conv") + " bwd"] = e0.elapsed_time(e1) * 1e3 / n
        if fused:
            tlb_fused = tlb
        if os.environ.get("COLD") == "1":
            # the same with the caches flushed between launches (in the training step a flow's kept activations were written milliseconds
            # and > 1 GB of traffic earlier: they come from HBM, not from the 256 MB MALL that a back-to-back loop over ONE flow enjoys)
            flush = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
            def timed(body):
                gg = torch.cuda.CUDAGraph()
                with torch.cuda.stream(st):
                    with torch.cuda.graph(gg, stream=st):
                        for i in range(n):
                            flush.fill_(i & 1)
                            body()
                    gg.replay()
                    e0.record(st)
                    gg.replay()
                    e1.record(st)
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / n
            base = timed(lambda: None)
            res[("fused" if fused else "per-conv") + " bwd COLD"] = timed(runb) - base
            res[("fused" if fused else "per-conv") + " fwd COLD"] = timed(run) - base
        res["fused" if fused else "per-conv"] = fwd_us
if os.environ.get("GLOWTTS_WN_ABL") in ("32", "96", "160", "288"):
    t = tl_fused.view(512, 12, 32)[:200, :, :18].cpu()
    t = (t - t[:, :1, :1]).double().view(200, 12, 6, 3)      # [wg][wave][step][before wait, after wait, after barrier]
    print("In_1, tap 2, steps kc = 0..5; clocks, median over 200 workgroups")
    for w in (0, 3, 6, 11):
        iss = (t[:, w, 1:, 0] - t[:, w, :-1, 2]).median().item()
        wait = (t[:, w, :, 1] - t[:, w, :, 0]).median().item()
        bar = (t[:, w, :, 2] - t[:, w, :, 1]).median().item()
        step = (t[:, w, 1:, 2] - t[:, w, :-1, 2]).median().item()
        print(f"  wave {w:2d}: barrier -> before wait (issue) {iss:6.0f} | waitcnt {wait:6.0f} | barrier {bar:6.0f} | step {step:6.0f}")
    arrive = t[:, :, :, 1]                                           # when each wave reaches the barrier
    print("  barrier arrival spread over the 12 waves (max - min), median:", (arrive.max(dim=1).values - arrive.min(dim=1).values).median().item())
if int(os.environ.get("GLOWTTS_WN_BWD_ABL", "0")) & 64:
    nwg = (R + 51) // 52
    t = tlb_fused.view(512, 64)[:nwg].cpu()
    n = int((t[0] != 0).sum())
    d = (t[:, 1:n] - t[:, : n - 1]).float()
    names = ["prologue", "end^T gemm", "end^T epi"] + sum([[f"L{l} gate loads", f"L{l} rs^T gemm", f"L{l} gate deriv", f"L{l} in^T pass 0", f"L{l} in^T pass 1", f"L{l} exchange", f"L{l} dx update"]
                                                            for l in range(dc.L - 1, -1, -1)], []) + ["start^T gemm", "dx_a rmw"]
    print("backward phase: median clocks over workgroups (min .. max)")
    for i in range(n - 1):
        print(f"  {names[i] if i < len(names) else i:18s} {d[:, i].median().item():9.0f} ({d[:, i].min().item():.0f} .. {d[:, i].max().item():.0f})")
    print(f"  total              {(t[:, n - 1] - t[:, 0]).float().median().item():9.0f}; first start .. last end {(t[:, n - 1].max() - t[:, 0].min()).item()}")
if os.environ.get("GLOWTTS_WN_ABL") == "16":
    tl = tl_fused
    t = tl.view(512 * 12, 32)[:512][: (R + 51) // 52].cpu()
    n = int((t[0] != 0).sum())
    d = (t[:, 1:n] - t[:, : n - 1]).float()
    names = ["prologue", "start gemm", "start epi"] + sum([[f"in{l} gemm", f"gate{l} epi", f"rs{l} gemm", f"rs{l} epi"] for l in range(dc.L)], []) + ["end gemm", "end epi"]
    print("phase: median clocks over workgroups (min .. max)")
    for i in range(n - 1):
        print(f"  {names[i] if i < len(names) else i:12s} {d[:, i].median().item():9.0f} ({d[:, i].min().item():.0f} .. {d[:, i].max().item():.0f})")
    print(f"  total        {(t[:, n - 1] - t[:, 0]).float().median().item():9.0f}; first start .. last end {(t[:, n - 1].max() - t[:, 0].min()).item()}")
print(f"B={B} Tm={TM} drop={DROP} abl={os.environ.get('GLOWTTS_WN_ABL', '0')}: " + " | ".join(f"{k} {v:7.1f} us/flow" for k, v in res.items()))
