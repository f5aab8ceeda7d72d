"""Bisects which part of the training step breaks hipGraph capture on this ROCm build: every case runs in its own process."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["full_model_fwd", "full_decoder_fwdbwd_F2", "full_decoder_fwdbwd_F12", "full_encoder_fwdbwd", "full_model_fwdbwd"] if os.environ.get("PROBE_FULL") else ["torch_linear", "conv_fwd", "conv_fwdbwd", "ln_fwdbwd", "attn_fwdbwd", "decoder_fwd", "decoder_fwdbwd", "encoder_fwdbwd", "model_fwd", "model_fwdbwd"]

def run_case(name):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch, yaml
    from glow_tts_amd import ops
    from glow_tts_amd.hparams import Recursive_Parse
    dev = "cuda"
    torch.manual_seed(0)
    def capture(fn, warm=2):
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm): fn()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        g.replay(); g.replay(); torch.cuda.synchronize()
        return out
    if name == "torch_linear":
        m = torch.nn.Linear(64, 64).to(dev); x = torch.randn(32, 64, device=dev)
        def fn():
            m.zero_grad(set_to_none=True); y = m(x).sum(); y.backward(); return y.detach()
        capture(fn)
    elif name in ("conv_fwd", "conv_fwdbwd"):
        from glow_tts_amd.conv_fn import conv_rows
        x = torch.randn(1024, 64, device=dev, requires_grad=True); w = torch.randn(64, 64, 3, device=dev, requires_grad=True); b = torch.zeros(64, device=dev, requires_grad=True)
        rm = torch.ones(1024, device=dev)
        def fn():
            y = conv_rows(x, w, b, rm, relu=True, mask_out=True)
            if name == "conv_fwdbwd":
                x.grad = w.grad = b.grad = None; y.sum().backward()
            return y.detach()
        capture(fn)
    elif name == "ln_fwdbwd":
        from glow_tts_amd.conv_fn import layernorm_rows
        x = torch.randn(1024, 64, device=dev, requires_grad=True); g_ = torch.ones(64, device=dev, requires_grad=True); b = torch.zeros(64, device=dev, requires_grad=True)
        rm = torch.ones(1024, device=dev)
        def fn():
            x.grad = g_.grad = b.grad = None
            y = layernorm_rows(x, None, g_, b, rm, relu=True); y.sum().backward(); return y.detach()
        capture(fn)
    elif name == "attn_fwdbwd":
        from glow_tts_amd.conv_fn import RPRAttention
        B, Tp, H, D = 2, 20, 2, 16
        qkv = torch.randn(B * Tp, 3 * H * D, device=dev, requires_grad=True); rk = torch.randn(1, 9, D, device=dev, requires_grad=True); rv = torch.randn(1, 9, D, device=dev, requires_grad=True)
        rm = torch.ones(B * Tp, device=dev)
        def fn():
            qkv.grad = rk.grad = rv.grad = None
            y = RPRAttention.apply(qkv, rk, rv, rm, B, Tp, H, 4, 0.0, 0, None); y.sum().backward(); return y.detach()
        capture(fn)
    elif name.startswith("full_"):
        from glow_tts_amd.modules import GlowTTS, MLE_Loss
        from glow_tts_amd import decoder as D, encoder as E
        hp = yaml.safe_load(open(os.path.join(REPO, "glow_tts_amd", "Hyper_Parameters.default.yaml")))
        if "_F2" in name: hp["Decoder"]["Stack"] = 2
        model = GlowTTS(Recursive_Parse(hp)).to(dev).train(); model.overlap_encoder = False
        B, Tt, Tm = int(os.environ.get("PROBE_B", "4")), 120, 800
        tokens = torch.randint(0, 35, (B, Tt), device=dev); tl = torch.full((B,), Tt, device=dev)
        mels = torch.randn(B, 80, Tm, device=dev); ml = torch.full((B,), Tm, device=dev)
        mle = MLE_Loss(model.hp)
        model(tokens, tl, mels, ml, None, None, None)          # ActNorm init, eager
        def fn():
            model.zero_grad(set_to_none=True)
            P = dict(model.named_parameters())
            if "decoder" in name:
                W = D.stack_decoder_weights(P, model.dec_cfg)
                z, ld, _ = D.DecoderFunction.apply(model.dec_cfg, mels, ml, None, 0.05, None, None, None, *W); (z.sum() + ld.sum()).backward(); return z.detach()
            if "encoder" in name:
                tm = model.Mask_Generate(tl, tokens.shape[1])
                mean, ls, dur = E.encoder_forward(P, model.hp, tokens, tm, None, None, True, precision=model.dec_cfg.precision)
                (mean.sum() + ls.sum() + dur.sum()).backward(); return mean.detach()
            out = model(tokens, tl, mels, ml, None, None, None)
            if name.endswith("_fwd"): return out[0].detach()
            loss = mle(z=out[0], mean=out[1], std=out[2], log_dets=out[3], lengths=ml) + torch.nn.functional.mse_loss(out[4], out[5])
            loss.backward(); return loss.detach()
        capture(fn)
    else:
        from helpers import load_case, tiny_hp_dict
        from glow_tts_amd.modules import GlowTTS, MLE_Loss
        sd, _, r = load_case("tiny_vanilla.npz")
        hp = tiny_hp_dict("Vanilla"); model = GlowTTS(Recursive_Parse(hp)); model.load_state_dict(sd)
        for f in model.layer_Dict["Decoder"].layer_Dict["Flows"]: f.layers[0].initialized = True
        model = model.to(dev).train(); model.overlap_encoder = False
        t = lambda k: torch.from_numpy(r[k]).to(dev)
        tokens, tl, mels, ml = t("tokens"), t("token_lengths"), t("mels"), t("mel_lengths")
        mle = MLE_Loss(model.hp)
        from glow_tts_amd import decoder as D, encoder as E
        def fn():
            model.zero_grad(set_to_none=True)
            if name.startswith("decoder"):
                P = dict(model.named_parameters()); W = D.stack_decoder_weights(P, model.dec_cfg)
                if name == "decoder_fwd":
                    with torch.no_grad(): z, ld, _ = D.DecoderFunction.apply(model.dec_cfg, mels, ml, None, 0.05, None, None, None, *W)
                else:
                    z, ld, _ = D.DecoderFunction.apply(model.dec_cfg, mels, ml, None, 0.05, None, None, None, *W); (z.sum() + ld.sum()).backward()
                return z.detach()
            if name == "encoder_fwdbwd":
                P = dict(model.named_parameters()); tm = model.Mask_Generate(tl, tokens.shape[1])
                mean, ls, dur = E.encoder_forward(P, model.hp, tokens, tm, None, None, True, precision=model.dec_cfg.precision)
                (mean.sum() + ls.sum() + dur.sum()).backward(); return mean.detach()
            out = model(tokens, tl, mels, ml, None, None, None)
            if name == "model_fwd": return out[0].detach()
            loss = mle(z=out[0], mean=out[1], std=out[2], log_dets=out[3], lengths=ml) + torch.nn.functional.mse_loss(out[4], out[5])
            loss.backward(); return loss.detach()
        capture(fn)
    print("CASE_OK", name)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for c in CASES:
            p = subprocess.run([sys.executable, "-W", "ignore", __file__, c], capture_output=True, text=True, timeout=600)
            ok = "CASE_OK" in p.stdout
            tail = (p.stderr.strip().splitlines() or [""])[-1][:160]
            print(f"{c:16s} {'OK' if ok else 'FAIL rc=' + str(p.returncode)}  {'' if ok else tail}")
