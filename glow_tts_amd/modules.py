"""Host-side mirror of the reference's `Modules.py` for the hot path: `GlowTTS` (forward = training graph,
inference = inverse flow) and `MLE_Loss`, with the reference's constructor convention (no arguments, hyper
parameters from ./Hyper_Parameters.yaml), call signatures, return tuples, attribute paths and state-dict keys
(Modules.py:16-229, 1020-1029), so checkpoints and yaml files drop in unchanged.  All arithmetic runs in the
HIP library (glow_tts_amd/csrc) or, for the parts still marked interim in DESIGN.md, in PyTorch-ROCm device ops.
"""
import math

import os
import torch

from . import alignment, decoder, encoder, ops
from .hparams import get_hp


class _Dict(torch.nn.Module):
    """A module that only owns a `layer_Dict` (the reference's universal container)."""

    def __init__(self):
        super().__init__()
        self.layer_Dict = torch.nn.ModuleDict()


class _Params(torch.nn.Module):
    """Leaf holding named parameters (weight / bias / weight_g / weight_v / ...)."""

    def __init__(self, **tensors):
        super().__init__()
        for k, v in tensors.items():
            self.register_parameter(k, torch.nn.Parameter(v))


def _xavier(shape, gain=1.0):
    w = torch.empty(shape)
    torch.nn.init.xavier_uniform_(w, gain=gain)
    return w


def _conv_params(o, i, k, gains="linear", bias=True, weight_norm=False, default_init=False):
    """Parameter leaf of a Conv1d.  `gains`: the reference's w_init_gain (Modules.py:983-1003): one gain or a list
    applied to equal chunks of the output channels; 'zero' -> zeros; default_init -> torch.nn.Conv1d's own init."""
    if default_init:
        c = torch.nn.Conv1d(i, o, k)
        w, b = c.weight.detach().clone(), c.bias.detach().clone()
    else:
        gl = [gains] if isinstance(gains, str) else list(gains)
        parts = []
        for g in gl:
            shape = (o // len(gl), i, k)
            if g == "zero":
                parts.append(torch.zeros(shape))
            elif g in ("relu", "leaky_relu"):
                w_ = torch.empty(shape)
                torch.nn.init.kaiming_uniform_(w_, nonlinearity=g)
                parts.append(w_)
            else:
                parts.append(_xavier(shape, torch.nn.init.calculate_gain(g)))
        w, b = torch.cat(parts, 0), torch.zeros(o)
    t = {}
    if bias:
        t["bias"] = b
    if weight_norm:                       # old-style torch weight_norm: g = ||v|| at init (Modules.py:766)
        t["weight_g"] = w.flatten(1).norm(dim=1).view(-1, 1, 1)
        t["weight_v"] = w
    else:
        t["weight"] = w
    return _Params(**t)


class _ActNorm(_Params):
    """Activation_Norm parameters + the reference's plain-attribute init flag (Modules.py:673)."""

    def __init__(self, C):
        super().__init__(logs=torch.zeros(1, C, 1), bias=torch.zeros(1, C, 1))
        self.initialized = False


class _Flow(torch.nn.Module):
    """AIA (Modules.py:653-660): layers = [Activation_Norm, Invertible_1x1_Conv, Affine_Coupling_Layer]."""

    def __init__(self, hp):
        super().__init__()
        C = hp.Sound.Mel_Dim * hp.Decoder.Num_Squeeze
        H = hp.Decoder.Affine_Coupling.Calc_Channels
        wn = hp.Decoder.Affine_Coupling.WaveNet
        ns = hp.Decoder.Num_Split
        w = torch.linalg.qr(torch.randn(ns, ns))[0]                     # Modules.py:718-725
        if torch.det(w) < 0:
            w[:, 0] = -w[:, 0]
        w = w.contiguous()                                              # (qr returns a column-major Q; collectives need contiguous tensors)
        coupling = _Dict()
        coupling.layer_Dict["Start"] = _conv_params(H, C // 2, 1, "linear", weight_norm=True)
        wavenet = _Dict()
        mode = hp.Mode.upper()
        for l in range(wn.Num_Layers):
            wavenet.layer_Dict[f"In_{l}"] = _conv_params(2 * H, H, wn.Kernel_Size, ["tanh", "sigmoid"], weight_norm=True)
            wavenet.layer_Dict[f"Res_Skip_{l}"] = _conv_params(2 * H if l < wn.Num_Layers - 1 else H, H, 1, "linear", weight_norm=True)
            if mode in ("SE", "GR"):
                wavenet.layer_Dict[f"Speaker_{l}"] = _conv_params(2 * H, hp.Speaker_Embedding.Embedding_Size, 1, ["tanh", "sigmoid"], weight_norm=True)
            if mode in ("PE", "GR"):
                wavenet.layer_Dict[f"Prosody_{l}"] = _conv_params(2 * H, hp.Prosody_Encoder.Size, 1, ["tanh", "sigmoid"], weight_norm=True)
            if mode == "GR":
                wavenet.layer_Dict[f"Pitch_{l}"] = _conv_params(2 * H, hp.Decoder.Num_Squeeze, 1, ["tanh", "sigmoid"], weight_norm=True)
        coupling.layer_Dict["WaveNet"] = wavenet
        coupling.layer_Dict["End"] = _conv_params(C, H, 1, "zero")      # zero-init: identity coupling (Modules.py:773-778)
        self.layers = torch.nn.ModuleList([_ActNorm(C), _Params(weight=w), coupling])


def _build_encoder(hp):
    e = hp.Encoder
    C = e.Channels
    enc = _Dict()
    enc.layer_Dict["Embedding"] = _Params(weight=torch.randn(e.Embedding_Tokens, C) * C ** -0.5)      # Modules.py:246-250
    pre = _Dict()
    for i in range(e.Prenet.Stacks):
        clrd = _Dict()
        clrd.layer_Dict["Conv"] = _conv_params(C, C, e.Prenet.Kernel_Size, default_init=True)
        clrd.layer_Dict["LayerNorm"] = _Params(weight=torch.ones(C), bias=torch.zeros(C))
        pre.layer_Dict[f"CLRD_{i}"] = clrd
    pre.layer_Dict["Conv1x1"] = _conv_params(C, C, 1, default_init=True)
    enc.layer_Dict["Prenet"] = pre
    tr = _Dict()
    heads = e.Transformer.Attention.Heads
    D = C // heads
    win = e.Transformer.Attention.Window_Size
    for i in range(e.Transformer.Stacks):
        blk = _Dict()
        att = _Params(weight_K=torch.randn(1, 2 * win + 1, D) * D ** -0.5, weight_V=torch.randn(1, 2 * win + 1, D) * D ** -0.5)
        att.layer_Dict = torch.nn.ModuleDict()
        for name in ("Query", "Key", "Value"):                           # xavier on Q/K/V only (RPR_MHA.py:45-47)
            pp = _conv_params(C, C, 1, default_init=True)
            torch.nn.init.xavier_uniform_(pp.weight)
            att.layer_Dict[name] = pp
        att.layer_Dict["Projection"] = _conv_params(C, C, 1, default_init=True)
        blk.layer_Dict["Attention"] = att
        blk.layer_Dict["LayerNorm_0"] = _Params(weight=torch.ones(C), bias=torch.zeros(C))
        blk.layer_Dict["Conv_0"] = _conv_params(e.Transformer.Conv.Calc_Channels, C, e.Transformer.Conv.Kernel_Size, default_init=True)
        blk.layer_Dict["Conv_1"] = _conv_params(C, e.Transformer.Conv.Calc_Channels, e.Transformer.Conv.Kernel_Size, default_init=True)
        blk.layer_Dict["LayerNorm_1"] = _Params(weight=torch.ones(C), bias=torch.zeros(C))
        tr.layer_Dict[f"ANCRDCN_{i}"] = blk
    enc.layer_Dict["Transformer"] = tr
    enc.layer_Dict["Project"] = _conv_params(hp.Sound.Mel_Dim * 2, C, 1, default_init=True)
    dp = _Dict()
    cin = C
    mode = hp.Mode.upper()
    if mode in ("SE", "GR"):
        cin += hp.Speaker_Embedding.Embedding_Size                          # Modules.py:583-590
    elif mode == "PE":
        cin += hp.Prosody_Encoder.Size
    for i in range(e.Duration_Predictor.Stacks):
        crnd = _Dict()
        crnd.layer_Dict["Conv"] = _conv_params(e.Duration_Predictor.Channels, cin, e.Duration_Predictor.Kernel_Size, default_init=True)
        dp.layer_Dict[f"CRND_{i}"] = crnd
        cin = e.Duration_Predictor.Channels
    dp.layer_Dict["Projection"] = _conv_params(1, cin, 1, default_init=True)
    enc.layer_Dict["Duration_Predictor"] = dp
    return enc


def _unshare_state_dict(module, state_dict, prefix, local_metadata):
    """Leaves that live in a flat per-class storage (decoder.LeafStack) are views of a larger tensor; torch.save would write the
    whole storage once per key.  Checkpoints get private copies (same keys, shapes and values as the reference's)."""
    for k, v in list(state_dict.items()):
        if torch.is_tensor(v) and v.untyped_storage().nbytes() > v.numel() * v.element_size():
            state_dict[k] = v.clone()
    return state_dict


def _drop_ge2e_keys(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
    """Reference checkpoints trained with Speaker_Embedding.Type 'GE2E' carry the pre-trained d-vector LSTM under `layer_Dict.GE2E.*`
    (Modules.py:30-35; loaded separately by Train.py:555-561).  That network is an un-vendored submodule of the reference: here the
    d-vectors are an input, so those keys are dropped explicitly instead of failing a strict load."""
    for k in [k for k in state_dict if k.startswith(prefix + "layer_Dict.GE2E.")]:
        del state_dict[k]


class GlowTTS(torch.nn.Module):
    """Drop-in for Modules.GlowTTS (Modules.py:16-229)."""

    def __init__(self, hp=None):
        super().__init__()
        self.hp = hp = hp if hp is not None else get_hp()
        mode = hp.Mode.upper()
        if mode not in ("VANILLA", "SE", "PE", "GR"):
            raise ValueError("Unsupported mode: {}".format(hp.Mode))
        self.layer_Dict = torch.nn.ModuleDict()
        if mode in ("SE", "GR"):
            if hp.Speaker_Embedding.Type.upper() == "LUT":
                self.layer_Dict["LUT"] = _Params(weight=torch.empty(hp.Speaker_Embedding.Num_Speakers,
                                                                    hp.Speaker_Embedding.Embedding_Size).uniform_(-1.0, 1.0))
            elif hp.Speaker_Embedding.Type.upper() == "GE2E":
                pass   # GE2E source is an un-vendored submodule of the reference: d-vectors are an input
            else:
                raise ValueError("Unsupported Speaker embedding type: {}".format(hp.Speaker_Embedding.Type))
        if mode in ("PE", "GR"):
            from .prosody import Prosody_Encoder
            self.layer_Dict["Prosody_Encoder"] = Prosody_Encoder(hp)
        if mode == "GR":                                                                   # Modules.py:44-46
            from .prosody import Pitch_Interpolater, Speaker_Classifier_GR
            assert hp.Speaker_Embedding.Embedding_Size == hp.Prosody_Encoder.Size, \
                "In GR mode, the size of speaker embeding and prosody encoder must be same."     # Modules.py:588-589
            self.layer_Dict["Speaker_Classifier_GR"] = Speaker_Classifier_GR(hp)
            self.layer_Dict["Pitch_Interpolater"] = Pitch_Interpolater()
        self.layer_Dict["Encoder"] = _build_encoder(hp)
        dec = _Dict()
        dec.layer_Dict["Flows"] = torch.nn.ModuleList([_Flow(hp) for _ in range(hp.Decoder.Stack)])
        self.layer_Dict["Decoder"] = dec
        wn = hp.Decoder.Affine_Coupling.WaveNet
        prec = {"bf16": ops.BF16, "f32": ops.F32}[str(getattr(hp, "HIP_Precision", "bf16")).lower()]
        self.dec_cfg = decoder.DecoderConfig(hp.Sound.Mel_Dim, hp.Decoder.Stack, hp.Decoder.Num_Squeeze, hp.Decoder.Num_Split,
                                             hp.Decoder.Affine_Coupling.Calc_Channels, wn.Num_Layers, wn.Kernel_Size, prec)
        if "Prosody_Encoder" in self.layer_Dict:
            self.layer_Dict["Prosody_Encoder"].hip_precision = prec
        self.actnorm_allreduce = None     # set by the data-parallel wrapper (glow_tts_amd.distributed)
        self._enc_stream = None
        self._pack_stream = None
        self._pcache = None               # name -> Parameter (see _params)
        self._dec_stacks = None           # decoder.DecoderStacks: flat per-class parameter storage, built on first use
        self._enc_cache = {}              # encoder leaf stacks (fused Query/Key/Value weights)
        self._register_state_dict_hook(_unshare_state_dict)
        self._register_load_state_dict_pre_hook(_drop_ge2e_keys)
        self.overlap_encoder = True       # the text encoder runs on its own HIP stream beside the flow decoder (set False to serialise them)

    # ---------------------------------------------------------------- helpers
    def __getstate__(self):
        # run-time caches (HIP stream, flat parameter storage views) are rebuilt on first use: keep them out of copies / pickles
        st = self.__dict__.copy()
        st["_enc_stream"], st["_pack_stream"], st["_dec_stacks"], st["_enc_cache"], st["_pcache"] = None, None, None, {}, None
        return st

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def _params(self):
        # name -> Parameter, cached: walking 515 parameters costs ~1 ms of host time per call (eager launches are host-bound)
        if self._pcache is None:
            self._pcache = dict(self.named_parameters())
        return self._pcache

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)           # .to() / .cuda() / .float(): parameter objects may be replaced
        self._pcache, self._dec_stacks, self._enc_cache = None, None, {}
        return out

    def _stacks(self, P):
        if self._dec_stacks is None:
            self._dec_stacks = decoder.DecoderStacks(P, self.dec_cfg)
        return self._dec_stacks

    def _flows(self):
        return self.layer_Dict["Decoder"].layer_Dict["Flows"]

    def set_precision(self, name):
        self.dec_cfg.precision = {"bf16": ops.BF16, "f32": ops.F32}[name]

    def Mask_Generate(self, lengths, max_lengths=None, dtype=torch.float):
        """Modules.py:206-211 (max_lengths avoids the hidden device->host sync of torch.max)."""
        T = int(max_lengths) if max_lengths is not None else int(torch.max(lengths))
        return (torch.arange(T, device=lengths.device)[None, :] < lengths[:, None]).unsqueeze(1).to(dtype)

    def _conditioning(self, P, speakers, mels_for_ge2e, prosody_mels, prosody_lengths, prosodies=None):
        """prosodies: a pre-computed prosody vector [B, Size] (GraphedInference runs the variable-length prosody encoder outside its graphs)."""
        mode = self.hp.Mode.upper()
        spk = pro = None
        if "LUT" in self.layer_Dict:
            spk = torch.nn.functional.embedding(speakers, P["layer_Dict.LUT.weight"])                   # Modules.py:73-74
        elif mode in ("SE", "GR"):
            # GE2E mode: the pre-computed, L2-normalised d-vectors arrive in `mels_for_ge2e` ([B, Embedding_Size])
            spk = mels_for_ge2e.detach()                                                                  # Modules.py:75-77
        if "Prosody_Encoder" in self.layer_Dict:
            pro = prosodies if prosodies is not None else self.layer_Dict["Prosody_Encoder"](prosody_mels, prosody_lengths)   # Modules.py:81-82
        return spk, pro

    def _maybe_init_actnorm(self, P, mels, mel_lengths, cond, pitch=None):
        """ActNorm data-dependent init on the first call (Modules.py:685-687), flag kept per flow like the reference."""
        flows = self._flows()
        if all(f.layers[0].initialized for f in flows):
            return
        W = dict(zip(decoder.WEIGHT_KEYS, [w.detach().contiguous().clone() for w in decoder.stack_decoder_weights(P, self.dec_cfg)]))
        decoder.actnorm_data_init(self.dec_cfg, W, mels, mel_lengths, cond=None if cond is None else cond.detach(),
                                  allreduce=self.actnorm_allreduce, pitch=pitch)
        with torch.no_grad():
            for i, f in enumerate(flows):
                if not f.layers[0].initialized:
                    f.layers[0].logs.copy_(W["an_logs"][i].view(1, -1, 1))
                    f.layers[0].bias.copy_(W["an_bias"][i].view(1, -1, 1))
                    f.layers[0].initialized = True

    # ---------------------------------------------------------------- training graph
    def forward(self, tokens, token_lengths, mels, mel_lengths, speakers=None, mels_for_ge2e=None, pitches=None):
        """Modules.py:50-126.  Returns the reference's 8-tuple."""
        hp = self.hp
        if not torch.cuda.is_current_stream_capturing():      # the check reads the lengths back (device sync): skipped inside a hipGraph capture
            assert bool(torch.all(mel_lengths % hp.Decoder.Num_Squeeze == 0)), "Mel lengths must be diviable by Num_Squeeze."
        if not mels.is_cuda:
            raise RuntimeError("glow_tts_amd runs on the GPU only (no CPU fallback)")
        P = self._params()
        decoder.stamp("fwd_begin")
        spk, pro = self._conditioning(P, speakers, mels_for_ge2e, mels, mel_lengths)
        ns_ = int(hp.Decoder.Num_Squeeze)
        if mels.shape[2] % ns_:                               # Squeeze cuts the frames that do not fill a group (Modules.py:897-898):
            t_cut = (mels.shape[2] // ns_) * ns_              # z, the masks and the attentions all have the cut length
            mels = mels[:, :, :t_cut].contiguous()
            if pitches is not None:
                pitches = pitches[:, :t_cut].contiguous()
        # Encoder and decoder are independent until the log-prior: the encoder runs on its own HIP stream, concurrently with the
        # flow decoder whose latency-bound kernels leave CUs idle.  autograd replays each backward on its forward stream, so the
        # two backward passes overlap as well.
        decoder.stamp("fwd_enter_main")
        # the decoder's weight preparation goes out first: see decoder.EARLY
        stacks = self._stacks(P)
        import weakref
        decoder.AUX["stacks"] = weakref.ref(stacks)
        use_gv = bool(decoder.TUNE["prep_fused"] and torch.is_grad_enabled() and decoder.fused_wn_supported(self.dec_cfg) and
                      self.dec_cfg.precision == ops.BF16)
        W = stacks.weights(gv=use_gv)
        main = torch.cuda.current_stream()
        if self._enc_stream is None:
            self._enc_stream = torch.cuda.Stream()      # (equal priority: a priority difference between the two branches of the replayed graph, in either direction, doubles the step - DESIGN.md section 5)
        side = self._enc_stream if self.overlap_encoder else main
        # the step's dropout seed word: one launch here, in front of the fork - both streams' dropout kernels read it (round 6; torch.randint drew one word on each
        # stream: two RNG launches inside the graph and two fills of the generator's offset words in front of every replay)
        seed_word = decoder.step_seed(mels.device) if self.training else None
        side.wait_stream(main)
        if seed_word is not None and side is not main:
            seed_word.record_stream(side)
        pack_aux = None
        if side is not main and decoder.TUNE["enc_pack_split"]:
            if self._pack_stream is None:
                self._pack_stream = torch.cuda.Stream()
            pack_aux = self._pack_stream
            pack_aux.wait_stream(main)                        # (forked HERE, from the origin stream: see encoder._PackSets.run)
        prior_ready = torch.cuda.Event() if side is not main else None
        # (behind the fork: the encoder's stream does not wait for it)
        if use_gv and decoder.TUNE["prep_early"] and all(f.layers[0].initialized for f in self._flows()):
            decoder.early_prepare(self.dec_cfg, W, mels.shape, fused_bwd_ok=(pitches is None or "Pitch_v" not in stacks.S),
                                  conditioned=(spk is not None or pro is not None))
        early_prep = decoder.EARLY["prep"]
        # (TUNE["prep_bwd_late"], off by default: the deferred launch writes buffers that early_prepare allocated on THIS stream behind the fork - the encoder's
        #  stream must see this stream's allocator history first: an event here, waited for in front of the deferred launch.  ADVICE r5)
        images_allocated = None
        if early_prep is not None and side is not main and getattr(early_prep, "bwd_pending", None) is not None:
            images_allocated = torch.cuda.Event()
            images_allocated.record(main)

        prepared = {}

        def prior_done(mean_, log_std_):
            # the log-prior GEMM's operands that need the encoder's outputs only: here, on the encoder's stream (one launch off the chain between the flow
            # decoder's forward and its backward)
            ns_sq = int(hp.Decoder.Num_Squeeze)
            prepared["p"] = alignment.log_prior_prepare(mean_.detach(), log_std_.detach(), token_lengths, mel_lengths, (mels.shape[2] // ns_sq) * ns_sq, ns_sq)
            prior_ready.record(side)
            # the decoder's backward-only weight images: on the encoder's stream right behind its projection (they need nothing but the weights) - under the
            # decoder's z / log-determinant passes and in front of the duration predictor, whose result only the losses read.  (Behind the duration
            # predictor they ran under the alignment search, a single-wave latency chain: mas_dp2 36 -> 52 us.)  Joined with this stream before the call returns.
            if early_prep is not None and getattr(early_prep, "bwd_pending", None) is not None:
                if images_allocated is not None:
                    side.wait_event(images_allocated)
                early_prep.launch_bwd_images(gentle=bool(decoder.TUNE["prep_bwd_gentle"]))
        with torch.cuda.stream(side):
            decoder.stamp("enc_branch_first_node")
            # (the token mask is the encoder's: built on its stream, so that the decoder's chain starts with its own weight preparation)
            token_mask, token_rowmask = encoder.token_masks(token_lengths, tokens.shape[1])
            mean, log_std, log_dur = encoder.encoder_forward(P, hp, tokens, token_mask, spk, pro, self.training, precision=self.dec_cfg.precision,
                                                             cache=self._enc_cache, rowmask=token_rowmask,
                                                             on_prior_ready=prior_done if prior_ready is not None else None,
                                                             pack_stream=pack_aux, seed_t=seed_word)
        decoder.stamp("main_after_enc_launch")
        # (the conditioning convs on a third stream, forked in front of the decoder's weight preparation and joined here, were measured in round 6: config 3 6.00
        #  against 5.11 ms/step, config 4 4.65 against 3.9 - a third branch in the replayed graph costs far more than the ~35 us it would hide)
        cond = stacks.conditioning(spk, pro)
        pitch_w, pitch_b = stacks.pitch_weights()
        if pitch_w is None:
            pitches = None                                                                              # Modules.py:89-90
        elif pitches is None:
            raise ValueError("GR mode needs `pitches` [Batch, Mel_t] (Modules.py:58, 867-869)")
        self._maybe_init_actnorm(P, mels, mel_lengths, cond, None if pitches is None else (pitches, pitch_w.detach(), pitch_b.detach()))
        # (training on the fused bf16 path: W holds the weight-norm pairs themselves - every weight image was prepared by one launch above)
        drop_p = float(hp.Decoder.Affine_Coupling.WaveNet.Dropout_Rate) if self.training else 0.0      # Modules.py:854-862
        decoder.AUX["stream"] = side if side is not main else None          # (z / log-determinant passes: off the chain to the log-prior, joined below)
        decoder.AUX["seed"] = seed_word
        try:
            z, log_dets, z_rows = decoder.DecoderFunction.apply(self.dec_cfg, mels, mel_lengths, cond, drop_p, pitches, pitch_w if pitches is not None else None,
                                                                pitch_b if pitches is not None else None, *W)
        finally:
            decoder.AUX["stream"] = decoder.AUX["seed"] = None
        if side is not main:
            # the log-prior needs mean / log_std only: the duration predictor still runs on the encoder's stream (joined below, before the
            # losses read log_dur) - the encoder's forward is what this point of the step waits for (timeline).  (One join behind
            # the duration predictor instead of this event + the join below: 4.86-4.91 against 4.84-4.87 ms/step, round 6.)
            main.wait_event(prior_ready)
            for t_ in (mean, log_std, log_dur):
                t_.record_stream(main)
        ns = int(hp.Decoder.Num_Squeeze)
        value_t, tx32, ty32 = alignment.log_prior_t(mean.detach(), log_std.detach(), z.detach(), token_lengths, mel_lengths, ns,
                                                    return_lengths=True, z_rows=z_rows, prepared=prepared.get("p"))        # Modules.py:107-114 (lengths of the squeezed z)
        if side is not main and prepared.get("p") is not None:
            for t_ in prepared["p"].values():
                if torch.is_tensor(t_):
                    t_.record_stream(main)
        idx = alignment.maximum_path_t(value_t, tx32, ty32)                                          # :115-116
        if idx.shape[1] != z.shape[2]:
            idx = idx[:, :z.shape[2]].contiguous()
        # The dense 0/1 attentions are only RETURNED (the losses use the per-frame token index): written by the expansion's launch below.  (Until round 6 they were
        # written on the encoder's stream, behind a second fork of it - that edge cost the replayed graph more than a launch on this stream.)
        bwd_side = side if (side is not main and torch.is_grad_enabled()) else None
        # Modules.py:116, 120-122 in one launch (the 0/1 matrix, the gathers by the MAS index, the run lengths); MLE_Loss on these two tensors differentiates through
        # the gather itself
        mel_mean, mel_log_std, log_dur_targets, attn = alignment.ExpandPair.apply(mean, log_std, idx, token_lengths, bwd_side, True)
        if cond is None:
            # (conditioned modes keep the expansion's own backward: there the decoder's weight-gradient tail runs on a third stream (decoder.TUNE["tail_aside"]), and
            #  with the encoder's backward hanging off MLE_Loss's node instead of off an explicit fork the replayed graph serialises it behind the conditioning
            #  encoders' backward - config 3 5.85 against 5.12 ms/step, measured; in Vanilla mode the fused backward is time-neutral and saves two passes)
            alignment.tag_prior(mel_mean, mel_log_std, mean, log_std, idx)
        log_dur_targets = log_dur_targets.unsqueeze(1)
        if side is not main:
            main.wait_stream(side)                                # (joins the duration predictor, which ran on behind the prior on the encoder's stream)
        classified = None
        if "Speaker_Classifier_GR" in self.layer_Dict:
            classified = self.layer_Dict["Speaker_Classifier_GR"](pro)                                    # Modules.py:84-87
        return z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_targets, attn, classified

    def aux_stream(self):
        """The stream the last forward ran the text encoder on, or None when that was the caller's own: work that only the encoder's side
        of the step consumes (the duration loss and its backward) can be queued there instead of in front of the flow decoder's backward
        (`Beside`)."""
        return self._enc_stream if (self.overlap_encoder and self._enc_stream is not None) else None

    # ---------------------------------------------------------------- inverse flow
    @torch.no_grad()
    def inference(self, tokens, token_lengths, mels_for_prosody=None, mel_lengths_for_prosody=None, speakers=None,
                  mels_for_ge2e=None, pitches=None, pitch_lengths=None, noise_scale=1.0, length_scale=1.0, noises=None):
        """Modules.py:128-204.  `noises` (optional, [B, Mel_Dim, >= max T_mel]) injects the Gaussian noise the reference
        draws with torch.randn_like (:187) so that results are reproducible."""
        front = self.inference_front(tokens, token_lengths, mels_for_prosody, mel_lengths_for_prosody, speakers, mels_for_ge2e, length_scale)
        return self.inference_back(front, None, noise_scale, noises, pitches=pitches, pitch_lengths=pitch_lengths)

    @torch.no_grad()
    def inference_front(self, tokens, token_lengths, mels_for_prosody=None, mel_lengths_for_prosody=None, speakers=None, mels_for_ge2e=None,
                        length_scale=1.0, prosodies=None):
        """First half of `inference` (Modules.py:128-174): conditioning, encoder, durations -> (mean, log_std, dur, mel_lengths, token_mask,
        spk, pro).  Everything stays on the device; nothing here depends on the mel length (glow_tts_amd.graph_infer replays it as a hipGraph)."""
        hp = self.hp
        P = self._params()
        spk, pro = self._conditioning(P, speakers, mels_for_ge2e, mels_for_prosody, mel_lengths_for_prosody, prosodies)
        token_mask = self.Mask_Generate(token_lengths, tokens.shape[1])
        mean, log_std, log_dur = encoder.encoder_forward(P, hp, tokens, token_mask, spk, pro, False, precision=self.dec_cfg.precision, cache=self._enc_cache)
        if not torch.is_tensor(length_scale):
            length_scale = torch.tensor([float(length_scale)], device=tokens.device)
        ls = length_scale.to(tokens.device).unsqueeze(-1).unsqueeze(-1)                                   # Modules.py:169
        dur = torch.ceil(torch.exp(log_dur) * token_mask * ls).squeeze(1)                                 # :173
        mel_lengths = torch.clamp_min(dur.sum(1), 1.0).long()                                             # :174
        return mean, log_std, dur, mel_lengths, token_mask, spk, pro

    @torch.no_grad()
    def inference_back(self, front, max_mel_length=None, noise_scale=1.0, noises=None, prep=None, pitches=None, pitch_lengths=None):
        """Second half (Modules.py:175-204): hard alignment, prior sample, inverse flow.  max_mel_length: padded frame count (None: the batch
        maximum, read back from the device like the reference's torch.max); prep: a decoder._Prepared kept by the caller (static weights)."""
        hp = self.hp
        mean, log_std, dur, mel_lengths, token_mask, spk, pro = front
        mel_mask = self.Mask_Generate(mel_lengths, max_mel_length)
        # Path_Generate (:181, :213-229) makes the hard alignment dense - attentions[x, y] = 1 iff cum[x-1] <= y < cum[x] - and the reference
        # expands mean / log_Std with two dense bmm (:183-184).  Here the per-frame token index comes from one search over the cumulative
        # durations, the dense matrix (a returned output) from the MAS path kernel and the expansion is the training path's gather.
        Tmax, Tx = mel_mask.shape[2], mean.shape[2]
        cum = torch.cumsum(dur, dim=1).contiguous()                                                         # integer-valued floats
        frames = torch.arange(Tmax, device=cum.device, dtype=cum.dtype)
        idx = torch.searchsorted(cum, frames.unsqueeze(0).expand(cum.shape[0], -1).contiguous(), right=True)      # tokens whose span ended at or before y
        idx = torch.where((frames.unsqueeze(0) < mel_lengths.unsqueeze(1)) & (idx < Tx), idx, torch.full_like(idx, -1)).to(torch.int32).contiguous()
        from .monotonic_align import path_from_idx
        attn = path_from_idx(idx, Tx, torch.float32)
        mel_mean = alignment.ExpandPrior.apply(mean, idx)
        mel_log_std = alignment.ExpandPrior.apply(log_std, idx)
        if noises is None:
            noises = torch.randn_like(mel_mean)
        z = (mel_mean + torch.exp(mel_log_std) * noises[:, :, :mel_mean.shape[2]] * noise_scale) * mel_mask   # :187-191
        P = self._params()
        stacks = self._stacks(P)
        cond = stacks.conditioning(spk, pro)
        W = None if prep is not None else dict(zip(decoder.WEIGHT_KEYS, [w.contiguous() for w in stacks.weights()]))
        pitch = None
        if "Pitch_Interpolater" in self.layer_Dict:                                                        # :193-196
            if pitches is None:
                raise ValueError("GR mode needs `pitches` and `pitch_lengths` (Modules.py:136-137)")
            pw, pb = stacks.pitch_weights()
            pitch = (self.layer_Dict["Pitch_Interpolater"](pitches, pitch_lengths, mel_lengths, z.shape[2]), pw, pb)
        mels = decoder.decoder_inverse(self.dec_cfg, W, z.contiguous(), mel_lengths, cond=cond, fill=-float(hp.Sound.Max_Abs_Mel), prep=prep,
                                       pitch=pitch)   # :198-202
        return mels, mel_lengths, attn

    def prepared_decoder_weights(self):
        """Packed weight images of the decoder for the CURRENT parameter values (inference with static weights: pack once, not per call)."""
        with torch.no_grad():
            P = self._params()
            W = dict(zip(decoder.WEIGHT_KEYS, [w.contiguous() for w in self._stacks(P).weights()]))
            return decoder._Prepared(self.dec_cfg, W, need_bwd=False, cond=None)

    def Path_Generate(self, durations, masks):
        """Modules.py:213-229."""
        B, Tx, Ty = masks.shape
        cum = torch.cumsum(durations, dim=1)
        upto = (torch.arange(Ty, device=masks.device)[None, None, :] < cum[:, :, None]).to(masks.dtype)
        prev = torch.nn.functional.pad(upto, [0, 0, 1, 0])[:, :-1]
        return (upto - prev) * masks


class Beside:
    """`with Beside(model) as b: length = ...` queues the block on model.aux_stream(), behind everything the caller's stream holds at that
    point; `b.join(*tensors)` makes the caller's stream wait for it (call it as late as the results are needed: what the caller queues in
    between runs concurrently).  autograd replays each backward on its forward's stream, so the block's backward leaves the caller's stream
    as well.  Without an auxiliary stream the block simply runs in place."""

    def __init__(self, model):
        self.main = torch.cuda.current_stream()
        aux = model.aux_stream() if hasattr(model, "aux_stream") else None
        self.aux = aux if (aux is not None and aux != self.main) else None
        self._ctx = None

    def __enter__(self):
        if self.aux is not None:
            self.aux.wait_stream(self.main)
            self._ctx = torch.cuda.stream(self.aux)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
            self._ctx = None
        return False

    def uses(self, *tensors):
        """Tensors of the caller's stream that the block reads (the caching allocator must not hand their memory out while it runs)."""
        if self.aux is not None:
            for t in tensors:
                if torch.is_tensor(t):
                    t.record_stream(self.aux)
        return self

    def join(self, *tensors):
        if self.aux is not None:
            self.main.wait_stream(self.aux)
            for t in tensors:
                if torch.is_tensor(t):
                    t.record_stream(self.main)


class MLE_Loss(torch.nn.modules.loss._Loss):
    """Modules.py:1020-1029."""

    def __init__(self, hp=None):
        super().__init__()
        self.hp = hp if hp is not None else get_hp()
        self._state = {}          # (this module's completion counter of the one-launch loss, alignment.PriorLoss)

    def forward(self, z, mean, std, log_dets, lengths):
        hp = self.hp
        return alignment.mle_loss(z, mean, std, log_dets, lengths, int(hp.Decoder.Num_Squeeze), int(hp.Sound.Mel_Dim), owner=self._state)
