"""`GlowTTS.inference` (Modules.py:128-204) as replayed hipGraphs, for serving.

Eagerly the inverse path is ~300 short launches and one device->host read per call: 2.7 ms per batch on an MI355X whatever the batch
size (host-bound).  Here a call is
    graph A  (per token shape)            conditioning, encoder, durations              -> mel lengths on the device
    one read of max(mel length)           picks the mel bucket
    graph B  (per token shape x bucket)   alignment expansion, prior sample, 12 inverse flows, unsqueeze
with the decoder's weight images packed once (`GlowTTS.prepared_decoder_weights`): serving weights are static.  Frames beyond an
utterance's length are masked exactly as the reference masks the batch padding, so the valid part of the result does not depend on the
bucket.  Call `refresh()` after the parameters changed.  Vanilla / LUT / pre-computed d-vector conditioning are tensor inputs of fixed shape;
PE mode: the GST prosody encoder, whose reference mels vary in length, runs eagerly before graph A and hands it the [B, Size] vector."""
import bisect

import torch


class GraphedInference:
    def __init__(self, model, mel_buckets=(128, 256, 384, 512, 640, 768, 896, 1024, 1280, 1536, 2048), warmup=2):
        self.model = model.eval()
        ns = int(model.hp.Decoder.Num_Squeeze)
        self.buckets = sorted(int(b) for b in mel_buckets)
        if any(b % ns for b in self.buckets):
            raise ValueError("mel buckets must be multiples of Decoder.Num_Squeeze")
        self.warmup = max(1, int(warmup))
        self.stream = torch.cuda.Stream()
        self.front, self.back = {}, {}
        self.prep = None
        self.refresh()

    def refresh(self):
        """Re-pack the decoder weights from the current parameters and drop the captured graphs (they point at the old images)."""
        self.prep = self.model.prepared_decoder_weights()
        self.front.clear()
        self.back.clear()

    def _graph(self, fn):
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(self.warmup):
                fn()
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        import torch.distributed as dist
        grouped = dist.is_available() and dist.is_initialized()
        if grouped:                                            # a data-parallel trainer's evaluation: see distributed.py "collectives and capture"
            from .distributed import before_capture
            before_capture()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local" if grouped else "global"):
            out = fn()
        return g, out

    @torch.no_grad()
    def __call__(self, tokens, token_lengths, speakers=None, mels_for_ge2e=None, noise_scale=1.0, length_scale=1.0, noises=None,
                 mels_for_prosody=None, mel_lengths_for_prosody=None):
        """-> (mels [B, Mel, T], mel_lengths [B], attentions [B, T_token, T]) like `GlowTTS.inference`; T = the batch's longest utterance
        rounded down to a multiple of Num_Squeeze (as the reference).  `noises` [B, Mel, >= T] replaces the internal Gaussian draw."""
        m = self.model
        dev = tokens.device
        if not torch.is_tensor(length_scale):
            length_scale = torch.tensor([float(length_scale)], device=dev)
        cond_in = speakers if speakers is not None else mels_for_ge2e
        pro = None
        if "Prosody_Encoder" in m.layer_Dict:                            # Modules.py:159-160, outside the graphs (variable-length input)
            pro = m.layer_Dict["Prosody_Encoder"](mels_for_prosody, mel_lengths_for_prosody)
        key = (tuple(tokens.shape), tuple(length_scale.shape), None if cond_in is None else (tuple(cond_in.shape), cond_in.dtype), pro is not None)
        if key not in self.front:
            st = [tokens.clone(), token_lengths.clone(), length_scale.to(dev).clone(), None if cond_in is None else cond_in.clone(),
                  None if pro is None else pro.clone()]
            fn = lambda: m.inference_front(st[0], st[1], None, None, st[3] if speakers is not None else None,
                                           st[3] if speakers is None else None, st[2], prosodies=st[4])
            g, out = self._graph(fn)
            self.front[key] = (g, st, out)
        g, st, front = self.front[key]
        st[0].copy_(tokens, non_blocking=True); st[1].copy_(token_lengths, non_blocking=True); st[2].copy_(length_scale.to(dev), non_blocking=True)
        if cond_in is not None:
            st[3].copy_(cond_in, non_blocking=True)
        if pro is not None:
            st[4].copy_(pro, non_blocking=True)
        g.replay()
        mel_lengths = front[3]
        tmax = int(mel_lengths.max())                                  # the one device -> host read of the call
        i = bisect.bisect_left(self.buckets, tmax)
        if i == len(self.buckets):
            raise ValueError(f"{tmax} frames exceed the largest mel bucket {self.buckets[-1]}")
        Tb = self.buckets[i]
        bkey = (key, Tb, noises is not None)
        if bkey not in self.back:
            ns_t = torch.tensor(float(noise_scale), device=dev)
            nz = torch.zeros(tokens.shape[0], int(m.hp.Sound.Mel_Dim), Tb, device=dev) if noises is not None else None
            fn = lambda: m.inference_back(front, Tb, ns_t, nz, prep=self.prep)
            g2, out2 = self._graph(fn)
            self.back[bkey] = (g2, ns_t, nz, out2)
        g2, ns_t, nz, (mels, lengths, attn) = self.back[bkey]
        ns_t.fill_(float(noise_scale))
        if noises is not None:
            nz.zero_()
            n = min(Tb, noises.shape[2])
            nz[:, :, :n].copy_(noises[:, :, :n], non_blocking=True)
        g2.replay()
        ns = int(m.hp.Decoder.Num_Squeeze)
        T = (tmax // ns) * ns
        return mels[:, :, :T], lengths, attn[:, :, :tmax]
