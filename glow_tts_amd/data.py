"""Batch layout on the host: the step immediately before the hot path.  `Collater` produces the reference's training
batch tuple (`Datasets.py:225-250`: tokens, token_lengths, mels [B, Mel, T], mel_lengths, speakers, mels_for_GE2E, pitches) with the
reference's padding (end-of-sequence token id, -Max_Abs_Mel, 0 for pitch; mels truncated to a multiple of Decoder.Num_Squeeze) and adds
what keeping an MI355X fed needs:
  * shape buckets: the padded lengths are rounded up to the next bucket, so a training run sees a handful of static shapes and every step
    can be a replayed hipGraph (glow_tts_amd.graph_step.GraphedTrainStep) instead of ~600 eager launches;
  * pinned, reused staging buffers: the host-to-device copy of a batch (8.2 MB of mels at B = 32) is one asynchronous DMA per tensor.
Pure host code (numpy / torch CPU tensors); the GE2E slice sampling (`Datasets.py:41-65`) is not part of this path: pass precomputed
d-vectors or leave `mels_for_GE2E` None."""
import bisect

import numpy as np
import torch


def _bucket(n, buckets, multiple):
    if buckets:
        i = bisect.bisect_left(buckets, n)
        if i == len(buckets):
            raise ValueError(f"length {n} exceeds the largest bucket {buckets[-1]}")
        return buckets[i]
    return -(-n // multiple) * multiple


class Collater:
    def __init__(self, num_squeeze=2, end_token_id=0, max_abs_mel=4.0, token_buckets=None, mel_buckets=None, token_multiple=1,
                 mel_multiple=None, pin_memory=False, ring=4):
        """token_buckets / mel_buckets: ascending lists of padded lengths (None: pad to the batch maximum rounded up to
        token_multiple / mel_multiple, the reference's behaviour for multiples of 1 / num_squeeze).  ring: number of pinned buffer sets that
        are cycled, i.e. how many batches may be in flight between the loader and the GPU copy."""
        self.ns, self.end, self.pad_mel = int(num_squeeze), int(end_token_id), -float(max_abs_mel)
        self.tb = sorted(token_buckets) if token_buckets else None
        self.mb = sorted(mel_buckets) if mel_buckets else None
        self.tm, self.mm = int(token_multiple), int(mel_multiple if mel_multiple else num_squeeze)
        if self.mb and any(b % self.ns for b in self.mb):
            raise ValueError("mel buckets must be multiples of Decoder.Num_Squeeze")
        self.pin, self.ring, self._bufs, self._turn = bool(pin_memory), int(ring), {}, 0

    @classmethod
    def from_hp(cls, hp, token_dict, **kw):
        """hp: the parsed Hyper_Parameters.yaml; token_dict: Token.yaml ({'<E>': id, ...}, Datasets.py:17-21)."""
        return cls(num_squeeze=hp.Decoder.Num_Squeeze, end_token_id=token_dict["<E>"], max_abs_mel=hp.Sound.Max_Abs_Mel, **kw)

    def _buffer(self, name, shape, dtype, fill):
        if not self.pin:
            return torch.full(shape, fill, dtype=dtype)
        key = (name, tuple(shape), self._turn % self.ring)
        buf = self._bufs.get(key)
        if buf is None:
            buf = self._bufs[key] = torch.empty(shape, dtype=dtype).pin_memory()
        buf.fill_(fill)
        return buf

    def __call__(self, batch):
        """batch: list of (token [Tt] int, mel [Tm, Mel] float, speaker int, pitch [Tm] float or None) like `Dataset.__getitem__`."""
        tokens, mels, speakers, pitches = zip(*batch)
        B = len(tokens)
        mels = [np.asarray(m)[:(len(m) // self.ns) * self.ns] for m in mels]                      # Datasets.py:230-233
        tl = [len(t) for t in tokens]
        ml = [len(m) for m in mels]
        Tt, Tm = _bucket(max(tl), self.tb, self.tm), _bucket(max(ml), self.mb, self.mm)
        mel_dim = mels[0].shape[1]
        out_tok = self._buffer("tok", (B, Tt), torch.int64, self.end)                              # Token_Stack, Datasets.py:23-30
        out_mel = self._buffer("mel", (B, mel_dim, Tm), torch.float32, self.pad_mel)               # Mel_Stack + transpose, :32-39, :244
        has_pitch = pitches[0] is not None
        out_pit = self._buffer("pit", (B, Tm), torch.float32, 0.0) if has_pitch else None          # Pitch_Stack, :67-74
        for b in range(B):
            out_tok[b, :tl[b]] = torch.as_tensor(np.asarray(tokens[b]), dtype=torch.int64)
            out_mel[b, :, :ml[b]] = torch.as_tensor(np.ascontiguousarray(mels[b].T), dtype=torch.float32)
            if has_pitch:
                n = min(len(pitches[b]), Tm)
                out_pit[b, :n] = torch.as_tensor(np.asarray(pitches[b][:n]), dtype=torch.float32)
        self._turn += 1
        return (out_tok, torch.tensor(tl, dtype=torch.int64), out_mel, torch.tensor(ml, dtype=torch.int64),
                torch.tensor(speakers, dtype=torch.int64), None, out_pit)


def to_device(batch, device, non_blocking=True):
    """Asynchronous host-to-device copy of a collated batch (pinned buffers make it a DMA that overlaps the running step)."""
    return tuple(t.to(device, non_blocking=non_blocking) if torch.is_tensor(t) else t for t in batch)
