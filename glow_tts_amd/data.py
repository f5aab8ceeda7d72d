"""Batch layout on the host: the step immediately before the hot path.  `Collater` produces the reference's training
batch tuple (`Datasets.py:225-250`: tokens, token_lengths, mels [B, Mel, T], mel_lengths, speakers, mels_for_GE2E, pitches) with the
reference's padding (end-of-sequence token id, -Max_Abs_Mel, 0 for pitch; mels truncated to a multiple of Decoder.Num_Squeeze) and adds
what keeping an MI355X fed needs:
  * shape buckets: the padded lengths are rounded up to the next bucket, so a training run sees a handful of static shapes and every step
    can be a replayed hipGraph (glow_tts_amd.graph_step.GraphedTrainStep) instead of ~600 eager launches;
  * pinned, reused staging buffers: the host-to-device copy of a batch (8.2 MB of mels at B = 32) is one asynchronous DMA per tensor.
Pure host code (numpy / torch CPU tensors).

On-disk formats of the reference, read here so that pattern directories written by its `Pattern_Generator.py` drop in unchanged:
  * `Token.yaml` (`Pattern_Generator.py:401-411`): {'<S>': 0, '<E>': 1, letter: id, ...}  -> `load_token_dict`, `text_to_token`;
  * pattern pickles (`Pattern_Generator.py:87-96`): {'Audio','Mel' [T, Mel],'Pitch' [T],'Speaker_ID','Speaker','Dataset','Text'};
  * `METADATA.PICKLE` (`Pattern_Generator.py:344-396`): 'File_List', '*_Length_Dict', 'File_List_by_Speaker_Dict', ... -> `PatternDataset`,
    `write_metadata` (the same keys, for pattern directories produced elsewhere);
  * the inference prompt TSV (`Datasets.py:137-144`) -> `read_inference_prompts`.
The audio front-end (wav -> mel / pitch, `Audio.py`, `yin.py`) stays out of scope: it needs librosa, which this image does not have."""
import bisect
import math
import os
import pickle
import re

import numpy as np
import torch

PATTERN_KEYS = ("Audio", "Mel", "Pitch", "Speaker_ID", "Speaker", "Dataset", "Text")
_TEXT_OK = re.compile(r"[A-Z,.?!'\-\s]+")                      # Pattern_Generator.py:19


def load_token_dict(path):
    """Token.yaml -> {symbol: id}."""
    import yaml
    with open(path, encoding="utf-8") as f:
        return yaml.load(f, Loader=yaml.Loader)


def make_token_dict(texts):
    """The dictionary `Token_Dict_Generate` writes (`Pattern_Generator.py:401-411`): '<S>', '<E>', then the sorted symbol set."""
    symbols = set()
    for t in texts:
        symbols |= set(t)
    return {tok: i for i, tok in enumerate(["<S>", "<E>"] + sorted(symbols))}


def text_filtering(text):
    """`Text_Filtering` (`Pattern_Generator.py:22-39`): upper-case, drop ()"[]:; , tidy spaces; None when the text holds anything
    outside [A-Z,.?!'- and whitespace] or starts with an apostrophe."""
    text = text.upper().strip()
    for ch in ("(", ")", '"', "[", "]", ":", ";"):
        text = text.replace(ch, "")
    for a, b in (("  ", " "), (" ,", ","), ("' ", "'")):
        text = text.replace(a, b)
    text = text.strip()
    found = _TEXT_OK.findall(text)
    if len(found) != 1 or text.startswith("'"):
        return None
    return found[0]


def text_to_token(text, token_dict):
    """`Text_to_Token` (`Datasets.py:17-21`): <S> + letters + <E> as int32 ids (KeyError on a symbol the dictionary lacks)."""
    return np.array([token_dict[c] for c in ["<S>"] + list(text) + ["<E>"]], dtype=np.int32)


class PatternDataset(torch.utils.data.Dataset):
    """`Datasets.Dataset` (`Datasets.py:78-131`): items are (token ids, mel [T, Mel], speaker id, pitch [T]) read from the pattern pickles
    listed in the metadata pickle, filtered by mel / text length, optionally repeated `accumulated_dataset_epoch` times and cached."""

    def __init__(self, pattern_path, metadata_file, token_dict, accumulated_dataset_epoch=1, mel_length_min=-math.inf, mel_length_max=math.inf,
                 text_length_min=-math.inf, text_length_max=math.inf, use_cache=False):
        self.pattern_path, self.token_dict, self.use_cache = pattern_path, token_dict, use_cache
        with open(os.path.join(pattern_path, metadata_file).replace("\\", "/"), "rb") as f:
            meta = pickle.load(f)
        mel_len, text_len = meta["Mel_Length_Dict"], meta["Text_Length_Dict"]
        self.files = [x for x in meta["File_List"]
                      if mel_length_min <= mel_len[x] <= mel_length_max and text_length_min <= text_len[x] <= text_length_max]
        self.base_length = len(self.files)
        self.files = self.files * int(accumulated_dataset_epoch)
        self.mel_lengths = [mel_len[x] for x in self.files]        # for length-bucketed samplers (not in the reference)
        self._cache = {}

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        key = idx % self.base_length
        if key in self._cache:
            return self._cache[key]
        with open(os.path.join(self.pattern_path, self.files[idx]).replace("\\", "/"), "rb") as f:
            pat = pickle.load(f)
        item = text_to_token(pat["Text"], self.token_dict), pat["Mel"], pat["Speaker_ID"], pat["Pitch"]
        if self.use_cache:
            self._cache[key] = item
        return item


def write_metadata(pattern_path, metadata_file, hp=None, use_text=True):
    """Scans a pattern directory and writes the metadata pickle with the reference's keys (`Metadata_Generate`,
    `Pattern_Generator.py:335-396`); files whose keys are not pattern keys are skipped.  Returns the dict."""
    meta = {"File_List": [], "Audio_Length_Dict": {}, "Mel_Length_Dict": {}, "Pitch_Length_Dict": {}, "Speaker_ID_Dict": {},
            "Speaker_Dict": {}, "Dataset_Dict": {}, "File_List_by_Speaker_Dict": {}}
    if hp is not None:
        s = hp.Sound
        meta.update({"Spectrogram_Dim": s.Spectrogram_Dim, "Mel_Dim": s.Mel_Dim, "Frame_Shift": s.Frame_Shift, "Frame_Length": s.Frame_Length,
                     "Sample_Rate": s.Sample_Rate, "Max_Abs_Mel": s.Max_Abs_Mel})
    if use_text:
        meta["Text_Length_Dict"] = {}
    target = os.path.join(pattern_path, metadata_file.upper())
    for root, _, files in sorted(os.walk(pattern_path)):
        for name in sorted(files):
            full = os.path.join(root, name)
            if os.path.abspath(full) == os.path.abspath(target):
                continue
            try:
                with open(full, "rb") as f:
                    pat = pickle.load(f)
                if not isinstance(pat, dict) or any(k not in PATTERN_KEYS for k in pat) or (use_text and "Text" not in pat):
                    continue
                rel = os.path.relpath(full, pattern_path).replace("\\", "/")
                meta["Audio_Length_Dict"][rel] = pat["Audio"].shape[0]
                meta["Mel_Length_Dict"][rel] = pat["Mel"].shape[0]
                meta["Pitch_Length_Dict"][rel] = pat["Pitch"].shape[0]
                meta["Speaker_ID_Dict"][rel] = pat["Speaker_ID"]
                meta["Speaker_Dict"][rel] = pat["Speaker"]
                meta["Dataset_Dict"][rel] = pat["Dataset"]
                meta["File_List"].append(rel)
                meta["File_List_by_Speaker_Dict"].setdefault(pat["Speaker"], []).append(rel)
                if use_text:
                    meta["Text_Length_Dict"][rel] = len(pat["Text"])
            except Exception:                                  # noqa: BLE001 - the reference ignores unreadable files too (:390-391)
                continue
    with open(target, "wb") as f:
        pickle.dump(meta, f, protocol=4)
    return meta


def mels_for_ge2e(mels, samples, slice_length, overlap_length, rng=np.random):
    """`Mel_for_GE2E_Stack` (`Datasets.py:41-65`): per utterance a window of samples * (slice - overlap) + overlap frames (random offset, or
    reflect-padded when the mel is shorter) cut into `samples` overlapping slices -> [B * samples, slice_length, Mel]."""
    hop = slice_length - overlap_length
    need = samples * hop + overlap_length
    out = []
    for mel in mels:
        mel = np.asarray(mel)
        if mel.shape[0] > need:
            off = rng.randint(0, mel.shape[0] - need)
            mel = mel[off:off + need]
        else:
            pad = (need - mel.shape[0]) / 2
            mel = np.pad(mel, [[int(np.floor(pad)), int(np.ceil(pad))], [0, 0]], mode="reflect")
        out.append(np.stack([mel[i:i + slice_length] for i in range(0, need - overlap_length, hop)]))
    return np.vstack(out)


def read_inference_prompts(path, token_dict=None):
    """The inference TSV (`Datasets.py:137-144`): a header line, then
    label, text, length scale, speaker id, wav for GE2E, wav for prosody, wav for pitch.  Text goes through `text_filtering`; with a token
    dictionary every record also carries its token ids."""
    out = []
    with open(path, "r", encoding="utf-8") as f:
        for line in f.readlines()[1:]:
            if not line.strip():
                continue
            label, text, scale, speaker, w_ge2e, w_pro, w_pitch = [x.strip() for x in line.strip().split("\t")]
            text = text_filtering(text)
            rec = {"label": label, "text": text, "length_scale": float(scale), "speaker": int(speaker),
                   "wav_for_ge2e": w_ge2e, "wav_for_prosody": w_pro, "wav_for_pitch": w_pitch}
            if token_dict is not None and text is not None:
                rec["token"] = text_to_token(text, token_dict)
            out.append(rec)
    return out



class InferenceDataset(torch.utils.data.Dataset):
    """`Datasets.Inference_Dataset` (Datasets.py:131-165): the prompts of the inference TSV (`hp.Train.Inference_Pattern_File_in_Train`) as
    (token, length scale, speaker, mel for GE2E, mel for prosody, pitch, label, text).  The reference runs `Pattern_Generate` on the three wav
    columns of every record whatever the Mode; here a reference wav is only decoded when the Mode reads it (GE2E d-vectors / PE-GR prosody /
    GR pitch) - the other entries are one-frame placeholders that `Trainer.Inference_Step` drops, so a Vanilla run needs no wav files."""

    def __init__(self, pattern_path, token_dict, hp, use_cache=False):
        self.records = [r for r in read_inference_prompts(pattern_path, token_dict) if r.get("token") is not None]
        self.hp, self.use_cache, self.cache = hp, use_cache, {}
        mode = hp.Mode.upper()
        self.need_ge2e = mode in ("SE", "GR") and hp.Speaker_Embedding.Type.upper() == "GE2E"
        self.need_prosody = mode in ("PE", "GR")
        self.need_pitch = mode == "GR"

    def __len__(self):
        return len(self.records)

    def __getitem__(self, idx):
        if idx in self.cache:
            return self.cache[idx]
        r = self.records[idx]
        mel_dim = int(self.hp.Sound.Mel_Dim)
        blank_mel, blank_pitch = np.full((1, mel_dim), -float(self.hp.Sound.Max_Abs_Mel), np.float32), np.zeros(1, np.float32)

        def ref(path, with_pitch=False):
            if str(path).lower().endswith(".npy"):             # a pre-computed mel [T, Mel] (as `Inferencer` accepts); no pitch track
                return np.load(path).astype(np.float32), blank_pitch
            from . import audio
            return audio.pattern_from_wav(path, self.hp, with_pitch=with_pitch)
        mel_ge2e = ref(r["wav_for_ge2e"])[0] if self.need_ge2e else blank_mel
        mel_pro = ref(r["wav_for_prosody"])[0] if self.need_prosody else blank_mel
        pitch = ref(r["wav_for_pitch"], with_pitch=True)[1] if self.need_pitch else blank_pitch
        pattern = (r["token"], r["length_scale"], r["speaker"], mel_ge2e, mel_pro, pitch, r["label"], r["text"])
        if self.use_cache:
            self.cache[idx] = pattern
        return pattern


class InferenceCollater:
    """`Datasets.Inference_Collater` (Datasets.py:252-275): -> tokens [B, Tt], token_lengths, mels_for_prosody [B, Mel, T], its lengths, speakers,
    mels_for_ge2e (the slice stack of `mels_for_ge2e`, [B * Samples, Mel, Slice]), pitches [B, T], pitch_lengths, length_scales, labels, texts."""

    def __init__(self, token_dict, hp):
        self.end, self.hp = int(token_dict["<E>"]), hp
        self.need_ge2e = hp.Mode.upper() in ("SE", "GR") and hp.Speaker_Embedding.Type.upper() == "GE2E"

    def __call__(self, batch):
        tokens, scales, speakers, mels_ge2e, mels_pro, pitches, labels, texts = zip(*batch)
        hp = self.hp
        tl = [t.shape[0] for t in tokens]
        tok = np.stack([np.pad(t, [0, max(tl) - t.shape[0]], constant_values=self.end) for t in tokens])
        pl = [m.shape[0] for m in mels_pro]
        pro = np.stack([np.pad(m, [[0, max(pl) - m.shape[0]], [0, 0]], constant_values=-float(hp.Sound.Max_Abs_Mel)) for m in mels_pro])
        pil = [p.shape[0] for p in pitches]
        pit = np.stack([np.pad(p, [0, max(pil) - p.shape[0]], constant_values=0.0) for p in pitches])
        ge = hp.Speaker_Embedding.GE2E.Inference
        ge2e = mels_for_ge2e(list(mels_ge2e), int(ge.Samples), int(ge.Slice_Length), int(ge.Overlap_Length)) if self.need_ge2e \
            else np.zeros((len(batch), 1, int(hp.Sound.Mel_Dim)), np.float32)
        return (torch.from_numpy(tok.astype(np.int64)), torch.tensor(tl, dtype=torch.int64), torch.from_numpy(pro.astype(np.float32)).transpose(2, 1).contiguous(),
                torch.tensor(pl, dtype=torch.int64), torch.tensor(speakers, dtype=torch.int64), torch.from_numpy(np.asarray(ge2e, np.float32)).transpose(2, 1).contiguous(),
                torch.from_numpy(pit.astype(np.float32)), torch.tensor(pil, dtype=torch.int64), torch.tensor(scales, dtype=torch.float32), list(labels), list(texts))


def _bucket(n, buckets, multiple):
    if buckets:
        i = bisect.bisect_left(buckets, n)
        if i == len(buckets):
            raise ValueError(f"length {n} exceeds the largest bucket {buckets[-1]}")
        return buckets[i]
    return -(-n // multiple) * multiple


class Collater:
    def __init__(self, num_squeeze=2, end_token_id=1, max_abs_mel=4.0, token_buckets=None, mel_buckets=None, token_multiple=1,
                 mel_multiple=None, pin_memory=False, ring=4, ge2e=None):
        """end_token_id: the id of '<E>' in Token.yaml (1 in every dictionary `Token_Dict_Generate` writes: '<S>' is 0; `from_hp` reads it
        from the dictionary).  token_buckets / mel_buckets: ascending lists of padded lengths (None: pad to the batch maximum rounded up to
        token_multiple / mel_multiple, the reference's behaviour for multiples of 1 / num_squeeze).  ring: number of pinned buffer sets that
        are cycled, i.e. how many batches may be in flight between the loader and the GPU copy; copy batches with `self.to_device`, which
        records an event per set so that a set is not refilled while its copy is still in flight.
        Pitches: without mel buckets they are padded like the reference's `Pitch_Stack` (to the longest UNtruncated track, so one frame
        longer than the mels when that utterance has an odd length; `Squeeze` drops the odd frame, Modules.py:897-898); with mel buckets to
        the mel bucket."""
        self.ns, self.end, self.pad_mel = int(num_squeeze), int(end_token_id), -float(max_abs_mel)
        self.tb = sorted(token_buckets) if token_buckets else None
        self.mb = sorted(mel_buckets) if mel_buckets else None
        self.tm, self.mm = int(token_multiple), int(mel_multiple if mel_multiple else num_squeeze)
        if self.mb and any(b % self.ns for b in self.mb):
            raise ValueError("mel buckets must be multiples of Decoder.Num_Squeeze")
        self.pin, self.ring, self._bufs, self._turn = bool(pin_memory), int(ring), {}, 0
        self._events = {}           # ring slot -> event recorded after the last host-to-device copy out of that slot
        self.ge2e = ge2e            # (samples, slice_length, overlap_length) of hp.Speaker_Embedding.GE2E.Inference, or None: no GE2E slices

    @classmethod
    def from_hp(cls, hp, token_dict, **kw):
        """hp: the parsed Hyper_Parameters.yaml; token_dict: Token.yaml ({'<E>': id, ...}, Datasets.py:17-21)."""
        return cls(num_squeeze=hp.Decoder.Num_Squeeze, end_token_id=token_dict["<E>"], max_abs_mel=hp.Sound.Max_Abs_Mel, **kw)

    def _buffer(self, name, shape, dtype, fill=None):
        """fill=None: the caller writes every element itself (data + padding per row: a batch is ~9 MB, filling it first doubles the traffic)."""
        if not self.pin:
            return torch.empty(shape, dtype=dtype) if fill is None else torch.full(shape, fill, dtype=dtype)
        slot = self._turn % self.ring
        ev = self._events.get(slot)
        if ev is not None:
            ev.synchronize()                                   # the copy that last read this buffer set has finished
        key = (name, tuple(shape), slot)
        buf = self._bufs.get(key)
        if buf is None:
            buf = self._bufs[key] = torch.empty(shape, dtype=dtype).pin_memory()
        if fill is not None:
            buf.fill_(fill)
        return buf

    def to_device(self, batch, device, non_blocking=True):
        """Host-to-device copy of the batch this collater produced LAST; with pinned buffers it records the event that guards the set."""
        out = to_device(batch, device, non_blocking)
        if self.pin and torch.cuda.is_available():
            ev = torch.cuda.Event()
            ev.record()
            self._events[(self._turn - 1) % self.ring] = ev
        return out

    def __call__(self, batch):
        """batch: list of (token [Tt] int, mel [Tm, Mel] float, speaker int, pitch [Tm] float or None) like `Dataset.__getitem__`."""
        tokens, mels, speakers, pitches = zip(*batch)
        B = len(tokens)
        ge2e = None
        if self.ge2e is not None:                                                                  # on the UNtruncated mels, Datasets.py:229
            ge2e = torch.from_numpy(np.ascontiguousarray(mels_for_ge2e(mels, *self.ge2e).astype(np.float32))).transpose(2, 1)
        mels = [np.asarray(m)[:(len(m) // self.ns) * self.ns] for m in mels]                      # Datasets.py:230-233
        tl = [len(t) for t in tokens]
        ml = [len(m) for m in mels]
        Tt, Tm = _bucket(max(tl), self.tb, self.tm), _bucket(max(ml), self.mb, self.mm)
        mel_dim = mels[0].shape[1]
        out_tok = self._buffer("tok", (B, Tt), torch.int64)                                        # Token_Stack, Datasets.py:23-30
        out_mel = self._buffer("mel", (B, mel_dim, Tm), torch.float32)                             # Mel_Stack + transpose, :32-39, :244
        has_pitch = pitches[0] is not None
        Tp = Tm if (self.mb or not has_pitch) else max(Tm, max(len(p) for p in pitches))             # Pitch_Stack pads to the longest track
        out_pit = self._buffer("pit", (B, Tp), torch.float32) if has_pitch else None               # Pitch_Stack, :67-74
        # rows are written through numpy views of the (pinned) buffers - data, then the row's own padding: one strided C loop per utterance
        # instead of torch's indexing machinery per row (24 -> ~6 ms per 32 x 800-frame batch on one core)
        tok_np, mel_np = out_tok.numpy(), out_mel.numpy()
        pit_np = out_pit.numpy() if has_pitch else None
        for b in range(B):
            tok_np[b, :tl[b]] = tokens[b]
            tok_np[b, tl[b]:] = self.end
            mel_np[b, :, :ml[b]] = mels[b].T
            mel_np[b, :, ml[b]:] = self.pad_mel
            if has_pitch:
                n = min(len(pitches[b]), Tp)
                pit_np[b, :n] = pitches[b][:n]
                pit_np[b, n:] = 0.0
        self._turn += 1
        return (out_tok, torch.tensor(tl, dtype=torch.int64), out_mel, torch.tensor(ml, dtype=torch.int64),
                torch.tensor(speakers, dtype=torch.int64), ge2e, out_pit)


def to_device(batch, device, non_blocking=True):
    """Asynchronous host-to-device copy of a collated batch (pinned buffers make it a DMA that overlaps the running step)."""
    return tuple(t.to(device, non_blocking=non_blocking) if torch.is_tensor(t) else t for t in batch)
