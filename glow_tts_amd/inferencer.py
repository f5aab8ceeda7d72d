"""The reference's inference entry point (`Inference.py:111-313`: `Inferencer(checkpoint_path).Inference(labels, texts, scales, speakers,
references, inference_path)`), thin: text -> tokens (`Token.yaml`), optional reference mels for the prosody encoder / pitch tracks, `S_*.pt`
checkpoint (ActNorm marked initialised, `Inference.py:258-270`), `GlowTTS.inference` (replayed as hipGraphs through `graph_infer.GraphedInference`
where the mode allows it), one `.npy` mel per utterance (`Inference.py:213-220`; the PNG plots are out of scope).  `references` are either
`.npy` files / arrays holding mels [T, Mel] (what `Pattern_Generator.Pattern_Generate` produces) or wav files, which go through
`glow_tts_amd.audio` (librosa-free restatement of `Audio.py`; resampling is not implemented: the wav must be at hp.Sound.Sample_Rate)."""
import logging
import os

import numpy as np
import torch

from . import data
from .hparams import get_hp
from .modules import GlowTTS


class Inferencer:
    def __init__(self, checkpoint_path, hp=None, device="cuda:0"):
        self.hp = hp if hp is not None else get_hp()
        self.device = torch.device(device)
        self.token_Dict = data.load_token_dict(self.hp.Token_Path)
        self.Model_Generate()
        self.Load_Checkpoint(checkpoint_path)

    def Model_Generate(self):                                                    # Inference.py:116-136
        # Arithmetic of the inverse flow: exact fp32 (the reference's; mels within 2e-4 of it) unless the yaml asks otherwise with the optional key
        # `HIP_Inference_Precision`.  bf16 - the TRAINING default, `HIP_Precision` - leaves ~1e-2 rms / 0.3 worst-element error on the generated mel
        # (tests/longform_check.py "BF16 INVERSE MEL"), above the 1e-3 the path promises, and inference is not the throughput-critical half.
        import copy
        self.hp = copy.copy(self.hp)
        self.hp.HIP_Precision = str(getattr(self.hp, "HIP_Inference_Precision", "f32"))
        self.model_Dict = {"GlowTTS": GlowTTS(self.hp).to(self.device).eval()}
        self._graphed = None

    def Load_Checkpoint(self, checkpoint_path):                                  # Inference.py:258-275
        state_Dict = torch.load(checkpoint_path, map_location="cpu")
        self.model_Dict["GlowTTS"].load_state_dict(state_Dict["Model"])
        for flow in self.model_Dict["GlowTTS"].layer_Dict["Decoder"].layer_Dict["Flows"]:
            flow.layers[0].initialized = True      # Activation_Norm is already initialized when checkpoint is loaded.
        self.model_Dict["GlowTTS"].to(self.device)
        logging.info("Checkpoint loaded at {} steps.".format(state_Dict.get("Steps")))

    def _reference(self, ref):
        """-> (mel [T, Mel] float32, pitch [T] float32 or None)"""
        if isinstance(ref, (tuple, list)):
            return np.asarray(ref[0], np.float32), np.asarray(ref[1], np.float32)
        if isinstance(ref, np.ndarray):
            return ref.astype(np.float32), None
        if str(ref).lower().endswith(".npy"):
            return np.load(ref).astype(np.float32), None
        from . import audio
        return audio.pattern_from_wav(ref, self.hp, with_pitch=self.hp.Mode.upper() == "GR")       # GR: the YIN pitch track (Inference.py:56-58, Pattern_Generator.py:41-52)

    @torch.no_grad()
    def Inference_Step(self, tokens, token_lengths, prosodies, prosody_lengths, speakers, ge2es, pitches, pitch_lengths, length_scales, labels, texts,
                       start_index=0, tag_index=False, inference_path="./inference", noise_scale=1.0):
        """Inference.py:139-223 without the plots: returns the list of written .npy files."""
        dev = self.device
        mv = lambda t: t if t is None else t.to(dev)
        model = self.model_Dict["GlowTTS"]
        kw = dict(mels_for_prosody=mv(prosodies), mel_lengths_for_prosody=mv(prosody_lengths), speakers=mv(speakers), mels_for_ge2e=mv(ge2es))
        if "Pitch_Interpolater" in model.layer_Dict:                             # GR: the pitch track rides the eager path
            mels, mel_Lengths, attentions = model.inference(mv(tokens), mv(token_lengths), pitches=mv(pitches), pitch_lengths=mv(pitch_lengths),
                                                            noise_scale=noise_scale, length_scale=mv(length_scales), **kw)
        else:
            if self._graphed is None:
                from .graph_infer import GraphedInference
                self._graphed = GraphedInference(model)
            mels, mel_Lengths, attentions = self._graphed(mv(tokens), mv(token_lengths), noise_scale=noise_scale, length_scale=mv(length_scales), **kw)
        os.makedirs(os.path.join(inference_path, "NPY").replace("\\", "/"), exist_ok=True)
        files = []
        for index, (label, mel, n) in enumerate(zip(labels, mels.cpu().numpy(), mel_Lengths.cpu().tolist())):
            tags = [str(label)] + (["IDX_{}".format(index + start_index)] if tag_index else [])
            path = os.path.join(inference_path, "NPY", ".".join(tags) + ".npy").replace("\\", "/")
            np.save(path, mel[:, :n].T, allow_pickle=False)                      # [T, Mel], Inference.py:216-220
            files.append(path)
        return files

    def Inference(self, labels, texts, scales, speakers=None, references=None, inference_path="./inference", noise_scale=1.0):
        """Inference.py:225-256."""
        logging.info("Start inference.")
        hp = self.hp
        bs = hp.Inference_Batch_Size or hp.Train.Batch_Size
        speakers = speakers or [None] * len(texts)
        references = references or [None] * len(texts)
        files = []
        for s in range(0, len(texts), bs):
            sl = slice(s, s + bs)
            toks = [data.text_to_token(data.text_filtering(t), self.token_Dict) for t in texts[sl]]
            tl = torch.tensor([len(t) for t in toks], dtype=torch.int64)
            tokens = torch.full((len(toks), int(tl.max())), self.token_Dict["<E>"], dtype=torch.int64)
            for b, t in enumerate(toks):
                tokens[b, :len(t)] = torch.from_numpy(t.astype(np.int64))
            refs = references[sl]
            pros = pl = pit = pitl = None
            if not any(r is None for r in refs):                                 # Inference.py:84-100
                mp = [self._reference(r) for r in refs]
                pl = torch.tensor([m.shape[0] for m, _ in mp], dtype=torch.int64)
                pros = torch.full((len(mp), hp.Sound.Mel_Dim, int(pl.max())), -float(hp.Sound.Max_Abs_Mel))
                for b, (m, _) in enumerate(mp):
                    pros[b, :, :m.shape[0]] = torch.from_numpy(m.T)
                if all(p is not None for _, p in mp):
                    pitl = torch.tensor([p.shape[0] for _, p in mp], dtype=torch.int64)
                    pit = torch.zeros(len(mp), int(pitl.max()))
                    for b, (_, p) in enumerate(mp):
                        pit[b, :p.shape[0]] = torch.from_numpy(p)
            spk = None if any(x is None for x in speakers[sl]) else torch.tensor(speakers[sl], dtype=torch.int64)
            files += self.Inference_Step(tokens, tl, pros, pl, spk, None, pit, pitl, torch.tensor(scales[sl], dtype=torch.float32), labels[sl], texts[sl],
                                         start_index=s, inference_path=inference_path, noise_scale=noise_scale)
        return files


def main(argv=None):
    import argparse
    logging.basicConfig(level=logging.INFO)
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--checkpoint", required=True)
    ap.add_argument("-t", "--prompts", default=None, help="inference prompt TSV (Datasets.py:137-144); default: the reference's two example sentences")
    ap.add_argument("-o", "--out", default="./inference")
    args = ap.parse_args(argv)
    inf = Inferencer(checkpoint_path=args.checkpoint)
    if args.prompts:
        recs = [r for r in data.read_inference_prompts(args.prompts) if r["text"] is not None]
        labels, texts, scales, speakers = [r["label"] for r in recs], [r["text"] for r in recs], [r["length_scale"] for r in recs], [r["speaker"] for r in recs]
    else:                                                                         # Inference.py:290-301
        labels, texts, scales, speakers = ["Alpha", "Bravo"], ["Birds of a feather flock together.",
                                                               "A creative artist works on his next composition because he was not satisfied with his previous one."], [1.0, 0.9], [0, 1]
    mode = inf.hp.Mode.upper()
    print("\n".join(inf.Inference(labels, texts, scales, speakers if mode in ("SE", "GR") else None, None, args.out)))
