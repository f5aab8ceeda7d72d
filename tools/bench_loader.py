"""Training throughput THROUGH THE REAL LOADER: `Trainer.Train_Step` fed by `data.PatternDataset` + `data.Collater` from a
generated LJSpeech-shaped pattern directory (pattern pickles + METADATA.PICKLE + Token.yaml in the reference's formats) at the bench batch
size, next to the resident synthetic batch of bench.py.  Reports steps/s for num_workers = 0 (collation in the training process) and for
hp.Train.Num_Workers worker processes.
    python tools/bench_loader.py [--utterances 1024] [--steps 120] [--workers 0 4 8]"""
import argparse
import copy
import os
import pickle
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import yaml


def make_corpus(root, n, seed=0):
    from glow_tts_amd import data
    rng = np.random.default_rng(seed)
    letters = list("abcdefghijklmnopqrstuvwxyz .,?!'")
    token_dict = {"<S>": 0, "<E>": 1}
    for ch in letters:
        token_dict.setdefault(ch.upper() if ch.isalpha() else ch, len(token_dict))
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "Token.yaml"), "w") as f:
        yaml.dump(token_dict, f)
    for split, count in (("Train", n), ("Eval", 64)):
        d = os.path.join(root, split, "LJ")
        os.makedirs(d, exist_ok=True)
        for i in range(count):
            T = int(rng.integers(650, 801))                   # two mel buckets (768, 896), one token bucket: steady state after three captures
            text = "".join(letters[j] for j in rng.integers(0, len(letters), int(rng.integers(100, 118)))).upper()
            pat = {"Audio": np.zeros(4, np.float32), "Mel": rng.normal(0, 1.5, (T, 80)).clip(-4, 4).astype(np.float32), "Pitch": rng.random(T).astype(np.float32),
                   "Speaker_ID": 0, "Speaker": "LJ", "Dataset": "LJ", "Text": text}
            with open(os.path.join(d, f"LJ.{i:05d}.PICKLE"), "wb") as f:
                pickle.dump(pat, f, protocol=4)
        data.write_metadata(os.path.join(root, split), "METADATA.PICKLE")
    return token_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=512)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--workers", type=int, nargs="*", default=[0, 4, 8])
    args = ap.parse_args()
    from glow_tts_amd import hparams
    from glow_tts_amd.trainer import Trainer
    root = tempfile.mkdtemp(prefix="glowtts_corpus_")
    t_gen = time.time()
    token_dict = make_corpus(root, args.utterances)
    print(f"corpus of {args.utterances} utterances written in {time.time() - t_gen:.1f} s", flush=True)
    d = copy.deepcopy(hparams.load_yaml(hparams.DEFAULT_YAML))
    d["Mode"] = "Vanilla"
    d["Token_Path"] = os.path.join(root, "Token.yaml")
    d["Encoder"]["Embedding_Tokens"] = len(token_dict)
    for split, key in (("Train", "Train_Pattern"), ("Eval", "Eval_Pattern")):
        d["Train"][key].update(Path=os.path.join(root, split), Metadata_File="METADATA.PICKLE")
    d["Train"].update(Batch_Size=32, Max_Step=10 ** 9, Checkpoint_Save_Interval=10 ** 9, Logging_Interval=10 ** 9, Evaluation_Interval=10 ** 9)
    d["Checkpoint_Path"] = os.path.join(root, "Checkpoint")
    out = {}
    for nw in args.workers:
        tr = Trainer(steps=0, hp=hparams.Recursive_Parse(copy.deepcopy(d)), workers=nw)
        it, n, t0, frames = None, 0, None, 0
        warm = 12                                             # captures of the three shapes happen here
        while n < warm + args.steps:
            for batch in tr.dataLoader_Dict["Train"]:
                if n == warm:
                    torch.cuda.synchronize()
                    t0, frames = time.time(), 0
                tr.Train_Step(*batch)
                frames += int(batch[3].sum())
                n += 1
                if n >= warm + args.steps:
                    break
        torch.cuda.synchronize()
        dt = time.time() - t0
        out[nw] = dict(ms_per_step=1e3 * dt / args.steps, frames_per_s=frames / dt, shapes=len(tr._graphed.graphs))
        print(f"num_workers={nw}: {out[nw]['ms_per_step']:.2f} ms/step, {out[nw]['frames_per_s'] / 1e6:.2f} M valid mel-frames/s, "
              f"{out[nw]['shapes']} captured shapes", flush=True)
        del tr
    print("LOADER", out)


if __name__ == "__main__":
    main()
