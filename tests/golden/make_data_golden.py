"""Golden vectors for the batch layout: runs the UNMODIFIED reference `Datasets.Collater` / `Inference_Collater` helpers
(/root/reference/Datasets.py:17-74, 225-250) on a seeded ragged batch and stores inputs + the 7 output tensors in collater_case.npz.

Datasets.py cannot be imported as is: it opens hp.Token_Path at import time (a Windows path) and imports Pattern_Generator,
which needs librosa.  Neither touches the collater's arithmetic, so: CWD = a temp dir with a generated Hyper_Parameters.yaml whose
Token_Path points at a generated Token.yaml, and a stub `Pattern_Generator` module (two unused names) is registered before the import.
Only runs in the build container.  Usage: python tests/golden/make_data_golden.py"""
import os
import sys
import tempfile
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GLOWTTS_REFERENCE", "/root/reference")


def load_reference_datasets(token_dict):
    with open(os.path.join(REF, "Hyper_Parameters.yaml"), encoding="utf-8") as f:
        cfg = yaml.load(f, Loader=yaml.Loader)
    tmp = tempfile.mkdtemp(prefix="glowtts_ref_data_")
    with open(os.path.join(tmp, "Token.yaml"), "w") as f:
        yaml.dump(token_dict, f)
    cfg["Token_Path"] = os.path.join(tmp, "Token.yaml")
    with open(os.path.join(tmp, "Hyper_Parameters.yaml"), "w", encoding="utf-8") as f:
        yaml.dump(cfg, f)
    stub = types.ModuleType("Pattern_Generator")
    stub.Pattern_Generate = stub.Text_Filtering = None          # imported by name at Datasets.py:6, unused by the collaters
    sys.modules["Pattern_Generator"] = stub
    for name in ("Datasets", "Arg_Parser"):
        sys.modules.pop(name, None)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        import Datasets
    finally:
        os.chdir(cwd)
    return Datasets, cfg


def main():
    symbols = sorted(set("ABCDEFGHIJKLMNOPQRSTUVWXYZ,.?!'- "))
    token_dict = {tok: i for i, tok in enumerate(["<S>", "<E>"] + symbols)}
    D, cfg = load_reference_datasets(token_dict)
    rng = np.random.default_rng(20260929)
    texts = ["HELLO, WORLD.", "A", "MI THREE FIFTY FIVE X - IS IT FAST?", "DON'T STOP!", "FLOW."]
    mel_lens = [351, 133, 260, 98, 201]                        # odd lengths: truncated to a multiple of Decoder.Num_Squeeze (:230-233)
    tokens = [D.Text_to_Token(t) for t in texts]
    mels = [rng.normal(0, 1.5, (n, 80)).clip(-4, 4).astype(np.float32) for n in mel_lens]
    speakers = [3, 0, 108, 7, 7]
    pitches = [np.abs(rng.normal(0.5, 0.3, n)).astype(np.float32) for n in mel_lens]
    np.random.seed(777)                                         # Mel_for_GE2E_Stack draws offsets from numpy's global generator (:49)
    out = D.Collater()(list(zip(tokens, mels, speakers, pitches)))
    names = ["tokens", "token_lengths", "mels", "mel_lengths", "speakers", "mels_for_ge2e", "pitches"]
    data = {"out/" + n: t.numpy() for n, t in zip(names, out)}
    data.update({"in/mel_cat": np.concatenate(mels, 0), "in/mel_lens": np.array(mel_lens), "in/pitch_cat": np.concatenate(pitches),
                 "in/token_cat": np.concatenate(tokens), "in/token_lens": np.array([len(t) for t in tokens]), "in/speakers": np.array(speakers),
                 "in/texts": np.array(texts), "in/seed": np.int64(777),
                 "in/token_symbols": np.array(list(token_dict.keys())), "in/token_ids": np.array(list(token_dict.values()))})
    ge = cfg["Speaker_Embedding"]["GE2E"]["Inference"]
    data["in/ge2e"] = np.array([ge["Samples"], ge["Slice_Length"], ge["Overlap_Length"]])
    np.savez_compressed(os.path.join(HERE, "collater_case.npz"), **data)
    print("wrote collater_case.npz:", {k: v.shape for k, v in data.items() if k.startswith("out/")})


if __name__ == "__main__":
    main()
