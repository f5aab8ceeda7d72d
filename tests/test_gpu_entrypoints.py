"""GPU: the preserved entry points end to end on a synthetic pattern directory in the reference's on-disk formats (pattern pickles,
METADATA.PICKLE, Token.yaml, Hyper_Parameters.yaml schema): `Trainer(steps).Train()` (Train.py:49-598) trains a tiny model for a few steps through
the graphed Train_Step (two batch-shape buckets), writes `S_{steps}.pt` in the reference's checkpoint layout, a second Trainer resumes from it, and
`Inferencer(checkpoint).Inference(...)` (Inference.py:111-282) synthesises mels from text (and from reference mels in PE mode)."""
import os
import pickle

import numpy as np
import pytest
import torch

from helpers import tiny_hp_dict

pytestmark = pytest.mark.gpu

TEXTS = ["HELLO WORLD.", "GLOW T T S ON M I THREE FIFTY FIVE X!", "A FLOW IS A BIJECTION, IS IT NOT?", "SHORT ONE.", "MONOTONIC ALIGNMENT SEARCH.",
         "WAVE NET COUPLING LAYERS.", "THE QUICK BROWN FOX.", "JUMPS OVER THE LAZY DOG?"]


def _make_dataset(root, mode):
    from glow_tts_amd import data
    rng = np.random.default_rng(5)
    token_dict = data.make_token_dict(TEXTS)
    import yaml
    with open(os.path.join(root, "Token.yaml"), "w") as f:
        yaml.dump(token_dict, f)
    for split in ("Train", "Eval"):
        d = os.path.join(root, split, "LJ", "LJ")
        os.makedirs(d)
        for i, text in enumerate(TEXTS):
            T = int(rng.integers(4 * len(text), 6 * len(text)))
            pat = {"Audio": np.zeros(T * 4, np.float32), "Mel": rng.normal(0, 1.5, (T, 12)).clip(-4, 4).astype(np.float32),
                   "Pitch": rng.random(T).astype(np.float32), "Speaker_ID": i % 5, "Speaker": "LJ", "Dataset": "LJ", "Text": text}
            with open(os.path.join(d, f"LJ.{i:03d}.PICKLE"), "wb") as f:
                pickle.dump(pat, f, protocol=4)
        data.write_metadata(os.path.join(root, split), "METADATA.PICKLE")
    hp = tiny_hp_dict(mode)
    hp["HIP_Precision"] = "f32"
    hp["Token_Path"] = os.path.join(root, "Token.yaml")
    hp["Encoder"]["Embedding_Tokens"] = len(token_dict)
    for split, key in (("Train", "Train_Pattern"), ("Eval", "Eval_Pattern")):
        hp["Train"][key].update(Path=os.path.join(root, split), Metadata_File="METADATA.PICKLE")
        hp["Train"][key]["Mel_Length"] = {"Min": 2, "Max": 400}
        hp["Train"][key]["Text_Length"] = {"Min": 1, "Max": 60}
    hp["Train"].update(Batch_Size=4, Max_Step=9, Checkpoint_Save_Interval=4, Logging_Interval=3, Evaluation_Interval=100, Use_Pattern_Cache=True)
    hp["Checkpoint_Path"] = os.path.join(root, "Checkpoint")
    hp["Inference_Batch_Size"] = 3
    hp["HIP_Buckets"] = {"Mel": [128, 256, 384], "Token": [32, 64]}
    return hp


@pytest.mark.parametrize("mode", ["Vanilla", "PE"])
def test_train_resume_and_inference(mode, tmp_path):
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.inferencer import Inferencer
    from glow_tts_amd.trainer import Trainer
    hp = Recursive_Parse(_make_dataset(str(tmp_path), mode))
    torch.manual_seed(0)
    tr = Trainer(steps=0, hp=hp)
    tr.Train()
    assert tr.steps >= hp.Train.Max_Step
    ckpts = sorted(os.listdir(hp.Checkpoint_Path))
    assert any(f.startswith("S_") and f.endswith(".pt") for f in ckpts), ckpts
    tr.Save_Checkpoint()
    state = torch.load(os.path.join(hp.Checkpoint_Path, f"S_{tr.steps}.pt"), map_location="cpu")
    assert set(state) >= {"Model", "Optimizer", "Scheduler", "Steps", "Epochs"} and state["Steps"] == tr.steps      # Train.py:538-544
    assert all(torch.isfinite(v).all() for v in state["Model"].values() if v.is_floating_point())
    # optimizer and trainer step counters agree (warm-up steps of a new batch shape are real steps, GraphedTrainStep.steps_taken)
    some = next(iter(tr.optimizer.state.values()))
    assert some["step"] == tr.steps
    # resume: -s <steps> (Train.py:499-533)
    hp.Train.Max_Step = tr.steps + 3
    tr2 = Trainer(steps=tr.steps, hp=hp)
    assert tr2.steps == tr.steps and all(f.layers[0].initialized for f in tr2.model_Dict["GlowTTS"].layer_Dict["Decoder"].layer_Dict["Flows"])
    tr2.Train()
    assert tr2.steps >= hp.Train.Max_Step
    # inference from the checkpoint
    inf = Inferencer(os.path.join(hp.Checkpoint_Path, f"S_{tr.steps}.pt"), hp=hp)
    refs = None
    if mode == "PE":
        rng = np.random.default_rng(1)
        refs = [rng.normal(0, 1.5, (n, 12)).clip(-4, 4).astype(np.float32) for n in (90, 64, 120)]
    files = inf.Inference(["Alpha", "Bravo", "Charlie"], ["Hello world.", "The quick brown fox jumps.", "A flow?"], [1.0, 0.9, 1.3], None, refs,
                          inference_path=str(tmp_path / "out"))
    assert len(files) == 3
    for f in files:
        mel = np.load(f)
        assert mel.ndim == 2 and mel.shape[1] == 12 and mel.shape[0] >= 2 and np.isfinite(mel).all()
