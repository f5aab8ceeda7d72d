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

pytestmark = [pytest.mark.gpu, pytest.mark.late(1)]

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
    # the prompts `Trainer.Inference_Epoch` synthesises every Inference_Interval steps (Train.py:91-93, 259-260, 445-461); reference mels as .npy
    np.save(os.path.join(root, "ref.npy"), rng.normal(0, 1.5, (57, 12)).clip(-4, 4).astype(np.float32))
    with open(os.path.join(root, "prompts.txt"), "w") as f:
        f.write("Label\tText\tLength_Scale\tSpeaker\tGE2E\tProsody\tPitch\n")
        for i, text in enumerate(TEXTS[:4]):
            f.write(f"P{i}\t{text}\t{1.0 + 0.1 * i}\t{i % 5}\t{root}/ref.npy\t{root}/ref.npy\t{root}/ref.npy\n")
    hp["Train"].update(Inference_Interval=5, Inference_Pattern_File_in_Train=os.path.join(root, "prompts.txt"))
    hp["Checkpoint_Path"] = os.path.join(root, "Checkpoint")
    hp["Inference_Batch_Size"] = 3
    hp["Inference_Path"] = os.path.join(root, "Inference")
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
    # Inference_Epoch ran when the step counter crossed Inference_Interval = 5: one .npy per prompt, batches of Inference_Batch_Size = 3
    got = sorted(os.listdir(os.path.join(hp.Inference_Path, "Step-5", "NPY")))
    assert got == [f"P{i}.npy" for i in range(4)], got
    for f in got:
        m = np.load(os.path.join(hp.Inference_Path, "Step-5", "NPY", f))
        assert m.ndim == 2 and m.shape[1] == 12 and m.shape[0] >= 1 and np.isfinite(m).all()
    # the evaluation epoch ran forward AND inference on the dev batches (Train.py:279-316); Inference_Step names its files like the reference
    mp, a_train, a_inf, _ = tr.last_Evaluation
    assert torch.isfinite(mp).all() and a_train.shape[:2] == a_inf.shape[:2] and mp.shape[1] == 12
    batch = next(iter(tr.dataLoader_Dict["Dev"]))
    tokens, tl, mels_b, ml_b, spk_b, ge2e_b, pit_b = batch
    nb = tokens.shape[0]
    files = tr.Inference_Step(tokens, tl, mels_b, ml_b, spk_b, ge2e_b, pit_b, ml_b, torch.ones(nb), [f"L{i}" for i in range(nb)], ["text"] * nb,
                              start_index=3, tag_step=True, tag_index=True)
    assert files[0] == f"Step-{tr.steps}.L0.IDX_3"
    for f in files:
        m = np.load(os.path.join(hp.Inference_Path, f"Step-{tr.steps}", "NPY", f + ".npy"))
        assert m.ndim == 2 and m.shape[1] == 12 and np.isfinite(m).all()
    ckpts = sorted(os.listdir(hp.Checkpoint_Path))
    assert any(f.startswith("S_") and f.endswith(".pt") for f in ckpts), ckpts
    tr.Save_Checkpoint()
    state = torch.load(os.path.join(hp.Checkpoint_Path, f"S_{tr.steps}.pt"), map_location="cpu")
    assert set(state) >= {"Model", "Optimizer", "Scheduler", "Steps", "Epochs"} and state["Steps"] == tr.steps      # Train.py:538-544
    assert all(torch.isfinite(v).all() for v in state["Model"].values() if v.is_floating_point())
    # optimizer and trainer step counters agree (a new batch shape costs dry warm-up passes, never extra optimizer steps)
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


def test_ge2e_mode_trains_and_runs_its_inference_epoch(tmp_path):
    """SE mode with GE2E d-vectors (config 4's conditioning): the loaders hand over the raw slice stack [B * Samples, Mel, Slice] and BOTH the
    training batches and the prompts of `Inference_Epoch` must go through `Trainer(speaker_encoder=...)` before the model sees them as d-vectors
    [B, Embedding_Size] (Modules.py:75-77, 154-156 run the GE2E network inside the model; here it is a caller-supplied callable)."""
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.trainer import Trainer
    d = _make_dataset(str(tmp_path), "SE")
    d["Speaker_Embedding"]["Type"] = "GE2E"
    d["Speaker_Embedding"]["GE2E"]["Inference"] = {"Samples": 3, "Slice_Length": 16, "Overlap_Length": 8}
    d["Train"].update(Max_Step=6, Inference_Interval=5, Checkpoint_Save_Interval=100)
    hp = Recursive_Parse(d)
    seen = []

    def speaker_encoder(stack):                                  # a stand-in for the un-vendored GE2E network: [B * 3, 12, 16] -> unit d-vectors [B, 16]
        assert stack.dim() == 3 and stack.shape[0] % 3 == 0 and tuple(stack.shape[1:]) == (12, 16), tuple(stack.shape)
        seen.append(stack.shape[0] // 3)
        v = stack.reshape(-1, 3, 12 * 16).mean(1)[:, :16]
        return torch.nn.functional.normalize(v, dim=1)
    torch.manual_seed(0)
    tr = Trainer(steps=0, hp=hp, speaker_encoder=speaker_encoder)
    assert "LUT" not in tr.model_Dict["GlowTTS"].layer_Dict
    tr.Train()
    assert tr.steps >= 6
    got = sorted(os.listdir(os.path.join(hp.Inference_Path, "Step-5", "NPY")))
    assert got == [f"P{i}.npy" for i in range(4)], got
    for f in got:
        m = np.load(os.path.join(hp.Inference_Path, "Step-5", "NPY", f))
        assert m.ndim == 2 and m.shape[1] == 12 and np.isfinite(m).all()
    assert 3 in seen and 1 in seen                               # the prompts' batches (Inference_Batch_Size = 3: 3 + 1) went through the encoder too
    with pytest.raises(RuntimeError, match="GE2E"):
        tr.speaker_encoder = None
        tr.Inference_Epoch()


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_graphed_training_reduces_the_loss(precision):
    """120 replayed Train_Steps (forward, MLE + duration loss, backward, clip, RAdam, Noam schedule - all inside the captured hipGraph) on one
    fixed batch of the tiny golden model: the negative log-likelihood must fall substantially and stay finite, and the graphed run must track
    an eager run of the same sequence (same seeds, dropout off) - a stale job table, hyper-parameter word or gradient buffer would show here."""
    from helpers import load_case
    from glow_tts_amd.graph_step import GraphedTrainStep
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS, MLE_Loss
    from glow_tts_amd.optim import Modified_Noam_Scheduler, RAdam, clip_grad_norm_
    sd, _, r = load_case("tiny_vanilla.npz")
    hp = tiny_hp_dict("Vanilla")
    hp["HIP_Precision"] = precision
    for k in ("Prenet", "Transformer", "Duration_Predictor"):
        hp["Encoder"][k]["Dropout_Rate"] = 0.0
    hp["Decoder"]["Affine_Coupling"]["WaveNet"]["Dropout_Rate"] = 0.0
    t = lambda k: torch.from_numpy(r[k]).cuda()
    batch = (t("tokens"), t("token_lengths"), t("mels"), t("mel_lengths"))

    def make():
        m = GlowTTS(Recursive_Parse(hp))
        m.load_state_dict(sd)
        for f in m.layer_Dict["Decoder"].layer_Dict["Flows"]:
            f.layers[0].initialized = True
        m = m.cuda().train()
        opt = RAdam(m.parameters(), lr=2e-3, eps=1e-6, weight_decay=1e-6)
        return m, opt, Modified_Noam_Scheduler(opt, base=4000), MLE_Loss(m.hp)

    def loss_fn_for(mle):
        def loss_fn(m, tokens, tl, mels, ml):
            z, mm, ms, ld, dur, durt, _, _ = m(tokens, tl, mels, ml, None, None, None)
            return mle(z=z, mean=mm, std=ms, log_dets=ld, lengths=ml) + torch.nn.functional.mse_loss(dur, durt)
        return loss_fn
    steps = 120
    mg2, og2, sg2, mle2 = make()
    step2 = GraphedTrainStep(mg2, loss_fn_for(mle2), warmup=2, optimizer=og2, scheduler=sg2, max_grad_norm=5.0)
    curve = []
    for _ in range(steps):
        curve.append(float(step2(*batch).detach()))
    me, oe, se, mlee = make()
    lf = loss_fn_for(mlee)
    eager = []
    for _ in range(steps):
        me.zero_grad(set_to_none=True)
        l = lf(me, *batch)
        l.backward()
        clip_grad_norm_(list(me.parameters()), 5.0)
        oe.step(); se.step()
        eager.append(float(l.detach()))
    assert all(np.isfinite(curve)) and all(np.isfinite(eager))
    assert curve[-1] < curve[0] - 0.5, (curve[0], curve[-1])             # it learns
    assert step2.steps_taken == steps
    tol = 2e-3 if precision == "f32" else 5e-2
    assert abs(curve[-1] - eager[-1]) <= tol * max(1.0, abs(eager[-1])), (curve[-1], eager[-1])      # graphed == eager trajectory
    for (k, pa), pb in zip(me.named_parameters(), mg2.parameters()):
        assert torch.isfinite(pb).all(), k
