"""CPU suite: the reference's on-disk formats read by glow_tts_amd.data - pattern pickles, METADATA.PICKLE, Token.yaml,
the inference TSV - and the GE2E slice stack.  The reference's Datasets.py / Pattern_Generator.py cannot be imported here (librosa, a
Windows token path opened at import), so the expectations are restated inline from Datasets.py:17-74,78-131,137-144 and
Pattern_Generator.py:22-39,87-96,335-411."""
import os
import pickle

import numpy as np
import pytest
import torch
import yaml

from glow_tts_amd import data as D


def _patterns(root, n=7, seed=0):
    rng = np.random.RandomState(seed)
    texts = ["HELLO, WORLD.", "IT'S A TEST!", "GLOW", "A B C D E F G H I J K", "WHY? BECAUSE.", "MI-THREE", "LAST ONE."]
    files = []
    for i in range(n):
        T = int(rng.randint(20, 90))
        pat = {"Audio": rng.randn(T * 256).astype(np.float32), "Mel": rng.randn(T, 80).astype(np.float32),
               "Pitch": rng.rand(T).astype(np.float32), "Speaker_ID": i % 3, "Speaker": f"spk{i % 3}", "Dataset": "LJ", "Text": texts[i]}
        sub = os.path.join(root, "LJ" if i % 2 else "VCTK")
        os.makedirs(sub, exist_ok=True)
        with open(os.path.join(sub, f"p{i}.pickle"), "wb") as f:
            pickle.dump(pat, f, protocol=4)
        files.append((os.path.relpath(os.path.join(sub, f"p{i}.pickle"), root), pat))
    with open(os.path.join(root, "notes.pickle"), "wb") as f:      # not a pattern: ignored
        pickle.dump({"something": 1}, f)
    return dict(files), texts[:n]


def test_text_filtering_and_tokens(tmp_path):
    assert D.text_filtering('  hello, (world)!  it\'s "fine" ') == "HELLO, WORLD! IT'S FINE"
    assert D.text_filtering("'tis") is None                      # leading apostrophe
    assert D.text_filtering("price: $5 or 6") is None            # two separate legal runs
    td = D.make_token_dict(["AB", "B,C"])
    assert td == {"<S>": 0, "<E>": 1, ",": 2, "A": 3, "B": 4, "C": 5}
    with open(tmp_path / "Token.yaml", "w") as f:
        yaml.dump(td, f)
    td2 = D.load_token_dict(str(tmp_path / "Token.yaml"))
    assert td2 == td
    tok = D.text_to_token("CAB", td2)
    assert tok.dtype == np.int32 and tok.tolist() == [0, 5, 3, 4, 1]
    with pytest.raises(KeyError):
        D.text_to_token("Z", td2)


def test_metadata_and_pattern_dataset(tmp_path):
    root = str(tmp_path)
    pats, texts = _patterns(root)
    meta = D.write_metadata(root, "metadata.pickle")
    assert os.path.exists(os.path.join(root, "METADATA.PICKLE"))   # upper-cased like Pattern_Generator.py:393
    with open(os.path.join(root, "METADATA.PICKLE"), "rb") as f:
        assert pickle.load(f) == meta
    assert sorted(meta["File_List"]) == sorted(pats) and "notes.pickle" not in meta["File_List"]
    for k, p in pats.items():
        assert meta["Mel_Length_Dict"][k] == p["Mel"].shape[0] and meta["Text_Length_Dict"][k] == len(p["Text"])
        assert meta["Audio_Length_Dict"][k] == p["Audio"].shape[0] and meta["Speaker_ID_Dict"][k] == p["Speaker_ID"]
        assert k in meta["File_List_by_Speaker_Dict"][p["Speaker"]]
    td = D.make_token_dict(texts)
    ds = D.PatternDataset(root, "METADATA.PICKLE", td, accumulated_dataset_epoch=3, mel_length_min=30, mel_length_max=80, use_cache=True)
    keep = [k for k in meta["File_List"] if 30 <= meta["Mel_Length_Dict"][k] <= 80]
    assert ds.base_length == len(keep) and len(ds) == 3 * len(keep)
    for i in (0, len(keep) - 1, len(keep), 2 * len(keep) + 1):
        tok, mel, spk, pitch = ds[i]
        p = pats[keep[i % len(keep)]]
        assert tok.tolist() == [td[c] for c in ["<S>"] + list(p["Text"]) + ["<E>"]]
        assert np.array_equal(mel, p["Mel"]) and spk == p["Speaker_ID"] and np.array_equal(pitch, p["Pitch"])
    assert ds[0] is ds[len(keep)]                                  # cache hit of the repeated epoch (Datasets.py:113-114)
    # through the collater: the reference's batch tuple
    batch = [ds[i] for i in range(min(4, len(ds)))]
    tokens, tl, mels, ml, spk, ge2e, pit = D.Collater(num_squeeze=2, end_token_id=td["<E>"], ge2e=(5, 16, 8))(batch)
    assert tokens.shape == (len(batch), int(tl.max())) and mels.shape == (len(batch), 80, int(ml.max()))
    assert all(int(m) % 2 == 0 for m in ml) and ge2e.shape == (len(batch) * 5, 80, 16) and pit.shape == (len(batch), int(ml.max()))
    for b, (tok, mel, _, _) in enumerate(batch):
        assert tokens[b, :len(tok)].tolist() == tok.tolist() and (tokens[b, len(tok):] == td["<E>"]).all()
        assert torch.equal(mels[b, :, :int(ml[b])], torch.from_numpy(mel[:int(ml[b])].T)) and (mels[b, :, int(ml[b]):] == -4.0).all()


def test_ge2e_slices_match_restatement():
    rng = np.random.RandomState(1)
    long_mel, short_mel = rng.randn(200, 8).astype(np.float32), rng.randn(30, 8).astype(np.float32)
    samples, sl, ov = 5, 16, 8
    need = samples * (sl - ov) + ov                                # 48
    out = D.mels_for_ge2e([long_mel, short_mel], samples, sl, ov, rng=np.random.RandomState(7))
    assert out.shape == (2 * samples, sl, 8)
    off = np.random.RandomState(7).randint(0, 200 - need)
    win = long_mel[off:off + need]
    for k in range(samples):
        assert np.array_equal(out[k], win[k * (sl - ov):k * (sl - ov) + sl])
    pad = (need - 30) / 2
    padded = np.pad(short_mel, [[int(np.floor(pad)), int(np.ceil(pad))], [0, 0]], mode="reflect")
    for k in range(samples):
        assert np.array_equal(out[samples + k], padded[k * (sl - ov):k * (sl - ov) + sl])


def test_inference_prompt_tsv(tmp_path):
    p = tmp_path / "prompts.txt"
    p.write_text("Label\tText\tLength_Scale\tSpeaker\tGE2E\tProsody\tPitch\n"
                 "a1\tHello, (world)!\t1.0\t3\tw1.wav\tw2.wav\tw3.wav\n"
                 "\n"
                 "a2\tit costs $5 and 6\t0.9\t0\tx.wav\ty.wav\tz.wav\n")
    td = D.make_token_dict(["HELLO, WORLD!"])
    recs = D.read_inference_prompts(str(p), td)
    assert [r["label"] for r in recs] == ["a1", "a2"]
    assert recs[0]["text"] == "HELLO, WORLD!" and recs[0]["length_scale"] == 1.0 and recs[0]["speaker"] == 3
    assert recs[0]["token"].tolist() == D.text_to_token("HELLO, WORLD!", td).tolist() and recs[0]["wav_for_pitch"] == "w3.wav"
    assert recs[1]["text"] is None and "token" not in recs[1]


def test_checkpoint_files_roundtrip(tmp_path):
    """glow_tts_amd.checkpoint: the reference trainer's `S_{steps}.pt` layout and discovery rule (Train.py:498-548)."""
    import sys, time
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import tiny_hp_dict
    from glow_tts_amd import checkpoint as C
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS
    from glow_tts_amd.optim import Modified_Noam_Scheduler, RAdam
    hp = Recursive_Parse(tiny_hp_dict("Vanilla"))
    torch.manual_seed(0)
    m1 = GlowTTS(hp)
    o1 = RAdam(m1.parameters(), lr=1e-3, eps=1e-6, weight_decay=1e-6)
    s1 = Modified_Noam_Scheduler(o1, base=4000)
    root = str(tmp_path / "ckpt")
    assert C.load_checkpoint(root, m1, o1, s1) is None             # nothing there: initial training
    f10 = C.save_checkpoint(root, m1, o1, s1, steps=10, epochs=1)
    assert os.path.basename(f10) == "S_10.pt"
    raw = torch.load(f10, map_location="cpu")
    assert set(raw) == {"Model", "Optimizer", "Scheduler", "Steps", "Epochs"} and raw["Steps"] == 10
    time.sleep(0.05)
    with torch.no_grad():
        for p in m1.parameters():
            p.add_(0.5)
    s1.step()
    C.save_checkpoint(root, m1, o1, s1, steps=20, epochs=2)
    torch.manual_seed(1)
    m2 = GlowTTS(hp)
    o2 = RAdam(m2.parameters(), lr=5e-4)
    s2 = Modified_Noam_Scheduler(o2, base=100)
    assert C.load_checkpoint(root, m2, o2, s2) == (20, 2)          # steps == 0: the newest file
    assert all(torch.equal(a, b) for a, b in zip(m1.state_dict().values(), m2.state_dict().values()))
    assert all(f.layers[0].initialized for f in m2.layer_Dict["Decoder"].layer_Dict["Flows"])
    assert s2.base == 4000 and s2.last_epoch == s1.last_epoch and o2.param_groups[0]["eps"] == 1e-6
    m3 = GlowTTS(hp)
    assert C.load_checkpoint(root, m3, steps=10) == (10, 1)        # an explicit step: exactly S_10.pt
    assert all(torch.equal(a, b) for a, b in zip(raw["Model"].values(), m3.state_dict().values()))


def test_collater_matches_the_reference_collater_bit_for_bit():
    """tests/golden/collater_case.npz: a ragged batch (odd mel lengths, one-letter text) through the UNMODIFIED reference
    `Datasets.Collater` (Datasets.py:225-250; generated by tests/golden/make_data_golden.py) - all seven tensors, dtypes included:
    '<E>' padding, -Max_Abs_Mel padding, truncation to a multiple of Num_Squeeze, the GE2E slice stack (same numpy RNG stream), pitch
    padding to the longest untruncated track."""
    import os
    import numpy as np
    import torch
    from glow_tts_amd import data
    from glow_tts_amd.hparams import DEFAULT_YAML, Recursive_Parse, load_yaml
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collater_case.npz"))
    token_dict = {str(k): int(v) for k, v in zip(d["in/token_symbols"], d["in/token_ids"])}
    texts = [str(t) for t in d["in/texts"]]
    def split(cat, lens):
        out, o = [], 0
        for n in lens:
            out.append(cat[o:o + n]); o += n
        return out
    tokens = split(d["in/token_cat"], d["in/token_lens"])
    for t, text in zip(tokens, texts):
        assert np.array_equal(t, data.text_to_token(text, token_dict))                       # Text_to_Token, Datasets.py:17-21
    mels, pitches = split(d["in/mel_cat"], d["in/mel_lens"]), split(d["in/pitch_cat"], d["in/mel_lens"])
    hp = Recursive_Parse(load_yaml(DEFAULT_YAML))
    col = data.Collater.from_hp(hp, token_dict, ge2e=tuple(int(x) for x in d["in/ge2e"]))
    np.random.seed(int(d["in/seed"]))
    out = col(list(zip(tokens, mels, [int(s) for s in d["in/speakers"]], pitches)))
    names = ["tokens", "token_lengths", "mels", "mel_lengths", "speakers", "mels_for_ge2e", "pitches"]
    for name, got in zip(names, out):
        want = torch.from_numpy(d["out/" + name])
        assert got.dtype == want.dtype and got.shape == want.shape, (name, got.dtype, want.dtype, got.shape, want.shape)
        assert torch.equal(got, want), name
    assert data.Collater().end == 1                                                          # '<E>' (Token.yaml: '<S>' 0, '<E>' 1)


def test_inference_dataset_and_collater_follow_the_reference_layout(tmp_path):
    """`Datasets.Inference_Dataset` / `Inference_Collater` (Datasets.py:131-165, 252-275): the 11-tuple `Trainer.Inference_Step` takes; a Vanilla
    run decodes no reference wav (placeholders of one frame)."""
    import yaml
    import torch
    from glow_tts_amd import data
    from glow_tts_amd.hparams import Recursive_Parse
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(repo, "glow_tts_amd", "Hyper_Parameters.default.yaml")) as f:
        hp = Recursive_Parse(yaml.safe_load(f))
    token_dict = data.make_token_dict(["HELLO WORLD.", "A LONGER SENTENCE, WITH COMMAS!"])
    tsv = tmp_path / "prompts.txt"
    tsv.write_text("Label\tText\tLength_Scale\tSpeaker\tGE2E\tProsody\tPitch\n"
                   "A\tHello world.\t1.0\t3\tx.wav\ty.wav\tz.wav\n"
                   "B\tA longer sentence, with commas!\t1.25\t0\tx.wav\ty.wav\tz.wav\n", encoding="utf-8")
    ds = data.InferenceDataset(str(tsv), token_dict, hp)
    assert len(ds) == 2
    out = data.InferenceCollater(token_dict, hp)([ds[0], ds[1]])
    tokens, tl, pro, pl, spk, ge2e, pit, pil, scales, labels, texts = out
    assert tokens.dtype == torch.int64 and tokens.shape == (2, int(tl.max())) and tl.tolist() == [len(ds[0][0]), len(ds[1][0])]
    assert int(tokens[0, 0]) == token_dict["<S>"] and int(tokens[0, tl[0] - 1]) == token_dict["<E>"] and bool((tokens[0, tl[0]:] == token_dict["<E>"]).all())
    assert pro.shape == (2, int(hp.Sound.Mel_Dim), 1) and pl.tolist() == [1, 1] and pit.shape == (2, 1) and pil.tolist() == [1, 1]
    assert spk.tolist() == [3, 0] and scales.tolist() == [1.0, 1.25] and labels == ["A", "B"] and texts[0] == ds[0][7]
