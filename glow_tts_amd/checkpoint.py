"""Checkpoint files of the reference trainer: `S_{steps}.pt` under `hp.Checkpoint_Path` holding
{'Model', 'Optimizer', 'Scheduler', 'Steps', 'Epochs'} (`Train.py:498-546`).  Files written here load in the reference's
`Trainer.Load_Checkpoint` and vice versa: the model's `state_dict()` has the reference's keys (glow_tts_amd/modules.py), `optim.RAdam` and
the Noam schedulers keep the reference's state layouts.  ('AMP' - apex mixed precision state - is carried through untouched when present.)"""
import os

import torch


def checkpoint_file(checkpoint_path, steps):
    return os.path.join(checkpoint_path, "S_{}.pt".format(steps)).replace("\\", "/")


def find_checkpoint(checkpoint_path, steps=0):
    """`Train.py:499-512`: steps == 0 -> the most recently created `*.pt` anywhere under `checkpoint_path` (None when there is none: an
    initial training); otherwise exactly `S_{steps}.pt`."""
    if steps != 0:
        return checkpoint_file(checkpoint_path, steps)
    paths = [os.path.join(root, f).replace("\\", "/") for root, _, files in os.walk(checkpoint_path) for f in files
             if os.path.splitext(f)[1] == ".pt"]
    return max(paths, key=os.path.getctime) if paths else None


def save_checkpoint(checkpoint_path, model, optimizer, scheduler, steps, epochs, extra=None):
    """`Trainer.Save_Checkpoint` (`Train.py:530-548`).  Returns the file name."""
    os.makedirs(checkpoint_path, exist_ok=True)
    state = {"Model": model.state_dict(), "Optimizer": optimizer.state_dict(), "Scheduler": scheduler.state_dict(),
             "Steps": int(steps), "Epochs": int(epochs)}
    if extra:
        state.update(extra)
    path = checkpoint_file(checkpoint_path, steps)
    torch.save(state, path)
    return path


def load_checkpoint(checkpoint_path, model, optimizer=None, scheduler=None, steps=0):
    """`Trainer.Load_Checkpoint` (`Train.py:498-524`): loads model / optimizer / scheduler state, marks every flow's ActNorm as
    initialised (a loaded model must not re-run the data-dependent init, `:526-527`) and returns (steps, epochs), or None when
    `steps == 0` and the directory holds no checkpoint."""
    path = find_checkpoint(checkpoint_path, steps)
    if path is None:
        return None
    state = torch.load(path, map_location="cpu")
    model.load_state_dict(state["Model"])
    if optimizer is not None:
        optimizer.load_state_dict(state["Optimizer"])
    if scheduler is not None:
        scheduler.load_state_dict(state["Scheduler"])
    for flow in model.layer_Dict["Decoder"].layer_Dict["Flows"]:
        flow.layers[0].initialized = True
    return state["Steps"], state["Epochs"]
