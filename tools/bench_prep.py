"""The decoder's one-launch weight preparation (csrc/prep_ops.hip) alone: us per launch of the full-width 12-flow job table (forward images + the backward
images of 9 flows), HIP events over 30 launches.  GLOWTTS_LIB_PATH=tools/_build/libglowtts_hip_prep.so + GLOWTTS_PREP_ABL=1|2|4|... for the ablations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import full_width_state  # noqa: E402
from glow_tts_amd import decoder as D  # noqa: E402

g = torch.Generator().manual_seed(5)
cfg, sd = full_width_state(12, g)
dc = D.DecoderConfig(cfg.mel_dim, 12, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, 1)
P = {k: v.cuda().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
st = D.DecoderStacks(P, dc)
with torch.no_grad():
    A = dict(zip(D.WEIGHT_KEYS_GV, [w.contiguous() for w in st.weights(gv=True)]))
    GV = {k: (A.pop("g" + k[1:]), A.pop("v" + k[1:])) for k in D.WN_KEYS}
    prep = D._Prepared(dc, A, need_bwd=True, rows=32 * 404, GV=GV)
    jobs = prep.prep_jobs
    torch.cuda.synchronize()
    for _ in range(5):
        jobs.launch(torch.device("cuda"))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        jobs.launch(torch.device("cuda"))
    e1.record()
    torch.cuda.synchronize()
print(f"prep launch: {e0.elapsed_time(e1) / 30 * 1000:.1f} us, {jobs.blocks.value} workgroups, {jobs.n.value} jobs, ABL={os.environ.get('GLOWTTS_PREP_ABL', '0')}")
