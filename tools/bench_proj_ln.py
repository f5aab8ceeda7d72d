"""The attention block's projection + LayerNorm alone (3 840 rows x 192): one launch (glowtts_proj_layernorm) against the two it replaces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import conv_fn as CF, ops, _lib
L = CF._L()
R, C, drop_p = 3840, 192, 0.1
att, x = torch.randn(R, C, device="cuda"), torch.randn(R, C, device="cuda")
w, b = torch.randn(C, C, 1, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
gamma, beta, rowmask = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), torch.ones(R, device="cuda")
pw = ops.pack_weight(w, precision=ops.BF16)
seed_t = torch.tensor([1], dtype=torch.int32, device="cuda")
proj, y, yb, s, st = (torch.empty(R, C, device="cuda"), torch.empty(R, C, device="cuda"), torch.empty(R, C, device="cuda", dtype=torch.bfloat16),
                      torch.empty(R, C, device="cuda"), torch.empty(R, 2, device="cuda"))
def two():
    CF._conv_launch(att, pw, C, R, 1, ops.F_BIAS | ops.F_DROPOUT, C, b, rowmask, proj, drop_p=drop_p, seed=7, seed_t=seed_t, a_bf=False)
    _lib.check(L.glowtts_layernorm_fwd_io(proj.data_ptr(), x.data_ptr(), s.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rowmask.data_ptr(), y.data_ptr(),
                                          st.data_ptr(), R, C, 1e-4, 0, 0.0, 0, None, yb.data_ptr(), _lib.stream()), "ln")
def one():
    _lib.check(L.glowtts_proj_layernorm(att.data_ptr(), C, pw.data.data_ptr(), pw.npad, b.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                        rowmask.data_ptr(), proj.data_ptr(), s.data_ptr(), st.data_ptr(), y.data_ptr(), yb.data_ptr(),
                                        R, C, 1e-4, drop_p, 7, seed_t.data_ptr(), _lib.stream()), "proj_ln")
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for _ in range(2):
    print(f"two launches {t(two):6.1f} us   one launch {t(one):6.1f} us")
