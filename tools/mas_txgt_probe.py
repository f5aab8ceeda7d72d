import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import mas_ref
from test_gpu_mas import hip_mas, hip_mas_t
rng = np.random.default_rng(1)
for (Tx, Ty, B) in [(5, 3, 2), (40, 17, 3), (130, 64, 2), (3, 1, 1)]:
    v = rng.normal(-100, 30, (B, Tx, Ty)).astype(np.float32)
    tx = np.full(B, Tx, np.int32); ty = np.full(B, Ty, np.int32)
    want, q_ref = mas_ref.maximum_path_c(v, tx, ty, return_q=True)
    try:
        path, idx, q = hip_mas(v, tx, ty, want_q=True)
        print(Tx, Ty, "plain: path equal", np.array_equal(path, want), "q equal", np.array_equal(q.view(np.uint32), q_ref.view(np.uint32)))
    except Exception as e:
        print(Tx, Ty, "plain raised", type(e).__name__, e)
    try:
        path, idx, q, kinds = hip_mas_t(v, tx, ty)
        print(Tx, Ty, "transposed: path equal", np.array_equal(path, want))
    except Exception as e:
        print(Tx, Ty, "transposed raised", type(e).__name__, e)
