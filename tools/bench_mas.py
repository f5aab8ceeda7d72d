"""MAS micro-benchmark: value ~ N(-100, 30^2) f32 [B,120,800], HIP events on the launch stream."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glow_tts_amd import monotonic_align as ma

def run(B, Tx=120, Ty=800, ragged=False, iters=50, transposed=False):
    rng = np.random.default_rng(1234)
    v = torch.from_numpy(rng.normal(-100, 30, (B, Tx, Ty)).astype(np.float32)).cuda()
    if ragged:
        ty = 2 * rng.integers(300, 401, B); tx = np.round(0.15 * ty)
    else:
        ty = np.full(B, Ty); tx = np.full(B, Tx)
    txd = torch.from_numpy(tx.astype(np.int32)).cuda(); tyd = torch.from_numpy(ty.astype(np.int32)).cuda()
    if transposed:                      # the product layout: scores [B, Ty, Tx] as the log-prior kernel writes them (alignment.maximum_path_t)
        from glow_tts_amd import alignment
        v = v.transpose(1, 2).contiguous()
        dp = lambda: alignment.maximum_path_t(v, txd, tyd)
    else:
        dp = lambda: ma.maximum_path_idx(v, txd, tyd)
    for _ in range(5):
        idx = dp(); p = ma.path_from_idx(idx, Tx)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_dp = t_all = 0.0
    for _ in range(iters):
        e[0].record(); idx = dp(); e[1].record(); p = ma.path_from_idx(idx, Tx); e[2].record()
        torch.cuda.synchronize()
        t_dp += e[0].elapsed_time(e[1]); t_all += e[0].elapsed_time(e[2])
    us_dp, us_all = 1e3 * t_dp / iters, 1e3 * t_all / iters
    print(json.dumps(dict(B=B, Tx=Tx, Ty=Ty, ragged=ragged, transposed=transposed, dp_us=round(us_dp, 2), total_us=round(us_all, 2),
                          us_per_utt=round(us_all / B, 3), GBps=round(8.0 * Tx * Ty * B / us_all / 1e3, 1))))

if __name__ == "__main__":
    for B in (1, 32, 256, 2048):
        run(B); run(B, ragged=True); run(B, transposed=True); run(B, ragged=True, transposed=True)
    run(32, 200, 1000, transposed=True)
