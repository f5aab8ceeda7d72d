"""What killed round 4's `bench.py --force-dist` (GPUTEST_r04), isolated: a one-rank RCCL group, one eager all-reduce, then a hipGraph capture that
outlasts the process group watchdog's 100-ms poll.  Variants (each in a child process, the abort takes the process down):
  sync        dist.all_reduce(async_op=False) with stream S current, capture on S           -> torch >= 2.7 records the Work's end event on S; the watchdog's
                                                                                               hipEventQuery answers hipErrorCapturedEvent while S captures: abort
  sync-other  the same collective on S, capture on ANOTHER stream                            -> survives (the event's stream is not capturing)
  async       glow_tts_amd.distributed._collective (async_op=True + wait) on S, capture on S -> survives (end event on the process group's own stream)
  drain       sync + glow_tts_amd.distributed.before_capture()                               -> survives (the watchdog holds no Work when the capture begins)
Usage: python tools/rccl_capture_race.py            (runs all four, prints one line each)"""
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(variant, port):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    from glow_tts_amd import distributed as gd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    x = torch.ones(1024, device="cuda")
    dist.all_reduce(x)                                   # communicator set-up outside the experiment
    torch.cuda.synchronize()
    time.sleep(0.5)                                      # the watchdog has retired it
    s, other = torch.cuda.Stream(), torch.cuda.Stream()
    y = torch.zeros(1 << 20, device="cuda")
    for rep in range(5):
        with torch.cuda.stream(s):
            if variant == "async":
                gd._collective(dist.all_reduce, x)
            else:
                dist.all_reduce(x)
        if variant == "drain":
            gd.before_capture()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=(other if variant == "sync-other" else s), capture_error_mode="thread_local"):
            for _ in range(40):
                y.add_(1.0)
                time.sleep(0.01)                         # 0.4 s under capture: at least three watchdog polls
        g.replay()
        torch.cuda.synchronize()
    print(f"SURVIVED {variant}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 2:
        child(sys.argv[1], sys.argv[2])
        raise SystemExit(0)
    for i, v in enumerate(("sync", "sync-other", "async", "drain")):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), v, str(29711 + i)], capture_output=True, text=True, timeout=300)
        why = ""
        if out.returncode != 0:
            hits = [l for l in out.stderr.splitlines() if "hipError" in l or "HIP error" in l]
            why = " | " + (hits[0].strip()[:200] if hits else out.stderr.strip().splitlines()[-1][:200] if out.stderr.strip() else "")
        print(f"[race] {v:10s} rc {out.returncode:4d} {'survived' if 'SURVIVED' in out.stdout else 'ABORTED'}{why}", flush=True)
