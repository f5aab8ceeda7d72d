"""bench.py as the driver runs it: `--gpus N` must produce N ranks by itself, refuse when the node has fewer
devices, and report `n_gpus` = the size of the process group that really ran.  One GPU on the test box: the N = 2 launch shares device 0
(`--one-device`, gloo: RCCL refuses two ranks on one device) and `--force-dist` takes the whole data-parallel path over RCCL with one rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.late(2)]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ["--steps", "3", "--warmup", "2", "--windows", "0", "--no-cpu-baseline"]


def _run(args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=REPO, env=env)


def _json_line(out):
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert lines, (out.stdout[-2000:], out.stderr[-2000:])
    return json.loads(lines[-1])


def test_bench_refuses_more_gpus_than_the_node_has():
    import torch
    n = torch.cuda.device_count() + 1
    out = _run(["--gpus", str(n)] + FAST, timeout=300)
    assert out.returncode != 0 and f"--gpus {n}" in (out.stderr + out.stdout), (out.returncode, out.stderr[-500:])


def test_bench_gpus_2_spawns_two_ranks():
    out = _run(["--gpus", "2", "--one-device", "--backend", "gloo"] + FAST)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _json_line(out)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 64
    assert d["launch_mode"] == "hipgraph" and d["value"] > 0 and "roofline" in d


def test_bench_force_dist_runs_rccl_with_one_rank():
    out = _run(["--force-dist"] + FAST)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _json_line(out)
    assert d["n_gpus"] == 1 and d["launch_mode"] == "hipgraph" and "RCCL" in d["config"]["workload"]


def test_bench_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"] + FAST, capture_output=True, text=True, timeout=300, cwd=REPO, env=env)
    assert out.returncode != 0 and "must agree" in out.stderr
