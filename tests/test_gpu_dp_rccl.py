"""GPU: the data-parallel step over RCCL itself (one rank: the test box has one GPU).  Round 4's driver run died in
`bench.py --force-dist`: the process group's watchdog thread queried the end event of an eager collective whose stream was capturing by then
(hipErrorCapturedEvent; mechanism and the two rules that rule it out: glow_tts_amd/distributed.py).  tests/rccl_capture_check.py drives exactly that
order, ten captures in one process, and the Trainer under a one-rank RCCL group; each check is a child process (a watchdog abort takes the process down)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.late(2)]
HERE = os.path.dirname(os.path.abspath(__file__))


def _check(what, ok):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(HERE, "rccl_capture_check.py"), what, str(port)], capture_output=True, text=True, timeout=600,
                         cwd=os.path.dirname(HERE), env=env)
    if out.returncode != 0 or ok not in out.stdout:          # (pytest abbreviates long assertion payloads: the child's own words go to the captured output)
        print(out.stdout[-3000:])
        print(out.stderr[-6000:], file=sys.stderr)
    assert out.returncode == 0 and ok in out.stdout, out.returncode
    print(out.stdout[-400:])


def test_ten_data_parallel_captures_behind_unsynchronised_collectives():
    _check("capture", "RCCL CAPTURE OK")


def test_trainer_trains_data_parallel_over_a_one_rank_rccl_group():
    _check("trainer", "RCCL TRAINER OK")
