"""`python3 bench.py --gpus N` from a plain shell starts its own ranks and carries one
fallback attempt (bench.py self_launch)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, timeout=600):
    env = dict(os.environ)
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(name, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=env, timeout=timeout)


def test_launcher_falls_back_twice_and_reports_failure():
    """No GPU here: every attempt must fail; the second one must be the all-gather form, the third the RCCL-free one, and the
    status must say so."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU: on a GPU box the first attempt would go on to RCCL")
    out = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--bodies", "1024", "--no-cpu-baseline", "--launch-timeout", "240")
    assert out.returncode != 0
    assert out.stderr.count("one more attempt with --exchange allgather") == 1, out.stderr[-2000:]
    assert out.stderr.count("one more attempt with --exchange staged") == 1, out.stderr[-2000:]
    assert '"metric"' not in out.stdout


def test_launcher_does_not_retry_an_exchange_the_caller_chose():
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU")
    out = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--bodies", "1024", "--no-cpu-baseline", "--exchange", "allgather", "--launch-timeout", "240")
    assert out.returncode != 0
    assert "one more attempt" not in out.stderr


@pytest.mark.gpu
def test_plain_shell_bench_starts_two_ranks_on_one_gpu():
    """The rehearsal form (gloo, ranks sharing the one GPU): the launcher, the rank set-up, the sharded step and the
    JSON line of the N-rank path all run; the number is not a performance figure."""
    out = run_bench("--gpus", "2", "--exchange", "host", "--steps", "3", "--warmup", "1", "--bodies", "16384", "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0
    assert "REHEARSAL" in line["config"]["exchange"]
