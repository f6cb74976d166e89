"""bench.py as a launcher (no GPU needed): `python bench.py --gpus N` without a launcher around it starts its N ranks
itself as child processes and relays their status.  In this container there is no GPU, so every rank must refuse to
run ("there is no CPU path") and the parent must hand that failure on -- which shows that the ranks were started with
the torch.distributed.run environment contract and that the parent never needed a GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return torch.cuda.device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="launcher failure relay is checked where no GPU is visible")
def test_bench_self_launch_relays_rank_failure():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""                                     # no result line is invented
    # both ranks ran and both refused (a rank still paging torch in when the other one has already failed is ended
    # by the parent after its grace period: then the parent says so)
    n = p.stderr.count("no GPU visible")
    assert n == 2 or (n == 1 and "ending the other ranks" in p.stderr), p.stderr
    assert "ending the other ranks" in p.stderr or "exited with status" in p.stderr


def test_bench_rejects_mismatched_launcher():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr
