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


def test_roofline_tool_prices_what_closed_shell_walkers_execute():
    """tools/roofline_configs.py prices the profiler's launches with the `closed_shell` object of the bench line the same run
    printed (the device's counts): executed columns of the large-system GEMM chain, one spin of the exchange energy, the
    library's own issued-flop count of either propagator deal -- and every committed round-6 row stays a utilisation (<= 1)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("roofline_configs", os.path.join(ROOT, "tools", "roofline_configs.py"))
    rc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rc)
    c = rc.CONFIGS['C5sd']
    taylor = 'void mfma_gemm_wg_kernel<2, 2, 2, 2, 4, TaylorProb, 5, true, 1, 5, 4>(TaylorProb)'
    full = rc.work(taylor, c)[1]
    closed = rc.work(taylor, c, {'large_chain_walker_step_share': 1.0})[1]
    assert closed == pytest.approx(0.5 * full) and rc.work(taylor, c, {'large_chain_walker_step_share': 0.5})[1] == pytest.approx(0.75 * full)
    # issued: the 64-wide column tiles that START below na run (50 columns of 100: one of two)
    assert rc.gemm_issued(taylor, c, cs={'large_chain_walker_step_share': 1.0}) == pytest.approx(0.5 * rc.gemm_issued(taylor, c))
    exq = 'void mfma_gemm_wg_kernel<2, 2, 2, 2, 4, ExxQProb<false>, 4, false, 1, 3, 1>(ExxQProb<false>)'
    assert rc.work(exq, c, {'exchange_one_spin': True})[1] == pytest.approx(0.5 * rc.work(exq, c)[1])
    c3 = rc.CONFIGS['bench_c3']
    cs = {'propagator_walker_step_share': 1.0, 'propagator_issued_flops_per_walker': {'open': 3.0, 'closed': 2.0}}
    assert rc.issued_flops('void prop_fused_kernel<false, 7>(PropFusedArgs)', c3, cs=cs) == pytest.approx(2.0 * c3['nw'])
    table = json.load(open(os.path.join(ROOT, "profiles", "r06_configs_roofline.json")))
    assert {t["config"] for t in table} >= {"C1", "C2", "C4", "C5sd", "C5", "bench_c3"}
    for t in table:
        for k in t["kernels"]:
            if k.get("bound") == "mfma" and "frac" in k:
                assert 0.0 < k["frac"] <= 1.0, (t["config"], k["kernel"], k["frac"])
