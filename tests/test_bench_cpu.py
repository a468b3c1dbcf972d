"""CPU tests of bench.py's multi-rank plumbing: `python bench.py --gpus 2` must start its own two ranks, reduce ONE tally
and print one JSON line with n_gpus 2.  The ranks use the gloo backend and a test-only oracle-backed engine stub
(tests/oracle_engine_stub.py); on the GPU box the same code path runs the HIP engine over RCCL."""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run_bench(*argv: str, timeout: int = 600) -> dict:
    env = dict(os.environ, FK_DIST_BACKEND="gloo", FK_BENCH_ENGINE="oracle_engine_stub:Engine",
               PYTHONPATH=f"{ROOT / 'tests'}:{os.environ.get('PYTHONPATH', '')}")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_bench_gpus_2_starts_its_own_ranks_and_reduces_one_tally(tmp_path):
    import golden_util as gu
    import pyoracle as po

    out = tmp_path / "tally.npy"
    line = _run_bench("--gpus", "2", "--shuffles", "120", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--dump-tally", str(out))
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["launcher"].startswith("self") and line["dist_backend"] == "gloo" and "TEST STUB" in line["engine"]
    assert "dist_fallback_reason" in line and line["dist_fallback_reason"] is None  # (gloo was asked for: nothing fell back)
    assert line["config"]["games_per_gpu_per_step"] == 120 * 32
    # steps 1..2 (after one warm-up step) of two ranks = shuffles [240, 720) of the single-process run
    table = gu.strategies_from_tuples(gu.load("grid_vectors.json")["g64"], po.STRATEGY_DTYPE)
    want = po.tournament(table, 2, 42, 240, 720)["tally"][0]
    assert np.array_equal(np.load(out), want)
    assert line["value"] == pytest.approx(2 * 2 * 120 * 32 / (line["ms_per_step"] * 2e-3), rel=1e-6)


def test_bench_single_process_and_rank_failure_propagates(tmp_path):
    line = _run_bench("--shuffles", "50", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert line["n_gpus"] == 1 and line["launcher"] == "single process" and line["dist_backend"] is None
    env = dict(os.environ, FK_DIST_BACKEND="gloo", FK_BENCH_ENGINE="oracle_engine_stub:NoSuchEngine",
               PYTHONPATH=f"{ROOT / 'tests'}:{os.environ.get('PYTHONPATH', '')}")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--shuffles", "10", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("config,extra", [(3, ("--shuffles", "3")), (4, ("--games", "3000")), (5, ("--games", "200")), (6, ("--games", "1500"))])
def test_bench_other_baseline_configs_two_ranks(config, extra):
    line = _run_bench("--gpus", "2", "--config", str(config), *extra, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", timeout=900)
    assert line["n_gpus"] == 2 and (f"config {config}" in line["metric"] or (config == 6 and "production player counts" in line["metric"]))
    assert line["scaling"] == ("weak" if config == 3 else "strong")
    assert line["roofline"]["ops_per_game"] > 0 and line["value"] > 0
    if config in (4, 6):  # sweeps: one record per player count; the headline record is the slowest player count's, whole
        per_k = line["roofline"]["per_k"]
        assert [r["k"] for r in per_k] == ([2, 4, 6, 8] if config == 4 else [2, 3, 4, 5, 6, 8, 10, 12])
        worst = min(per_k, key=lambda r: r["frac"])
        assert all(line["roofline"][key] == worst[key] for key in ("frac", "kernel", "kernel_ms", "games_per_launch", "ops_per_game"))
        assert line["roofline"]["frac_mean_over_k"] == pytest.approx(np.mean([r["frac"] for r in per_k]))


def test_bench_under_torch_distributed_run_like_the_driver(tmp_path):
    """The driver's N > 1 launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` — ranks come from the environment, bench.py must NOT start ranks of its own."""
    import socket

    import golden_util as gu
    import pyoracle as po

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "tally.npy"
    env = dict(os.environ, FK_DIST_BACKEND="gloo", FK_BENCH_ENGINE="oracle_engine_stub:Engine",
               PYTHONPATH=f"{ROOT / 'tests'}:{os.environ.get('PYTHONPATH', '')}")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--shuffles", "100", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--dump-tally", str(out)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["launcher"] == "torch.distributed.run" and line["dist_backend"] == "gloo"
    table = gu.strategies_from_tuples(gu.load("grid_vectors.json")["g64"], po.STRATEGY_DTYPE)
    assert np.array_equal(np.load(out), po.tournament(table, 2, 42, 200, 600)["tally"][0])
