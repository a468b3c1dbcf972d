#!/usr/bin/env python3
"""bench.py — simulated Farkle games/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: at N=1 the workload is
BASELINE.json configs[1] — k=2, the 64-strategy grid, 10^7 games (312 500 shuffles x 32 games),
root seed 42, counts-only tallies ([S][26] int64 resident in HBM).  For N>1 (one rank per GPU,
launched by torch.distributed.run) every rank plays its own range of 312 500 shuffles per step
(weak scaling: the shuffle space is partitioned, no data-path collective); ranks add their step tallies locally
and ONE RCCL reduce of the [S][26] int64 tally to rank 0 at the end of the job, inside the timed region, plays the
role of OutcomeCounter.absorb (run_tournament.py:197-213).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (fk_play_kernel): the path is
integer VALU work, so the bound is the vector-ALU issue roof, not HBM or MFMA; `cpu_baseline` is the
CPU oracle (a C port of the reference's algorithm, test infrastructure) timed on the host's cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SHUFFLES_PER_STEP = 312_500  # x 32 games = 10^7 games (BASELINE.json configs[1])
ROOT_SEED = 42
K = 2


def grid64():
    from farkle_ii_amd.strategies import generate_strategy_grid, pack_strategies

    strategies, _ = generate_strategy_grid(
        score_thresholds=[250, 300, 350, 400], dice_thresholds=[0, 1, 2, 3], smart_five_opts=[True], smart_one_opts=[True],
        consider_score_opts=[True], consider_dice_opts=[True], auto_hot_dice_opts=[True], run_up_score_opts=[True])
    table = pack_strategies(strategies)
    assert len(table) == 64
    return table


def work_per_game(rows: np.ndarray, k: int) -> dict:
    """W(game) = 229*R + 30*T + 850*k int32 lane-ops, R = sum of seat rolls, T = sum of seat turns."""
    R = rows["seats"]["rolls"].astype(np.int64).sum(axis=1)
    T = rows["seats"]["n_turns"].astype(np.int64).sum(axis=1)
    W = 229 * R + 30 * T + 850 * k
    return {"rolls_per_game": float(R.mean()), "turns_per_game": float(T.mean()), "ops_per_game": float(W.mean())}


def cpu_baseline(table: np.ndarray, seconds_target: float = 12.0) -> dict:
    """Time the CPU oracle (checker, never the product) on a bounded sample of the same workload."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import pyoracle as po

    # a 1-GPU box gives this job a 16-core CPU share however many hardware threads the host shows
    threads = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("FK_CPU_THREADS", "16"))))
    t = table.view(po.STRATEGY_DTYPE)
    po.tournament(t, K, ROOT_SEED, 0, 64, n_threads=threads)  # warm the thread pool
    t0 = time.perf_counter()
    po.tournament(t, K, ROOT_SEED, 0, 400, n_threads=threads)
    rate = 400 * 32 / (time.perf_counter() - t0)
    n_sh = int(max(400, min(SHUFFLES_PER_STEP, rate * seconds_target / 32)))
    t0 = time.perf_counter()
    res = po.tournament(t, K, ROOT_SEED, 0, n_sh, n_threads=threads)
    dt = time.perf_counter() - t0
    # the same oracle on ONE core (SURVEY section 8d asks for both): first shuffles of the workload, about 3 s
    t0 = time.perf_counter()
    po.tournament(t, K, ROOT_SEED, 0, 200, n_threads=1)
    n1 = int(max(200, min(n_sh, 200 * 3.0 / (time.perf_counter() - t0))))
    t0 = time.perf_counter()
    po.tournament(t, K, ROOT_SEED, 0, n1, n_threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": n_sh * 32 / dt, "unit": "games/s", "cores": threads, "kind": "port",
            "sample": f"shuffles 0..{n_sh - 1} of the same workload ({n_sh * 32} games, {dt:.1f} s, OpenMP over shuffles)",
            "single_core": {"value": n1 * 32 / dt1, "unit": "games/s", "cores": 1,
                            "sample": f"shuffles 0..{n1 - 1} ({n1 * 32} games, {dt1:.1f} s)"},
            "_tally": res["tally"][0], "_n_sh": n_sh}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shuffles", type=int, default=SHUFFLES_PER_STEP, help="shuffles per rank per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    # FK_DIST_BACKEND=gloo rehearses the multi-rank path on a one-GPU box (ranks share GPU 0, tallies reduced on CPU);
    # the real runs use nccl = RCCL over xGMI with one GPU per rank.
    backend = os.environ.get("FK_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if distributed:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {n_gpus}", file=sys.stderr)

    from farkle_ii_amd.backend import Engine

    table = grid64()
    S = len(table)
    eng = Engine(local_rank)
    info = eng.device_info()
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the tally reduction runs

    def sync() -> None:
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def step(index: int):
        first = (index * n_gpus + rank) * args.shuffles
        res = eng.tournament(table, K, ROOT_SEED, first, first + args.shuffles)
        return res["tally"][0], eng.timing()

    def reduce_to_rank0(local: np.ndarray):
        """The path's only exchange: one SUM of the per-strategy int64 tally to rank 0 at the end of the
        job, the analogue of OutcomeCounter.absorb — RCCL over xGMI when there is more than one rank."""
        t = torch.from_numpy(local).to(red_dev)
        if distributed:
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        return t

    # one-time initialisation outside any step: device workspace, lazily loaded torch kernels, RCCL communicator
    eng.tournament(table, K, ROOT_SEED, 0, args.shuffles)
    warm = torch.zeros((S, 26), dtype=torch.int64, device=red_dev)
    warm += torch.from_numpy(np.zeros((S, 26), dtype=np.int64)).to(red_dev)
    if distributed:
        dist.reduce(warm, dst=0, op=dist.ReduceOp.SUM)
    local = np.zeros((S, 26), dtype=np.int64)
    for i in range(args.warmup):
        tally, _ = step(i)
        local += tally
    reduce_to_rank0(local)
    local[:] = 0
    play_ms, seed_ms, perm_ms, launches = 0.0, 0.0, 0.0, 0
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tally, t = step(args.warmup + i)
        local += tally  # [64][26] int64 host add
        play_ms += t["play_ms"]
        seed_ms += t["seed_ms"]
        perm_ms += t["perm_ms"]
        launches += t["play_launches"]
    total = reduce_to_rank0(local)  # inside the timed region
    sync()
    elapsed = time.perf_counter() - t0
    if distributed:
        e = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        elapsed = float(e.item())

    games_per_rank_step = args.shuffles * (S // K)
    total_games = games_per_rank_step * n_gpus * args.steps
    value = total_games / elapsed

    if rank == 0:
        tot = total.cpu().numpy()
        assert int(tot[:, 1].sum()) == total_games * K, "exposure conservation failed"
        assert np.array_equal(tot[:, 1], tot[:, 2] + tot[:, 3]) and int(tot[:, 0].sum()) * K == int(tot[:, 2].sum())

        # live per-game work (R, T of SURVEY section 8d) from the rows of one launch of the SAME size, so that every
        # fk_play_kernel launch of this process has the step's shape (the rocprofv3 per-kernel average stays comparable)
        sample = eng.tournament(table, K, ROOT_SEED, 0, args.shuffles, want_rows=True)
        wpg = work_per_game(sample["rows"], K)
        del sample
        kernel_ms = play_ms / max(launches, 1)
        kernel_games_per_s = games_per_rank_step / (kernel_ms * 1e-3)
        # VALU roof: CUs x 4 SIMD x 32 lanes/clk x clock (MI355X_MICROARCH.md: wave64 issues over 2 cycles on a SIMD-32)
        peak_ops = info["compute_units"] * 4 * 32 * info["clock_mhz"] * 1e6
        achieved_ops = kernel_games_per_s * wpg["ops_per_game"]
        # algorithmic HBM bytes of the game kernel in counts-only mode: seat seeds read once (32 B x k per game)
        # + 2 B x k permutation entries; the tally is [S][26] int64 written once per launch
        hbm_bytes_per_game = 32 * K + 2 * K + 4
        # HBM bytes per launch from the PMC passes of the same launch (rocprofv3 cannot run inside this process):
        # profiles/r01_hbm_traffic.json, FETCH_SIZE x2-corrected per MI355X_MICROARCH.md
        traffic = None
        tpath = ROOT / "profiles" / "r01_hbm_traffic.json"
        if tpath.exists() and args.shuffles == SHUFFLES_PER_STEP:
            traffic = json.loads(tpath.read_text())["kernels"]["fk_play_kernel"]["hbm_bytes_corrected"]
        roofline = {
            "bound": "valu", "kernel": "fk_play_kernel",
            "achieved": achieved_ops / 1e12, "peak": peak_ops / 1e12, "unit": "Tlane-op/s (int32)", "frac": achieved_ops / peak_ops,
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC, separate pass)",
            "kernel_ms": kernel_ms, "kernel_games_per_s": kernel_games_per_s, **wpg,
            "hbm": {"achieved": kernel_games_per_s * hbm_bytes_per_game / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": kernel_games_per_s * hbm_bytes_per_game / 8e12, "bytes_per_game": hbm_bytes_per_game},
            "seed_kernel_ms": seed_ms / max(launches, 1), "perm_kernel_ms": perm_ms / max(launches, 1),
            "launch": {k2: t[k2] for k2 in ("play_block", "play_grid", "play_lds_bytes")},
        }
        cpu = None
        if not args.no_cpu_baseline and n_gpus == 1:  # the CPU leg runs on rank 0 of the single-GPU run only
            cpu = cpu_baseline(table)
            n_sh = cpu.pop("_n_sh")
            ref_tally = cpu.pop("_tally")
            if n_sh == args.shuffles:
                got = eng.tournament(table, K, ROOT_SEED, 0, n_sh)["tally"][0]
            else:  # slower host: compare on the sample's shuffles only (per-batch tallies of one same-size launch)
                spb = n_sh
                got = eng.tournament(table, K, ROOT_SEED, 0, (args.shuffles // spb) * spb, shuffles_per_batch=spb)["tally"][0]
            assert np.array_equal(got, ref_tally), "GPU tally differs from the CPU oracle on the baseline sample"
            cpu["parity"] = f"GPU tally == oracle tally on the sample ({n_sh * 32} games)"
        line = {
            "metric": "simulated games/sec (whole node) at k=2, fixed strategy-grid size",
            "value": value, "unit": "games/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: k=2, 64-strategy grid, 10^7 games per GPU per step, root_seed 42, counts-only tallies",
                       "k": K, "n_strategies": S, "games_per_gpu_per_step": games_per_rank_step, "parallelism": f"shuffle-range split x{n_gpus}",
                       "device": info["name"], "arch": info["arch"], "compute_units": info["compute_units"], "clock_mhz": info["clock_mhz"]},
            "roofline": roofline, "cpu_baseline": cpu,
            # not `vs_baseline` (BASELINE.json publishes nothing for this exact config): the reference's own report of the
            # k=2 tournament path on its 80-strategy grid, Ryzen 7 3700X, 12 workers = 1 142.9 games/s
            "vs_reference_published_12_workers": value / 1142.9,
        }
        # the Python reference itself timed on this same workload in the build container (oracle/time_reference.py; the
        # reference cannot travel to the GPU box, so this is a committed fixture, not a measurement of this run)
        fixture = ROOT / "tests" / "golden" / "reference_cpu_timing.json"
        if fixture.exists():
            ref = json.loads(fixture.read_text())
            line["reference_python_fixture"] = {
                "tournament_loop_games_per_s_1_process": ref["tournament_loop_1_process"]["games_per_s"],
                "tournament_loop_games_per_s_8_processes": ref["tournament_loop_8_processes"]["games_per_s"],
                "where": ref["host"]["note"]}
        print(json.dumps(line))
    eng.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
