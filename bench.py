#!/usr/bin/env python3
"""bench.py — simulated Farkle games/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config {2,3,4,5}]

A "step" is one pass of the hot path over one batch of synthetic input.  `--config` picks the BASELINE.json workload
(default 2 = configs[1], the configuration the metric is quoted on):

  2  k=2, 64-strategy grid, 10^7 games per GPU per step (312 500 shuffles x 32 games), root seed 42, counts-only
  3  k=4, default 5 160-strategy grid, 10^8 games per GPU per step (77 520 shuffles x 1 290 games), root seed 0
  4  k in {2,4,6,8} on the 5 160 grid, 2.5*10^8 games per k per step (10^9 in all), every k's shuffle range split over
     the ranks (strong scaling), one tally reduce per k
  5  H2H: all 66 pairs of 12 candidate strategies x orders {0,1}, 10^8 completed games per pair, blocks dealt over
     the ranks, one reduce of the block results

One process per GPU.  Under `torch.distributed.run` the ranks come from the environment; WITHOUT it, `--gpus N` (N > 1)
makes this process start N rank processes itself — before anything here touches the GPU — and wait for them (the
reference's scaling table is likewise one command per worker count, run_tournament.py:1576-1586).  Ranks add their step
tallies locally and ONE int64 SUM reduce of the tally to rank 0 at the end of the job (RCCL over xGMI), inside the timed
region, plays the role of OutcomeCounter.absorb (run_tournament.py:197-213).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (fk_play_kernel): the path is integer VALU work,
so the bound is the vector-ALU issue roof, not HBM or MFMA; `cpu_baseline` is the CPU oracle (a C port of the
reference's algorithm, test infrastructure) timed on the host's cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SHUFFLES_PER_STEP = 312_500  # x 32 games = 10^7 games (BASELINE.json configs[1])
ROOT_SEED = 42
K = 2
METRIC = "simulated games/sec (whole node) at k=2, fixed strategy-grid size"


def grid64():
    from farkle_ii_amd.strategies import generate_strategy_grid, pack_strategies

    strategies, _ = generate_strategy_grid(
        score_thresholds=[250, 300, 350, 400], dice_thresholds=[0, 1, 2, 3], smart_five_opts=[True], smart_one_opts=[True],
        consider_score_opts=[True], consider_dice_opts=[True], auto_hot_dice_opts=[True], run_up_score_opts=[True])
    table = pack_strategies(strategies)
    assert len(table) == 64
    return table


def grid5160():
    from farkle_ii_amd.strategies import STRATEGY_DTYPE, default_grid_tuples

    tuples = default_grid_tuples()
    table = np.zeros(len(tuples), dtype=STRATEGY_DTYPE)
    for i, t in enumerate(tuples):
        table[i] = tuple(t)
    assert len(table) == 5160
    return table


def work_per_game(rows: np.ndarray, k: int) -> dict:
    """W(game) = 229*R + 30*T + 850*k int32 lane-ops, R = sum of seat rolls, T = sum of seat turns."""
    R = rows["seats"]["rolls"].astype(np.int64).sum(axis=1)
    T = rows["seats"]["n_turns"].astype(np.int64).sum(axis=1)
    W = 229 * R + 30 * T + 850 * k
    return {"rolls_per_game": float(R.mean()), "turns_per_game": float(T.mean()), "ops_per_game": float(W.mean())}


def work_from_seat_stats(stats: np.ndarray, games: int, k: int) -> dict:
    """The same R, T, W from the all-seat integer statistics of a launch (fk_tournament_run_stats: column 15 = sum of seat
    rolls, column 6 = sum of seat turns) — exact over ALL games of a full-size launch, no rows materialised."""
    R = float(stats[..., 15].sum()) / games
    T = float(stats[..., 6].sum()) / games
    return {"rolls_per_game": R, "turns_per_game": T, "ops_per_game": 229.0 * R + 30.0 * T + 850.0 * k}


def add_timing(acc: dict, t: dict) -> dict:
    """Accumulate fk_timing records (HIP events on the engine's stream) over the engine calls of a step."""
    for key in ("play_ms", "seed_ms", "perm_ms", "play_launches", "games", "prefetched_chunks"):
        acc[key] = acc.get(key, 0) + t.get(key, 0)
    for key in ("play_block", "play_grid", "play_lds_bytes", "play_mixed_flags"):
        acc[key] = t.get(key)
    if t.get("play_launches") == 1 and t.get("play_ms"):  # spread of the single-launch calls (what a rocprofv3 minimum / maximum would show)
        acc["play_ms_min"] = min(acc.get("play_ms_min", t["play_ms"]), t["play_ms"])
        acc["play_ms_max"] = max(acc.get("play_ms_max", t["play_ms"]), t["play_ms"])
    if t.get("play_block_end_max_ms"):  # the launch's drain tail (same option): last workgroup's end - median workgroup's end
        acc["tail_ms_sum"] = acc.get("tail_ms_sum", 0.0) + (t["play_block_end_max_ms"] - t["play_block_end_p50_ms"])
        acc["tail_n"] = acc.get("tail_n", 0) + 1
    elif t.get("tail_n"):
        acc["tail_ms_sum"] = acc.get("tail_ms_sum", 0.0) + t["tail_ms_sum"]
        acc["tail_n"] = acc.get("tail_n", 0) + t["tail_n"]
    if t.get("play_clock_mhz"):  # option "clock_stamps": the shader clock the call's last game kernel ran at, measured inside it
        acc["clock_sum"] = acc.get("clock_sum", 0) + t["play_clock_mhz"]
        acc["clock_n"] = acc.get("clock_n", 0) + 1
    elif t.get("clock_n"):
        acc["clock_sum"] = acc.get("clock_sum", 0) + t["clock_sum"]
        acc["clock_n"] = acc.get("clock_n", 0) + t["clock_n"]
    for k, sub in (t.get("per_k") or {}).items():  # config 4: one record per player count
        add_timing(acc.setdefault("per_k", {}).setdefault(k, {}), sub)
    return acc


def _oracle():
    sys.path.insert(0, str(ROOT / "oracle"))
    import pyoracle as po

    return po


def _cpu_threads() -> int:
    # a 1-GPU box gives this job a 16-core CPU share however many hardware threads the host shows
    return max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("FK_CPU_THREADS", "16"))))


# ---------------------------------------------------------------------------------------------------------------
# workloads: every one exposes  local_shape, step(eng, index, rank, world) -> (local tally, games played by THIS rank),
# games_per_step(world), verify(total, steps, world), sample(eng) -> (rows, k, games) for W, cpu_baseline(eng)
# ---------------------------------------------------------------------------------------------------------------
class Tournament:
    """BASELINE configs[1] / configs[2]: one (grid, k, root) cell, every rank plays its own shuffle range per step."""

    scaling = "weak"

    def __init__(self, config: int, table: np.ndarray, k: int, root: int, shuffles: int, label: str, sample_shuffles: int):
        self.config, self.table, self.k, self.root, self.shuffles, self.label = config, table, k, root, shuffles, label
        self.S = len(table)
        self.gps = self.S // k
        self.local_shape = (self.S, 26)
        self.sample_shuffles = min(sample_shuffles, shuffles)

    def games_per_step(self, world: int) -> int:
        return self.shuffles * self.gps * world

    def step(self, eng, index: int, rank: int, world: int, next_index: int | None = None):
        first = (index * world + rank) * self.shuffles
        if next_index is not None and hasattr(eng, "hint_next"):
            # the step after this one: its permutations and seat seeding are prepared in the drain tail of this step's game
            # kernel (both steps are inside the timed region; the last timed step hints nothing)
            nxt = (next_index * world + rank) * self.shuffles
            eng.hint_next(nxt, nxt + self.shuffles)
        res = eng.tournament(self.table, self.k, self.root, first, first + self.shuffles)
        return res["tally"][0], self.shuffles * self.gps, add_timing({}, eng.timing())

    def verify(self, tot: np.ndarray, games: int) -> None:
        assert int(tot[:, 1].sum()) == games * self.k, "exposure conservation failed"
        assert np.array_equal(tot[:, 1], tot[:, 2] + tot[:, 3]) and int(tot[:, 0].sum()) * self.k == int(tot[:, 2].sum())

    def sample(self, eng):
        # measured R, T of SURVEY 8d over ALL games of one launch of the step's own size (every fk_play_kernel launch of the
        # process has the step's shape, so the rocprofv3 per-kernel average stays comparable): the all-seat statistics
        res = eng.tournament(self.table, self.k, self.root, 0, self.shuffles, want_seat_stats=True, want_seat_ratios=False)
        games = self.shuffles * self.gps
        return work_from_seat_stats(res["seat_stats"], games, self.k), None, f"all {games} games of shuffles 0..{self.shuffles - 1} (all-seat statistics of one launch)"

    def hbm_bytes_per_game(self) -> int:
        # counts-only: seat seeds read once (32 B x k) + 2 B x k permutation entries + 4 B schedule entry per game
        return 32 * self.k + 2 * self.k + 4

    def cpu_baseline(self, eng, seconds_target: float = 12.0) -> dict:
        """Time the CPU oracle (checker, never the product) on a bounded sample of the same workload."""
        po = _oracle()
        threads = _cpu_threads()
        t = self.table.view(po.STRATEGY_DTYPE)
        probe = max(2, 12_800 // self.gps)
        po.tournament(t, self.k, self.root, 0, max(1, probe // 4), n_threads=threads)  # warm the thread pool
        t0 = time.perf_counter()
        po.tournament(t, self.k, self.root, 0, probe, n_threads=threads)
        rate = probe * self.gps / (time.perf_counter() - t0)
        n_sh = int(max(probe, min(self.shuffles, rate * seconds_target / self.gps)))
        t0 = time.perf_counter()
        res = po.tournament(t, self.k, self.root, 0, n_sh, n_threads=threads)
        dt = time.perf_counter() - t0
        # the same oracle on ONE core (SURVEY section 8d asks for both): first shuffles of the workload, about 3 s
        p1 = max(1, probe // 8)
        t0 = time.perf_counter()
        po.tournament(t, self.k, self.root, 0, p1, n_threads=1)
        n1 = int(max(p1, min(n_sh, p1 * 3.0 / (time.perf_counter() - t0))))
        t0 = time.perf_counter()
        po.tournament(t, self.k, self.root, 0, n1, n_threads=1)
        dt1 = time.perf_counter() - t0
        # parity on the sample: per-batch tallies of one launch of the step's size
        if n_sh == self.shuffles or self.config != 2:
            got = eng.tournament(self.table, self.k, self.root, 0, n_sh)["tally"][0]
        else:  # config 2: keep every fk_play_kernel launch of this process at the step's size (rocprofv3 per-kernel average)
            got = eng.tournament(self.table, self.k, self.root, 0, (self.shuffles // n_sh) * n_sh, shuffles_per_batch=n_sh)["tally"][0]
        assert np.array_equal(got, res["tally"][0]), "GPU tally differs from the CPU oracle on the baseline sample"
        return {"value": n_sh * self.gps / dt, "unit": "games/s", "cores": threads, "kind": "port",
                "sample": f"shuffles 0..{n_sh - 1} of the same workload ({n_sh * self.gps} games, {dt:.1f} s, OpenMP over shuffles)",
                "single_core": {"value": n1 * self.gps / dt1, "unit": "games/s", "cores": 1,
                                "sample": f"shuffles 0..{n1 - 1} ({n1 * self.gps} games, {dt1:.1f} s)"},
                "parity": f"GPU tally == oracle tally on the sample ({n_sh * self.gps} games)"}

    def describe(self, world: int) -> dict:
        return {"workload": self.label, "k": self.k, "n_strategies": self.S, "root_seed": self.root,
                "games_per_gpu_per_step": self.shuffles * self.gps, "parallelism": f"shuffle-range split x{world}"}


class KSweep:
    """BASELINE configs[3]: k in {2,4,6,8} on the 5 160 grid; each k's shuffle range of the step is split over the ranks."""

    scaling = "strong"

    def __init__(self, table: np.ndarray, games_per_k: int, ks=(2, 4, 6, 8), root: int = 0, config: int = 4, label: str | None = None):
        self.table, self.ks, self.root = table, tuple(ks), root
        self.S = len(table)
        self.n_sh = {k: max(1, games_per_k // (self.S // k)) for k in self.ks}
        self.local_shape = (len(self.ks), self.S, 26)
        self.config = config
        self.k = self.ks[0]
        self.label = label or "BASELINE configs[3]: k in {2,4,6,8}, 5 160-strategy grid, equal games per k, counts-only tallies"

    def games_per_step(self, world: int) -> int:
        return sum(self.n_sh[k] * (self.S // k) for k in self.ks)

    def step(self, eng, index: int, rank: int, world: int, next_index: int | None = None):
        from farkle_ii_amd.distributed import shard_shuffle_range

        out = np.zeros(self.local_shape, dtype=np.int64)
        games, timing = 0, {}
        for i, k in enumerate(self.ks):
            lo, hi = shard_shuffle_range(index * self.n_sh[k], (index + 1) * self.n_sh[k], rank, world)
            if hi > lo:
                out[i] = eng.tournament(self.table, k, self.root, lo, hi)["tally"][0]
                games += (hi - lo) * (self.S // k)
                add_timing(timing, {**eng.timing(), "per_k": {k: eng.timing()}})
        return out, games, timing

    def verify(self, tot: np.ndarray, games: int) -> None:
        for i, k in enumerate(self.ks):
            assert np.array_equal(tot[i][:, 1], tot[i][:, 2] + tot[i][:, 3]) and int(tot[i][:, 0].sum()) * k == int(tot[i][:, 2].sum())
        assert sum(int(tot[i][:, 1].sum()) // k for i, k in enumerate(self.ks)) == games, "exposure conservation failed"

    def sample(self, eng):
        # W of the sweep = games-weighted mean over k (equal games per k): all-seat statistics of one full-size launch per k
        parts = []
        for k in self.ks:
            res = eng.tournament(self.table, k, self.root, 0, self.n_sh[k], want_seat_stats=True, want_seat_ratios=False)
            parts.append(work_from_seat_stats(res["seat_stats"], self.n_sh[k] * (self.S // k), k))
        mean = {key: float(np.mean([p[key] for p in parts])) for key in parts[0]}
        mean["per_k"] = {k: p for k, p in zip(self.ks, parts)}
        return mean, None, "per k: all games of one full-size launch (all-seat statistics)"

    def hbm_bytes_per_game(self) -> int:
        return int(np.mean([32 * k + 2 * k + 4 for k in self.ks]))

    def cpu_baseline(self, eng, seconds_target: float = 12.0) -> dict:
        po = _oracle()
        threads = _cpu_threads()
        t = self.table.view(po.STRATEGY_DTYPE)
        games = 0
        t0 = time.perf_counter()
        for k in self.ks:  # ~1/4 of the budget per k: equal GAMES per k as in the sweep
            n_sh = max(1, int(25_000 * seconds_target / 12.0 * 4 / len(self.ks)) // (self.S // k))
            ref = po.tournament(t, k, self.root, 0, n_sh, n_threads=threads)["tally"][0]
            got = eng.tournament(self.table, k, self.root, 0, n_sh)["tally"][0]
            assert np.array_equal(got, ref), f"GPU tally differs from the CPU oracle on the k={k} sample"
            games += n_sh * (self.S // k)
        dt = time.perf_counter() - t0
        return {"value": games / dt, "unit": "games/s", "cores": threads, "kind": "port",
                "sample": f"first shuffles of every k, {games} games in all, {dt:.1f} s incl. the GPU parity launches",
                "parity": "GPU tally == oracle tally on every k's sample"}

    def describe(self, world: int) -> dict:
        return {"workload": self.label,
                "k": list(self.ks), "n_strategies": self.S, "root_seed": self.root, "shuffles_per_k_per_step": self.n_sh,
                "parallelism": f"every k's shuffle range split x{world}, one tally reduce per k"}


class H2H:
    """BASELINE configs[4]: all pairs of M candidate strategies x orders, `target` completed games per pair."""

    scaling = "strong"

    def __init__(self, table: np.ndarray, candidates, games_per_pair: int, root: int = 42):
        self.table, self.root = table, root
        self.cand = list(candidates)
        self.per_block = games_per_pair // 2
        self.blocks = []
        pair_id = 0
        for i in range(len(self.cand)):
            for j in range(i + 1, len(self.cand)):
                for order in (0, 1):
                    a, b = (self.cand[i], self.cand[j]) if order == 0 else (self.cand[j], self.cand[i])
                    self.blocks.append((pair_id, order, a, b))
                pair_id += 1
        self.local_shape = (len(self.blocks), 5)
        self.config = 5
        self.k = 2
        self._manifest = None
        self.last_attempts = None

    def games_per_step(self, world: int) -> int:
        return len(self.blocks) * self.per_block  # completed games required; attempts are reported separately

    def block_dicts(self, index: int) -> list[dict]:
        """The step's blocks as the dictionaries the reference's schedule holds (h2h_schedule.py:1149-1243); a fresh pair-id
        range per step = fresh attempt coordinates."""
        return [{"block_id": f"s{index}-p{pid}-o{order}", "family_hash": "bench", "schedule_hash": "bench", "pair_id": pid + index * 10_000,
                 "root_index": 0, "root_seed": self.root, "order": order, "seat1_strategy": a, "seat2_strategy": b,
                 "n_completed_required": self.per_block, "max_attempts": 2 * self.per_block}
                for pid, order, a, b in self.blocks]

    def manifest(self):
        """Strategy manifest frame of the candidates (the columns strategies.py:762-800 decodes)."""
        if self._manifest is None:
            import pandas as pd

            t = self.table[self.cand]
            cols = {"strategy_id": list(self.cand), "score_threshold": t["score_threshold"].astype(int), "dice_threshold": t["dice_threshold"].astype(int)}
            for name in ("smart_five", "smart_one", "consider_score", "consider_dice", "require_both", "auto_hot_dice", "run_up_score"):
                cols[name] = t[name].astype(bool)
            cols["favor_dice_or_score"] = np.where(t["favor_score"] != 0, "score", "dice")
            self._manifest = pd.DataFrame(cols)
        return self._manifest

    def step(self, eng, index: int, rank: int, world: int, next_index: int | None = None):
        # the product's schedule-level runner: blocks dealt over the ranks, one fk_h2h_run_blocks per root and rank, every
        # rank gets every block's result back (h2h.run_blocks); this rank's own blocks go into the tally the job reduces
        from farkle_ii_amd.h2h import run_blocks

        res = run_blocks(self.block_dicts(index), self.manifest(), None, engine=eng, rank=rank, world=world)
        out = np.zeros(self.local_shape, dtype=np.int64)
        for b in range(rank, len(self.blocks), world):
            out[b] = [res[b][key] for key in ("games_attempted", "games_completed", "games_safety_limit", "wins_seat1", "wins_seat2")]
        self.last_attempts = np.array([r["games_attempted"] for r in res], dtype=np.int64)
        timing = add_timing({}, eng.timing()) if len(range(rank, len(self.blocks), world)) else {}
        return out, int(out[:, 0].sum()), timing

    def verify(self, tot: np.ndarray, games: int) -> None:
        assert (tot[:, 0] == tot[:, 1] + tot[:, 2]).all() and (tot[:, 1] == tot[:, 3] + tot[:, 4]).all()

    def sample(self, eng, per_block: int = 1500):
        """R, T, W of the attempts the timed launches play: the first `per_block` attempts of EVERY block (attempts of a block
        are i.i.d.), weighted by the attempts each block needed in the last timed step — never-banking pairings run every
        attempt to the round limit and need twice the attempts, so an unweighted or one-block sample is not this workload."""
        from farkle_ii_amd.backend import make_coords

        nb = len(self.blocks)
        pair = np.repeat(np.array([b[0] for b in self.blocks], dtype=np.uint64), per_block)
        order = np.repeat(np.array([b[1] for b in self.blocks], dtype=np.uint64), per_block)
        attempt = np.tile(np.arange(per_block, dtype=np.uint64), nb)
        coords = make_coords(203, self.root, 2, 0, pair, order, attempt)
        pos = {sid: i for i, sid in enumerate(self.cand)}
        seats = np.repeat(np.array([[pos[b[2]], pos[b[3]]] for b in self.blocks], dtype=np.int32), per_block, axis=0)
        rows = eng.play_games(coords, self.table[self.cand], seats, 2)
        R = rows["seats"]["rolls"].astype(np.int64).sum(axis=1).reshape(nb, per_block).mean(axis=1)
        T = rows["seats"]["n_turns"].astype(np.int64).sum(axis=1).reshape(nb, per_block).mean(axis=1)
        w = self.last_attempts.astype(np.float64) if getattr(self, "last_attempts", None) is not None else np.ones(nb)
        w = w / w.sum()
        Rm, Tm = float((R * w).sum()), float((T * w).sum())
        wpg = {"rolls_per_game": Rm, "turns_per_game": Tm, "ops_per_game": 229.0 * Rm + 30.0 * Tm + 850.0 * 2,
               "rolls_per_attempt_by_block_min_max": [float(R.min()), float(R.max())]}
        return wpg, None, (f"first {per_block} attempts of each of the {nb} blocks ({nb * per_block} attempts), block means weighted by the "
                           "attempts each block played in the last timed step")

    def hbm_bytes_per_game(self) -> int:
        return 64

    def cpu_baseline(self, eng, seconds_target: float = 12.0) -> dict:
        po = _oracle()
        target = 20_000
        t0 = time.perf_counter()
        n = 0
        for bi in range(0, len(self.blocks), max(1, len(self.blocks) // 12)):
            b = self.blocks[bi]
            seats = self.table[[b[2], b[3]]]
            ref = po.h2h_block(seats.view(po.STRATEGY_DTYPE), self.root, b[0], b[1], target, 2 * target, 10**9)
            got = eng.h2h_blocks(seats[None], self.root, [b[0]], [b[1]], target, 2 * target)[0]
            assert np.array_equal(np.asarray(got, dtype=np.uint64), ref), f"H2H block {b[:2]} differs from the CPU oracle"
            n += int(ref[0])
            if time.perf_counter() - t0 > seconds_target:
                break
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "attempts/s", "cores": 1, "kind": "port",
                "sample": f"{n} attempts of {target}-game blocks spread over the pair list, {dt:.1f} s incl. the GPU parity launches",
                "parity": "GPU block state == oracle block state on every sampled block"}

    def describe(self, world: int) -> dict:
        return {"workload": f"BASELINE configs[4]: H2H, {len(self.cand)} candidates of the 5 160 grid -> {len(self.blocks) // 2} pairs x "
                            f"orders {{0,1}}, {2 * self.per_block} completed games per pair",
                "k": 2, "n_blocks": len(self.blocks), "root_seed": self.root, "completed_games_per_block": self.per_block,
                "parallelism": f"blocks dealt round-robin x{world}, one reduce of the block results"}


def make_workload(args):
    if args.config == 2:
        return Tournament(2, grid64(), K, ROOT_SEED, args.shuffles or SHUFFLES_PER_STEP,
                          "BASELINE configs[1]: k=2, 64-strategy grid, 10^7 games per GPU per step, root_seed 42, counts-only tallies",
                          sample_shuffles=args.shuffles or SHUFFLES_PER_STEP)
    if args.config == 3:
        return Tournament(3, grid5160(), 4, 0, args.shuffles or 77_520,
                          "BASELINE configs[2]: k=4, default 5 160-strategy grid, 10^8 games per GPU per step, root_seed 0, counts-only tallies",
                          sample_shuffles=4_000)
    if args.config == 4:
        return KSweep(grid5160(), args.games or 250_000_000)
    if args.config == 6:
        # the player counts the reference's production configuration runs (configs/farkle_mega_config.yaml:10 n_players_list), on its
        # default 5 160-strategy grid (5 160 = 2^3 x 3 x 5 x 43 is divisible by every one of them)
        return KSweep(grid5160(), args.games or 100_000_000, ks=(2, 3, 4, 5, 6, 8, 10, 12), config=6,
                      label="the reference's production player counts (farkle_mega_config.yaml n_players_list [2,3,4,5,6,8,10,12]), 5 160-strategy "
                            "grid, equal games per k, counts-only tallies")
    if args.config == 5:
        table = grid5160()
        # candidates: twelve ids spread over the grid (a plausible h2h_2p candidate family)
        cand = [int(i) for i in np.linspace(0, len(table) - 1, 12).astype(int)]
        return H2H(table, cand, args.games or 100_000_000)
    raise SystemExit(f"unknown --config {args.config}")


# ---------------------------------------------------------------------------------------------------------------
# launching ranks
# ---------------------------------------------------------------------------------------------------------------
def launch_ranks(n: int, argv: list[str]) -> int:
    """Start n rank processes of this script (one per GPU) and wait.  Runs BEFORE anything in this process touches
    the GPU; the children are ordinary child processes (no exec of a GPU-initialised process anywhere)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FK_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:  # one rank failed: the others would wait in a collective until their timeout
                    rc = rc or code
                    for o in pending:
                        procs[o].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def load_engine_factory():
    """The product engine (HIP through the C-ABI).  FK_BENCH_ENGINE=module:attr swaps in another factory — used by the
    CPU tests of this file's multi-rank plumbing only; the JSON line then says so and is not a measurement."""
    spec = os.environ.get("FK_BENCH_ENGINE")
    if not spec:
        from farkle_ii_amd.backend import Engine

        return Engine, "farkle_ii_amd.backend.Engine (HIP C-ABI, libfarkle_hip.so)"
    import importlib

    mod, attr = spec.split(":")
    return getattr(importlib.import_module(mod), attr), f"{spec} (TEST STUB - not a measurement)"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5, 6))
    ap.add_argument("--shuffles", type=int, default=0, help="configs 2/3: shuffles per rank per step (default: the BASELINE size)")
    ap.add_argument("--games", type=int, default=0, help="configs 4 / 6: games per k per step; config 5: completed games per pair")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-tally", type=Path, default=None, help="rank 0 saves the reduced tally here (.npy; tests)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {2: 20, 3: 2, 4: 1, 5: 1, 6: 1}[args.config]
    if args.warmup is None:
        # config 2's steps are 9 ms: a fresh process reaches its steady shader clock (2.39 GHz) only after ~50 ms of launches (measured:
        # 2.0 - 2.2 GHz in the first launches), so five warm-up steps, like the driver's own command; the other configurations' steps are
        # 170 ms and longer, one warm-up step is past the ramp
        args.warmup = 5 if args.config == 2 else 1

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries ONE line, the JSON: whatever a library writes to file descriptor 1 meanwhile (gloo announces "[Gloo] Rank 0 is connected
    # to 7 peer ranks" there from C++, once per rank) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    backend_note = None
    fallback_reasons: list[str] = []  # THIS rank's reasons for every fallback taken on the way to the tally reduce (gathered into the JSON line)
    have_gpu = torch.cuda.is_available()
    # FK_DIST_BACKEND=gloo rehearses the multi-rank path on a one-GPU box (ranks share GPU 0, tallies reduced on CPU);
    # the real runs use nccl = RCCL over xGMI with one GPU per rank.
    backend = os.environ.get("FK_DIST_BACKEND", "nccl")
    if backend == "gloo" or os.environ.get("FK_BENCH_SHARE_GPU"):
        # rehearsals on a box with fewer GPUs than ranks: ranks share the devices (FK_BENCH_SHARE_GPU with the RCCL backend makes
        # RCCL refuse the duplicate device — which exercises the collective fallback on real hardware)
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    data_group = None  # the process group the tally reduction runs on (RCCL when it came up on EVERY rank)
    if distributed:
        import datetime

        import torch.distributed as dist

        if have_gpu:
            torch.cuda.set_device(local_rank)
        # Control plane (rendezvous, barriers, agreement flags): gloo over TCP.  The data path's one exchange gets its own
        # RCCL group, and whether it is used is decided COLLECTIVELY: every rank tries it, the ranks agree on the minimum of
        # their success flags over gloo, so no rank can end up on another backend than its peers (a per-rank fallback inside
        # an `except` would split the job and hang it).  FK_DIST_BACKEND=gloo skips RCCL altogether (CPU rehearsals).
        dist.init_process_group("gloo")
        if backend == "nccl" and have_gpu:
            ok, why = 1, ""
            try:
                data_group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180))
                probe = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", local_rank))
                dist.all_reduce(probe, group=data_group)  # eager: RCCL trouble shows up here, not inside the timed region
                torch.cuda.synchronize()
                ok = int(int(probe.item()) == world)
            except Exception as exc:
                ok, why = 0, f"{type(exc).__name__}: {str(exc)[:200]}"
            if not ok:
                fallback_reasons.append(f"torch.distributed nccl group: {why or 'all_reduce probe returned a wrong sum'}")
            flag = torch.tensor([ok], dtype=torch.int64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                data_group = None
                backend = "gloo"
                backend_note = f"RCCL group did not come up on every rank ({why or 'failed on another rank'}); tally reduced over gloo"
                print(f"rank {rank}: {backend_note}", file=sys.stderr)
        else:
            backend = "gloo"
    n_gpus = dist.get_world_size() if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"note: --gpus {args.gpus} but the process group has {n_gpus} ranks; reporting n_gpus={n_gpus}", file=sys.stderr)

    make_engine, engine_name = load_engine_factory()
    wl = make_workload(args)
    eng = make_engine(local_rank)
    info = eng.device_info()
    try:  # every game-kernel workgroup stamps s_memtime / s_memrealtime at its first and last instruction (two scalar reads and one
        # (FK_BENCH_CLOCK_STAMPS=0 turns them off: the A/B of profiles/r06_ab_clock_stamps.json — their cost inside the timed steps)
        eng.set_option("clock_stamps", 0 if os.environ.get("FK_BENCH_CLOCK_STAMPS") == "0" else 1)  # 32-byte store per workgroup and launch): the clock the fractions below are also priced at
    except Exception:
        pass  # (the CPU test stub has no such option)
    dev = torch.device("cuda", local_rank) if have_gpu else torch.device("cpu")
    red_dev = dev if data_group is not None else torch.device("cpu")  # where the tally reduction runs

    def sync() -> None:
        if distributed:
            dist.barrier()
        if have_gpu:
            torch.cuda.synchronize(dev)

    # The tally reduction itself.  Default under RCCL: the engine's own communicator (fk_comm_init) — ncclReduce on the engine's
    # stream through the C-ABI, no PyTorch in the data path; for the tournament workloads the tally stays in HBM from the
    # game kernels' post-pass to the reduce (fk_tally_resident_reduce: one D2H, on the root).  FK_TALLY_REDUCE=torch keeps
    # torch.distributed's reduce on the RCCL group; both are RCCL over xGMI.  Again decided collectively.
    tally_reduce = "single process"
    use_fk_comm = False
    resident = False
    if distributed:
        tally_reduce = f"torch.distributed.reduce ({'nccl = RCCL' if data_group is not None else 'gloo'})"
        if data_group is not None and os.environ.get("FK_TALLY_REDUCE", "rccl") == "rccl":
            from farkle_ii_amd.distributed import init_engine_comm

            ok = 0
            try:
                ok = int(bool(init_engine_comm(eng)))
            except Exception as exc:
                fallback_reasons.append(f"engine communicator (fk_comm_init): {type(exc).__name__}: {str(exc)[:200]}")
                print(f"rank {rank}: fk_comm_init failed ({type(exc).__name__}: {str(exc)[:200]}); torch.distributed reduce instead", file=sys.stderr)
            else:
                if not ok:
                    fallback_reasons.append("engine communicator (fk_comm_init): returned no communicator")
            flag = torch.tensor([ok], dtype=torch.int64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            use_fk_comm = int(flag.item()) == 1
            if use_fk_comm:
                resident = isinstance(wl, Tournament) and hasattr(eng, "reduce_resident_tally")
                tally_reduce = ("fk_tally_resident_reduce (tally resident in HBM, ncclReduce int64 sum through the C-ABI, one D2H on the root)"
                                if resident else "fk_reduce_tally (ncclReduce int64 sum through the C-ABI)")
                if resident:
                    eng.set_option("resident_tally", 1)
    rccl_ranks = (eng.comm_ranks() if use_fk_comm and hasattr(eng, "comm_ranks") else
                  (dist.get_world_size(data_group) if data_group is not None else 0))

    def reduce_to_rank0(local: np.ndarray):
        """The path's only exchange: one SUM of the int64 tally to rank 0 at the end of the job, the
        analogue of OutcomeCounter.absorb — RCCL over xGMI when there is more than one rank."""
        if resident:
            tot = eng.reduce_resident_tally((1,) + tuple(local.shape), 0)
            return torch.from_numpy(tot[0] if tot is not None else local)
        if use_fk_comm:
            return torch.from_numpy(eng.reduce_tally(local, 0))
        t = torch.from_numpy(local).to(red_dev)
        if distributed:
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, group=data_group)
        return t

    # one-time initialisation outside any step: device workspace, lazily loaded torch kernels, RCCL communicator
    wl.step(eng, 0, rank, n_gpus)
    warm = torch.zeros(wl.local_shape, dtype=torch.int64, device=red_dev)
    warm += torch.from_numpy(np.zeros(wl.local_shape, dtype=np.int64)).to(red_dev)
    if distributed:
        dist.reduce(warm, dst=0, op=dist.ReduceOp.SUM, group=data_group)
    local = np.zeros(wl.local_shape, dtype=np.int64)
    for i in range(args.warmup):  # (the last warm-up step does not prepare the first timed step)
        tally, _, _ = wl.step(eng, i, rank, n_gpus, next_index=i + 1 if i + 1 < args.warmup else None)
        local += tally
    # The warm-up reduce is also the engine communicator's FIRST collective.  If it fails or runs into the deadline on any rank
    # (fk_reduce_tally / fk_tally_resident_reduce return FK_ERR_COMM and abort the communicator), every rank learns of it over
    # gloo and the job continues on torch.distributed's reduce — decided collectively, outside the timed region.
    reduce_ok, reduce_why = 1, ""
    try:
        reduce_to_rank0(local)
    except Exception as exc:
        if not (distributed and use_fk_comm):
            raise
        reduce_ok, reduce_why = 0, f"{type(exc).__name__}: {str(exc)[:200]}"
        fallback_reasons.append(f"engine communicator's first reduce: {reduce_why}")
    if distributed and use_fk_comm:
        flag = torch.tensor([reduce_ok], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            print(f"rank {rank}: the engine communicator's first reduce failed ({reduce_why or 'on another rank'}); "
                  "torch.distributed reduce instead", file=sys.stderr)
            use_fk_comm = resident = False
            eng.set_option("resident_tally", 0)
            try:
                eng.comm_destroy()
            except Exception:
                pass
            tally_reduce = f"torch.distributed.reduce ({'nccl = RCCL' if data_group is not None else 'gloo'}) after the engine communicator's first reduce failed"
            rccl_ranks = dist.get_world_size(data_group) if data_group is not None else 0
    local[:] = 0
    my_games = 0
    t: dict = {}
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tally, g, st = wl.step(eng, args.warmup + i, rank, n_gpus, next_index=args.warmup + i + 1 if i + 1 < args.steps else None)
        local += tally  # host add of the step's int64 tally
        my_games += g
        add_timing(t, st)  # HIP events on the engine's stream, summed over every engine call of the timed steps
    total = reduce_to_rank0(local)  # inside the timed region
    sync()
    elapsed = time.perf_counter() - t0
    if distributed:
        e = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        elapsed = float(e.item())
        g_all = torch.tensor([my_games], dtype=torch.int64)
        dist.all_reduce(g_all, op=dist.ReduceOp.SUM)
        played = int(g_all.item())
    else:
        played = my_games
    dist_fallback_reason = None
    if distributed:  # every rank's reasons to rank 0 (gloo, outside the timed region)
        gathered = [None] * n_gpus if rank == 0 else None
        dist.gather_object(fallback_reasons, gathered, dst=0)
        if rank == 0 and any(gathered):
            dist_fallback_reason = {f"rank {r}": why for r, why in enumerate(gathered) if why}

    total_games = wl.games_per_step(n_gpus) * args.steps
    value = total_games / elapsed

    if rank == 0:
        tot = total.cpu().numpy()
        wl.verify(tot, played if wl.config != 5 else total_games)
        if args.dump_tally is not None:
            np.save(args.dump_tally, tot)

        # live per-game work (R, T of SURVEY section 8d) from the rows of a launch of the same workload
        rows, k_s, sample_desc = wl.sample(eng)
        wpg = dict(rows) if isinstance(rows, dict) else work_per_game(rows, k_s)
        del rows
        wpg["sample"] = sample_desc
        # dominant kernel: average launch duration and games per launch over the last engine call of every timed step
        launches = max(int(t.get("play_launches", 0)), 1)
        kernel_ms = t.get("play_ms", 0.0) / launches
        games_per_launch = t.get("games", 0) / launches
        kernel_games_per_s = games_per_launch / max(kernel_ms * 1e-3, 1e-12)
        # VALU roof: CUs x 4 SIMD x 32 lanes/clk x clock (MI355X_MICROARCH.md: wave64 issues over 2 cycles on a SIMD-32)
        peak_ops = info["compute_units"] * 4 * 32 * info["clock_mhz"] * 1e6
        achieved_ops = kernel_games_per_s * wpg["ops_per_game"]
        hbm_bpg = wl.hbm_bytes_per_game()
        # HBM bytes per launch from the PMC passes of the same launch shape (rocprofv3 cannot run inside this process):
        # profiles/r*_hbm_traffic.json carries the commit it was taken at; dropped when that is not an ancestor's kernel
        def kernel_of(shape: dict, k_seats: int) -> str:
            """fk_play_hc_kernel launches are recognisable by their LDS size: 16 bytes per seat and lane (+ the 12 656-byte table
            image), or 32 with the cold records in LDS (csrc/fk_play_hc.h); everything else is fk_play_kernel."""
            block, lds = shape.get("play_block"), shape.get("play_lds_bytes")
            hc = bool(block) and lds in (block * 16 * k_seats, block * 16 * k_seats + 12656, block * 32 * k_seats)
            return "fk_play_hc_kernel" if hc else "fk_play_kernel"

        dominant = kernel_of(t, int(wl.k)) if isinstance(getattr(wl, "k", None), int) else "fk_play_kernel"  # (sweep / H2H lines: per_k / k = 2)
        def traffic_for(kernel: str, k_seats: int | None = None):
            """HBM bytes per launch of `kernel` from the newest PMC stamp of this workload (per player count for the sweep), or None
            when the kernel sources have changed since it was taken."""
            suffix = f"config{wl.config}" + (f"_k{k_seats}" if k_seats is not None else "")
            for tpath in sorted((ROOT / "profiles").glob(f"r*_hbm_traffic_{suffix}.json"), reverse=True):
                rec = json.loads(tpath.read_text())
                if rec.get("kernel_source_sha256") == kernel_source_sha():
                    return rec["kernels"].get(kernel, {}).get("hbm_bytes_corrected")
                break
            return None

        def mix_for(kernel: str, shape: dict, k_seats: int, frac: float):
            """Static instruction mix of the instance's roll loop priced with the two measured issue classes (tools/isa_mix.py ->
            profiles/r*_isa_mix.json, valid while the kernel sources are the ones it was taken on): `ceiling_frac` is the fraction of the
            NOMINAL peak a SIMD issuing this mix back to back would reach, `frac_of_mix_ceiling` = frac / that.  None when unknown."""
            for mpath in sorted((ROOT / "profiles").glob("r*_isa_mix.json"), reverse=True):
                doc = json.loads(mpath.read_text())
                if doc.get("kernel_source_sha256") != kernel_source_sha():
                    break
                block, mixed = shape.get("play_block"), shape.get("play_mixed_flags")
                hits = []
                for name, rec in doc["instances"].items():
                    a = [int(x) for x in name[name.index("<") + 1:-1].split(",")]
                    if not name.startswith(kernel + "<") or a[0] != block:
                        continue
                    if kernel == "fk_play_kernel":  # <BLOCK, LEAN, WPE, MIXED, GS, BLK, KC>
                        ok = a[1] == 1 and a[3] == mixed and a[4] == 0 and a[5] == (1 if wl.config == 5 else 0) and a[6] == (2 if k_seats == 2 else 0)
                    else:  # <BLOCK, MIXED, LT, KI, WPE, PKR, CL, NS, CR, IL>: KI seats in registers (0 = the cold-in-LDS instance of k <= 4)
                        ki = 0 if k_seats <= 4 else k_seats if k_seats <= 8 else 10 if k_seats <= 10 else 12
                        ok = a[1] == mixed and a[3] == ki
                    if ok:
                        hits.append((name, rec))
                if len(hits) == 1:
                    name, rec = hits[0]
                    return {"instance": name, "source": f"profiles/{mpath.name} (static mix of the roll loop; classes: profiles/r05_valu_issue_rates.txt)",
                            **{key: rec[key] for key in ("valu_static", "full_rate", "half_rate", "mean_issue_cycles", "ceiling_frac")},
                            "frac_of_mix_ceiling": frac / rec["ceiling_frac"]}
                break
            return None

        traffic = traffic_for(dominant)
        roofline = {
            "bound": "valu", "kernel": dominant,
            "achieved": achieved_ops / 1e12, "peak": peak_ops / 1e12, "unit": "Tlane-op/s (int32)", "frac": achieved_ops / peak_ops,
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC, separate pass; null when the kernel sources changed since)",
            "kernel_ms": kernel_ms, "games_per_launch": games_per_launch, "kernel_games_per_s": kernel_games_per_s, **wpg,
            # ONE number, stated once: kernel_ms is the mean duration of the game kernel over the launches of the TIMED steps only (HIP
            # events on the engine's stream).  A rocprofv3 --stats average of this command also counts the warm-up steps (clock still
            # ramping), the statistics launch that measures W and the CPU-baseline parity launch: compare with its rows for the timed
            # launches, not with its mean (round 5: 9.31 ms average over nine launches against 8.88 ms here for the same kernel).
            "kernel_ms_is": "mean over the launches of the timed steps (HIP events); a rocprofv3 average of this command also includes warm-up, statistics and parity launches",
            "kernel_ms_min": t.get("play_ms_min"), "kernel_ms_max": t.get("play_ms_max"), "timed_launches": t.get("play_launches"),
            "hbm": {"achieved": kernel_games_per_s * hbm_bpg / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": kernel_games_per_s * hbm_bpg / 8e12, "bytes_per_game": hbm_bpg},
            # permutations: every launch (they run on the main stream, in front of the previous game kernel when pipelined);
            # seat seeding: only the launches that were NOT seeded on the side stream behind the previous game kernel
            "seed_kernel_ms": t.get("seed_ms", 0.0) / max(launches - int(t.get("prefetched_chunks", 0)), 1),
            "perm_kernel_ms": t.get("perm_ms", 0.0) / max(launches, 1),
            "launches_seeded_behind_the_previous_kernel": int(t.get("prefetched_chunks", 0)),
            "launch": {k2: t.get(k2) for k2 in ("play_block", "play_grid", "play_lds_bytes", "play_mixed_flags")},
        }
        roofline["mix"] = mix_for(dominant, t, int(wl.k) if isinstance(getattr(wl, "k", None), int) else 2, roofline["frac"])

        def with_clock(rec: dict, tt: dict) -> None:
            """`frac` stays defined against the NOMINAL clock (as every round before); next to it the clock the game kernels
            really held (in-kernel s_memtime / s_memrealtime, median over workgroups, mean over the timed launches) and the fraction of
            the peak at THAT clock — the chip lowers its clock under load (MI355X_MICROARCH.md, DVFS give-back)."""
            if tt.get("clock_n"):
                mhz = tt["clock_sum"] / tt["clock_n"]
                rec["clock_mhz_measured"] = mhz
                rec["frac_at_measured_clock"] = rec["frac"] * info["clock_mhz"] / mhz
            if tt.get("tail_n"):  # how long the last launch of a call took to drain behind its longest games
                rec["launch_tail_ms"] = tt["tail_ms_sum"] / tt["tail_n"]

        with_clock(roofline, t)
        if "per_k" in wpg:  # config 4: one roofline record per player count (kernel time, W and fraction of each k's launches)
            per_k = []
            for k2, w2 in wpg.pop("per_k").items():
                tk = (t.get("per_k") or {}).get(k2, {})
                n_l = max(int(tk.get("play_launches", 0)), 1)
                ms = tk.get("play_ms", 0.0) / n_l
                gpl = tk.get("games", 0) / n_l
                rate = gpl / max(ms * 1e-3, 1e-12)
                per_k.append({"k": int(k2), "kernel": kernel_of(tk, int(k2)), "kernel_ms": ms,
                              "games_per_launch": gpl, "kernel_games_per_s": rate, **w2,
                              "frac": rate * w2["ops_per_game"] / peak_ops, "traffic": traffic_for(kernel_of(tk, int(k2)), int(k2)),
                              "launch": {k3: tk.get(k3) for k3 in ("play_block", "play_grid", "play_lds_bytes", "play_mixed_flags")}})
                per_k[-1]["mix"] = mix_for(per_k[-1]["kernel"], tk, int(k2), per_k[-1]["frac"])
                with_clock(per_k[-1], tk)
            roofline["per_k"] = per_k
            # The line's headline record is ONE kernel's, whole: the player count furthest below its roofline (every field — kernel, time,
            # games, W, achieved, frac, clock, traffic, launch, hbm — from that k's launches; round 4 mixed the mean fraction with the worst
            # k's kernel fields).  The mean over the player counts stands beside it under its own name.
            worst = min(per_k, key=lambda r: r["frac"])
            roofline["frac_mean_over_k"] = float(np.mean([r["frac"] for r in per_k]))
            hbm_k = 32 * worst["k"] + 2 * worst["k"] + 4
            roofline.update({key: worst[key] for key in worst if key != "k"})
            roofline.update({"kernel_note": f"slowest player count of the sweep: k = {worst['k']} (every headline field is that k's; mean over k in frac_mean_over_k)",
                             "achieved": worst["frac"] * peak_ops / 1e12,
                             "hbm": {"achieved": worst["kernel_games_per_s"] * hbm_k / 1e9, "peak": 8000.0, "unit": "GB/s",
                                     "frac": worst["kernel_games_per_s"] * hbm_k / 8e12, "bytes_per_game": hbm_k}})
        cpu = None
        if not args.no_cpu_baseline and n_gpus == 1:  # the CPU leg runs on rank 0 of the single-GPU run only
            cpu = wl.cpu_baseline(eng)
        line = {
            "metric": METRIC if wl.config in (2,) else (f"{METRIC} [variant: BASELINE config {wl.config}]" if wl.config != 6 else
                                                          f"{METRIC} [variant: sweep over the reference's production player counts]"),
            "value": value, "unit": "games/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": wl.scaling, "vs_baseline": None,
            "dtype": "int32", "data": "synthetic", "engine": engine_name, "tally_reduce": tally_reduce,
            "dist_backend": (("nccl (RCCL over xGMI) for the tally reduce; gloo control plane" if data_group is not None
                              else (backend_note or "gloo")) if distributed else None),
            "rccl_ranks": rccl_ranks if distributed else None,
            # why the job is NOT on "engine communicator over RCCL", per rank that had a reason (stderr carried these before round 6): null
            # when every rank took the first choice
            "dist_fallback_reason": dist_fallback_reason,
            "launcher": "self (bench.py started the ranks)" if os.environ.get("FK_BENCH_SELF_LAUNCHED") else
                        ("torch.distributed.run" if distributed else "single process"),
            "config": {**wl.describe(n_gpus), "device": info["name"], "arch": info["arch"], "compute_units": info["compute_units"],
                       "clock_mhz": info["clock_mhz"]},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if wl.config == 5:
            line["attempts_per_s"] = played / elapsed
            line["value_note"] = "completed games required by the schedule per second; attempts (incl. safety-limit games) in attempts_per_s"
        if wl.config == 2:
            # not `vs_baseline` (BASELINE.json publishes nothing for this exact config): the reference's own report of the
            # k=2 tournament path on its 80-strategy grid, Ryzen 7 3700X, 12 workers = 1 142.9 games/s
            line["vs_reference_published_12_workers"] = value / 1142.9
            # the Python reference itself timed on this same workload in the build container (oracle/time_reference.py; the
            # reference cannot travel to the GPU box, so this is a committed fixture, not a measurement of this run)
            fixture = ROOT / "tests" / "golden" / "reference_cpu_timing.json"
            if fixture.exists():
                ref = json.loads(fixture.read_text())
                line["reference_python_fixture"] = {
                    "tournament_loop_games_per_s_1_process": ref["tournament_loop_1_process"]["games_per_s"],
                    "tournament_loop_games_per_s_8_processes": ref["tournament_loop_8_processes"]["games_per_s"],
                    "where": ref["host"]["note"]}
            e2e = sorted((ROOT / "profiles").glob("r*_farkle_run_end_to_end.json"))
            if e2e:  # the latest committed measurement of `farkle run` end to end (rows off / on), not of this run
                line["farkle_run_end_to_end_fixture"] = {"file": f"profiles/{e2e[-1].name}", **json.loads(e2e[-1].read_text())}
        os.write(json_fd, (json.dumps(line) + "\n").encode("utf-8"))
    eng.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def kernel_source_sha() -> str:
    """sha256 over the CODE of the device sources (kernels + per-roll device functions; comments and white space left out, so that
    a reworded comment does not orphan a measurement): profiles carry it so that an HBM-traffic figure or an instruction mix taken on
    other kernels is never attached."""
    import hashlib
    import re

    h = hashlib.sha256()
    for name in ("fk_kernels.h", "fk_play_hc.h", "fk_device.h"):
        text = (ROOT / "farkle_ii_amd" / "csrc" / name).read_text(encoding="utf-8")
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)   # (no string literal of these files holds a comment marker)
        text = re.sub(r"//[^\n]*", " ", text)
        h.update(" ".join(text.split()).encode("utf-8"))
    return h.hexdigest()


if __name__ == "__main__":
    main()
