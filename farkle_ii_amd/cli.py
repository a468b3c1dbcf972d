"""``farkle`` command line for the simulation path: ``run`` and ``time``.

Mirrors ``src/farkle/cli/main.py`` (:53-140 parser, :325-464 dispatch) for the two commands on this path; the
analysis/orchestration commands of the reference are out of scope and are rejected with a clear message.

    python -m farkle_ii_amd --config configs/fast.yaml --set sim.n_players_list=[2] --set sim.seed_list=[42] run --metrics
    python -m farkle_ii_amd time --players 2 --n-games 1000 --seed 42
    torchrun --nproc-per-node 8 -m farkle_ii_amd --config cfg.yaml run      (one process per GPU, RCCL tally reduce)
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
from pathlib import Path
from typing import Sequence

LOGGER = logging.getLogger("farkle_ii_amd.cli")
_OUT_OF_SCOPE = ("watch", "analyze", "two-seed-pipeline")


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(prog="farkle")
    parser.add_argument("--config", type=Path, help="Path to YAML configuration")
    parser.add_argument("--set", dest="overrides", action="append", default=[], metavar="KEY=VALUE",
                        help="Override configuration values")
    parser.add_argument("--log-level", default="INFO", help="Root logging level")
    sub = parser.add_subparsers(dest="command", required=True)
    run = sub.add_parser("run", help="Run a tournament")
    run.add_argument("--metrics", action="store_true", help="Collect per-strategy metrics in addition to win counts")
    run.add_argument("--row-dir", type=Path, help="Write full per-game rows to this directory")
    run.add_argument("--all-player-batches", nargs="?", const=Path("all_player_batches"), type=Path, default=None, metavar="DIR",
                     help="Write the unconditional all-player batch metrics (integer columns of the reference's "
                          "all_player_batch_schema) per deterministic batch, from device accumulators, without rows")
    run.add_argument("--rng-lag-sums", action="store_true",
                     help="Write the lag sufficient statistics of the RNG diagnostics' strategy family (per strategy and "
                          "analysis.rng_diagnostic_lags: pairs, sums, square sums and cross sums of the win indicator and of n_rounds) "
                          "and the autocorrelation rows computed from them, from device accumulators, without rows.  The series spans every "
                          "shuffle of the run and is held in memory until the end: an interrupted --rng-lag-sums run cannot resume (the next "
                          "invocation asks for --force), and a run completed without it must be replayed with --force")
    run.add_argument("--sidecars", action="store_true",
                     help="Write <artifact>.sidecar.json (producer contract + SHA-256 / size of the artifact) beside every output")
    run.add_argument("--code-identity", metavar="COMMIT[:DIRTY_SHA256[:POLICY]]",
                     help="Write artifact-contract-v3 sidecars, sealed shard manifests and the authenticated simulation.done.json, signed with "
                          "this code identity: the one the checkout that will run the reference's `analyze ingest` resolves (its Git commit and, "
                          "for a dirty tree, the worktree fingerprint).  Implies --sidecars")
    run.add_argument("--reference-checkout", type=Path, metavar="PATH",
                     help="Like --code-identity, with the identity resolved from the Git checkout at PATH the way the reference does "
                          "(rev-parse HEAD; staged + worktree diff + inventoried untracked files for a dirty tree)")
    run.add_argument("--force", action="store_true", help="Recompute even when existing run artifacts are available")
    t = sub.add_parser("time", help="Benchmark simulation throughput")
    t.add_argument("--players", type=int, default=5, help="Players per game (default: 5)")
    t.add_argument("--n-games", dest="n_games", type=int, default=1000, help="Number of games to run (default: 1000)")
    t.add_argument("--jobs", type=int, default=1, help="Parallel jobs (accepted for compatibility)")
    t.add_argument("--seed", type=int, default=42, help="Seed (default: 42)")
    for name in _OUT_OF_SCOPE:
        sub.add_parser(name, help="(reference command outside the simulation path)")
    return parser


def _maybe_init_distributed() -> None:
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            local = int(os.environ.get("LOCAL_RANK", "0"))
            if torch.cuda.is_available():
                torch.cuda.set_device(local)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group("gloo")


def main(argv: Sequence[str] | None = None) -> None:
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=getattr(logging, str(args.log_level).upper(), logging.INFO),
                        format="%(asctime)s %(levelname)s %(name)s: %(message)s")
    if args.command in _OUT_OF_SCOPE:
        raise SystemExit(f"farkle {args.command}: outside the simulation path this engine replaces; use the reference CLI")
    if args.command == "time":
        from .time_farkle import measure_sim_times

        out = measure_sim_times(n_games=args.n_games, players=args.players, seed=args.seed, jobs=args.jobs)
        print(f"{args.n_games} games, {args.players} players: {out['games_per_sec']:.1f} games/s; winners {out['winners']}")
        return
    from . import runner
    from .config import AppConfig, apply_dot_overrides, load_app_config

    cfg = load_app_config(args.config, seed_list_len=None) if args.config is not None else AppConfig()
    cfg = apply_dot_overrides(cfg, list(args.overrides or []))
    if cfg.sim.seed_list is not None and len(cfg.sim.seed_list) != 1:
        raise ValueError(f"sim.seed_list must contain exactly 1 seeds, got {cfg.sim.seed_list!r}")
    cfg.sim.populate_seed_list(1)
    if args.metrics:
        cfg.sim.expanded_metrics = True
    if args.row_dir is not None:
        cfg.sim.row_dir = args.row_dir
    if args.sidecars:
        cfg.sim.sidecars = True
    if args.code_identity and args.reference_checkout:
        raise SystemExit("farkle run: --code-identity and --reference-checkout are two ways to say the same thing; pass one")
    if args.code_identity or args.reference_checkout:
        from .contract_v3 import parse_code_identity, resolve_code_identity

        cfg._code_identity = parse_code_identity(args.code_identity) if args.code_identity else resolve_code_identity(args.reference_checkout)
        cfg.sim.sidecars = True
    if args.all_player_batches is not None:
        cfg.sim.all_player_batch_dir = args.all_player_batches
    if args.rng_lag_sums:
        cfg.sim.rng_lag_sums = True
    _maybe_init_distributed()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("FK_TALLY_REDUCE", "rccl") == "rccl":
        # The per-group tally reduction through the C-ABI's own RCCL communicator (fk_comm_init / fk_reduce_tally): the default
        # whenever it comes up on EVERY rank (the ranks agree on the minimum of their success flags, so none is left on another
        # path); otherwise — and with FK_TALLY_REDUCE=torch — torch.distributed's reduce on the process group.
        from .distributed import agree, init_engine_comm
        from .engine import get_engine

        ok = False
        try:
            ok = bool(init_engine_comm(get_engine()))
        except Exception as exc:  # librccl missing, ncclCommInitRank failed, a CPU-only engine ...
            LOGGER.warning("fk_comm_init failed (%s: %s)", type(exc).__name__, exc)
        everywhere = agree(ok)
        if not everywhere:
            from . import distributed

            distributed._ENGINE_COMM = None
        LOGGER.info("tally reduce: %s", "fk_reduce_tally (RCCL through the C-ABI)" if everywhere else "torch.distributed.reduce")
    rank = int(os.environ.get("RANK", "0"))
    # the resolved configuration beside the results: written on the runner's helper thread, under the first engine call (ctypes drops
    # the GIL for its duration) — 2 ms of a 25-ms `farkle run`; any error of the write surfaces when the run has finished
    active = runner._helper_thread().submit(runner.write_active_config, cfg, cfg.results_root) if rank == 0 else None
    LOGGER.info("Dispatching run command: seed=%s n_players_list=%s results_dir=%s", cfg.sim.seed, cfg.sim.n_players_list,
                cfg.results_root)
    try:
        if len(cfg.sim.n_players_list) > 1:
            out = runner.run_multi(cfg, force=args.force)
        else:
            out = {cfg.sim.n_players_list[0]: runner.run_single_n(cfg, cfg.sim.n_players_list[0], force=args.force)}
    finally:
        if active is not None:
            active.result()
    if rank == 0:
        print({f"{k}p_games": v for k, v in out.items()})


if __name__ == "__main__":
    main(sys.argv[1:])
