"""Eight ranks without hardware (round-4 review: the CPU tests used world_size 2 only).  The pool offers no multi-GPU box, so what CAN be
proven here is proven here: the shard arithmetic at 8 ranks with batch counts that do not divide by 8, the one tally reduce, the H2H prefix
cut with the cut in the first, a middle and the last rank, and bench.py's own 8-rank launch (configs 3 and 4) — all over gloo with the
CPU oracle behind the engine interface (tests/oracle_engine_stub.py).  On an 8-GPU node the same code runs the HIP engine over RCCL."""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))

H2H_SEATS = [(300, 2, 1, 1, 1, 1, 0, 1, 1, 1, 0), (0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1)]  # seat 2 never banks: safety-limit games occur


def test_shard_arithmetic_covers_every_batch_exactly_once():
    """shard_shuffle_range: contiguous, disjoint, whole deterministic batches, union = the range — for the BASELINE shapes at 8 ranks
    (config 3: 77 520 shuffles; config 4: 2.5 x 10^8 games per k = 96 899 / 193 798 / 290 697 / 387 596 shuffles of the 5 160 grid),
    for step-shifted ranges as bench.py builds them, and for batch counts below / not divisible by the rank count."""
    from farkle_ii_amd.distributed import shard_shuffle_range

    cases = [(0, 77_520, 1), (0, 96_899, 1), (193_798, 2 * 193_798, 1), (290_697 * 3, 290_697 * 4, 1), (0, 387_596, 1),
             (0, 1_000, 30), (0, 1_001, 30), (60, 1_001, 30), (0, 5, 1), (0, 7 * 16 + 3, 16), (0, 3, 8), (0, 0, 4)]
    for begin, end, bs in cases:
        for world in (1, 2, 3, 5, 7, 8):
            shards = [shard_shuffle_range(begin, end, r, world, batch_size=bs) for r in range(world)]
            assert shards[0][0] == begin and shards[-1][1] == end, (begin, end, bs, world, shards)
            for (lo, hi), (lo2, _) in zip(shards, shards[1:]):
                assert lo <= hi == lo2, (begin, end, bs, world, shards)
            for lo, hi in shards:  # whole batches: every cut sits on a batch boundary (or at the end of the range)
                assert (lo - begin) % bs == 0 and ((hi - begin) % bs == 0 or hi == end)
            sizes = [-(-(hi - lo) // bs) for lo, hi in shards]
            assert max(sizes) - min(sizes) <= 1, (begin, end, bs, world, sizes)  # balanced to one batch
    with pytest.raises(ValueError):
        shard_shuffle_range(5, 100, 0, 8, batch_size=30)  # a range that does not start on a batch boundary


def _tally_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import golden_util as gu
    import pyoracle as po
    from farkle_ii_amd.distributed import reduce_tally, shard_shuffle_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = gu.strategies_from_tuples(gu.load("grid_vectors.json")["g64"], po.STRATEGY_DTYPE)
    for name, n_sh, spb in (("thirteen", 13 * 8 - 3, 8), ("five", 5 * 6, 6)):  # 13 batches (the last one short) and 5 batches over 8 ranks
        n_batches = -(-n_sh // spb)
        lo, hi = shard_shuffle_range(0, n_sh, rank, world, batch_size=spb)
        local = np.zeros((n_batches, 64, 26), dtype=np.int64)
        if hi > lo:
            mine = po.tournament(table, 2, 42, lo, hi, shuffles_per_batch=spb)["tally"]
            local[lo // spb: lo // spb + len(mine)] = mine
        total = reduce_tally(local, dst=0)
        if rank == 0:
            np.save(os.path.join(out_dir, f"{name}.npy"), total)
    dist.destroy_process_group()


def test_eight_rank_tally_reduce_with_uneven_batch_counts(tmp_path):
    import torch.multiprocessing as mp

    sys.path.insert(0, str(ROOT / "oracle"))
    import golden_util as gu
    import pyoracle as po

    port = 33500 + os.getpid() % 2000
    mp.spawn(_tally_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    table = gu.strategies_from_tuples(gu.load("grid_vectors.json")["g64"], po.STRATEGY_DTYPE)
    for name, n_sh, spb in (("thirteen", 13 * 8 - 3, 8), ("five", 5 * 6, 6)):
        assert np.array_equal(np.load(tmp_path / f"{name}.npy"), po.tournament(table, 2, 42, 0, n_sh, shuffles_per_batch=spb)["tally"]), name


def _h2h_cases():
    """(target, max_attempts, chunk) whose serial stop falls into the first, a middle and the last of eight attempt sub-ranges, plus
    a target that is not reached and a resumed block — found by running the serial block (the oracle) and locating its last attempt."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import pyoracle as po

    seats = np.zeros(2, dtype=po.STRATEGY_DTYPE)
    seats[0], seats[1] = H2H_SEATS
    chunk = 800
    want_ranks, cases, seen = [0, 3, 7], [], {}
    for target in range(1, 800):  # (at a 12-round limit most attempts hit the safety limit: completed games are rare, so are the targets)
        out = po.h2h_block(seats, 7, 3, 1, target, chunk, chunk, max_rounds=12)
        if int(out[1]) < target:
            break
        seen.setdefault((int(out[0]) - 1) * 8 // chunk, target)
    assert all(r in seen for r in want_ranks), f"serial stops found in sub-ranges {sorted(seen)} only"
    cases = [(seen[r], chunk, chunk, None, r) for r in want_ranks]
    cases.append((10_000, 700, 700, None, None))                      # not reached: every sub-range counts in full
    cases.append((400, 900, 333, (120, 110, 10, 60, 50), None))       # a resumed block, sub-ranges of uneven length
    return seats, cases


def _h2h_worker(rank: int, world: int, port: int, out_path: str, cases) -> None:
    for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import pyoracle as po
    from farkle_ii_amd.distributed import h2h_block_distributed

    dist.init_process_group("gloo", rank=rank, world_size=world)
    seats = np.zeros(2, dtype=po.STRATEGY_DTYPE)
    seats[0], seats[1] = H2H_SEATS
    results = []
    for target, max_attempts, chunk, start, _ in cases:
        st = np.zeros(5, dtype=np.uint64) if start is None else np.array(start, dtype=np.uint64)
        results.append(h2h_block_distributed(po.h2h_block, seats, 7, 3, 1, target, max_attempts, chunk, state=st, max_rounds=12).astype(np.int64))
    if rank == 0:
        np.save(out_path, np.stack(results))
    dist.destroy_process_group()


def test_eight_rank_h2h_prefix_cut_in_the_first_a_middle_and_the_last_rank(tmp_path):
    import torch.multiprocessing as mp

    import pyoracle as po

    seats, cases = _h2h_cases()
    assert [c[4] for c in cases[:3]] == [0, 3, 7]
    out = str(tmp_path / "h2h8.npy")
    port = 35500 + os.getpid() % 2000
    mp.spawn(_h2h_worker, args=(8, port, out, cases), nprocs=8, join=True)
    want = []
    for target, max_attempts, chunk, start, _ in cases:
        st = np.zeros(5, dtype=np.uint64) if start is None else np.array(start, dtype=np.uint64)
        want.append(po.h2h_block(seats, 7, 3, 1, target, max_attempts, chunk, max_rounds=12, state=st).astype(np.int64))
    assert np.array_equal(np.load(out), np.stack(want)), (np.load(out), want)
    assert (np.stack(want)[:, 2] > 0).any()  # safety-limit games among the attempts


def test_bench_with_eight_self_launched_ranks_configs_3_and_4(tmp_path):
    """`python bench.py --gpus 8` (gloo, oracle-backed stub): config 3 — every rank its own shuffle range of the step; config 4 — every
    k's shuffle range (11 / 23 / 34 / 46 shuffles: none divisible by 8) split over the ranks — one reduced tally, equal to the
    single-process oracle's."""
    import pyoracle as po
    from test_bench_cpu import _run_bench

    from bench import grid5160

    table = grid5160().view(po.STRATEGY_DTYPE)
    out3 = tmp_path / "c3.npy"
    line = _run_bench("--gpus", "8", "--config", "3", "--shuffles", "3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                      "--dump-tally", str(out3), timeout=1200)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["games_per_gpu_per_step"] == 3 * 1290
    # step 0 of eight ranks = shuffles [0, 24) of the single-process run
    assert np.array_equal(np.load(out3), po.tournament(table, 4, 0, 0, 24, n_threads=8)["tally"][0])
    out4 = tmp_path / "c4.npy"
    line = _run_bench("--gpus", "8", "--config", "4", "--games", "30000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                      "--dump-tally", str(out4), timeout=1200)
    n_sh = line["config"]["shuffles_per_k_per_step"]
    assert line["n_gpus"] == 8 and {int(k): v for k, v in n_sh.items()} == {2: 11, 4: 23, 6: 34, 8: 46}
    got = np.load(out4)
    for i, k in enumerate((2, 4, 6, 8)):
        assert np.array_equal(got[i], po.tournament(table, k, 0, 0, n_sh[str(k)], n_threads=8)["tally"][0]), k
