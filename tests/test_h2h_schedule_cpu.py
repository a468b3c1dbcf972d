"""Host logic of the schedule-level H2H runner (farkle_ii_amd/h2h.py: run_blocks, PrefetchingBlockRunner) on the
oracle-backed stub engine (CPU): equality with the per-block runner, chunking, resumed blocks, two gloo ranks."""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
import golden_util as gu
from h2h_schedule_util import make_schedule, manifest_frame, serial_schedule_loop


def _table():
    from farkle_ii_amd.strategies import STRATEGY_DTYPE

    t = gu.strategies_from_tuples(gu.load("grid_vectors.json")["g64"], STRATEGY_DTYPE)
    t["strategy_id"] = np.arange(len(t))
    return t


def _per_block_runner(engine):
    from farkle_ii_amd.h2h import simulate_block, strategy_from_manifest

    def runner(block, manifest, attempt_count):
        return simulate_block(block, strategy_from_manifest(block["seat1_strategy"], manifest),
                              strategy_from_manifest(block["seat2_strategy"], manifest), attempt_count, None, engine)

    return runner


def test_run_blocks_equals_the_per_block_runner_on_the_stub():
    from oracle_engine_stub import Engine

    from farkle_ii_amd.h2h import run_blocks

    eng, table = Engine(), _table()
    manifest = manifest_frame(table)
    blocks = make_schedule(table, 40, seed=1)
    per_block = _per_block_runner(eng)
    for chunk in (None, 7):
        got = run_blocks(blocks, manifest, chunk, engine=eng)
        for b, g in zip(blocks, got):
            want = per_block(b, manifest, int(b["max_attempts"]) if chunk is None else chunk)
            assert g == want
    # resumed blocks: progress recorded in the dict is where the chunk starts; terminal blocks come back unchanged
    half = run_blocks(blocks, manifest, 5, engine=eng)
    rest = run_blocks(half, manifest, None, engine=eng)
    assert rest == run_blocks(blocks, manifest, None, engine=eng)
    assert run_blocks(rest, manifest, 9, engine=eng) == rest


@pytest.mark.parametrize("chunk", [4, 1000])
def test_prefetching_runner_serves_the_serial_loop(chunk):
    from oracle_engine_stub import Engine

    from farkle_ii_amd.h2h import PrefetchingBlockRunner

    eng, table = Engine(), _table()
    manifest = manifest_frame(table)
    blocks = make_schedule(table, 60, seed=2)
    want, calls = serial_schedule_loop(blocks, _per_block_runner(eng), manifest, chunk)
    runner = PrefetchingBlockRunner(blocks, engine=eng)
    got, calls2 = serial_schedule_loop(blocks, runner, manifest, chunk)
    assert got == want and calls == calls2
    assert runner.single_block_calls == 0 and runner.generations < calls  # every call served from a shared launch group
    # a block the runner was not built with, and one asked for at a progress it never returned: played alone, same result
    stranger = make_schedule(table, 64, seed=3)[-1]
    assert runner(stranger, manifest, 5) == _per_block_runner(eng)(stranger, manifest, 5)
    assert runner.single_block_calls == 1


def _rank_worker(rank: int, world: int, port: int, out_path: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle_engine_stub import Engine

    from farkle_ii_amd.h2h import run_blocks

    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = _table()
    blocks = make_schedule(table, 30, seed=4)
    got = run_blocks(blocks, manifest_frame(table), None, engine=Engine(), rank=rank, world=world)
    if rank == 1:  # every rank holds every result
        import json

        Path(out_path).write_text(json.dumps(got))
    dist.barrier()
    dist.destroy_process_group()


def test_run_blocks_two_gloo_ranks_equal_one_process(tmp_path):
    import json

    import torch.multiprocessing as mp
    from oracle_engine_stub import Engine

    from farkle_ii_amd.h2h import run_blocks

    out = tmp_path / "blocks.json"
    mp.spawn(_rank_worker, args=(2, 34100 + os.getpid() % 2000, str(out)), nprocs=2, join=True)
    table = _table()
    want = run_blocks(make_schedule(table, 30, seed=4), manifest_frame(table), None, engine=Engine())
    assert json.loads(out.read_text()) == json.loads(json.dumps(want))
