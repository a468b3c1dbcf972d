"""world_size-2 gloo test of the multi-GPU path's structure: shard the shuffle range by whole batches, reduce the
int64 tally once, compare with the unsharded run.  The per-rank games come from the CPU oracle here (no GPU);
on the GPU box each rank calls Engine.tournament on its shard instead (bench.py)."""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank: int, world: int, port: int, out_path: str) -> None:
    for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import golden_util as gu
    import pyoracle as po
    from farkle_ii_amd.distributed import reduce_tally, shard_shuffle_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = gu.strategies_from_tuples(gu.load("grid_vectors.json")["g64"], po.STRATEGY_DTYPE)
    lo, hi = shard_shuffle_range(0, 96, rank, world, batch_size=8)
    local = np.zeros((12, 64, 26), dtype=np.int64)
    if hi > lo:
        mine = po.tournament(table, 2, 42, lo, hi, shuffles_per_batch=8)["tally"]
        local[lo // 8: lo // 8 + len(mine)] = mine
    total = reduce_tally(local, dst=0)
    if rank == 0:
        np.save(out_path, total)
    dist.destroy_process_group()


def test_two_rank_tally_reduce_matches_single_process(tmp_path):
    import torch.multiprocessing as mp

    sys.path.insert(0, str(ROOT / "oracle"))
    import golden_util as gu
    import pyoracle as po

    out = str(tmp_path / "total.npy")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    table = gu.strategies_from_tuples(gu.load("grid_vectors.json")["g64"], po.STRATEGY_DTYPE)
    full = po.tournament(table, 2, 42, 0, 96, shuffles_per_batch=8)["tally"]
    assert np.array_equal(np.load(out), full)


def _h2h_worker(rank: int, world: int, port: int, out_path: str) -> None:
    for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import pyoracle as po
    from farkle_ii_amd.distributed import h2h_block_distributed

    dist.init_process_group("gloo", rank=rank, world_size=world)
    seats = np.zeros(2, dtype=po.STRATEGY_DTYPE)
    seats[0] = (300, 2, 1, 1, 1, 1, 0, 1, 1, 1, 0)
    seats[1] = (0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1)  # never banks: every game it does not lose early runs to the round limit
    results = []
    # (target, max_attempts, chunk, start state): cut inside rank 0 / rank 1 / not reached / attempt cap / resumed block
    for target, max_attempts, chunk, start in [(40, 400, 400, None), (150, 400, 200, None), (10_000, 300, 120, None),
                                               (90, 100, 1000, None), (120, 500, 77, (60, 55, 5, 30, 25))]:
        st = np.zeros(5, dtype=np.uint64) if start is None else np.array(start, dtype=np.uint64)
        got = h2h_block_distributed(po.h2h_block, seats, 7, 3, 1, target, max_attempts, chunk, state=st, max_rounds=12)
        results.append(got.astype(np.int64))
    if rank == 0:
        np.save(out_path, np.stack(results))
    dist.destroy_process_group()


def test_two_rank_h2h_prefix_cut_matches_serial_block(tmp_path):
    """contiguous attempt ranges per rank + exclusive scan of completed counts reproduces the serial prefix."""
    import torch.multiprocessing as mp

    sys.path.insert(0, str(ROOT / "oracle"))
    import pyoracle as po

    out = str(tmp_path / "h2h.npy")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_h2h_worker, args=(2, port, out), nprocs=2, join=True)
    seats = np.zeros(2, dtype=po.STRATEGY_DTYPE)
    seats[0] = (300, 2, 1, 1, 1, 1, 0, 1, 1, 1, 0)
    seats[1] = (0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1)
    want = []
    for target, max_attempts, chunk, start in [(40, 400, 400, None), (150, 400, 200, None), (10_000, 300, 120, None),
                                               (90, 100, 1000, None), (120, 500, 77, (60, 55, 5, 30, 25))]:
        st = np.zeros(5, dtype=np.uint64) if start is None else np.array(start, dtype=np.uint64)
        want.append(po.h2h_block(seats, 7, 3, 1, target, max_attempts, chunk, max_rounds=12, state=st).astype(np.int64))
    got = np.load(out)
    assert np.array_equal(got, np.stack(want)), (got, want)
    assert (np.stack(want)[:, 2] > 0).any()  # the cases include safety-limit games (attempted != completed)


def test_tcp_rendezvous_ships_the_communicator_id_to_every_rank():
    """The stdlib rendezvous of the 128-byte RCCL communicator id (no torch.distributed group): rank 0 serves, the other
    ranks connect (also when they arrive first)."""
    import threading
    import time

    from farkle_ii_amd.distributed import tcp_broadcast

    port = 36500 + os.getpid() % 2000
    payload = bytes(range(128))
    got: dict[int, bytes] = {}

    def run(rank: int, delay: float) -> None:
        time.sleep(delay)
        got[rank] = tcp_broadcast(payload if rank == 0 else None, rank, 4, "127.0.0.1", port, timeout=30)

    threads = [threading.Thread(target=run, args=(r, 0.3 if r == 0 else 0.0)) for r in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert got == {r: payload for r in range(4)}


def test_engine_comm_is_skipped_for_engines_without_it_and_single_rank():
    from farkle_ii_amd.distributed import init_engine_comm

    class NoComm:
        pass

    assert init_engine_comm(NoComm()) is False


def test_engine_comm_id_failure_on_rank_0_reaches_every_rank():
    """`fk_comm_unique_id` failing on rank 0 must not leave the other ranks waiting for an id: the error travels with the
    rendezvous and every rank raises it (stdlib TCP rendezvous, two threads standing in for two ranks)."""
    import socket
    import threading

    from farkle_ii_amd.distributed import init_engine_comm

    class Broken:
        def comm_unique_id(self):
            raise RuntimeError("no RCCL here")

        def comm_init(self, *a):  # pragma: no cover - must not be reached
            raise AssertionError("comm_init called without an id")

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    errors = [None, None]

    def run(rank):
        import os

        os.environ["FK_COMM_PORT"] = str(port)
        try:
            init_engine_comm(Broken(), rank=rank, world=2)
        except RuntimeError as exc:
            errors[rank] = str(exc)

    threads = [threading.Thread(target=run, args=(r,)) for r in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    assert all(e and "no RCCL here" in e for e in errors), errors
