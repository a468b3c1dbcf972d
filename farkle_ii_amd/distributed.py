"""Multi-GPU partitioning of the (seed x shuffle x game) space and the single tally reduction.

One process per GPU (``torch.distributed``; backend ``nccl`` is RCCL on ROCm, ``gloo`` in CPU tests).  Shuffles are
independent (``docs/rng_contract.md:3-8``), so ranks take contiguous WHOLE deterministic batches and the only
exchange is one integer SUM of the ``[n_batches][S][26]`` tally — the analogue of ``OutcomeCounter.absorb`` +
``_reduce_metric_chunk_payloads`` (``run_tournament.py:197-213, 1023-1042``).  Integer sums are order independent,
so the result is bit-identical to a single-process run.
"""
from __future__ import annotations

import numpy as np


def shard_shuffle_range(shuffle_begin: int, shuffle_end: int, rank: int, world_size: int, batch_size: int = 1) -> tuple[int, int]:
    """Contiguous range of whole batches for ``rank``; batch b = shuffles [b*batch_size, (b+1)*batch_size)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("rank must be in [0, world_size)")
    if batch_size < 1 or shuffle_end < shuffle_begin:
        raise ValueError("bad shuffle range / batch size")
    if shuffle_begin % batch_size:
        raise ValueError("shuffle_begin must be aligned to the deterministic batch size")
    n_batches = -(-(shuffle_end - shuffle_begin) // batch_size)
    lo = (n_batches * rank) // world_size
    hi = (n_batches * (rank + 1)) // world_size
    return (min(shuffle_begin + lo * batch_size, shuffle_end), min(shuffle_begin + hi * batch_size, shuffle_end))


def collective_device(device=None):
    """Where tensors of a collective must live: the rank's GPU under ``nccl`` (RCCL rejects CPU tensors), the CPU under
    ``gloo``.  An explicit ``device`` wins."""
    import torch
    import torch.distributed as dist

    if device is not None:
        return torch.device(device)
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


_ENGINE_COMM = None  # the engine whose RCCL communicator (fk_comm_init) carries the tally reduction


def tcp_broadcast(payload: bytes | None, rank: int, world: int, addr: str, port: int, timeout: float = 300.0) -> bytes:
    """Rank 0's bytes to every rank over plain TCP (stdlib only): the rendezvous of the RCCL communicator id when no
    torch.distributed process group exists."""
    import socket
    import struct
    import time

    if world == 1:
        return payload or b""
    if rank == 0:
        assert payload is not None
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(struct.pack("<I", len(payload)) + payload)
        return payload
    deadline = time.monotonic() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                head = b""
                while len(head) < 4:
                    piece = conn.recv(4 - len(head))
                    if not piece:  # the peer closed before the header was complete: retry the rendezvous, do not spin
                        raise ConnectionError("rendezvous closed before the header")
                    head += piece
                n, = struct.unpack("<I", head)
                data = b""
                while len(data) < n:
                    chunk = conn.recv(n - len(data))
                    if not chunk:
                        raise ConnectionError("rendezvous closed early")
                    data += chunk
                return data
        except (ConnectionError, OSError):
            if time.monotonic() > deadline:
                raise
            time.sleep(0.05)


def init_engine_comm(engine, rank: int | None = None, world: int | None = None) -> bool:
    """Give ``engine`` an RCCL communicator over all ranks (``fk_comm_init``) so that :func:`reduce_tally` runs as one
    ``ncclReduce`` on the engine's stream, through the C-ABI, with no PyTorch in the data path.  The 128-byte
    communicator id travels over the torch.distributed process group when one exists (launched by torchrun), otherwise
    over a stdlib TCP rendezvous on ``MASTER_ADDR``:``FK_COMM_PORT`` (default ``MASTER_PORT`` + 17).  Collective: every
    rank calls it.  Returns False (and leaves the torch path in place) when the engine has no communicator support."""
    import os

    global _ENGINE_COMM
    if not hasattr(engine, "comm_init"):
        return False
    try:
        import torch.distributed as dist

        group = dist.is_available() and dist.is_initialized()
    except ImportError:
        group = False
    if group:
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    if world == 1:
        return False
    # rank 0 makes the id; a failure there travels with the broadcast, so that every rank raises it instead of waiting for an id
    comm_id = error = None
    if rank == 0:
        try:
            comm_id = engine.comm_unique_id()
        except Exception as exc:  # noqa: BLE001 - re-raised on every rank below
            error = f"{type(exc).__name__}: {exc}"
    if group:
        comm_id, error = gather_objects((comm_id, error), broadcast_from=0)
    else:
        port = int(os.environ.get("FK_COMM_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 17))
        import pickle

        blob = tcp_broadcast(pickle.dumps((comm_id, error)) if rank == 0 else None, rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), port)
        comm_id, error = pickle.loads(blob)
    if error is not None:
        raise RuntimeError(f"fk_comm_unique_id failed on rank 0: {error}")
    engine.comm_init(comm_id, rank, world)
    _ENGINE_COMM = engine
    return True


def reduce_tally(tally: np.ndarray, dst: int = 0, device=None) -> np.ndarray:
    """SUM-reduce an int64 tally over all ranks; returns the total on ``dst`` (own tally elsewhere).  With an engine
    communicator (:func:`init_engine_comm`) this is ``fk_reduce_tally`` — RCCL through the C-ABI; otherwise the default
    torch.distributed process group (``nccl`` = RCCL on GPU tensors, ``gloo`` on the CPU in tests)."""
    if _ENGINE_COMM is not None and getattr(_ENGINE_COMM, "comm_world", 1) > 1:
        return _ENGINE_COMM.reduce_tally(tally, dst)
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tally
    t = torch.from_numpy(np.ascontiguousarray(tally, dtype=np.int64)).to(collective_device(device))
    dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_objects(obj, dst: int | None = None, broadcast_from: int | None = None):
    """Small picklable objects across ranks (manifest records, recovered batch sets).  ``dst``: list of every rank's
    object on ``dst`` (None elsewhere); ``broadcast_from``: that rank's object on every rank.  Single process: identity."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj] if dst is not None else obj
    if dist.get_backend() == "nccl":
        import torch

        torch.cuda.set_device(collective_device())
    if broadcast_from is not None:
        box = [obj]
        dist.broadcast_object_list(box, src=broadcast_from)
        return box[0]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(obj, out, dst=dst)
    return out


def agree(ok: bool) -> bool:
    """True iff ``ok`` is true on EVERY rank (one MIN all-reduce on the process group): how optional fast paths are switched
    on without ever leaving ranks on different paths.  Single process: ``ok``."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=collective_device())
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


def barrier() -> None:
    """Process-group barrier (no-op without one)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def h2h_block_distributed(h2h, seats, root_seed: int, pair_id: int, order: int, target: int, max_attempts: int,
                          chunk_games: int, *, state, device=None, **limits) -> np.ndarray:
    """One H2H block advanced by all ranks together; the result equals the single-process ``h2h(...)`` call.

    ``h2h`` is ``Engine.h2h`` (same signature).  The stop rule of a block is a PREFIX (the first ``target`` completed
    games in attempt order, h2h_schedule.py:1165-1235), while attempts themselves are independent, so:
      1. the chunk's attempt range ``[attempted, min(max_attempts, attempted + chunk_games))`` is cut into one
         contiguous sub-range per rank and every rank plays its sub-range to the end (no stop rule);
      2. one all-gather of the five per-range counts; every rank takes the exclusive scan of the completed counts and
         finds the first sub-range in which the running total reaches ``target``;
      3. only that rank replays its sub-range with the remaining target, which stops it at the exact attempt the serial
         loop would have stopped at; a second all-gather publishes the cut.
    Sub-ranges after the cut are discarded (their attempts were never consumed by the serial loop)."""
    import torch
    import torch.distributed as dist

    state = np.ascontiguousarray(state, dtype=np.uint64).copy()
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not distributed:
        return h2h(seats, root_seed, pair_id, order, target, max_attempts, chunk_games, state=state, **limits)
    rank, world = dist.get_rank(), dist.get_world_size()
    attempted, completed = int(state[0]), int(state[1])
    end = min(int(max_attempts), attempted + int(chunk_games))
    if completed >= target or end <= attempted:
        return state
    span = end - attempted
    lo = attempted + (span * rank) // world
    hi = attempted + (span * (rank + 1)) // world

    def play(first: int, last: int, remaining: int) -> np.ndarray:
        """Counts of attempts [first, last) stopped after `remaining` completed games: attempted, completed, safety, w1, w2."""
        if last <= first or remaining <= 0:
            return np.zeros(5, dtype=np.int64)
        # a consistent block state whose next attempt index is `first` (attempted = completed + safety, wins = completed)
        base = np.array([first, first, 0, first, 0], dtype=np.uint64)
        out = h2h(seats, root_seed, pair_id, order, first + remaining, last, last - first, state=base, **limits)
        return out.astype(np.int64) - base.astype(np.int64)

    def gather(vec: np.ndarray) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.int64)).to(collective_device(device))
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        return np.stack([p.cpu().numpy() for p in parts])

    need = target - completed
    full = gather(play(lo, hi, 1 << 62))  # [world][5]
    before = np.concatenate([[0], np.cumsum(full[:, 1])[:-1]])  # exclusive scan of completed games
    cut = next((r for r in range(world) if before[r] + full[r, 1] >= need), None)
    if cut is None:  # target not reached inside this chunk: every sub-range counts in full
        total = full.sum(axis=0)
    else:
        mine = play(lo, hi, int(need - before[cut])) if rank == cut else np.zeros(5, dtype=np.int64)
        partial = gather(mine)[cut]
        total = full[:cut].sum(axis=0) + partial
    return (state.astype(np.int64) + total).astype(np.uint64)
