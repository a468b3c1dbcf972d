"""Row-shard writer process of rows mode: ``python -m farkle_ii_amd.shard_writer``.

One parquet file per shuffle of (for k = 2, 64 strategies) 32 games is the reference's row-shard format
(``run_tournament.py:530-558``).  Such a file costs ~0.75 ms of Arrow encoding (46 column chunks, footer) and ~0.6 ms of file
creation on one core, most of it under the GIL: writer THREADS do not scale (sixteen of them took 5.8 ms of thread time per
shard in round 2).  The runner therefore keeps a few of these processes for the length of a run and feeds them runs of
shuffles over pipes (length-prefixed pickles on stdin / stdout): fresh interpreters that never touch the GPU, started as
ordinary child processes.  The work itself is ``tournament._write_shard_group``.
"""
from __future__ import annotations

import pickle
import struct
import subprocess
import sys
import threading
from typing import Sequence


def _serve() -> None:
    from .tournament import _shard_worker_init, _write_shard_group

    inp, out = sys.stdin.buffer, sys.stdout.buffer
    _shard_worker_init()
    while True:
        head = inp.read(8)
        if len(head) < 8:
            return
        job = pickle.loads(inp.read(struct.unpack("<Q", head)[0]))
        try:
            res = ("ok", _write_shard_group(*job))
        except Exception as exc:  # reported to the parent, which raises
            res = ("err", f"{type(exc).__name__}: {exc}")
        blob = pickle.dumps(res, protocol=pickle.HIGHEST_PROTOCOL)
        out.write(struct.pack("<Q", len(blob)) + blob)
        out.flush()


class ShardWriters:
    """``n`` writer processes; ``map(jobs)`` returns the results in job order."""

    def __init__(self, n: int):
        import os
        from pathlib import Path

        root = str(Path(__file__).resolve().parent.parent)  # the children import this package whatever their working directory
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([root] + [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p]))
        self.procs = [subprocess.Popen([sys.executable, "-m", "farkle_ii_amd.shard_writer"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
                      for _ in range(max(1, n))]

    def map(self, jobs: Sequence[tuple]) -> list:
        results: list = [None] * len(jobs)
        errors: list[str] = []
        cursor = iter(range(len(jobs)))
        lock = threading.Lock()

        def feed(proc) -> None:
            while not errors:
                with lock:
                    i = next(cursor, None)
                if i is None:
                    return
                blob = pickle.dumps(jobs[i], protocol=pickle.HIGHEST_PROTOCOL)
                try:
                    proc.stdin.write(struct.pack("<Q", len(blob)) + blob)
                    proc.stdin.flush()
                    head = proc.stdout.read(8)
                    if len(head) < 8:
                        raise ConnectionError("row-shard writer process ended")
                    status, payload = pickle.loads(proc.stdout.read(struct.unpack("<Q", head)[0]))
                except Exception as exc:
                    errors.append(f"{type(exc).__name__}: {exc}")
                    return
                if status != "ok":
                    errors.append(str(payload))
                    return
                results[i] = payload

        threads = [threading.Thread(target=feed, args=(p,), daemon=True) for p in self.procs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise RuntimeError(f"row-shard writer failed: {errors[0]}")
        return results

    def close(self) -> None:
        for p in self.procs:
            try:
                p.stdin.close()
            except OSError:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        self.procs = []


if __name__ == "__main__":
    _serve()
