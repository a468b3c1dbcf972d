"""Diagnostic: row-shard writer processes — shards per second against the number of writers, with and without the
tmp + rename per shard (all writers create their files in ONE directory).
usage: python tools/time_shard_pool.py [n_shuffles=25600]"""
import shutil, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd import tournament as rt
from farkle_ii_amd.backend import row_dtype

if __name__ == "__main__":
    k, gps = 2, 32
    n_sh = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
    rows = np.zeros(n_sh * gps, dtype=row_dtype(k))
    idx = np.arange(n_sh, dtype=np.int64)
    tasks = rt.ShuffleRange(1, k, idx, idx, idx // 512)
    seeds = np.zeros((n_sh, gps), dtype=np.uint32)
    for workers in (8, 16, 24, 32):
        for atomic in (True, False):
            for group in (64, 256):
                d = Path(tempfile.mkdtemp(prefix="fk_shards_"))
                rt.write_row_shards(d, rt.ShuffleRange(1, k, idx[:workers * group], idx[:workers * group], idx[:workers * group]),
                                    rows[:workers * group * gps], np.arange(64), threads=workers, group=group, game_seeds=seeds[:workers * group], as_lines=True, atomic=atomic)  # start the pool
                t0 = time.perf_counter()
                rt.write_row_shards(d, tasks, rows, np.arange(64), threads=workers, group=group, game_seeds=seeds, as_lines=True, atomic=atomic)
                dt = time.perf_counter() - t0
                print(f"writers {workers:2d} atomic {int(atomic)} group {group:3d}: {dt:6.3f} s = {dt / n_sh * 1e3:6.4f} ms of wall per shard, {dt * workers / n_sh * 1e3:6.3f} ms of writer time", flush=True)
                shutil.rmtree(d, ignore_errors=True)
