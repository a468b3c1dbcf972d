"""Diagnostic (round 3): rows mode against counts-only on BASELINE config 2 (10^7 k=2 games per call): wall time of the whole
Engine.tournament call (kernels + PCIe + host copy-out), rows into a fresh pageable array, into a reused pageable array and
into a page-locked buffer (fk_host_alloc), for several rows-mode chunk sizes.
usage: python tools/time_rows.py [shuffles=312500]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from bench import grid64
from farkle_ii_amd.backend import Engine, row_dtype

n_sh = int(sys.argv[1]) if len(sys.argv) > 1 else 312500
table = grid64()
eng = Engine(0)
games = n_sh * 32


def timed(label, reps=4, **kw):
    best, t = None, None
    for rep in range(reps):
        t0 = time.perf_counter()
        r = eng.tournament(table, 2, 42, rep * n_sh, (rep + 1) * n_sh, **kw)
        dt = time.perf_counter() - t0
        tt = eng.timing()
        if best is None or dt < best:
            best, t = dt, tt
    print(f"{label:44s} wall {best * 1e3:8.2f} ms  ({games / best / 1e6:7.1f} M games/s)  device {t['total_ms']:7.2f} play {t['play_ms']:7.2f} "
          f"launches {t['play_launches']}", flush=True)
    return r


base = timed("counts only")
pageable = np.zeros(games, dtype=row_dtype(2))
pinned = eng.pinned_empty(games, row_dtype(2))
for chunk in (1 << 30, 8_000_000, 4_000_000, 2_000_000, 1_000_000):
    eng.set_option("rows_chunk_games", chunk)
    timed(f"rows, fresh pageable array, chunk {chunk}", want_rows=True)
    timed(f"rows, reused pageable array, chunk {chunk}", want_rows=True, rows_out=pageable)
    r = timed(f"rows, page-locked buffer, chunk {chunk}", want_rows=True, rows_out=pinned)
assert int(r["rows"]["status"].max()) <= 1
