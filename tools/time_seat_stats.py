import sys, time
sys.path.insert(0, "/root/repo")
import os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for
import numpy as np
eng = Engine(0)
for grid, k, n_sh in ((5160, 4, 16000), (64, 2, 312500)):
    table = table_for(grid)
    eng.tournament(table, k, 0, 0, n_sh, want_seat_stats=True)
    t0 = time.perf_counter(); r = eng.tournament(table, k, 0, 0, n_sh, want_seat_stats=True); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); r2 = eng.tournament(table, k, 0, 0, n_sh); dt2 = time.perf_counter() - t1
    print(grid, k, n_sh, f"with stats {dt*1e3:.2f} ms, without {dt2*1e3:.2f} ms, stats cost {1e3*(dt-dt2):.2f} ms; checksum {int(r['seat_stats'].sum())}")
