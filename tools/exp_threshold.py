"""Diagnostic: hand-over threshold (lanes of a wave that must be waiting before the wave fetches new games) against player count
on the 5 160-strategy grid."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for

table = table_for(5160)
eng = Engine(0)
for k, n_sh in ((2, 6000), (4, 12000), (6, 18000), (8, 24000)):
    tk = table[:5160 - 5160 % k]
    out = []
    for thr in (4, 6, 8, 10, 12, 16, 8):
        eng.set_option("batch_threshold", thr)
        best = None
        for rep in range(3):
            eng.tournament(tk, k, 0, 0, n_sh)
            t = eng.timing()
            best = t["play_ms"] if best is None else min(best, t["play_ms"])
        out.append(f"{thr}: {best:.3f}")
    print(f"k={k} play ms by threshold  " + "  ".join(out), flush=True)
