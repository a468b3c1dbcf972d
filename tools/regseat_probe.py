"""Diagnostic: register-resident seats (k=2) vs LDS seats; block / resident-wave sweeps."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import grid64
from farkle_ii_amd.backend import Engine

table = grid64()
eng = Engine(0)
n = 312500
eng.tournament(table, 2, 42, 0, n)
def run(label):
    ts = []
    for i in range(3):
        eng.tournament(table, 2, 42, 0, n); ts.append(eng.timing()["play_ms"])
    t = eng.timing()
    print(f"{label:34s} play {min(ts):8.3f} ms  block {t['play_block']} grid {t['play_grid']} lds {t['play_lds_bytes']}", flush=True)
eng.set_option("reg_seats", 0); run("LDS seats (block auto)")
eng.set_option("reg_seats", 1)
for blk in (256, 512, 1024, 128):
    eng.set_option("block", blk); run(f"register seats block {blk}")
eng.set_option("block", 256)
for thr in (4, 6, 8, 12):
    eng.set_option("batch_threshold", thr); run(f"register seats thr {thr}")
