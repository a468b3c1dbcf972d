"""Diagnostic: play-kernel time vs tunables (batch_threshold, block, LDS tally)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from bench import grid64
from farkle_ii_amd.backend import Engine

table = grid64()
eng = Engine(0)
n = 312500
eng.tournament(table, 2, 42, 0, n)
def run(label):
    ts = []
    for i in range(3):
        eng.tournament(table, 2, 42, 0, n)
        ts.append(eng.timing()["play_ms"])
    t = eng.timing()
    print(f"{label:40s} play {min(ts):8.3f} ms  block {t['play_block']} grid {t['play_grid']} lds {t['play_lds_bytes']}", flush=True)
for thr in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
    eng.set_option("batch_threshold", thr); run(f"threshold {thr}")
eng.set_option("batch_threshold", 8)
for blk in (1024, 512, 256):
    eng.set_option("block", blk); run(f"block {blk} lds-tally")
eng.set_option("use_lds_tally", 0)
for blk in (1024, 512, 256):
    eng.set_option("block", blk); run(f"block {blk} global-tally")
