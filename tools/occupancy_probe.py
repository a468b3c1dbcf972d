"""Diagnostic: play-kernel time vs resident waves per SIMD (is the kernel latency- or issue-bound?)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import grid64
from farkle_ii_amd.backend import Engine

table = grid64()
eng = Engine(0)
n = 312500
eng.tournament(table, 2, 42, 0, n)
for blk, per_cu in ((1024, 1), (512, 2), (512, 1), (256, 4), (256, 3), (256, 2), (256, 1)):
    eng.set_option("block", blk); eng.set_option("blocks_per_cu", per_cu)
    ts = []
    for i in range(3):
        eng.tournament(table, 2, 42, 0, n); ts.append(eng.timing()["play_ms"])
    t = eng.timing()
    print(f"block {blk:5d} x {per_cu}/CU = {blk*per_cu//256:2d} waves/SIMD-ish ({blk*per_cu//64} waves/CU): play {min(ts):8.3f} ms grid {t['play_grid']}", flush=True)
