"""Diagnostic: summary of a `FK_RUN_TRACE=1 FK_SHARD_WRITER_TIMING=1 python tools/time_farkle_run.py ...` stderr log — the shard writer's
thread time per phase, the launching thread's holes, the shard thread's idle / waiting / writing time.  usage: python tools/trace_summary.py LOG"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
L = [l for l in src if "fk trace" in l]
W = [l for l in src if "fk shards" in l]
rows = []
for l in W:
    m = re.search(r"(\d+) shards, (\d+) threads: build ([\d.]+) ms, sha256 ([\d.]+) ms, write ([\d.]+) ms.*wall ([\d.]+) ms, began at ([\d.]+)", l)
    rows.append(tuple(float(x) for x in m.groups()))
h = len(rows) // 2
for half, name in ((rows[:h], "v2"), (rows[h:], "v3")):
    tot_wall = sum(r[5] for r in half)
    thr = sum(r[2] + r[3] + r[4] for r in half)
    print(name, "calls", len(half), "sum wall %.1f ms" % tot_wall, "thread time %.1f ms" % thr, "eff %.2f" % (thr / (16 * tot_wall)),
          "build %.0f sha %.0f write %.0f" % tuple(sum(r[i] for r in half) for i in (2, 3, 4)))
runs, cur = [], []
for l in L:
    m = re.match(r"\[fk trace\]\s+([\d.]+) ms\s+(\S+)\s+(.*)", l)
    t, th, lab = float(m.group(1)), m.group(2), m.group(3).strip()
    if t == 0.0 and cur: runs.append(cur); cur = []
    cur.append((t, th, lab))
runs.append(cur)
for r in runs[1:]:
    print("==== run, ends", r[-1][0])
    main = [e for e in r if e[1] == "MainThread"]
    for a, b in zip(main, main[1:]):
        if b[0] - a[0] > 12 and not ("engine call" in a[2]): print("  main hole %.1f ms: %s -> %s" % (b[0] - a[0], a[2], b[2]))
    calls = [(b[0] - a[0]) for a, b in zip(main, main[1:]) if "engine call" in a[2]]
    print("  engine calls total %.1f; per call" % sum(calls), [round(c, 1) for c in calls])
    sh = [e for e in r if e[1].startswith("fk-shards")]
    idle = [(b[0] - a[0], a[0]) for a, b in zip(sh, sh[1:]) if a[2] == "shard job ends"]
    wait = sum(b[0] - a[0] for a, b in zip(sh, sh[1:]) if a[2] == "shard job begins")
    busy = [b[0] - a[0] for a, b in zip(sh, sh[1:]) if a[2] == "shard job: images here"]
    print("  shard thread: first job at %.1f; idle %.1f, waiting for images %.1f, writing %.1f; last end %.1f" % (sh[0][0], sum(x[0] for x in idle), wait, sum(busy), sh[-1][0]))
    print("  writes:", [round(x, 1) for x in busy])
    print("  idles:", [round(x[0], 1) for x in idle])
    print("  first events:", [(round(e[0], 1), e[2]) for e in r[:6]])
    print("  last events:", [(round(e[0], 1), e[1][:6], e[2]) for e in r[-5:]])
