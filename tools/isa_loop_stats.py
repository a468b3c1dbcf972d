"""Static instruction mix of a kernel's roll loop from a --save-temps .s file.
usage: python tools/isa_loop_stats.py <file.s> <kernel symbol substring> [loop header comment regex]
Counts instructions per class between the Depth=2 loop header that holds the roll step and the end of that loop."""
import re, sys
from collections import Counter

def kernel_text(path, sym):
    out, on = [], False
    for line in open(path):
        if not on and line.startswith("_ZN") and sym in line and line.rstrip().endswith(("E", ":", ")")) or (not on and line.startswith("_ZN") and sym in line and ":" in line):
            on = True
        if on:
            out.append(line.rstrip("\n"))
            if "s_endpgm" in line:
                break
    return out

def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = kernel_text(path, sym)
    # the roll loop: the LAST "This Loop Header: Depth=2" whose body holds ds_read2_b64 / ds_read_b128 (the seat record load)
    heads = [i for i, l in enumerate(lines) if "Loop Header: Depth=2" in l]
    best = None
    for h in heads:
        m = re.search(r"Header=(BB\d+_\d+)", " ".join(lines[h - 3:h + 1]))
        # label of this header = nearest label line above
        j = h
        while j > 0 and not lines[j].startswith(".LBB"):
            j -= 1
        label = lines[j].split(":")[0][1:]
        body = [l for l in lines[j:] if True]
        end = j
        for t in range(j + 1, len(lines)):
            if lines[t].startswith(".LBB") or lines[t].lstrip().startswith(";"):
                if f"Header={label[1:] if label.startswith('L') else label}" in lines[t] or f"Header={label.replace('LBB','BB')}" in lines[t]:
                    end = t
        seg = lines[j:end + 200]
        if any("ds_read" in l for l in lines[j:j + 40]):
            best = (j, end)
    if best is None:
        print("roll loop not found"); return
    j, end = best
    # extend to the end of the last block that belongs to the loop
    t = end + 1
    while t < len(lines) and not lines[t].startswith(".LBB"):
        t += 1
    seg = [l.strip() for l in lines[j:t] if l.startswith("\t") and not l.strip().startswith(";")]
    ops = Counter(l.split()[0] for l in seg)
    valu = sum(n for o, n in ops.items() if o.startswith("v_"))
    print(f"loop lines {j}..{t}: {len(seg)} instructions, VALU {valu}, v_mov {sum(n for o, n in ops.items() if o.startswith('v_mov'))}, "
          f"SALU {sum(n for o, n in ops.items() if o.startswith('s_') and not o.startswith(('s_nop', 's_cbranch', 's_branch', 's_waitcnt')))}, "
          f"s_nop {ops.get('s_nop', 0)}, branches {sum(n for o, n in ops.items() if o.startswith(('s_cbranch', 's_branch')))}, "
          f"mad64 {ops.get('v_mad_u64_u32', 0)}, mul_lo {ops.get('v_mul_lo_u32', 0)}, cndmask {sum(n for o,n in ops.items() if o.startswith('v_cndmask'))}, "
          f"LDS {sum(n for o, n in ops.items() if o.startswith('ds_'))}, VMEM {sum(n for o, n in ops.items() if o.startswith(('global_', 'buffer_', 'flat_')))}, "
          f"readlane/writelane {sum(n for o,n in ops.items() if o.startswith(('v_readlane','v_writelane')))}")

main()
