"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel instance whose name matches the filter."""
import re, sys
log, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
t = open(log).read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split(" ")[0].split("\n")[0]
    if pat and not re.search(pat, name):
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    sc, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print("%-75s VGPR %4s AGPR %3s SGPR %4s scratch %4s occ %s" % (name[20:95], g("VGPRs"), g("AGPRs"), g("SGPRs"), sc, occ))
