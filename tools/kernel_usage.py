"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel (registers, spills, scratch, occupancy)."""
import re
import sys

txt = open(sys.argv[1]).read()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]

    def f(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    nm = re.sub(r"_ZN\d+[a-z]+\d+_GLOBAL__N_1\d+", "", name)
    print("%-46s sgpr %3s vgpr %3s agpr %3s scratch %4s spillS %3s spillV %3s occ %s" % (
        nm[:46], f("TotalSGPRs"), f("VGPRs"), f("AGPRs"), f(r"ScratchSize \[bytes/lane\]"), f("SGPRs Spill"), f("VGPRs Spill"), f(r"Occupancy \[waves/SIMD\]")))
