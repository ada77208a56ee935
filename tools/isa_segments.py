"""Instruction mix between consecutive s_barriers of one kernel's gfx950 assembly (see tools/isa_loops.py for how to cut a kernel out)."""
import re
import sys

L = open(sys.argv[1]).read().split("\n")
bars = [i for i, l in enumerate(L) if "s_barrier" in l]
print(len(L), "lines;", len(bars), "barriers")
VALU, SALU = re.compile(r"^\s+v_"), re.compile(r"^\s+s_")
prev = 0
for b in bars + [len(L)]:
    seg = L[prev:b]

    def c(p):
        return sum(1 for x in seg if re.search(p, x))

    print(f"lines {prev}-{b}: mfma={c('v_mfma')} exp={c('v_exp')} valu={sum(1 for x in seg if VALU.search(x))} salu={sum(1 for x in seg if SALU.search(x))} "
          f"ds={c('ds_read')} glds={c('global_load_lds')} scr_ld={c('scratch_load')} scr_st={c('scratch_store')} cndmask={c('v_cndmask')} "
          f"branch={c('s_cbranch|s_branch|s_setpc')} waitcnt={c('s_waitcnt')} nop={c('s_nop')}")
    prev = b
