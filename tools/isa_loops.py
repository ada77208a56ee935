"""Per-loop instruction mix of one kernel's gfx950 assembly (hipcc -S --cuda-device-only output, one kernel cut out with awk):
python tools/isa_loops.py kernel.s  ->  for every backward branch: MFMA / AccVGPR moves / exp / LDS reads / VALU / scratch counts."""
import re
import sys

L = open(sys.argv[1]).read().split("\n")
labels = {}
for i, l in enumerate(L):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
VALU = re.compile(r"^\s+v_")
for i, l in enumerate(L):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        body = L[a:i]

        def c(p):
            return sum(1 for x in body if re.search(p, x))

        print(f"loop {m.group(1)} lines {a}-{i}: mfma={c('v_mfma')} acc_read={c('v_accvgpr_read')} acc_write={c('v_accvgpr_write')} "
              f"exp={c('v_exp_f32')} ds_read={c('ds_read')} branches={c('s_cbranch')} valu={sum(1 for x in body if VALU.search(x))} "
              f"scratch={c('scratch_')} s_nop={c('s_nop')}")
