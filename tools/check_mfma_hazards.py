"""Static check of an hipcc -S listing for the hazards an inline-asm MFMA kernel has to avoid by construction (hipcc does not know that the asm
statements are MFMAs): a non-MFMA instruction that READS a register an MFMA wrote, fewer than MIN_GAP MFMA issues (each >= 32 cycles for
32x32x16) after it, reads stale data; a VALU / accvgpr write to a register that an MFMA reads as A / B / C directly behind it (< 2 wait states).
Usage: python tools/check_mfma_hazards.py listing.s [kernel-name-substring]"""
import re
import sys

MIN_GAP = 2


def regs(tok):
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.startswith(".") or line.endswith(":"):
        return None
    parts = line.replace(",", " ").split()
    op, toks = parts[0], parts[1:]
    return op, [regs(t) for t in toks]


def main():
    text = open(sys.argv[1]).read().split("\n")
    sel = sys.argv[2] if len(sys.argv) > 2 else None
    inside = sel is None
    pending = []  # (dst regs, mfma count at issue, line no, instruction index at issue)
    nm = 0
    last_write = {}  # reg -> (instr index, line no) of the last non-MFMA VALU write
    idx = 0
    bad = 0
    seen_kernel = sel is None
    for ln, raw in enumerate(text, 1):
        m = re.match(r"^(_Z\w+):", raw)
        if sel is not None and m:
            inside = sel == m.group(1) or sel in m.group(1)
            pending, nm = [], 0
            if inside:
                seen_kernel = True
        if not inside:
            continue
        p = parse(raw)
        if p is None:
            continue
        op, ops = p
        if op in ("s_endpgm", "s_branch", "s_setpc_b64"):  # the next instruction in the listing is not reached from here
            pending = []
            last_write = {}
            continue
        if op.startswith("s_nop"):
            m = re.search(r"s_nop\s+(\d+)", raw)
            idx += int(m.group(1)) + 1
            continue
        idx += 1
        if op.startswith("v_mfma"):
            dst, srcs = ops[0], set().union(*ops[1:]) if len(ops) > 1 else set()
            for r in srcs:
                if r in last_write and idx - last_write[r][0] < 3:
                    print(f"line {ln}: MFMA reads {r[0]}{r[1]} written {idx - last_write[r][0] - 1} wait states earlier (line {last_write[r][1]})")
                    bad += 1
            nm += 1
            pending = [(d, c, l, i0) for (d, c, l, i0) in pending if nm - c <= MIN_GAP + 1]
            pending.append((dst, nm, ln, idx))
            continue
        if op.startswith("s_") or op.startswith("buffer_") or op.startswith("global_store") or op.startswith("scratch_store"):
            continue
        # reads: every operand but the first (stores / ds_write excluded above for simplicity)
        rd = set().union(*ops[1:]) if len(ops) > 1 else set()
        wr = ops[0] if ops else set()
        for d, c, l, i0 in pending:
            if nm - c < MIN_GAP and idx - i0 < 20 and (rd & d):  # (20 wait states settle any MFMA)
                print(f"line {ln}: {op} reads {sorted(rd & d)[:2]} {nm - c} MFMA(s) after the MFMA at line {l} that writes it")
                bad += 1
        if op.startswith("v_"):
            for r in wr:
                last_write[r] = (idx, ln)
            # a register rewritten by this instruction no longer holds the MFMA's (late) result as far as later READERS are concerned
            pending = [(d - wr, c, l, i0) for (d, c, l, i0) in pending]
    if not seen_kernel:
        print("kernel not found:", sel)
        sys.exit(2)
    print("hazards:", bad)


main()
