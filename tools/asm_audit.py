"""Audit of a -save-temps .s: no compiler instruction may touch a VGPR that an inline-asm ds_read_b128 has in flight
(the compiler counts the asm's outputs as written at ASMEND).  A batch of asm reads is retired by the SECOND asm
`s_waitcnt lgkmcnt` after it (the first one only covers the batch before), or by any lgkmcnt(0)."""
import re, sys

def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()

pending = []  # list of [regset, waits_seen]
in_asm = False
bad = 0
for ln, line in enumerate(open(sys.argv[1]), 1):
    t = line.strip()
    if t.startswith(";;#ASMSTART"):
        in_asm = True; continue
    if t.startswith(";;#ASMEND"):
        in_asm = False; continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    toks = re.split(r"[ ,]+", t)
    if in_asm and toks[0] == "ds_read_b128":
        r = regs(toks[1])
        if pending and pending[-1][1] == 0 and pending[-1][2] == ln - 1:
            pending[-1][0] |= r; pending[-1][2] = ln
        else:
            pending.append([set(r), 0, ln])
        continue
    if toks[0] == "s_waitcnt" and "lgkmcnt" in t:
        n = int(re.search(r"lgkmcnt\((\d+)\)", t).group(1))
        if n == 0:
            pending = []
        else:
            for p in pending:
                p[1] += 1
            pending = [p for p in pending if p[1] < 2]
        continue
    used = set()
    for tok in toks[1:]:
        used |= regs(tok)
    for p in pending:
        if used & p[0]:
            # the consuming MFMAs legitimately read a batch after ONE wait (the batch before the newest)
            if p[1] >= 1:
                continue
            bad += 1
            print(f"line {ln}: {t}   <- touches in-flight asm read registers {sorted(used & p[0])[:4]}")
print("violations:", bad)
