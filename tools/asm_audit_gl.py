"""Audit of a hipcc .s (-S --cuda-device-only) for in-flight inline-asm global loads: a `global_load_dwordx4 v[a:b], ..` issued inside an
asm statement writes its destination registers when the data arrives, not at the statement, and hipcc does not know (cdna_hip_programming
5.7 item 1).  Between such a load and the `; GL16_USE v[a:b]` marker of its first use (gemm256.hip touch(), placed behind the counted
s_waitcnt that retires the load) NO instruction may name any of those registers: a compiler copy, spill or re-use there reads garbage or
gets overwritten when the data lands.  Usage: asm_audit_gl.py file.s  -> prints violations, exit status 1 if any."""
import re, sys

def regs(tok):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"(?<![\w\[])v(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out

def audit(path):
    pending = []  # [regset, line]
    in_asm = False
    bad = 0
    kernel = None
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if re.match(r"^_Z\w+:", t):
            kernel = t.split(":")[0]; pending = []
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        if in_asm and t.startswith("; GL16_USE"):
            r = regs(t)
            pending = [p for p in pending if not (p[0] & r)]
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        body = t.split(";")[0]
        toks = body.split(None, 1)
        if in_asm and toks[0] == "global_load_dwordx4":
            dst = regs(toks[1].split(",")[0])
            for p in pending:
                if p[0] & dst:
                    bad += 1; print(f"{path}:{ln} [{kernel}]: {t}   <- reloads registers of the load at line {p[1]} before its use marker")
            pending.append([dst, ln]); continue
        if toks[0] == "s_endpgm":
            pending = []; continue
        used = regs(toks[1]) if len(toks) > 1 else set()
        for p in pending:
            if used & p[0]:
                bad += 1
                print(f"{path}:{ln} [{kernel}]: {t}   <- touches v{sorted(used & p[0])[0]}.. of the in-flight asm load at line {p[1]}")
    return bad

if __name__ == "__main__":
    n = sum(audit(f) for f in sys.argv[1:])
    print("violations:", n)
    sys.exit(1 if n else 0)
